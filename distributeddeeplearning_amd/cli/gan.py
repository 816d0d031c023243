"""DCGAN entry point (the reference's GAN_distributed.py main(), :127-145).

    python -m distributeddeeplearning_amd.cli.gan                              # 1 GPU
    torchrun --nproc-per-node 8 -m distributeddeeplearning_amd.cli.gan         # data parallel over RCCL

Per iteration: train_discriminator() then train_generator(); every `log_period` iterations print
`generator loss`, `discriminator loss`, the iteration, and dump `sample_images` JPEGs to
./images/<k>.jpg.  `--reference-exact` reproduces the script's accidental optimizer/BN semantics
(SURVEY A.1/A.2); the default is persistent Adam + per-half batch-statistics BN (`--bn-policy split`,
models/dcgan.py).  Unlike the reference -- whose name says "distributed" but which contains no
distribution -- running under torchrun shards the batch and all-reduces one flat gradient buffer per
network per phase.

Observability the reference lacks: `--log-jsonl` records both losses every
`--log-every` iterations and, every `log_period`, the discriminator's mean logits on the three batches
of a step plus a sample-quality score (mean nearest-neighbour distance of generated samples to the
training images); `--sample-grid` writes a PNG sheet of samples; step phases are rocTX ranges.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

from ..config import GANSettings, apply_overrides


def nn_distance(samples: torch.Tensor, real: torch.Tensor) -> float:
    """Mean over generated samples of the per-pixel RMS distance to the nearest training image (lower =
    closer to the data manifold; uniform noise scores ~0.4 on MNIST-like data, held-out data ~0.1)."""
    a = samples.reshape(samples.shape[0], -1).float()
    b = real.reshape(real.shape[0], -1).float()
    d = torch.cdist(a, b)
    return float(d.min(1).values.mean()) / (a.shape[1] ** 0.5)


def _resize(images: torch.Tensor, size: int) -> torch.Tensor:
    """[N,28,28,1] uint8 -> [N,size,size,1] float in [0,1] (bilinear), for the 64x64 configuration."""
    x = images.float().div(255.0)
    if size == images.shape[1]:
        return x
    x = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(size, size), mode="bilinear",
                                        align_corners=False)
    return x.permute(0, 2, 3, 1).contiguous()


def save_grid(path, imgs, cols: int = 8):
    """imgs [N,H,W,1] in [0,1] -> one PNG sheet."""
    import numpy as np
    from PIL import Image
    n, h, w = imgs.shape[0], imgs.shape[1], imgs.shape[2]
    rows = (n + cols - 1) // cols
    sheet = np.zeros((rows * h, cols * w), np.uint8)
    a = (imgs.reshape(n, h, w).clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()
    for k in range(n):
        r, c = divmod(k, cols)
        sheet[r * h:(r + 1) * h, c * w:(c + 1) * w] = a[k]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(sheet).save(path)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--set", action="append", default=[], metavar="FIELD=VALUE")
    ap.add_argument("--device", default=None)
    ap.add_argument("--reference-exact", action="store_true")
    ap.add_argument("--bn-policy", default=None, choices=["split", "inference", "mixed"])
    ap.add_argument("--graph", default="on", choices=["on", "off"])
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--no-plot", action="store_true")
    ap.add_argument("--checkpoint", default=None, help="path of a resumable training-state file")
    ap.add_argument("--checkpoint-every", type=int, default=0)
    ap.add_argument("--log-jsonl", default=None, help="append loss / diagnostics records to this JSONL file")
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--sample-grid", default=None, help="PNG sheet of 32 generated samples, rewritten every log period")
    a = ap.parse_args(argv)
    cfg = apply_overrides(GANSettings(reference_exact=a.reference_exact), a.set)
    if a.bn_policy:
        cfg.bn_policy = a.bn_policy

    from ..models.dcgan import GANConfig, GANTrainer, GraphedGANStep, discriminator_spec, generator_spec
    from ..parallel import ddp
    from ..utils import image, plot, trace
    from ..utils.data import Data

    device = a.device or ("cuda" if torch.cuda.is_available() else "cpu")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(f"cuda:{local}") if device == "cuda" else torch.device("cpu")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    rank, world, _ = ddp.init_process_group(device=dev)
    dp = ddp.FlatDataParallel() if world > 1 else None

    if rank == 0 and not a.no_plot:                                    # GAN_distributed.py:130-131
        plot.plot_model(discriminator_spec(cfg.size), "discriminator.png", show_shapes=True)
        plot.plot_model(generator_spec(cfg.size, noise_dim=cfg.noise_dimension), "generator.png", show_shapes=True)

    data = Data(dev, channels=1, n_train=cfg.train_images * world, n_test=16,
                synthetic=True if a.synthetic else None)
    # x_train[:1000] / 255 (GAN_distributed.py:10-11); under DP every rank takes its own 1000.
    # size 64 [R-ns]: the same images, bilinearly up-scaled.
    real = _resize(data.train_x[rank * cfg.train_images:(rank + 1) * cfg.train_images], cfg.size)
    trainer = GANTrainer(GANConfig(size=cfg.size, batch=cfg.train_images, noise_dim=cfg.noise_dimension,
                                   lr=cfg.learning_rate, reference_exact=cfg.reference_exact, seed=cfg.seed,
                                   bn_policy=cfg.bn_policy), dev, dp, rank)
    start = 0
    if a.checkpoint and os.path.exists(a.checkpoint):
        start = trainer.load_checkpoint(a.checkpoint)
        if rank == 0:
            print(f"resumed from {a.checkpoint} at iteration {start}", flush=True)
    real_p = trainer.D.prepare_input(real)
    log = trace.JsonlLogger(a.log_jsonl if rank == 0 else None)
    log.log(event="start", iteration=start, world=world, bn_policy=trainer.cfg.bn_policy,
            reference_exact=cfg.reference_exact, size=cfg.size, data=data.source, device=str(dev))
    step = None
    i = start
    remaining = max(cfg.iterations - start, 0)
    full = False
    if dp is not None and dev.type == "cuda" and remaining > 0:
        # kernel plans tuned without collectives and shared by all ranks, then: may the collectives be captured into
        # the step graph (gradient buckets under the backward pass need that)?  Agreed across ranks.
        from ..parallel import dp_tune, selfcheck
        dp_tune.pretune_gan(trainer, real_p)
        full = selfcheck.captured_collective_ok(dp, dev)
    first = None                    # losses of the first executed iteration, logged as the reference does at i == 0
    if dev.type == "cuda" and a.graph == "on" and remaining > 0:
        # the graph's warm-up launches are real training steps: they count as iterations (nothing is captured when no
        # iteration remains: a capture without warm-up would record autotuning launches)
        warm = min(3, remaining)
        step = GraphedGANStep(trainer, real_p, warmup=warm, capture_collectives=full)
        first = (trainer.d_loss, trainer.g_loss, i + warm - 1)
        i += warm
    meter = trace.StepMeter(images_per_step=cfg.train_images * world)

    def report(it, d_loss, g_loss):
        """GAN_distributed.py:137-142: losses, iteration, sample JPEGs (+ diagnostics the reference lacks)."""
        with trace.range("gan_log"):
            trainer.sync_bn_stats()
            n_s = max(cfg.sample_images, 32)
            imgs = trainer.generate(n_s)
            if rank == 0:
                print("generator loss : ", float(g_loss))              # GAN_distributed.py:138-140
                print("discriminator loss : ", float(d_loss))
                print(it, f"({meter.images_per_sec():.0f} images/s)", flush=True)
                if log.enabled:
                    diag = trainer.diagnose(real_p)
                    log.log(event="eval", iteration=it, d_loss=float(d_loss), g_loss=float(g_loss),
                            nn_distance=nn_distance(imgs, real), images_per_sec=meter.images_per_sec(), **diag)
                out = imgs.cpu().numpy() * 255.0
                for k in range(cfg.sample_images):
                    image.save_img(os.path.join(cfg.image_directory, f"{k}.jpg"), out[k])
                if a.sample_grid:
                    save_grid(a.sample_grid, imgs[:32])

    if first is not None and start == 0:        # the reference prints and samples at iteration 0
        report(first[2], first[0], first[1])
    while i < cfg.iterations:
        with meter, trace.range("gan_step"):
            d_loss, g_loss = step.step() if step is not None else trainer.train_step(real_p, prepared=True)
        if log.enabled and a.log_every and i % a.log_every == 0:    # float() is a host sync: only when somebody reads it
            log.log(event="loss", iteration=i, d_loss=float(d_loss), g_loss=float(g_loss))
        if i % cfg.log_period == 0 or i == cfg.iterations - 1:
            report(i, d_loss, g_loss)
        i += 1
        if a.checkpoint and a.checkpoint_every and i % a.checkpoint_every == 0:
            trainer.save_checkpoint(a.checkpoint, i, write=rank == 0)
    log.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
