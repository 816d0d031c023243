"""DCGAN entry point (the reference's GAN_distributed.py main(), :127-145).

    python -m distributeddeeplearning_amd.cli.gan                              # 1 GPU
    torchrun --nproc-per-node 8 -m distributeddeeplearning_amd.cli.gan         # data parallel over RCCL

Per iteration: train_discriminator() then train_generator(); every `log_period` iterations print
`generator loss`, `discriminator loss`, the iteration, and dump `sample_images` JPEGs to
./images/<k>.jpg.  `--reference-exact` reproduces the script's accidental optimizer/BN semantics
(SURVEY A.1/A.2); the default is persistent Adam + batch-statistics BN.  Unlike the reference --
whose name says "distributed" but which contains no distribution -- running under torchrun shards
the batch and all-reduces one flat gradient buffer per network per phase.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

from ..config import GANSettings, apply_overrides


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--set", action="append", default=[], metavar="FIELD=VALUE")
    ap.add_argument("--device", default=None)
    ap.add_argument("--reference-exact", action="store_true")
    ap.add_argument("--graph", default="on", choices=["on", "off"])
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--no-plot", action="store_true")
    ap.add_argument("--checkpoint", default=None, help="path of a resumable training-state file")
    ap.add_argument("--checkpoint-every", type=int, default=0)
    a = ap.parse_args(argv)
    cfg = apply_overrides(GANSettings(reference_exact=a.reference_exact), a.set)

    from ..models.dcgan import GANConfig, GANTrainer, GraphedGANStep, discriminator_spec, generator_spec
    from ..parallel import ddp
    from ..utils import image, plot
    from ..utils.data import Data
    from ..utils.trace import StepMeter

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
    if cfg.size != 28:
        raise SystemExit("MNIST-shaped data is 28x28; use bench.py --config dcgan64 for the 64x64 stress config")
    # x_train[:1000] / 255 (GAN_distributed.py:10-11); under DP every rank takes its own 1000
    real = data.train_x[rank * cfg.train_images:(rank + 1) * cfg.train_images].float() / 255.0
    trainer = GANTrainer(GANConfig(size=cfg.size, batch=cfg.train_images, noise_dim=cfg.noise_dimension,
                                   lr=cfg.learning_rate, reference_exact=cfg.reference_exact, seed=cfg.seed),
                         dev, dp, rank)
    start = 0
    if a.checkpoint and os.path.exists(a.checkpoint):
        start = trainer.load_checkpoint(a.checkpoint)
        if rank == 0:
            print(f"resumed from {a.checkpoint} at iteration {start}", flush=True)
    real_p = trainer.D.prepare_input(real)
    step = None
    if dev.type == "cuda" and a.graph == "on":
        step = GraphedGANStep(trainer, real_p)
    meter = StepMeter(images_per_step=cfg.train_images * world)
    for i in range(start, cfg.iterations):
        with meter:
            d_loss, g_loss = step.step() if step is not None else trainer.train_step(real_p, prepared=True)
        if i % cfg.log_period == 0:
            if rank == 0:
                print("generator loss : ", float(g_loss))              # GAN_distributed.py:138-140
                print("discriminator loss : ", float(d_loss))
                print(i, f"({meter.images_per_sec():.0f} images/s)", flush=True)
                imgs = trainer.generate(cfg.sample_images).cpu().numpy() * 255.0
                for k in range(cfg.sample_images):
                    image.save_img(os.path.join(cfg.image_directory, f"{k}.jpg"), imgs[k])
        if a.checkpoint and a.checkpoint_every and (i + 1) % a.checkpoint_every == 0 and rank == 0:
            trainer.save_checkpoint(a.checkpoint, i + 1)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
