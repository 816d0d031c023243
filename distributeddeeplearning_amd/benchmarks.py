"""Benchmark runners behind bench.py (one per BASELINE.json config)."""
from __future__ import annotations

import torch


class _DCGAN:
    def __init__(self, size, device, dp, rank, args):
        self.size, self.device, self.dp, self.rank, self.args = size, device, dp, rank, args
        self.batch = args.batch or (1000 if size == 28 else 32768)     # 64x64: ~59 GB of HBM per GPU (65536 -> 118 GB)

    def setup(self):
        from .models.dcgan import GANConfig, GANTrainer, GraphedGANStep
        cfg = GANConfig(size=self.size, batch=self.batch, seed=0)
        self.t = GANTrainer(cfg, self.device, self.dp, self.rank, shard_optimizer=getattr(self.args, 'zero', False))
        gen = torch.Generator().manual_seed(1234 + self.rank)
        real = torch.rand(self.batch, self.size, self.size, 1, generator=gen)    # synthetic [0,1] images
        self.real = self.t.D.prepare_input(real)
        self.graphed = None
        if self.device.type == "cuda" and self.args.graph != "off":
            self.graphed = GraphedGANStep(self.t, self.real, capture_collectives=self.args.graph == "full")

    def step(self):
        if self.graphed is not None:
            return self.graphed.step()
        return self.t.train_step(self.real, prepared=True)

    def describe(self):
        return {"metric": f"dcgan{self.size}_train_images_per_sec",
                "images_per_step_per_rank": self.batch,
                "config": {"model": f"DCGAN {self.size}x{self.size} (GAN_distributed generator+discriminator)",
                           "per_gpu_batch": self.batch, "seq_len": None,
                           "step": "D update on [B real; B fake] + G update on B noise, Adam 2e-4, "
                                   f"BN policy {self.t.cfg.bn_policy} (batch statistics, per half in D)",
                           "graph": self.args.graph},
                "extra": {"d_loss": float(self.t.d_loss), "g_loss": float(self.t.g_loss),
                          "hbm_high_water_gb": (round(torch.cuda.max_memory_allocated(self.device) / 2 ** 30, 2)
                                                if self.device.type == "cuda" else None)}}


def make(name, device, dp, rank, args):
    if name == "dcgan28":
        return _DCGAN(28, device, dp, rank, args)
    if name == "dcgan64":
        return _DCGAN(64, device, dp, rank, args)
    if name == "classifier":
        from .bench_classifier import ClassifierBench
        return ClassifierBench(device, dp, rank, args)
    if name == "localsgd":
        from .bench_classifier import LocalSGDBench
        return LocalSGDBench(device, dp, rank, args)
    if name == "ensemble32":
        from .bench_classifier import EnsembleBench
        return EnsembleBench(device, dp, rank, args)
    raise ValueError(name)
