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
        self.graphed, self.dp_report = None, {}
        comms = getattr(self.args, "comms", None) or {}
        if self.device.type == "cuda" and self.args.graph != "off":
            tune = (self.dp is not None and comms and not getattr(self.args, "zero", False)
                    and getattr(self.args, "dp_variant", "default") == "auto" and self.size == 28)
            tuned, can_merge = False, self.t.merged
            if tune:        # communicator x generator pass x bucketing, timed on this machine (parallel/dp_tune.py)
                from .parallel import dp_tune
                err = None
                try:
                    self.graphed, self.dp_report = dp_tune.tune_gan_step(self.t, self.real, comms, self.args.full_ok,
                                                                         verbose=self.rank == 0)
                except Exception as e:      # noqa: BLE001 - every rank falls back together, the record says why
                    err = e
                if dp_tune.agree_max(0.0 if err is None else 1.0, self.device) > 0.0:
                    self.graphed = None
                    self.dp_report = {"dp_tune_failed": repr(err) if err is not None else "failed on another rank"}
                    self.t.set_comm(self.dp)
                    self.t.merged, self.t.buckets, self.t.comm_off, self.t._fake = can_merge, False, False, None
                    comms = {n: c for n, c in comms.items() if c is self.dp}       # the plain, unbucketed RCCL step
                    torch.cuda.synchronize()
                else:
                    tuned = True
            if not tuned:
                from .parallel import dp_tune
                if self.dp is not None and len(comms) > 1:
                    # no step tuning (64x64: a step is ~170 ms, the collectives are noise): the communicator with
                    # the lower latency on the two real gradient sizes
                    lat = {n: sum(dp_tune.allreduce_latency_us(c, net.G.numel(), self.device) for net in (self.t.D, self.t.G))
                           for n, c in comms.items()}
                    name = min(lat, key=lat.get)
                    self.t.set_comm(comms[name])
                    self.dp_report.update(comm_chosen=name,
                                          allreduce_us_D_plus_G={k: round(v, 2) for k, v in lat.items()})
                full = self.args.graph == "full" or getattr(self.t.dp, "captures_safely", False)
                self.graphed = GraphedGANStep(self.t, self.real, capture_collectives=full)
                if self.dp is not None:
                    self.dp_report.setdefault("comm_chosen", next((n for n, c in comms.items() if c is self.t.dp), None))
                    self.dp_report["dp_variant"] = {"merged_generator_pass": self.t.merged,
                                                    "bucketed_allreduce": bool(self.t.buckets and self.graphed.full),
                                                    "fused_optimizer_in_allreduce": bool(self.t.fused)}

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
                                                if self.device.type == "cuda" else None), **self.dp_report}}


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
