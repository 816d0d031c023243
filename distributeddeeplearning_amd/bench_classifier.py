"""Benchmark runners for the classifier configs of BASELINE.json:

* `classifier` -- config 2: the five exported replica architectures (model0..model4,
  /root/reference/models/*/architecture.json, shipped as package assets) trained synchronously
  data-parallel: one step = one optimizer step on EACH of the five networks (per-GPU batch B each),
  with one flat RCCL gradient all-reduce per network per step;
* `localsgd`   -- config 5: the same five networks with NO per-step collective; weights are averaged
  every k steps by an all-reduce overlapped on a side HIP stream (parallel/averaging.py).

Images are synthetic MNIST-shaped uint8 resident in HBM; a batch is one on-device gather.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from .config import get_space
from .models.classifier import RandomModel, ReplicaGroupStep
from .models.ir import ModelSpec
from .models.optim import OptimConfig
from .models.search import CompiledSpec
from .utils.data import Data

ASSETS = Path(__file__).resolve().parent / "assets"


def replica_specs():
    return [ModelSpec.from_keras_json((ASSETS / f"model{i}.architecture.json").read_text()) for i in range(5)]


class ClassifierBench:
    local_sgd = False

    def __init__(self, device, dp, rank, args):
        self.device, self.dp, self.rank, self.args = device, dp, rank, args
        self.batch = args.batch or 512

    def setup(self):
        from .parallel.averaging import PeriodicAverager
        n_pool = max(4096, self.batch * 4)
        space = get_space("ddl_normal4", [f"data_train_size={n_pool}", f"batch_size={self.batch}"])
        self.data = Data(self.device, channels=1, n_train=n_pool, n_test=256, synthetic=True, seed=self.rank)
        self.members = []
        use_graph = self.device.type == "cuda" and self.args.graph != "off"
        self.averager = None
        if self.local_sgd and self.dp is not None:
            self.averager = PeriodicAverager(self.dp, every=self.args.avg_every, overlap=True)
        for i, spec in enumerate(replica_specs()):
            compiled = CompiledSpec(spec, OptimConfig.keras("adam", 1e-3), "categorical_crossentropy")
            sample = self.data.get_random_sample(n_pool, seed=100 + self.rank)
            m = RandomModel(self.data, sample, space, compiled, self.device, number=i, seed=i,
                            dp=self.dp, averager=self.averager, use_graph=use_graph, verbose=False)
            if self.dp is not None:
                self.dp.broadcast_(m.net.P, m.net.S)
                m.net.refresh_working_copies()
            self.members.append(m)
        g = torch.Generator().manual_seed(7 + self.rank)
        self.idx = [torch.randint(0, n_pool, (self.batch,), generator=g).to(self.device) for _ in range(8)]
        self.t = 0
        self.loss = None
        # synchronous DP: the five networks' all-reduces overlap the next network's compute (ReplicaGroupStep)
        self.group = None
        if self.dp is not None and not self.local_sgd:
            mode = "eager" if not use_graph else ("full" if self.args.graph == "full" else "segments")
            self.group = ReplicaGroupStep(self.members, self.dp, self.idx[0], mode)

    def step(self):
        idx = self.idx[self.t % len(self.idx)]
        self.t += 1
        if self.group is not None:
            self.loss = self.group.step(idx)
            return self.loss
        for m in self.members:
            self.loss = m.train_step(idx)
        return self.loss

    def describe(self):
        name = "localsgd" if self.local_sgd else "classifier"
        cfg = {"model": "5 replicas model0..model4 (DDL_Normal4 random_layer CNNs, reference architecture.json)",
               "per_gpu_batch": self.batch, "seq_len": None, "networks_per_step": 5,
               "step": "fwd + softmax-CE + bwd + Adam on each of the 5 replicas",
               "sync": (f"weight average every {self.args.avg_every} steps, overlapped" if self.local_sgd
                        else "flat gradient all-reduce per network per step, overlapped with the next network"),
               "graph": self.args.graph}
        return {"metric": f"{name}_train_images_per_sec", "images_per_step_per_rank": self.batch * 5,
                "config": cfg, "extra": {"loss": float(self.loss)}}


class LocalSGDBench(ClassifierBench):
    local_sgd = True
