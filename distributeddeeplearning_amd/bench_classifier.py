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

import os
import time
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
        # one GPU, no collectives: the five replicas are independent -- their step graphs go to separate streams,
        # placed longest-first (models/classifier.py member_streams), so one replica's latency-bound tail layers run
        # beside another's bandwidth-bound front layers
        self.streams = None
        if (self.dp is None and use_graph and self.device.type == "cuda"
                and os.environ.get("DDL_CLS_CONCURRENT", "1") == "1"):
            from .models.classifier import member_streams
            for m in self.members:
                for _ in range(3):
                    m.train_step(self.idx[0])           # warm-up + capture
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    m.train_step(self.idx[0])
                torch.cuda.synchronize()
                m.chunk_ms = (time.perf_counter() - t0) * 1e3 / 3
            k = int(os.environ.get("DDL_CLS_STREAMS", "4"))
            self.streams, self.stream_pool = member_streams(self.members, self.device, k)
            self.replica_ms = [round(m.chunk_ms, 3) for m in self.members]

    def step(self):
        idx = self.idx[self.t % len(self.idx)]
        self.t += 1
        if self.group is not None:
            self.loss = self.group.step(idx)
            return self.loss
        if self.streams is not None:
            main = torch.cuda.current_stream()
            for st in self.stream_pool:
                st.wait_stream(main)
            for m, st in zip(self.members, self.streams):
                with torch.cuda.stream(st):
                    self.loss = m.train_step(idx)
            for st in self.stream_pool:
                main.wait_stream(st)
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
        extra = {"loss": float(self.loss),
                 "replica_streams": len(self.stream_pool) if self.streams is not None else 0,
                 "replica_ms": getattr(self, "replica_ms", None),
                 "replica_group_mode": self.group.mode if self.group is not None else None,
                 "averages_done": self.averager.n_averages if self.averager is not None else None}
        return {"metric": f"{name}_train_images_per_sec", "images_per_step_per_rank": self.batch * 5,
                "config": cfg, "extra": extra}


class LocalSGDBench(ClassifierBench):
    local_sgd = True


class EnsembleBench:
    """`--config ensemble32`: the reference's real regime -- an ensemble generation of random architectures trained
    at batch 32 (Keras fit() default, /root/reference/DDL.py:92) -- on ONE GPU per rank.

    One bench step = `RandomModel.CHUNK` (32) optimizer steps of EVERY member: each member replays its multi-step
    graph on its own HIP stream, so the members' tiny kernels share the GPU (models/classifier.py).  With N ranks
    every rank trains its own N-member generation (ensemble-member parallelism has no gradient exchange).
    `DDL_SERIAL_MEMBERS=1` issues the members on one stream, `DDL_CHUNK=1` falls back to one graph per step:
    the two knobs price the two optimisations separately."""

    def __init__(self, device, dp, rank, args):
        self.device, self.rank, self.args = device, rank, args
        self.batch = args.batch or 32
        self.n_members = getattr(args, "members", 10)
        self.preset = getattr(args, "preset", "ddl")

    def setup(self):
        from .models import search
        self.serial = os.environ.get("DDL_SERIAL_MEMBERS", "0") == "1"
        chunk = max(RandomModel.CHUNK, 1)
        n_pool = max(8192, self.batch * chunk * 2)
        space = get_space(self.preset, [f"data_train_size={n_pool}", f"batch_size={self.batch}", "seed=0"])
        self.data = Data(self.device, channels=space.channels, n_train=n_pool, n_test=256, synthetic=True,
                         seed=self.rank)
        rng = np.random.default_rng(space.seed)
        self.members = []
        use_graph = self.device.type == "cuda" and self.args.graph != "off"
        for i in range(self.n_members):
            compiled = search.make_new_model(space, rng, self.data.input_shape, 10)
            sample = self.data.get_random_sample(n_pool, seed=100 + i)
            m = RandomModel(self.data, sample, space, compiled, self.device, number=i, seed=i, use_graph=use_graph,
                            verbose=False)
            m.prepare(self.batch)
            self.members.append(m)
        self.chunk = getattr(self.members[0], "_chunk_S", 0) if use_graph else 0
        self.steps_per_call = self.chunk if self.chunk else 1
        on_gpu = self.device.type == "cuda"
        self.streams = None
        if on_gpu and not self.serial:  # K streams (hardware queues), members placed longest-first (models/classifier.py)
            from .models.classifier import member_streams
            self.streams, self.stream_pool = member_streams(self.members, self.device)
        g = torch.Generator().manual_seed(7 + self.rank)
        self.idx = [torch.randint(0, n_pool, (self.steps_per_call, self.batch), generator=g).to(self.device)
                    for _ in range(4)]
        self.t = 0
        self.params = [m.net.num_params(True) for m in self.members]
        if on_gpu:
            torch.cuda.synchronize()

    def _one(self, m, idx):
        if self.chunk:
            m._idx_chunk.copy_(idx)
            m._chunk_graph.replay()
        else:
            for k in range(idx.shape[0]):
                m.train_step(idx[k])

    def step(self):
        idx = self.idx[self.t % len(self.idx)]
        self.t += 1
        if self.streams is None:
            for m in self.members:
                self._one(m, idx)
            return
        main = torch.cuda.current_stream()
        for st in self.stream_pool:
            st.wait_stream(main)
        for m, st in zip(self.members, self.streams):
            with torch.cuda.stream(st):
                self._one(m, idx)
        for st in self.stream_pool:          # the step ends when every member's chunk has: bench timing stays honest
            main.wait_stream(st)

    def describe(self):
        cfg = {"model": f"{self.n_members} random members of preset {self.preset} (DDL.py search space), batch {self.batch}",
               "per_gpu_batch": self.batch, "seq_len": None, "members": self.n_members,
               "optimizer_steps_per_member_per_bench_step": self.steps_per_call,
               "member_params": self.params,
               "step": "fwd + softmax-CE + bwd + optimizer, every member, "
                       + ("members one after another" if self.streams is None else
                          f"members concurrent on {len(self.stream_pool)} streams, placed longest-first by measured step time"),
               "graph": self.args.graph}
        return {"metric": "ensemble32_train_images_per_sec",
                "images_per_step_per_rank": self.batch * self.steps_per_call * self.n_members,
                "config": cfg, "extra": {"member_streams": len(self.stream_pool) if self.streams is not None else 0,
                                         "steps_per_graph": self.chunk}}
