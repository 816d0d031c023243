"""Flat-buffer optimizers (one fused multi-tensor launch per network on device).

Covers the optimizer set of the reference: Keras `Adam` / `SGD` / `RMSprop` with an lr override
(/root/reference/DDL.py:39-42, eps 1e-7), TF-1 `AdamOptimizer(2e-4)` (eps 1e-8,
/root/reference/GAN_distributed.py:93), and the GAN's accidental "fresh Adam applied twice"
semantics as `adam_fresh_double`.  On device the update kernel also emits the bf16
working copy of the parameters; the periodic-averaging / local-SGD mode [R-ns] folds its weight
average in with one fused kernel of its own (`avg_delta`, parallel/averaging.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from ..ops import cpu_ref
from ..ops import functional as F


@dataclass
class OptimConfig:
    kind: str = "adam"            # adam | sgd | rmsprop | adam_fresh_double
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-7             # Keras default; TF-1 AdamOptimizer uses 1e-8
    rho: float = 0.9

    @staticmethod
    def keras(name: str, lr: float) -> "OptimConfig":
        name = name.lower()
        if name not in ("adam", "sgd", "rmsprop"):
            raise ValueError(name)
        return OptimConfig(kind=name, lr=lr, eps=1e-7)

    @staticmethod
    def tf_adam(lr: float = 2e-4, fresh_double: bool = False) -> "OptimConfig":
        return OptimConfig(kind="adam_fresh_double" if fresh_double else "adam", lr=lr, eps=1e-8)


class FlatOptimizer:
    def __init__(self, net, cfg: OptimConfig):
        self.net, self.cfg = net, cfg
        self.t = 0
        dev = net.P.device
        need_m = cfg.kind == "adam"
        need_v = cfg.kind in ("adam", "rmsprop")
        self.m = torch.zeros_like(net.P) if need_m else None
        self.v = torch.zeros_like(net.P) if need_v else None
        # device-side step counter so a captured hipGraph advances bias correction on replay
        # [steps taken, scratch counter of the fused kernel's last-block increment]
        self.step_dev = torch.zeros(2, dtype=torch.int32, device=dev) if net.on_gpu else None

    def zero_grad(self):
        self.net.G.zero_()

    def step(self, grad_scale: float = 1.0):
        net, c = self.net, self.cfg
        self.t += 1       # host mirror; the device counter `step_dev` is what the kernel reads
        if net.on_gpu:
            F.hip_backend().optimizer_step(self, grad_scale)
            net.refresh_working_copies(cast=False)
            return
        g = net.G if grad_scale == 1.0 else net.G * grad_scale
        if c.kind == "adam":
            cpu_ref.adam_step(net.P, g, self.m, self.v, self.t, c.lr, c.beta1, c.beta2, c.eps)
        elif c.kind == "sgd":
            cpu_ref.sgd_step(net.P, g, c.lr)
        elif c.kind == "rmsprop":
            cpu_ref.rmsprop_step(net.P, g, self.v, c.lr, c.rho, c.eps)
        elif c.kind == "adam_fresh_double":
            cpu_ref.adam_fresh_double_step(net.P, g, c.lr, c.beta1, c.beta2, c.eps)
        else:
            raise ValueError(c.kind)

    def step_allreduce_fused(self, dp, lo: int = 0, hi: Optional[int] = None, first: bool = True, last: bool = True):
        """Data-parallel step with the gradient all-reduce, the 1/world factor and this update in ONE kernel
        (parallel/p2p.py: xGMI direct all-reduce with the optimizer in its epilogue).  Replaces
        `dp.all_reduce_sum_(net.G); opt.step(dp.grad_scale)` and produces the same bits as that pair when the
        communicator sums in rank order.  With gradient buckets the call is made once per range [lo, hi): the
        `first` one advances the step counter, the `last` one redoes the weight relayouts."""
        if first:
            self.t += 1
            F.hip_backend().step_inc(self.step_dev)
        dp.all_reduce_update_(self, lo, hi)
        if last:
            self.net.refresh_working_copies(cast=False)

    def steps_taken(self) -> int:
        """Optimizer steps actually executed.  On device the truth is the device-side counter: hipGraph
        replays advance `step_dev` only, and a capture bumps the host mirror without executing anything."""
        if self.step_dev is not None:
            self.t = int(self.step_dev[0].item())
        return self.t

    def state_dict(self):
        return {"t": self.steps_taken(), "m": self.m, "v": self.v, "cfg": vars(self.cfg)}

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
        if self.m is not None and sd.get("m") is not None:
            self.m.copy_(sd["m"])
        if self.v is not None and sd.get("v") is not None:
            self.v.copy_(sd["v"])
        if self.step_dev is not None:
            self.step_dev.zero_()
            self.step_dev[0] = self.t


class ShardedFlatOptimizer(FlatOptimizer):
    """ZeRO-1 style data-parallel update [R-ns, SURVEY 2.3 "ZeRO/FSDP sharding"]: the flat gradient
    is reduce-scattered, every rank keeps optimizer state for and updates only its 1/world slice
    of the flat master parameters, and the slices are all-gathered back.  On a ring over xGMI this
    moves the same bytes as the all-reduce it replaces, but the update reads 1/world of P/m/v.

    `step()` contains the collectives: callers must not all-reduce the gradients themselves.
    """

    def __init__(self, net, cfg: OptimConfig, dp):
        self.net, self.cfg, self.dp = net, cfg, dp
        self.t = 0
        dev, n, w = net.P.device, net.P.numel(), dp.world
        self.chunk = ((-(-n // w)) + 7) // 8 * 8          # equal 32-byte aligned shards
        self.lo = min(dp.rank * self.chunk, n)
        self.hi = min(self.lo + self.chunk, n)
        self._gbuf = torch.zeros(self.chunk * w, dtype=net.G.dtype, device=dev)
        self._pbuf = torch.zeros(self.chunk * w, dtype=net.P.dtype, device=dev)
        self.pshard = self._pbuf[dp.rank * self.chunk:(dp.rank + 1) * self.chunk]
        self.pshard[:self.hi - self.lo].copy_(net.P[self.lo:self.hi])
        self.m = torch.zeros(self.chunk, dtype=net.P.dtype, device=dev) if cfg.kind == "adam" else None
        self.v = torch.zeros(self.chunk, dtype=net.P.dtype, device=dev) if cfg.kind in ("adam", "rmsprop") else None
        self.step_dev = torch.zeros(2, dtype=torch.int32, device=dev) if net.on_gpu else None

    def sync_from_net(self):
        """Re-read this rank's parameter shard from `net.P` (after a checkpoint load / broadcast rewrote it)."""
        self.pshard.zero_()
        self.pshard[:self.hi - self.lo].copy_(self.net.P[self.lo:self.hi])

    def state_dict(self):
        """Full-size optimizer state: every rank's m / v shard all-gathered (a collective: all ranks call it)."""
        n = self.net.P.numel()
        out = {"t": self.steps_taken(), "cfg": vars(self.cfg), "m": None, "v": None}
        for k in ("m", "v"):
            sh = getattr(self, k)
            if sh is not None:
                full = torch.empty(self.chunk * self.dp.world, dtype=sh.dtype, device=sh.device)
                self.dp.all_gather_flat(sh, full)
                out[k] = full[:n].clone()
        return out

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
        for k in ("m", "v"):
            sh, full = getattr(self, k), sd.get(k)
            if sh is not None and full is not None:
                sh.zero_()
                sh[:self.hi - self.lo].copy_(full[self.lo:self.hi])
        if self.step_dev is not None:
            self.step_dev.zero_()
            self.step_dev[0] = self.t
        self.sync_from_net()

    def step(self, grad_scale: Optional[float] = None):
        net, c, dp = self.net, self.cfg, self.dp
        scale = dp.grad_scale if grad_scale is None else grad_scale
        n = net.P.numel()
        self.t += 1
        self._gbuf[:n].copy_(net.G)
        g = dp.reduce_scatter_sum(self._gbuf)
        if net.on_gpu:
            F.hip_backend().optimizer_step_flat(self.pshard, g, self.m, self.v, None, self.step_dev, c, scale)
        else:
            g = g * scale
            if c.kind == "adam":
                cpu_ref.adam_step(self.pshard, g, self.m, self.v, self.t, c.lr, c.beta1, c.beta2, c.eps)
            elif c.kind == "sgd":
                cpu_ref.sgd_step(self.pshard, g, c.lr)
            elif c.kind == "rmsprop":
                cpu_ref.rmsprop_step(self.pshard, g, self.v, c.lr, c.rho, c.eps)
            elif c.kind == "adam_fresh_double":
                cpu_ref.adam_fresh_double_step(self.pshard, g, c.lr, c.beta1, c.beta2, c.eps)
            else:
                raise ValueError(c.kind)
        dp.all_gather_flat(self.pshard, self._pbuf)
        net.P.copy_(self._pbuf[:n])
        if net.on_gpu:
            net.refresh_working_copies(cast=True)
