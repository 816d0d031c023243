"""xGMI direct all-reduce (csrc/comm/p2p_allreduce.hip) behind the flat-buffer data-parallel facade.

    base = ddp.FlatDataParallel()                      # torch.distributed: bootstrap + every other collective
    dp = P2PAllReduce(base, device, capacity_elems)    # all-reduce of fp32 flat buffers through peer mappings
    dp.all_reduce_sum_(net.G); opt.step(dp.grad_scale)             # as with any other communicator, or
    opt.step_allreduce_fused(dp)                                   # reduce + 1/world + optimizer update in ONE kernel

One-shot (every rank pulls and reduces everything) up to `oneshot_bytes`, two-shot (pull my slice over all links,
push the reduced slice over all links) above; sums are taken in rank order, so every rank holds the same bits.
The kernel is an ordinary launch on the current stream: it is captured into hipGraphs like any other kernel, with
no dependence on RCCL's capture support.  IPC handles travel through `torch.distributed.all_gather_object`, so any
backend bootstraps it -- including gloo with several processes sharing ONE GPU, which is how the kernel is tested
on a single-GPU box (RCCL refuses duplicate devices; peer mappings of the same device do not care).

`validate()` compares the kernel against the base communicator on a rank-dependent buffer and agrees on the verdict
across ranks; callers keep RCCL when it fails (bench.py --comm p2p).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import torch
import torch.distributed as dist

_LIB = Path(__file__).resolve().parent.parent / "_C" / "libddl_comm.so"
_lib = None
_OPT_KIND = {"adam": 0, "sgd": 1, "rmsprop": 2, "adam_fresh_double": 3}


def _load():
    global _lib
    if _lib is None:
        if not _LIB.exists():
            raise ImportError(f"{_LIB} is missing: run `python build.py --only comm`")
        _lib = C.CDLL(str(_LIB))
        vp, ci, cl, cf, cd = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double
        _lib.ddl_p2p_last_error.restype = C.c_char_p
        _lib.ddl_p2p_create.restype = vp
        _lib.ddl_p2p_create.argtypes = [ci, ci, ci, cl, vp]
        _lib.ddl_p2p_connect.argtypes = [vp, vp]
        _lib.ddl_p2p_configure.argtypes = [vp, ci, cl, cd]
        _lib.ddl_p2p_all_reduce.argtypes = [vp, vp, vp, cl, cf, vp]
        _lib.ddl_p2p_all_reduce_update.argtypes = [vp, vp, cl, cf, vp, vp, vp, vp, vp, ci, cf, cf, cf, cf, cf, vp]
        _lib.ddl_p2p_status.argtypes = [vp]
        _lib.ddl_p2p_destroy.argtypes = [vp]
    return _lib


class P2PError(RuntimeError):
    pass


def _err():
    return _load().ddl_p2p_last_error().decode("utf8", "replace")


class P2PAllReduce:
    supports_fused_update = True
    captures_safely = True          # plain kernel launches: whole-step graph capture needs no RCCL capture support

    def __init__(self, base, device, capacity_elems: int, max_blocks: Optional[int] = None,
                 oneshot_bytes: Optional[int] = None, timeout_s: Optional[float] = None):
        lib = _load()
        self.base = base
        self.world, self.rank = base.world, base.rank
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise P2PError("the xGMI all-reduce needs a GPU")
        if self.world > lib.ddl_p2p_max_world():
            raise P2PError(f"world {self.world} exceeds the kernel's limit {lib.ddl_p2p_max_world()}")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.capacity = int(capacity_elems)
        hb = lib.ddl_p2p_handle_bytes()
        mine = C.create_string_buffer(hb)
        self._h = lib.ddl_p2p_create(self.world, self.rank, idx, self.capacity, mine)
        # Every rank takes part in every collective of this constructor whatever happened locally, and the ranks
        # agree on each stage's outcome before the next one: a rank that cannot allocate or map raises TOGETHER with
        # its peers instead of leaving them inside all_gather_object / barrier (mismatched collectives hang).
        err = None if self._h else _err()
        handles = [None] * self.world
        dist.all_gather_object(handles, (self.rank, err, mine.raw if self._h else b""), group=base.group)
        bad = [(r, e) for r, e, _ in handles if e is not None]
        if bad:
            self.close()
            raise P2PError(f"xGMI buffer allocation failed on rank(s) {[r for r, _ in bad]}: {bad[0][1]}")
        blob = b"".join(h for _, _, h in sorted(handles))
        rc = lib.ddl_p2p_connect(self._h, C.create_string_buffer(blob, len(blob)))
        err = None if rc == 0 else _err()
        verdicts = [None] * self.world
        dist.all_gather_object(verdicts, (self.rank, err), group=base.group)
        bad = [(r, e) for r, e in verdicts if e is not None]
        if bad:
            self.close()
            raise P2PError(f"mapping the peers' buffers failed on rank(s) {[r for r, _ in bad]}: {bad[0][1]}")
        mb = int(os.environ.get("DDL_P2P_BLOCKS", "0")) if max_blocks is None else max_blocks
        lib.ddl_p2p_configure(self._h, mb or 0, -1 if oneshot_bytes is None else oneshot_bytes, timeout_s or 0.0)
        base.barrier()              # every peer has mapped every buffer before the first kernel touches one

    # ---- the facade ------------------------------------------------------------------------------------------
    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    @property
    def group(self):
        return self.base.group

    def _eligible(self, t: torch.Tensor) -> bool:
        return (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0
                and 0 < t.numel() <= self.capacity and t.data_ptr() % 16 == 0)

    def all_reduce_sum_(self, t: torch.Tensor, async_op: bool = False, scale: float = 1.0):
        if not self._eligible(t):
            return self.base.all_reduce_sum_(t)
        rc = _load().ddl_p2p_all_reduce(self._h, t.data_ptr(), t.data_ptr(), t.numel(), scale,
                                        torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise P2PError(_err())
        return None

    def all_reduce_mean_(self, t: torch.Tensor):
        if self._eligible(t):
            return self.all_reduce_sum_(t, scale=1.0 / self.world)
        return self.base.all_reduce_mean_(t)

    def all_reduce_update_(self, opt, lo: int = 0, hi: Optional[int] = None):
        """scale * sum(G[lo:hi]) -> fused optimizer update of that range of opt.net's flat buffers (P, m, v, bf16
        copy) in one kernel.  The caller has already advanced the device step counter
        (FlatOptimizer.step_allreduce_fused); `lo` / `hi` are multiples of 4 elements (gradient buckets)."""
        net, c = opt.net, opt.cfg
        hi = net.G.numel() if hi is None else hi
        g = net.G[lo:hi]
        if lo % 4 or not self._eligible(g):
            raise P2PError("gradient range is not eligible for the fused xGMI all-reduce (size / alignment)")
        p = lambda t, b: None if t is None else t.data_ptr() + lo * b     # noqa: E731
        rc = _load().ddl_p2p_all_reduce_update(self._h, g.data_ptr(), g.numel(), self.grad_scale,
                                               p(net.P, 4), p(opt.m, 4), p(opt.v, 4), p(net.Pc, 2),
                                               opt.step_dev.data_ptr(), _OPT_KIND[c.kind], c.lr, c.beta1, c.beta2,
                                               c.eps, c.rho, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise P2PError(_err())

    # everything that is not a big flat all-reduce goes through the base communicator
    def broadcast_(self, *tensors, src: int = 0):
        return self.base.broadcast_(*tensors, src=src)

    def all_gather_scalars(self, values):
        return self.base.all_gather_scalars(values)

    def reduce_scatter_sum(self, t):
        return self.base.reduce_scatter_sum(t)

    def all_gather_flat(self, shard, out):
        return self.base.all_gather_flat(shard, out)

    def send(self, t, dst):
        return self.base.send(t, dst)

    def recv(self, t, src):
        return self.base.recv(t, src)

    def barrier(self):
        return self.base.barrier()

    # ---- health ------------------------------------------------------------------------------------------------
    def healthy(self) -> bool:
        """False once any wait inside the kernel timed out (synchronises the device)."""
        return _load().ddl_p2p_status(self._h) == 0

    def healthy_everywhere(self) -> bool:
        """`healthy()` agreed across ranks (a collective): after a time-out the kernel skips every reduction and
        update on the rank that saw it, so one unhealthy rank invalidates the whole job's results."""
        ok = self.healthy()
        on_dev = dist.get_backend(self.group) == "nccl"
        f = torch.tensor([0.0 if ok else 1.0], device=self.device if on_dev else "cpu")
        dist.all_reduce(f, op=dist.ReduceOp.MAX, group=self.group)
        return float(f.item()) == 0.0

    def validate(self, numel: Optional[int] = None) -> bool:
        """Kernel vs base communicator on rank-dependent data, both size classes; verdict agreed across ranks."""
        ok = True
        try:
            for n in {min(self.capacity, 4096), min(self.capacity, numel or self.capacity)}:
                n -= n % 4
                x = (torch.arange(n, device=self.device, dtype=torch.float32) % 127.0) * float(self.rank + 1) \
                    + float(self.rank)
                on_dev = dist.get_backend(self.group) == "nccl"
                ref = x.clone() if on_dev else x.cpu()         # gloo bootstrap (tests): reference on the host
                self.base.all_reduce_sum_(ref)
                for _ in range(3):              # three calls: both parities of the double buffers
                    y = x.clone()
                    self.all_reduce_sum_(y)
                    torch.cuda.synchronize()
                    ok = ok and torch.equal(y if on_dev else y.cpu(), ref)
            ok = ok and self.healthy()
        except Exception:       # noqa: BLE001 - the verdict is "no"
            ok = False
        f = torch.tensor([0.0 if ok else 1.0], device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(f, op=dist.ReduceOp.MAX, group=self.group)
        return float(f.item()) == 0.0

    def close(self):
        if self._h:
            _load().ddl_p2p_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
