"""Persistent whole-network trainer for the MLP members of a batch-32 ensemble (csrc/hip/mlp_persistent.hip).

The reference's Ray scripts train MLP-only ensembles (/root/reference/DDL_Ray.py:21: `conv2D_size` is always 0, so
a member is Flatten -> [Dense(act) -> BatchNorm]* -> Dense(10, softmax), :76-85) with Keras' default batch of 32.
For such members the multi-kernel executor is ~60 launch-bound kernels per step.  `MlpGroup` instead hands a whole
GROUP of members to one kernel launch per epoch: one workgroup per member runs every optimizer step of the epoch
(gather, forward, softmax-CE, backward, Adam / SGD / RMSprop) with the step's activations in LDS, and updates the
members' flat fp32 buffers (`Network.P`, optimizer moments, BN moving statistics) in place -- so checkpoints,
`predict`, transfer learning and the rest of the framework see an ordinary trained `RandomModel`.

`train_group(models)` reproduces `RandomModel.train()` for every member: the same per-member shuffles (each
member's own generator), the reference's early-stop rule per member (DDL.py:92-96), the same loss bookkeeping
(Keras' sample-weighted epoch mean, ragged last batch included).
"""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np
import torch

from ..ops import functional as F
from .network import BNOp, ConvOp, ViewOp

_OPT = {"adam": 0, "sgd": 1, "rmsprop": 2}
MAX_LAYERS = 11     # MLP_MAXL of csrc/hip/mlp_persistent.hip (checked against the library in `limits`)
_limits = None


def _lib():
    hip = F.hip_backend()
    lib = hip._lib
    if not hasattr(lib, "_mlp_ready"):
        lib.ddl_mlp_member_bytes.restype = C.c_long
        lib.ddl_mlp_limits.argtypes = [C.POINTER(C.c_int)] * 3
        lib.ddl_mlp_persistent.restype = C.c_int
        lib.ddl_mlp_persistent.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p]
        lib._mlp_ready = True
    return lib


def limits():
    global _limits
    if _limits is None:
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _lib().ddl_mlp_limits(C.byref(a), C.byref(b), C.byref(c))
        _limits = (a.value, b.value, c.value)
        if a.value != MAX_LAYERS or _lib().ddl_mlp_member_bytes() != C.sizeof(_Member):
            raise ImportError("libddl_hip.so's MlpMember layout differs from models/mlp_fast.py: rebuild (python build.py)")
    return _limits


class _Layer(C.Structure):
    _fields_ = [("w_off", C.c_int), ("b_off", C.c_int), ("N", C.c_int), ("Kp", C.c_int), ("act", C.c_int),
                ("has_bn", C.c_int), ("g_off", C.c_int), ("be_off", C.c_int), ("mm_off", C.c_int), ("mv_off", C.c_int),
                ("alpha", C.c_float), ("momentum", C.c_float), ("bn_eps", C.c_float), ("pad_", C.c_int)]


class _Member(C.Structure):
    _fields_ = [("P", C.c_void_p), ("M", C.c_void_p), ("V", C.c_void_p), ("S", C.c_void_p), ("step", C.c_void_p),
                ("loss_acc", C.c_void_p), ("idx", C.c_void_p), ("n_layers", C.c_int), ("pixels", C.c_int),
                ("cpad0", C.c_int), ("ctrue0", C.c_int), ("opt_kind", C.c_int), ("lr", C.c_float), ("b1", C.c_float),
                ("b2", C.c_float), ("eps", C.c_float), ("rho", C.c_float), ("L", _Layer * MAX_LAYERS)]


def _chain(net):
    """[(dense ConvOp, BNOp | None)] if `net` is Flatten -> [Dense (-> BN)]* -> Dense(softmax), else None."""
    ops = net.ops
    if not ops or not isinstance(ops[0], ViewOp) or not net.softmax_head:
        return None
    out, i = [], 1
    while i < len(ops):
        op = ops[i]
        if not (isinstance(op, ConvOp) and op.kind == "dense"):
            return None
        bn = None
        if i + 1 < len(ops) and isinstance(ops[i + 1], BNOp):
            bn = ops[i + 1]
            if bn.post_act not in (None, "linear") or bn.mode != "batch":
                return None
            i += 1
        out.append((op, bn))
        i += 1
    return out or None


def eligible(model) -> bool:
    """Can this RandomModel be trained by the persistent kernel?  (device MLP chain within the kernel's limits,
    cross-entropy loss, no data parallelism, dataset of 1-channel uint8 images)."""
    net = model.net
    if not (net.on_gpu and model.dp is None and model.averager is None):
        return False
    if model.loss_name != "categorical_crossentropy" or model.opt.cfg.kind not in _OPT:
        return False
    chain = _chain(net)
    if chain is None:
        return False
    max_l, max_n, batch = limits()
    if len(chain) > max_l or any(op.g.O > max_n for op, _ in chain) or model.space.batch_size != batch:
        return False
    if len(net.in_shape) != 3 or net.in_shape[-1] > 4:
        return False
    x = model.data.train_x
    if x.dtype != torch.uint8 or x.shape[-1] != 1 or not x.is_contiguous():
        return False
    first = chain[0][0]
    if first.kstruct is None:
        return False
    if chain[-1][0].act not in (None, "linear") or chain[-1][1] is not None:
        return False
    return True


class MlpGroup:
    def __init__(self, models: List):
        if not models or not all(eligible(m) for m in models):
            raise ValueError("MlpGroup needs eligible MLP members (mlp_fast.eligible)")
        self.models = models
        self.device = models[0].device
        self.data = models[0].data
        if any(m.data is not self.data for m in models):
            raise ValueError("members of one group must share the resident dataset")
        self.pixels = int(np.prod(self.data.train_x.shape[1:]))
        self.loss = torch.zeros(len(models), dtype=torch.float32, device=self.device)
        self._keep = []

    def _record(self, k: int, m, idx: torch.Tensor) -> _Member:
        from ..ops.hip import ACT_CODE
        net, opt = m.net, m.opt
        chain = _chain(net)
        H, W, Cc = net.in_shape
        r = _Member()
        r.P, r.S = net.P.data_ptr(), net.S.data_ptr()
        r.M = opt.m.data_ptr() if opt.m is not None else None
        r.V = opt.v.data_ptr() if opt.v is not None else None
        r.step = opt.step_dev.data_ptr()
        r.loss_acc = self.loss[k:k + 1].data_ptr()
        r.idx = idx.data_ptr()
        r.n_layers, r.pixels, r.cpad0, r.ctrue0 = len(chain), H * W, net.cpad, Cc
        c = opt.cfg
        r.opt_kind, r.lr, r.b1, r.b2, r.eps, r.rho = _OPT[c.kind], c.lr, c.beta1, c.beta2, c.eps, c.rho
        pm = net._pmap
        for i, (op, bn) in enumerate(chain):
            L = r.L[i]
            L.w_off, L.b_off = pm[(op.name, "kernel")].offset, pm[(op.name, "bias")].offset
            L.N = op.g.O
            L.Kp = int(np.prod(pm[(op.name, "kernel")].shape[1:]))
            L.act, L.alpha = ACT_CODE[op.act], op.alpha
            L.has_bn = int(bn is not None)
            if bn is not None:
                L.g_off, L.be_off = pm[(bn.name, "gamma")].offset, pm[(bn.name, "beta")].offset
                L.mm_off = pm[(bn.name, "moving_mean")].offset
                L.mv_off = pm[(bn.name, "moving_variance")].offset
                L.momentum, L.bn_eps = bn.momentum, bn.eps
        return r

    def run_epoch(self, active: List[int], batch: int):
        """One epoch of every member in `active` (indices into self.models): ONE kernel launch.  Returns the
        device tensor of per-member loss sums (mean batch loss * rows, summed over the epoch)."""
        lib = _lib()
        ms = [self.models[i] for i in active]
        n = ms[0].sample.train_idx.numel()
        if any(m.sample.train_idx.numel() != n for m in ms):
            raise ValueError("members of one launch must have equally sized training splits")
        steps = -(-n // batch)
        last = n - (steps - 1) * batch
        recs = (_Member * len(ms))()
        keep = []
        for k, (i, m) in enumerate(zip(active, ms)):
            tr = m.sample.train_idx
            perm = tr[torch.randperm(n, generator=m.gen).to(tr.device)]
            idx = torch.zeros(steps * batch, dtype=torch.int64, device=self.device)
            idx[:n] = perm
            keep.append(idx)
            recs[k] = self._record(i, m, idx)
        self.loss.zero_()
        table = torch.from_numpy(np.frombuffer(bytes(recs), dtype=np.uint8).copy()).to(self.device)
        rc = lib.ddl_mlp_persistent(table.data_ptr(), len(ms), self.data.train_x.data_ptr(),
                                    self.data.train_labels.data_ptr(), self.pixels, steps, last,
                                    torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"mlp_persistent rejected the launch (rc={rc})")
        self._keep = [keep, table]          # alive until the next launch on this stream
        return self.loss, n

    def finish(self):
        """bf16 working copies back in sync with the fp32 masters the kernel updated."""
        for m in self.models:
            m.net.refresh_working_copies(cast=True)
            m.opt.steps_taken()


def train_group_gen(models: List, verbose: bool = False):
    """`RandomModel.train()` for every member of an eligible group, one kernel launch per epoch for all of them.
    A generator in the protocol of RandomModel._train_gen: yields None after issuing work and an event when it
    needs the round's losses back (models/classifier.py train_concurrently keeps other members going meanwhile)."""
    grp = MlpGroup(models)
    sp = models[0].space
    active = list(range(len(models)))
    for rnd in range(int(sp.epochs / sp.mini_epochs)):
        if not active:
            break
        sums = []
        for _ in range(sp.mini_epochs):
            loss, n = grp.run_epoch(active, sp.batch_size)
            sums.append(loss[list(active)].clone() / n)
            yield None
        stacked = torch.stack(sums)
        ev = torch.cuda.Event()
        ev.record()
        yield ev
        vals = stacked.cpu()                        # one synchronisation per round, already complete
        nxt = []
        for col, i in enumerate(active):
            m, h = models[i], [float(v) for v in vals[:, col]]
            m.history.append(h)
            if m.verbose or verbose:
                print(rnd, h, flush=True)
            if not (h[0] == 0 or 1 - (h[-1] / h[0]) < sp.early_stop_rate):
                nxt.append(i)
        active = nxt
    grp.finish()


def train_group(models: List, verbose: bool = False):
    for ev in train_group_gen(models, verbose):
        if ev is not None:
            ev.synchronize()
