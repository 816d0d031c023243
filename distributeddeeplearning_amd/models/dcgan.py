"""DCGAN on 28x28 (reference-exact graphs) and 64x64 [R-ns], with a data-parallel trainer.

Graphs: /root/reference/GAN_distributed.py:15-72 (shapes in SURVEY 2.6.1).  Step semantics:
/root/reference/GAN_distributed.py:86-117 -- per iteration a D-step on [real ; G(noise)] with
labels [1 ; 0] in ONE pass, then a G-step through the frozen D with labels 1; sigmoid-BCE on
logits; Adam(2e-4).

Optimizer: `reference_exact=True` is the script's literal "fresh TF-Adam applied twice
per phase" (== 2*lr*sign(g)); the default is one persistent Adam per network.

BatchNorm policy (`bn_policy`).  The batch composition differs between the two phases
([real ; fake] in the D phase, fake only in the G phase) and `generator.predict` is an inference-phase
call, so WHICH statistics normalise WHAT decides whether the game is adversarial at all:
  * "split" (default): batch statistics everywhere, standard DCGAN practice -- the generator runs in
    training mode for the D-phase fakes too, and the discriminator normalises the real half and the
    fake half of its batch separately.  A fake batch then looks the same to D in both phases.
  * "inference": the reference's own semantics -- every BN in inference phase on its (never updated)
    moving statistics, i.e. a learned per-channel affine.  Forced by `reference_exact`.
  * "mixed": round-1 behaviour, kept for study only: D normalises [real ; fake] as ONE batch and the
    D-phase fakes come from inference-phase G.  D separates real from fake through the batch
    statistics, a fake-only batch then reads as "real", and both losses collapse to ~1e-4 within a
    few hundred iterations while G learns nothing (docs/GAN_TRAINING.md has the measurements).

No host round trips inside a step: noise comes from a device RNG, the generated batch is written
next to the real batch on device, the loss stays on device until someone asks for it.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from ..ops import functional as F
from ..utils import hipgraph
from . import ir
from .network import ConvOp, Network
from .optim import FlatOptimizer, OptimConfig, ShardedFlatOptimizer
from ..utils.streams import comm_stream

NOISE_DIM = 128


def discriminator_spec(size: int = 28, base: int = 32) -> ir.ModelSpec:
    """28: Conv 5x5 s2 1->32, LeakyReLU, Conv 32->64, BN, LeakyReLU, Conv 64->128, BN, LeakyReLU,
    Flatten, Dense 1 (GAN_distributed.py:16-43).  64: one more stride-2 stage [R-ns]."""
    n = ir.NameScope()
    m = ir.ModelSpec("discriminator", [ir.Input(n("input"), (size, size, 1))])
    stages = 3 if size == 28 else 4
    for s in range(stages):
        m.append(ir.Conv2D(n("conv2d"), base << s, 5, 2, 1, "same"))
        if s > 0:
            m.append(ir.BatchNorm(n("batch_norm")))
        m.append(ir.LeakyReLU(n("leaky_relu"), 0.3))
    m.append(ir.Flatten(n("flatten")))
    m.append(ir.Dense(n("dense"), 1))
    return m


def generator_spec(size: int = 28, base: int = 32, noise_dim: int = NOISE_DIM) -> ir.ModelSpec:
    """28: Dense 128->7*7*128, Reshape, ConvT 5x5 s2 ->64 relu, BN, ConvT s2 ->32 relu, BN,
    ConvT s1 ->1 tanh, Lambda (x+1)/2 (GAN_distributed.py:46-72).  64: 8x8x256 seed and three
    stride-2 stages [R-ns]."""
    n = ir.NameScope()
    m = ir.ModelSpec("generator", [ir.Input(n("input"), (noise_dim,))])
    if size == 28:
        seed_hw, chans = 7, [base * 4, base * 2, base]
    elif size == 64:
        seed_hw, chans = 8, [base * 8, base * 4, base * 2, base]
    else:
        raise ValueError("size must be 28 or 64")
    m.append(ir.Dense(n("dense"), seed_hw * seed_hw * chans[0]))
    m.append(ir.Reshape(n("reshape"), (seed_hw, seed_hw, chans[0])))
    for c in chans[1:]:
        m.append(ir.Conv2DTranspose(n("conv2d_transpose"), c, 5, 2, 1, "same", "relu"))
        m.append(ir.BatchNorm(n("batch_norm")))
    m.append(ir.Conv2DTranspose(n("conv2d_transpose"), 1, 5, 1, 1, "same", "tanh"))
    m.append(ir.LambdaAffine(n("lambda_affine"), 0.5, 0.5))
    return m


@dataclass
class GANConfig:
    size: int = 28
    batch: int = 1000               # real images per step per rank (reference: 1000, full batch)
    noise_dim: int = NOISE_DIM
    lr: float = 2e-4
    reference_exact: bool = False
    seed: int = 0
    base: int = 32
    bn_policy: str = "split"        # split | inference | mixed (module docstring)

    def __post_init__(self):
        if self.reference_exact:
            self.bn_policy = "inference"
        if self.bn_policy not in ("split", "inference", "mixed"):
            raise ValueError(f"bn_policy {self.bn_policy!r}")


class DeviceRNG:
    """Uniform(-1,1) noise.  Device: counter-based Philox kernel whose offset lives in device
    memory (replay-safe under hipGraph capture).  CPU: torch.Generator."""

    def __init__(self, device, seed: int):
        self.device = torch.device(device)
        self.seed = int(seed)
        if self.device.type == "cuda":
            self.state = torch.zeros(3, dtype=torch.int64, device=self.device)     # seed, offset, kernel scratch
            self.state[0] = self.seed
        else:
            self.gen = torch.Generator().manual_seed(self.seed)

    def uniform(self, shape, dtype, lo=-1.0, hi=1.0):
        if self.device.type == "cuda":
            return F.hip_backend().philox_uniform(shape, dtype, self.state, lo, hi)
        return (torch.rand(shape, generator=self.gen) * (hi - lo) + lo).to(dtype)

    def state_dict(self):
        if self.device.type == "cuda":
            return {"philox": self.state.detach().cpu()}
        return {"generator": self.gen.get_state()}

    def load_state_dict(self, sd, keep_seed: bool = False):
        """`keep_seed`: restore the stream position only (a state saved by ANOTHER rank: this rank keeps its own
        seed so the ranks' noise stays different)."""
        if self.device.type == "cuda":
            seed = self.state[0].clone()
            self.state.copy_(sd["philox"])
            if keep_seed:
                self.state[0] = seed
        elif not keep_seed:
            self.gen.set_state(sd["generator"])


class GANTrainer:
    """One rank's DCGAN trainer.  `dp` (parallel.ddp.FlatDataParallel or None) averages the flat
    gradient buffer of the network being updated across ranks -- one collective per phase."""

    def __init__(self, cfg: GANConfig, device="cpu", dp=None, rank: int = 0, shard_optimizer: bool = False):
        self.cfg, self.device, self.dp = cfg, torch.device(device), dp
        self.sharded = bool(shard_optimizer and dp is not None)
        bn_mode = "inference" if cfg.bn_policy == "inference" else "batch"
        self._split = cfg.bn_policy == "split"
        # same seed on every rank -> identical initial replicas (dp.broadcast makes it certain)
        self.G = Network(generator_spec(cfg.size, cfg.base, cfg.noise_dim), device, cfg.seed + 1, bn_mode)
        self.D = Network(discriminator_spec(cfg.size, cfg.base), device, cfg.seed + 2, bn_mode)
        self.G.set_no_input_grad()      # nothing differentiates with respect to the noise
        # weight-gradient chains as parallel branches of the captured step (models/network.py SideBranch)
        br = os.environ.get("DDL_GAN_BRANCH", "0")
        self.D.branch_wgrad = br in ("1", "D", "DG")
        self.G.branch_wgrad = br in ("1", "G", "DG")
        ocfg = OptimConfig.tf_adam(cfg.lr, fresh_double=cfg.reference_exact)
        self.optG, self.optD = FlatOptimizer(self.G, ocfg), FlatOptimizer(self.D, ocfg)
        self._ocfg = ocfg
        self.rng = DeviceRNG(device, cfg.seed * 7919 + 17 + rank)
        self._fake = None
        self.head: ConvOp = self.D.ops[-1]
        assert self.head.kind == "dense" and self.head.g.O == 1
        self.d_loss = self.g_loss = None
        if dp is not None:
            dp.broadcast_(self.G.P, self.G.S, self.D.P, self.D.S)
            self.G.refresh_working_copies()
            self.D.refresh_working_copies()
        if self.sharded:        # reduce-scatter / shard update / all-gather instead of all-reduce (ZeRO-1)
            self.optG = ShardedFlatOptimizer(self.G, ocfg, dp)
            self.optD = ShardedFlatOptimizer(self.D, ocfg, dp)
        # xGMI direct all-reduce with the optimizer in its epilogue (parallel/p2p.py): reduce + 1/world + update
        # of a network are ONE kernel, so the separate optimizer launches disappear from the step
        self.fused = bool(dp is not None and not self.sharded and self.G.on_gpu
                          and getattr(dp, "supports_fused_update", False))
        self._comm_stream = comm_stream() if (dp is not None and self.G.on_gpu) else None
        # Gradient buckets (Network.grad_buckets): the tail of each flat gradient buffer -- D: conv3 + BN + head, 0.8
        # of its 1.04 MB; G: the transposed convolutions -- is all-reduced on the communication stream as soon as
        # its last layer's backward has been issued, under the rest of the backward pass; only the bucket that
        # finishes last (D: conv1 + conv2, 0.2 MB; G: the Dense layer) stays exposed.  One launch per bucket in a
        # fixed order on every rank; sums are elementwise, so bucketed == unbucketed bit for bit for any
        # communicator whose per-element summation order does not depend on the buffer length (the xGMI kernel: rank
        # order; two ranks: commutative).  DDL_DP_BUCKETS=1 restores one collective per network.
        nb = int(os.environ.get("DDL_DP_BUCKETS", "2"))
        self.buckets = bool(dp is not None and not self.sharded and nb > 1)
        self._bk = {id(n): n.grad_buckets(max(nb, 1)) for n in (self.D, self.G)}
        self.comm_off = False           # measurement knob (parallel/dp_tune.py): same step, collectives skipped
        # "split" policy on device: the generator's weights are the same in the D phase and in the G phase of an
        # iteration (G is updated last), so both forwards run as ONE pass over [z1 ; z2] with per-half batch
        # statistics: half the generator launches, each on twice the rows (these kernels do not fill 256 CUs at
        # B = 1000).  The pass writes into one [3B] image buffer laid out [real | G(z1) | G(z2)]: the D phase reads
        # the first 2B images, the G phase the last B -- no concat, no copy of the real images per step.
        self.merged = bool(self._split and self.G.on_gpu and self.G.can_forward_into()
                           and os.environ.get("DDL_GAN_MERGED_G", "1") == "1")
        self._buf3 = None
        # D ends BN -> LeakyReLU -> Flatten -> Dense(1): with batch statistics on device the head kernel applies
        # that BatchNorm on load and emits its backward reduction (ops/hip.py dense1_bce_fwd_bwd `bn=`)
        self._head_bn = self._find_head_bn()

    def _find_head_bn(self):
        from .network import BNOp, ViewOp
        ops = self.D.ops
        if not (self.D.on_gpu and os.environ.get("DDL_GAN_HEAD_BN", "1") == "1" and len(ops) >= 4):
            return None
        bn, view, head = ops[-3], ops[-2], ops[-1]
        if not (isinstance(bn, BNOp) and isinstance(view, ViewOp) and head is self.head):
            return None
        K = int(self.head.w_main.numel())
        if bn.mode != "batch" or not bn.rank4 or bn.Cp != bn.C or bn.bias_grad_for is not None:
            return None
        if 512 % bn.Cp or K % bn.Cp or K > 2048 or (bn.Cp // 8) * 64 > K:
            return None
        return bn

    # ---- the two phases -------------------------------------------------------------------
    def _d_forward_head(self, xs, groups, n_first, label_first, label_second, want_dw):
        """D forward + loss head (+ the head's own backward).  Returns (loss, resume) where `resume(**kw)` runs
        the rest of D's backward with the keyword arguments of `Network.backward`."""
        bn = self._head_bn
        if bn is None:
            feat = self.D.forward(xs, training=True, stop=-1, bn_groups=groups)
            loss, dfeat = self._d_head(feat, n_first, label_first, label_second, True, want_dw)
            return loss, lambda **kw: self.D.backward(dfeat, start=-1, **kw)
        h = self.head
        x = self.D.forward(xs, training=True, stop=-3, bn_groups=groups)
        bn.groups = groups
        scale, shift, mean, invstd = bn.fwd_stats(x)
        info = dict(scale=scale, shift=shift, mean=mean, invstd=invstd, C=bn.Cp, groups=groups, act=bn.post_act,
                    alpha=bn.post_alpha)
        loss, g, _, _, _ = F.dense1_bce_fwd_bwd(x.reshape(x.shape[0], -1), h.w_main, h.bias, label_first,
                                                label_second, n_first, True, want_dw,
                                                h.gkernel if want_dw else None, h.gbias if want_dw else None,
                                                bn=info)

        def resume(**kw):
            dx = bn.bwd_from_part(g.reshape(x.shape), info["part"], info["nb"], need_dw=kw.get("need_dw", True))
            return self.D.backward(dx, start=-3, **kw)
        return loss, resume

    def _d_head(self, feat, n_first, label_first, label_second, want_dx, want_dw):
        h = self.head
        loss, dfeat, _, _, _ = F.dense1_bce_fwd_bwd(feat, h.w_main, h.bias, label_first, label_second,
                                                    n_first, want_dx, want_dw,
                                                    h.gkernel if want_dw else None,
                                                    h.gbias if want_dw else None)
        return loss, dfeat

    def real_slot(self, like: torch.Tensor) -> torch.Tensor:
        """The [B] real-image part of the merged pass's [real | G(z1) | G(z2)] buffer, allocated for batches shaped
        like `like` (a prepared batch).  A caller that WRITES its minibatch here and passes this very tensor to
        `train_step` pays no copy per step (static-buffer idiom of a captured graph: refill between replays, see
        GraphedGANStep.set_real); any other tensor is copied in on every step."""
        B = like.shape[0]
        if self._buf3 is None or self._buf3.shape[0] != 3 * B or self._buf3.shape[1:] != like.shape[1:]:
            self._buf3 = torch.empty((3 * B,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        return self._buf3[:B]

    def _d_backward_merged(self, real):
        B = real.shape[0]
        slot = self.real_slot(real)
        if real.data_ptr() != slot.data_ptr():       # a foreign tensor: always copied (its CONTENTS may have changed
            slot.copy_(real)                         # under the same address and version counter)
        z = self.rng.uniform((2 * B, self.cfg.noise_dim), self.G.cdtype)
        self.G.forward(z, training=True, out=self._buf3[B:], bn_groups=2)
        self.G.narrow_saved(B, 2 * B, group=1)       # the G phase trains on z2 only
        self._fake = self._buf3[2 * B:]
        loss, resume = self._d_forward_head(self._buf3[:2 * B], 2, B, 1.0, 0.0, True)
        resume(need_input_grad=False, after_op=self._bucket_hook(self.D, self.optD))
        self.d_loss = loss
        return loss

    def d_backward(self, real):
        """Forward/backward of the discriminator phase; leaves averaged grads in D.G."""
        if self.merged and real.is_contiguous():
            return self._d_backward_merged(real)
        B = real.shape[0]
        z = self.rng.uniform((B, self.cfg.noise_dim), self.G.cdtype)
        g_train = self._split           # "split": the D-phase fakes see batch statistics like the G-phase ones
        if self.G.can_forward_into() and real.is_contiguous():
            # [real ; fake] assembled in place: the generator's last kernel writes straight into the
            # second half of the discriminator batch (no 2B-image concat copy)
            xs = torch.empty((2 * B,) + tuple(real.shape[1:]), dtype=real.dtype, device=real.device)
            xs[:B].copy_(real)
            self.G.forward(z, training=g_train, out=xs[B:])     # generator.predict(noise)
        else:
            fake = self.G.forward(z, training=g_train)
            xs = torch.cat([real, fake], 0)
        if g_train:
            self.G.release()
        loss, resume = self._d_forward_head(xs, 2 if self._split else 1, B, 1.0, 0.0, True)
        resume(need_input_grad=False, after_op=self._bucket_hook(self.D, self.optD))
        self.d_loss = loss
        return loss

    def g_forward(self, B):
        """Generator forward of the G phase (independent of D, so it can overlap D's all-reduce).  Already done
        by the merged pass of the D phase when `self.merged`."""
        if self.merged and self._fake is not None:
            return self._fake
        z = self.rng.uniform((B, self.cfg.noise_dim), self.G.cdtype)
        self._fake = self.G.forward(z, training=True)
        return self._fake

    def g_backward(self, B, fake=None):
        fake = self.g_forward(B) if fake is None else fake
        loss, resume = self._d_forward_head(fake, 1, B, 1.0, 1.0, False)
        # the generator's output activation backward rides in the epilogue of D's input-gradient kernel
        from .network import FUSE_ACT_BWD
        mul = Network.act_owner(self.G.ops[-1]) if (FUSE_ACT_BWD and isinstance(self.D.ops[0], ConvOp)) else None
        dfake = resume(need_input_grad=True, need_dw=False, input_mul=mul)
        self.G.backward(dfake, need_input_grad=False, dy_premultiplied=mul is not None,
                        after_op=self._bucket_hook(self.G, self.optG))
        self.g_loss = loss
        self._fake = None
        return loss

    # ---- one reference iteration = train_discriminator() then train_generator() ------------
    def phase_a(self, real):
        """D forward/backward (local gradients in D.G)."""
        return self.d_backward(real)

    def phase_b(self, B):
        """D update, then G forward / D dgrad / G backward (local gradients in G.G)."""
        if not self.fused or self.comm_off:
            self.optD.step(self._gscale)
        return self.g_backward(B)

    def phase_b1(self, B):
        """The part of phase B that does not need the averaged D gradients: G's forward."""
        return self.g_forward(B)

    def phase_b2(self, B):
        if not self.fused or self.comm_off:
            self.optD.step(self._gscale)
        return self.g_backward(B, self._fake)

    def phase_c(self):
        if not self.fused or self.comm_off:
            self.optG.step(self._gscale)
        self.D.release()
        self.G.release()

    # ---- gradient exchange ------------------------------------------------------------------
    def set_comm(self, dp):
        """Swap the communicator (start-up selection between RCCL and the xGMI kernel, parallel/dp_tune.py)."""
        self.dp = dp
        self.fused = bool(dp is not None and not self.sharded and self.G.on_gpu
                          and getattr(dp, "supports_fused_update", False))

    def _on_comm(self, fn):
        cs = self._comm_stream if not self.sharded else None
        if cs is None:
            return fn()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            return fn()

    def _join_comm(self):
        if self._comm_stream is not None and not self.sharded:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    def _bucketed(self, net) -> bool:
        return bool(self.buckets and self.dp is not None and not self.sharded and not self.comm_off
                    and len(self._bk[id(net)]) > 1)

    def _reduce_range(self, net, opt, lo, hi, first, last):
        if self.fused:
            opt.step_allreduce_fused(self.dp, lo, hi, first, last)
        else:
            self.dp.all_reduce_sum_(net.G if (lo == 0 and hi == net.G.numel()) else net.G[lo:hi])

    def _bucket_hook(self, net, opt):
        """`after_op` callback of `Network.backward`: launches every bucket but the last-finishing one on the
        communication stream the moment its gradients are complete."""
        if not self._bucketed(net):
            return None
        bk = self._bk[id(net)]
        ready = {b[2]: k for k, b in enumerate(bk[:-1])}

        def hook(i):
            k = ready.get(i)
            if k is not None:
                lo, hi, _ = bk[k]
                self._on_comm(lambda: self._reduce_range(net, opt, lo, hi, k == 0, False))
        return hook

    def _reduce_rest(self, net, opt):
        """What the hooks have not sent yet: the last bucket, or the whole buffer (one collective / one kernel)."""
        if self.dp is None or self.sharded or self.comm_off:
            return
        if self._bucketed(net):
            lo, hi, _ = self._bk[id(net)][-1]
            self._reduce_range(net, opt, lo, hi, False, True)
        else:
            self._reduce_range(net, opt, 0, net.G.numel(), True, True)

    def reduce_D(self):
        """Average D's gradients over the ranks (and, fused, update D)."""
        self._reduce_rest(self.D, self.optD)

    def reduce_G(self):
        self._reduce_rest(self.G, self.optG)

    @property
    def _gscale(self):
        return 1.0 if self.dp is None else self.dp.grad_scale

    def train_step(self, real, prepared: bool = False):
        """One iteration.  Data parallel on device: gradient buckets are reduced on a communication stream under
        the rest of the backward pass that produces them; D's last bucket additionally hides behind the
        generator's forward of the G phase when that is a pass of its own (`merged` off) -- captured into a
        hipGraph the forks / joins become graph branches."""
        if not prepared:
            real = self.D.prepare_input(real)
        B = real.shape[0]
        self.phase_a(real)
        self._on_comm(self.reduce_D)
        self.phase_b1(B)                # a no-op after the merged generator pass
        self._join_comm()
        self.phase_b2(B)
        if self._bucketed(self.G):      # G's first bucket is already on the communication stream: stay in order
            self._on_comm(self.reduce_G)
            self._join_comm()
        else:
            self.reduce_G()
        self.phase_c()
        return self.d_loss, self.g_loss

    # ---- checkpoint / resume (the reference never checkpoints its GAN) ---------
    def sync_bn_stats(self):
        """Synchronous DP exchanges gradients only, so every rank's BN moving statistics follow its own data
        shard: average them (one tiny all-reduce per network) before anything rank-dependent reads them --
        a checkpoint, sample generation, evaluation.  A collective: every rank calls it."""
        if self.dp is not None and self.dp.world > 1:
            self.dp.all_reduce_mean_(self.G.S)
            self.dp.all_reduce_mean_(self.D.S)

    def save_checkpoint(self, path, iteration: int, write: bool = True):
        """Every rank calls this (BN statistics are averaged, sharded optimizer state is gathered);
        `write=False` on all ranks but the one that owns the file."""
        from ..utils import ckpt
        self.sync_bn_stats()
        extra = {"iteration": iteration, "rng": self.rng.state_dict(), "bn_policy": self.cfg.bn_policy}
        ckpt.save_training_state(str(path) + ".G", self.G, self.optG, extra, write)
        ckpt.save_training_state(str(path) + ".D", self.D, self.optD, {"iteration": iteration}, write)
        if self.dp is not None and self.dp.world > 1:       # every rank has its own noise stream: one small file each
            torch.save(self.rng.state_dict(), f"{path}.rng{self.dp.rank}")
        if write:
            with open(path, "w") as f:
                f.write(str(iteration))

    def load_checkpoint(self, path) -> int:
        from ..utils import ckpt
        extra = ckpt.load_training_state(str(path) + ".G", self.G, self.optG)
        ckpt.load_training_state(str(path) + ".D", self.D, self.optD)
        import os
        own = f"{path}.rng{self.dp.rank}" if (self.dp is not None and self.dp.world > 1) else None
        if own and os.path.exists(own):
            self.rng.load_state_dict(torch.load(own, map_location="cpu", weights_only=True))
        elif "rng" in extra:
            self.rng.load_state_dict(extra["rng"], keep_seed=own is not None)
        return int(extra["iteration"])

    def generate(self, n: int, training_bn: Optional[bool] = None):
        """`generator.predict(noise)` (GAN_distributed.py:119-124).  Inference-phase BN (moving statistics) by
        default; `training_bn=True` normalises the sample batch with its own statistics instead."""
        z = self.rng.uniform((n, self.cfg.noise_dim), self.G.cdtype)
        y = self.G.forward(z, training=bool(training_bn))
        if training_bn:
            self.G.release()
        return self.G.to_logical(y)

    def diagnose(self, real):
        """Mean discriminator logits on the three batches a step sees -- the real half and the fake half of
        the D phase and the fake-only batch of the G phase -- computed exactly as the step computes them
        (same BN policy).  Leaves weights, BN statistics, RNG and optimizer untouched.  A healthy game has
        logit_fake ~= logit_gphase; a BN batch-composition leak shows as logit_fake << 0 << logit_gphase."""
        B = real.shape[0]
        keepS = (self.G.S.clone(), self.D.S.clone())
        keepR = self.rng.state_dict()
        if "philox" in keepR:
            keepR = {"philox": keepR["philox"].clone()}
        z = self.rng.uniform((B, self.cfg.noise_dim), self.G.cdtype)
        fake = self.G.forward(z, training=self._split)
        xs = torch.cat([real, fake], 0)
        feat = self.D.forward(xs, training=True, stop=-1, bn_groups=2 if self._split else 1)
        h = self.head
        lg = F.dense1_bce_fwd_bwd(feat, h.w_main, h.bias, 1.0, 0.0, B, False, False)[4].float()
        z = self.rng.uniform((B, self.cfg.noise_dim), self.G.cdtype)
        feat2 = self.D.forward(self.G.forward(z, training=True), training=True, stop=-1)
        lg2 = F.dense1_bce_fwd_bwd(feat2, h.w_main, h.bias, 1.0, 1.0, B, False, False)[4].float()
        out = {"logit_real": float(lg[:B].mean()), "logit_fake": float(lg[B:].mean()),
               "logit_gphase": float(lg2.mean())}
        self.G.release()
        self.D.release()
        self.G.S.copy_(keepS[0])
        self.D.S.copy_(keepS[1])
        self.rng.load_state_dict(keepR)
        return out


class GraphedGANStep:
    """The training iteration captured as hipGraphs (torch.cuda.CUDAGraph == hipGraph on ROCm).

    A DCGAN step at the reference's batch is ~100 short kernels: launch-bound when issued
    eagerly.  Shapes are static, the noise RNG and the Adam step counter live
    in device memory, so the whole iteration replays from a graph.  Single rank: one graph.
    Multi rank: three graph segments with the two flat-gradient all-reduces issued between them
    (`capture_collectives=True` captures the RCCL calls too and replays a single graph).
    """

    def __init__(self, trainer: GANTrainer, real_prepared: torch.Tensor, warmup: int = 3,
                 capture_collectives: bool = False):
        self.t, self.real = trainer, real_prepared
        self.B = real_prepared.shape[0]
        if trainer.merged and real_prepared.is_contiguous():
            # the captured step reads the real images from the trainer's own static slot: no copy inside the graph,
            # and `set_real` (refill between replays) reaches the buffer the discriminator actually reads
            slot = trainer.real_slot(real_prepared)
            slot.copy_(real_prepared)
            self.real = slot
        self.full = (trainer.dp is None or capture_collectives or trainer.sharded
                     or getattr(trainer.dp, "captures_safely", False))
        t = trainer
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):          # populates table caches / allocator before capture
                t.train_step(self.real, prepared=True)
        torch.cuda.current_stream().wait_stream(side)
        hipgraph.quiesce()
        if t.G.on_gpu:
            hip = F.hip_backend()
            hip.sync_choices(t.dp)          # every rank captures rank 0's tuned plans
            hip.save_choice_cache()
        pool = torch.cuda.graph_pool_handle()
        self.graphs = []
        if self.full:
            g, _ = hipgraph.capture(lambda: t.train_step(self.real, prepared=True), pool)
            self.graphs = [g]
        else:
            # A | all-reduce(D grads) on the comm stream || B1 (G forward) | B2 | all-reduce(G grads) | C
            # (segments cannot contain an eager collective, so the bucket hooks are off in this mode)
            t.buckets = False
            self.comm_stream = comm_stream()
            fns = [lambda: t.phase_a(self.real), lambda: t.phase_b1(self.B), lambda: t.phase_b2(self.B), t.phase_c]
            if t.merged:            # the G-phase forward is part of phase A's merged generator pass
                fns.pop(1)
            for fn in fns:
                self.graphs.append(hipgraph.capture(fn, pool)[0])

    def set_real(self, real_prepared: torch.Tensor):
        """New real minibatch for the following replays (same shape; prepared layout)."""
        if real_prepared.data_ptr() != self.real.data_ptr():
            self.real.copy_(real_prepared)

    def step(self):
        t = self.t
        if self.full:
            self.graphs[0].replay()
        else:
            main = torch.cuda.current_stream()
            self.graphs[0].replay()
            self.comm_stream.wait_stream(main)
            with torch.cuda.stream(self.comm_stream):      # D's 1 MB all-reduce hides behind G's forward
                t.reduce_D()
            k = 1
            if not t.merged:
                self.graphs[1].replay()
                k = 2
            main.wait_stream(self.comm_stream)
            self.graphs[k].replay()
            t.reduce_G()
            self.graphs[k + 1].replay()
        return t.d_loss, t.g_loss
