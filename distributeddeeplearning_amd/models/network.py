"""Static-graph executor: explicit forward/backward over a `ModelSpec` chain.

Design (MI355X-first, not a Keras clone):
  * one flat fp32 master buffer per network for all trainable parameters, one flat gradient
    buffer of the same layout, one flat buffer for BN moving statistics -> a single RCCL
    collective per network and a single fused multi-tensor optimizer launch;
  * on device, a flat bf16 working copy (`Pc`, same offsets) plus an alternate-layout copy
    (`Palt`) for the transposed operand the dgrad kernels want; both are refreshed by the fused
    optimizer / relayout kernels, never by host code;
  * activations NHWC, bf16 on device / fp32 on CPU; on device every channel dimension is
    physically padded to a multiple of 8 with zero lanes (16-byte vector access everywhere, MFMA
    friendly reduction dims, odd channel counts of the random search space need no slow path);
    parameters are stored padded the same way and stripped on export;
  * shapes are static so a whole training step can be captured in a hipGraph;
  * no autograd: every op has a hand-written backward that calls the HIP kernels.

Replaces what Keras' `Model`/`fit` machinery did for the reference (/root/reference/DDL.py:58-96,
/root/reference/GAN_distributed.py:15-117).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ..ops import functional as F
from ..ops.geometry import ConvGeom, dense_geom
from .ir import LayerSpec, ModelSpec

ONECH = os.environ.get("DDL_ONECH", "1") == "1"      # single-channel conv fast path (device)
BN_BWD_FUSE = os.environ.get("DDL_BN_BWD_FUSE", "1") == "1"       # BN backward column sums from the kernel that makes dy
BIAS_IN_WGRAD = os.environ.get("DDL_BIAS_IN_WGRAD", "1") == "1"   # one-channel input conv: bias gradient out of the weight-gradient launch
BN_ON_LOAD_PCONV = os.environ.get("DDL_BN_ON_LOAD_PCONV", "1") == "1"   # ... in front of a multi-channel transposed conv
BN_ON_LOAD = os.environ.get("DDL_BN_ON_LOAD", "1") == "1"         # BN in front of a 1-channel ConvT applied on load
FUSE_ACT_BWD = os.environ.get("DDL_FUSE_ACT_BWD", "1") == "1"   # act'(y) of the layer below in the dgrad epilogue
STREAM1 = os.environ.get("DDL_STREAM1", "1") == "1"     # Conv2D 1 -> N with k <= 3: streaming kernels (csrc/hip/upconv_1ch.hip)
STREAM1_MIN_ELEMS = int(os.environ.get("DDL_STREAM1_MIN_ELEMS", str(4 << 20)))
BN_BWD_ON_LOAD = os.environ.get("DDL_BN_BWD_ON_LOAD", "1") == "1"   # first layer's wgrad applies the BatchNorm backward on load
DW_BN_FUSE = os.environ.get("DDL_DW_BN_FUSE", "1") == "1"     # depthwise dgrad + backward of the BatchNorm in front of it
BRANCH = os.environ.get("DDL_BRANCH", "1") == "1"    # weight-gradient branch on a side stream (captured steps)
ALIGN = 8      # every tensor starts at a multiple of 8 elements (16 B in bf16, 32 B in fp32)


class SideBranch:
    """Weight-gradient branch of the backward pass on a second HIP stream.

    Given dY, a layer's weight/bias gradient and its input gradient are independent; neither fills
    256 CUs at the reference's batch sizes (persistent wgrad workers: 256-1024 blocks, dgrad tiles of a
    few hundred blocks), so inside a captured step the wgrad chain (wgrad -> slab reduce -> bias
    column sums) is forked onto a side stream and becomes a parallel branch of the hipGraph, joined
    once at the end of `Network.backward`.  Eager launches stay single-stream: plan autotuning times
    kernels undisturbed and the eager path remains the bit-exact oracle for the captured one
    (every reduction order is fixed, so both produce identical bits).

    Cross-stream lifetime: tensors produced on the main stream and read by the branch (dY) are kept
    referenced until the join, so the caching allocator cannot hand their blocks to a later
    main-stream allocation while the branch still reads them.  Tensors the branch allocates itself
    belong to the side stream and are only ever reused there.

    Opt-in per network (`Network.branch_wgrad`), from measurements on one MI355X: the 5-replica
    classifier step (B=512, many small launches) gains 2 % (6.31 -> 6.19 ms); the DCGAN steps do not
    (28x28: 1.871 -> 1.898 ms, 64x64: unchanged) -- their wgrad workers and patch-resident dgrad
    blocks each claim most of a CU's LDS, so the branches cannot co-reside and the extra
    cross-stream dependencies only add latency."""

    def __init__(self):
        self.stream = None
        self.active = False
        self.enabled = False        # set per backward pass from Network.branch_wgrad
        self.keep = []

    def on(self) -> bool:
        return (BRANCH and self.enabled and self.stream is not None
                and torch.cuda.is_current_stream_capturing())

    def ensure_stream(self):
        if BRANCH and self.stream is None:
            self.stream = torch.cuda.Stream()

    def run(self, fn, *keep):
        """Run `fn()` on the branch (or inline when branching is off)."""
        if not self.on():
            fn()
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        self.keep.extend(keep)
        self.active = True
        with torch.cuda.stream(self.stream):
            fn()

    def join(self):
        if self.active:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.active = False
        self.keep.clear()


_BR = SideBranch()


def _round_up(n, a):
    return (n + a - 1) // a * a


class _Param:
    """One tensor inside a flat buffer.  `shape` is the physical (padded) shape, `lshape` the
    logical one (same rank): logical data lives in the leading corner, the rest is zero."""
    __slots__ = ("layer", "wname", "shape", "lshape", "numel", "offset", "trainable", "keras_shape")

    def __init__(self, layer, wname, shape, lshape, trainable, keras_shape):
        self.layer, self.wname, self.trainable = layer, wname, trainable
        self.shape, self.lshape, self.keras_shape = tuple(shape), tuple(lshape), tuple(keras_shape)
        self.numel = int(np.prod(shape))
        self.offset = -1

    def corner(self):
        return tuple(slice(0, n) for n in self.lshape)


# =============================================================================== exec ops
class Op:
    name = "op"
    has_params = False

    def fwd(self, x, training):
        raise NotImplementedError

    def bwd(self, dy, need_dx=True, need_dw=True):
        raise NotImplementedError

    def release(self):
        pass


# The streaming 5x5 one-channel weight gradient (ops/hip.py stream1_wgrad) from this many elements of the N-channel
# operand up.  Below it the taps tensor + split-K GEMM path is as fast or faster (dcgan28, 6-12 M elements: 1.120 vs
# 1.124 ms per step); at dcgan64's 2-4 G elements the streaming kernel saves the 8.6 GB taps tensor and 1 ms per step.
S1_WGRAD_MIN_ELEMS = int(os.environ.get("DDL_S1_WGRAD_MIN_ELEMS", str(1 << 27)))
_NO = object()        # sentinel: "an affine exists but cannot be applied on load"


class ConvOp(Op):
    """Conv2D / Conv2DTranspose / DepthwiseConv2D / Dense with fused bias + activation."""
    has_params = True

    def __init__(self, spec: LayerSpec, kind: str, g: ConvGeom, act: str, alpha: float, cpad: int,
                 kstruct: Optional[Tuple[int, int, int]] = None):
        self.spec, self.kind, self.g, self.act, self.alpha = spec, kind, g, act, alpha
        self.name = spec.name
        self.cpad = cpad
        self.kstruct = kstruct          # Dense fed by Flatten of (H, W, C): K is structured
        self.act_bwd_fused = False      # a following BN op applies act'(y) for us
        self.next_bn = None             # BN op fed directly by this layer (if any)
        self.kernel = self.bias = None            # fp32 master views
        self.gkernel = self.gbias = None          # fp32 gradient views
        self.w_main = self.w_alt = None           # compute-dtype views
        self.x = self.y = self.x_img = None
        # single-channel fast path (device only): see ops/hip.py im2col_1ch / col2im_1ch
        self.onech = None               # None | "in" (Conv2D 1->N) | "out" (Conv2DTranspose C->1) | "up" (ConvT 1->N)
        self.g1 = None                  # the equivalent 1x1 geometry on the taps-as-channels tensor
        self.w_tap = self.w_tapT = None
        self.dy_premul = False          # the incoming dY already carries act'(y) (fused by the layer above)
        self._gbias_is_zero = False     # gbias currently holds the exact-zero gradient written by _bias_grad
        self.out_buffer = None          # optional destination of the next forward's output ("out" path only)
        # BatchNorm applied on load ("out" path): the BN in front of this layer hands over (scale, shift, groups,
        # c_true) instead of writing its output; the taps kernel and the weight-gradient kernel read x through it
        self._slabs = {}                # persistent zeroed split-K slabs of the boxed weight gradient ("in" path)
        self.in_bn = None               # the BatchNorm whose apply this layer performs on load (BNOp.defer_to is us)
        self.in_affine = None           # set by BNOp.fwd for the next forward only
        self.x_affine = None            # what the last training forward used (for backward)
        self.bn_deferred = None         # "up" path: the BatchNorm behind us left its backward apply step to our wgrad
        self.stream1 = False            # "in" path, kernel <= 3 x 3: the streaming kernels of the "up" path, forward direction
        self.stream_used = False        # ... and the last training forward took them

    def _p(self, c):
        return _round_up(c, self.cpad)

    def in_channels(self):       # logical, layer view
        return self.g.O if self.kind == "conv2d_transpose" else self.g.I

    def out_channels(self):
        return self.g.I if self.kind == "conv2d_transpose" else self.g.O

    def shapes(self):
        """(physical, logical) canonical kernel shapes:
        [layer_out][KH][KW][layer_in] ; Dense [N][K] (or [N][H][W][C]) ; depthwise [KH][KW][C]."""
        g, p = self.g, self._p
        cin, cout = self.in_channels(), self.out_channels()
        if self.kind == "dense":
            if self.kstruct is not None:
                H, W, Cc = self.kstruct
                return (cout, H, W, p(Cc)), (cout, H, W, Cc)
            return (cout, p(cin)), (cout, cin)
        if self.kind == "depthwise_conv2d":
            return (g.KH, g.KW, p(cin)), (g.KH, g.KW, cin)
        return (cout, g.KH, g.KW, p(cin)), (cout, g.KH, g.KW, cin)

    def relayout_dims(self):
        """(R, T, C, Rp) for canonical [R][T][C] -> alternate [C][T][Rp]."""
        phys, _ = self.shapes()
        R = phys[0]
        Cc = phys[-1] if self.kstruct is None else int(np.prod(phys[1:]))
        T = 1 if self.kind == "dense" else self.g.KH * self.g.KW
        return R, T, Cc, self._p(R)

    def alt_shape(self):
        R, T, Cc, Rp = self.relayout_dims()
        if self.kind == "dense":
            return (Cc, Rp)
        return (Cc, self.g.KH, self.g.KW, Rp)

    def _w4(self, w, rows_first=True):
        """Dense weights as a 1x1 conv kernel."""
        return w.reshape(w.shape[0], 1, 1, -1)

    def _bias_grad(self, dy):
        """Column sum of dY -- except straight in front of a batch-statistics BatchNorm, where the
        bias gradient is identically zero (BN subtracts the batch mean): no reduction pass at all."""
        bn = self.next_bn
        if bn is not None and bn.used_batch_stats and self.act in (None, "linear"):
            if not self._gbias_is_zero:         # only this method ever writes the slot: zero it once, not per step
                self.gbias.zero_()
                self._gbias_is_zero = True
            return
        if bn is not None and bn.bias_grad_for is self:
            return                      # already written by the BN's backward reduction (F.bn_bwd out_dbias)
        self._gbias_is_zero = False
        F.bias_grad(dy, out=self.gbias)

    def _bias_plain(self):
        """True when the bias gradient is a plain sum of dY (not zero by construction in front of a batch-statistics
        BatchNorm, not written by that BatchNorm's backward): a weight-gradient kernel may produce it."""
        bn = self.next_bn
        if self.gbias is None or self.gbias.dtype != torch.float32 or not self.gbias.is_contiguous():
            return False
        return not (bn is not None and ((bn.used_batch_stats and self.act in (None, "linear")) or bn.bias_grad_for is self))

    def _bias_tail(self, T, nbias=None):
        """(offset, flat view over [kernel gradient | alignment gap | bias gradient]) when this layer's bias gradient
        is a plain column sum of dY and can be produced by the weight-gradient launch itself, else None."""
        bn = self.next_bn
        if not BIAS_IN_WGRAD or T % 8 == 0 or self.gbias is None or self.gkernel is None:
            return None
        if bn is not None and ((bn.used_batch_stats and self.act in (None, "linear")) or bn.bias_grad_for is self):
            return None                 # zero by construction / written by the BatchNorm's backward (_bias_grad)
        gk, gb = self.gkernel, self.gbias
        if gk.untyped_storage().data_ptr() != gb.untyped_storage().data_ptr() or not gk.is_contiguous():
            return None
        off = gb.storage_offset() - gk.storage_offset()
        # the launch zero-fills whatever lies between the kernel and the bias gradient: that may only ever be the flat
        # buffer's alignment padding (< ALIGN elements), never another parameter's gradient
        if gk.dtype != torch.float32 or gb.dtype != torch.float32:
            return None
        if off < gk.numel() or off - gk.numel() >= ALIGN or gb.numel() != (self.g.O if nbias is None else nbias):
            return None
        # every parameter starts at a multiple of ALIGN elements: what follows the bias up to the next multiple of 4 is
        # alignment padding of the flat buffer (it receives zeros; the wide slab reduce wants float4 columns)
        total = _round_up(off + gb.numel(), 4)
        if ALIGN % 4 or gk.storage_offset() + total > gk.untyped_storage().nbytes() // gk.element_size():
            return None
        return off, torch.as_strided(gk, (total,), (1,))

    def _centre_tap(self):
        """Index of the tap whose im2col column IS the image (stride 1, the tap that reads pixel (h, w) for output
        (h, w)), or None: its pixel sum is the sum of the image."""
        g = self.g
        if g.sh != 1 or g.sw != 1 or g.OH != g.H or g.OW != g.W or g.pt % g.dh or g.pl % g.dw:
            return None
        kh, kw = g.pt // g.dh, g.pl // g.dw
        return kh * g.KW + kw if (kh < g.KH and kw < g.KW) else None

    def _fwd_onech(self, x, training):
        g, g1, hip = self.g, self.g1, F.hip_backend()
        if self.onech == "in":          # direct conv from the 1-channel image (or im2col -> 1x1 conv)
            xc = None
            y = hip.tapconv_1ch(x, self.w_tap, g, self.bias, self.act, self.alpha)
            if y is None:
                xc = hip.im2col_1ch(x, g)
                y = hip.conv_fprop(xc, self.w_tap, self.bias, g1, self.act, self.alpha)
            if training:                # the taps tensor is only the weight gradient's operand: built there if needed
                self.x, self.x_img, self.y = xc, x, y
        else:                           # 1x1 conv to taps -> col2im fold with bias + activation (one fused kernel)
            aff, self.in_affine = self.in_affine, None
            y = hip.tapfold_1ch(x, self.w_main.reshape(g1.O, -1), g, self.bias, self.act, self.alpha,
                                out=self.out_buffer, in_affine=aff)
            if y is None:
                if aff is not None:     # outside the fused kernel's range: materialise the BN output after all
                    x = hip.bn_apply(x, aff[0], aff[1], None, 0.3, aff[3], aff[2])
                    aff = None
                z = hip.conv_fprop(x, self.w_main.reshape(g1.O, 1, 1, -1), None, g1)
                y = hip.col2im_1ch(z, g, self.bias, self.act, self.alpha, out=self.out_buffer)
            if training:
                self.x, self.y, self.x_affine = x, y, aff
        return y

    def _bwd_onech(self, dy, need_dx, need_dw, mul=None):
        g, g1, hip = self.g, self.g1, F.hip_backend()
        T = g.KH * g.KW
        dx = None
        if self.onech == "in":
            if need_dw:                 # dW[n][t] -> canonical [O][KH][KW][8] slot (n, t, channel 0)
                def wg():
                    # 4x4 / 5x5: the streaming kernel reads dY once against the image in LDS (bias gradient included)
                    if (self.x is None and self.x_img is not None and self.gkernel.dtype == torch.float32
                            and dy.numel() >= S1_WGRAD_MIN_ELEMS):
                        plain = self._bias_plain()
                        if hip.stream1_wgrad(self.x_img, dy, g, self.gkernel.reshape(-1), T * 8, 8, zfill=8, n_out=g.O,
                                             bias=self.gbias if plain else None, bias_mode=1 if plain else 0):
                            if plain:
                                self._gbias_is_zero = False
                            else:
                                self._bias_grad(dy)
                            return
                    # The bias gradient (column sums of dY) rides in the same MFMAs: the taps tensor gets a column of
                    # ones in its first padding tap and the kernel routes that column of dW to the bias slot, which
                    # sits behind the kernel's in the flat gradient buffer (one launch covers both).
                    tail = self._bias_tail(T) if self.x is None else None
                    if self.x is None:
                        self.x = hip.im2col_1ch(self.x_img, g, ones=tail is not None)
                    if tail is not None:
                        off, flat = tail
                        hip.conv_wgrad(self.x, dy, g1, "A", out=flat, box=(g.O, T, T * 8, 0, 8, flat.numel()),
                                       slab_holder=self._slabs, bias_tail=(T, off))
                        self._gbias_is_zero = False
                    else:
                        hip.conv_wgrad(self.x, dy, g1, "A", out=self.gkernel,
                                       box=(g.O, T, T * 8, 0, 8, self.gkernel.numel()), slab_holder=self._slabs)
                        self._bias_grad(dy)
                _BR.run(wg, dy)
            if need_dx:
                dx = hip.tapfold_1ch(dy, self.w_tapT.reshape(T, -1), g, mul=mul)
                if dx is None:
                    dx = hip.col2im_1ch(hip.conv_dgrad(dy, self.w_tapT, g1), g, mul=mul)
        else:
            dyc = [None]                # the taps tensor of dY: the weight gradient's operand (side branch)
            if need_dx:                 # input gradient straight from the dY image when nothing is fused behind it
                red, bn = None, self.in_bn
                if (mul is None and BN_BWD_FUSE and bn is not None and bn.used_batch_stats and bn.groups_used == 1
                        and bn.x is not None and bn.defer_to is self and self.x_affine is not None):
                    # dx is the gradient at that BatchNorm's output: the kernel also emits its backward column sums
                    sums = 5 if (need_dw and bn.bias_grad_for is not None) else 2
                    red = dict(x=bn.x, mean=bn.mean.reshape(-1), invstd=bn.invstd.reshape(-1), sums=sums,
                               prev_act=bn.prev_act, prev_alpha=bn.prev_alpha)
                dx = hip.tapconv_1ch(dy, self.w_tapT, g, bn_red=red) if mul is None else None
                if dx is not None and red is not None and "part" in red:
                    bn.pre_bwd = (red["part"], red["nb"], red["sums"])
                if dx is None:
                    dyc[0] = hip.im2col_1ch(dy, g)
                    dx = hip.conv_dgrad(dyc[0], self.w_tapT, g1, mul=mul)
            if need_dw:
                x, xa, xa_any = self.x, None, None
                if self.x_affine is not None:
                    sc, sh, groups, ctrue = self.x_affine
                    if groups == 1:
                        xa = xa_any = (sc.reshape(-1), sh.reshape(-1), ctrue)
                    else:               # several statistics groups in one backward: materialise
                        x = hip.bn_apply(x, sc, sh, None, 0.3, ctrue, groups)
                        xa_any = _NO

                def wg():
                    if (dyc[0] is None and xa_any is not _NO and self.gkernel.dtype == torch.float32
                            and self.x.numel() >= S1_WGRAD_MIN_ELEMS):
                        # 4x4 / 5x5: streaming kernel, the dY image in LDS, x read once with its BatchNorm on load
                        plain = self._bias_plain()
                        Ip = self.x.shape[-1]
                        if hip.stream1_wgrad(dy, self.x, g, self.gkernel.reshape(-1), 1, Ip, affine=xa_any,
                                             bias=self.gbias if plain else None, bias_mode=2 if plain else 0):
                            if plain:
                                self._gbias_is_zero = False
                            else:
                                self._bias_grad(dy)
                            return
                    d = dyc[0] if dyc[0] is not None else hip.im2col_1ch(dy, g)
                    # the scalar bias gradient sum(dY) = the pixel sum of d's centre tap: one more MFMA in this launch
                    ctr = self._centre_tap()
                    tail = self._bias_tail(1, nbias=1) if ctr is not None else None
                    if tail is not None:
                        off, flat = tail
                        Ip = x.shape[-1]
                        hip.conv_wgrad(x, d, g1, "A", out=flat, x_affine=xa, slab_holder=self._slabs,
                                       box=(g1.O, Ip, Ip, Ip, 1, flat.numel()), bias_tail=("row", ctr, off))
                        self._gbias_is_zero = False
                    else:
                        hip.conv_wgrad(x, d, g1, "A", out=self.gkernel, x_affine=xa)
                        self._bias_grad(dy)
                _BR.run(wg, dy)
        return dx

    def _fwd_up(self, x, training):
        """Conv2DTranspose from a one-channel image (the first layer of a `random_layer` model,
        /root/reference/DDL_Normal4.py:39-43): the streaming kernel of csrc/hip/upconv_1ch.hip, which also leaves the
        statistics partials of a batch-statistics BatchNorm behind it.  None: outside its range (generic path)."""
        bn = self.next_bn
        self.stream_used = False
        if self.in_affine is not None or (not training and bn is not None):
            return None                 # (inference folds the BatchNorm into the generic kernels' epilogue)
        # forward direction: from ~4 M output elements up (batch-32 ensemble members keep the taps kernels: at 0.8 M
        # elements the streaming pair was 2 % slower per 32-step graph, at 12.8 M -- batch 512 -- it wins)
        if self.onech == "in" and x.shape[0] * self.g.OH * self.g.OW * _round_up(self.g.O, 8) < STREAM1_MIN_ELEMS:
            return None
        so = {"groups": bn.groups} if (training and bn is not None and bn.mode == "batch") else None
        y = F.hip_backend().upconv_1ch(x, self.w_main, self.g, self.bias, self.act, self.alpha, stats_out=so,
                                       fwdconv=self.onech == "in")
        if y is None:
            return None
        if so and "part" in so and (bn.groups == 1 or x.shape[0] % (so["img"] * bn.groups) == 0):
            bn.pre_stats = (so["part"], so["nb"])
        self.stream_used = True
        if training:
            self.x, self.y, self.x_affine, self.x_img = x, y, None, x
            if self.onech == "in":
                self.x = None           # (the "in" path's taps tensor: built by its weight gradient if that runs)
        return y

    def fwd(self, x, training):
        g = self.g
        if self.onech == "up" or self.stream1:
            y = self._fwd_up(x, training)
            if y is not None:
                return y
        if self.onech and self.onech != "up":
            return self._fwd_onech(x, training)
        # a batch-statistics BN right behind this layer: let the conv kernel emit the statistics' partial sums
        bn = self.next_bn
        so = {"groups": bn.groups} if (training and bn is not None and bn.mode == "batch" and x.is_cuda) else None
        # pure inference with a plain BN behind: the conv epilogue can apply the BN's affine itself
        bi = None
        if not training and bn is not None and bn.post_act in (None, "linear") and x.is_cuda:
            so, bi = {}, (bn.gamma, bn.beta, bn.moving_mean, bn.moving_var, bn.eps)
        aff, self.in_affine = self.in_affine, None
        sink = {} if aff is not None else None      # filled when the dispatch had to materialise the BatchNorm output
        if self.kind == "dense":
            B = x.shape[0]
            x4 = x.reshape(B, 1, 1, -1)
            y = F.conv_fprop(x4, self._w4(self.w_main), self.bias, g, self.act, self.alpha, stats_out=so,
                             bn_infer=bi).reshape(B, -1)
        elif self.kind == "conv2d_transpose":
            y = F.conv_dgrad(x, self.w_main, g, self.bias, self.act, self.alpha, stats_out=so, bn_infer=bi,
                             in_affine=aff, aff_sink=sink)
        else:
            y = F.conv_fprop(x, self.w_main, self.bias, g, self.act, self.alpha, stats_out=so, bn_infer=bi,
                             in_affine=aff, aff_sink=sink)
        if sink:                        # keep the materialised tensor: the weight gradient reads it as it is
            x, aff = sink["x"], None
        if so and "part" in so:
            # per-group statistics need every producer block inside one batch group
            if bn.groups == 1 or (x.shape[0] % (so["img"] * bn.groups) == 0):
                bn.pre_stats = (so["part"], so["nb"])
        elif so and so.get("affine"):
            bn.applied_by_producer = True
        if training:
            self.x, self.y, self.x_affine = x, y, aff
        return y

    def bwd(self, dy, need_dx=True, need_dw=True, mul=None):
        """`mul` = (y, act, alpha) of the activation-carrying layer below: dx leaves already multiplied by
        act'(y) (epilogue of the input-gradient kernel), and that layer skips its own act_bwd pass."""
        g = self.g
        if self.act not in (None, "linear") and not self.act_bwd_fused:
            if self.dy_premul:
                self.dy_premul = False
            else:
                dy = F.act_bwd(dy, self.y, self.act, self.alpha)
        x = self.x
        dx = None
        if self.onech == "up" or (self.stream1 and self.stream_used):
            bnd, self.bn_deferred = self.bn_deferred, None
            if need_dw and not need_dx and self.x_affine is None:
                done = [False]

                def wg_up():
                    done[0] = F.hip_backend().upconv_1ch_wgrad(self.x_img if self.onech == "in" else x, dy, self.w_main,
                                                               g, self.gkernel, bn=bnd, fwdconv=self.onech == "in")
                    if done[0]:
                        self._bias_grad(dy)
                _BR.run(wg_up, dy)
                if done[0]:
                    return None
            if bnd is not None:         # (not reached with the shipped shapes) the generic kernels want the real dx
                dy, _, _ = F.hip_backend().bn_bwd(dy, bnd["x"], None, bnd["mean"], bnd["invstd"], bnd["gamma"], None, 0.3,
                                                  True, bnd["act"], bnd["alpha"])
        if self.onech and self.onech != "up":
            return self._bwd_onech(dy, need_dx, need_dw, mul)
        if self.kind == "dense":
            B = x.shape[0]
            dy4, x4 = dy.reshape(B, 1, 1, -1), x.reshape(B, 1, 1, -1)
            if need_dw:
                def wg():
                    # the bias gradient (column sums of dY) as row sums of the weight gradient's dY operand: one more
                    # MFMA per sub-tile against a fragment of ones, no reduction pass of its own
                    tail = self._bias_tail(1) if dy.is_cuda else None
                    if tail is not None:
                        off, flat = tail
                        Ip = x4.shape[-1]
                        F.hip_backend().conv_wgrad(x4, dy4, g, "A", out=flat, slab_holder=self._slabs,
                                                   box=(g.O, Ip, Ip, Ip, 1, flat.numel()), bias_tail=("rows", off))
                        self._gbias_is_zero = False
                    else:
                        F.conv_wgrad(x4, dy4, g, "A", out=self._w4(self.gkernel))
                        self._bias_grad(dy)
                _BR.run(wg, dy)
            if need_dx:
                dx = F.conv_dgrad(dy4, self._w4(self.w_alt), g, mul=mul).reshape(x.shape)
        elif self.kind == "conv2d_transpose":
            # core X = our output (dy), core Y = our input (x)
            if need_dw:
                def wg():
                    xx = x
                    if self.x_affine is not None:       # the BatchNorm output was never written
                        sc, sh, groups, ctrue = self.x_affine
                        hip = F.hip_backend()
                        if groups == 1 and hip._is_1x1(g):
                            # a 1x1 layer is the same GEMM seen from either side: the streaming weight-gradient kernel
                            # takes OUR input as its X operand and applies the BatchNorm as it stages it
                            gs = ConvGeom(g.OH, g.OW, g.O, g.H, g.W, g.I, 1, 1).check()
                            hip.conv_wgrad(x, dy, gs, "A", out=self.gkernel,
                                           x_affine=(sc.reshape(-1), sh.reshape(-1), ctrue))
                            self._bias_grad(dy)
                            return
                        if groups == 1:
                            # our input is the core problem's dY operand: the persistent kernel applies the BatchNorm
                            # as it stages the tile (no 12 MB .. 8 GB tensor rebuilt just to be read once)
                            hip.conv_wgrad(dy, x, g, "B", out=self.gkernel,
                                           dy_affine=(sc.reshape(-1), sh.reshape(-1), ctrue))
                            self._bias_grad(dy)
                            return
                        xx = hip.bn_apply(x, sc, sh, None, 0.3, ctrue, groups)      # rebuild it (on the wgrad branch)
                    F.conv_wgrad(dy, xx, g, "B", out=self.gkernel)
                    self._bias_grad(dy)
                _BR.run(wg, dy)
            if need_dx:
                dx = F.conv_fprop(dy, self.w_alt, None, g, mul=mul)
        else:
            # depthwise behind a batch-statistics BatchNorm: input gradient + that BatchNorm's backward as two fused
            # passes -- issued BEFORE the weight gradient: the data-gradient chain is the critical path of the step and
            # the weight gradient only has to be done by the optimizer (3.48 -> 3.40 ms per classifier step)
            if need_dx and g.depthwise and mul is None:
                dx = self._dw_bn_bwd(dy, need_dw)
            if need_dw:
                def wg():
                    xx, xa = x, None
                    if self.x_affine is not None:       # depthwise behind a BatchNorm applied on load
                        sc, sh, groups, ctrue = self.x_affine
                        if g.depthwise and groups == 1:
                            xa = (sc.reshape(-1), sh.reshape(-1), ctrue)
                        else:
                            xx = F.hip_backend().bn_apply(x, sc, sh, None, 0.3, ctrue, groups)
                    F.conv_wgrad(xx, dy, g, "A", out=self.gkernel, x_affine=xa)
                    self._bias_grad(dy)
                _BR.run(wg, dy)
            if need_dx and dx is None:
                dx = F.conv_dgrad(dy, self.w_main if g.depthwise else self.w_alt, g, mul=mul)
        return dx

    def _dw_bn_bwd(self, dy, need_dw):
        """Depthwise layer behind a batch-statistics BatchNorm (every `random_layer` of DDL_Normal4.py:52 is followed
        by one): the input gradient IS that BatchNorm's dY, so both backward steps run as two fused passes that
        recompute it from the small dY instead of writing and re-reading it (ops/hip.py dw_dgrad_bn_bwd).  Returns the
        gradient at the BatchNorm's input and marks the BatchNorm's backward as done, or None (unfused path)."""
        bn = self.in_bn
        if not (DW_BN_FUSE and dy.is_cuda and need_dw and bn is not None and bn.defer_to is self
                and bn.used_batch_stats and bn.groups_used == 1 and bn.x is not None and bn.rank4
                and bn.post_act in (None, "linear") and bn.pre_bwd is None):
            return None
        # the fused passes compute the stencil twice: worth it while a pixel has few live taps (stride 2: 2.25 of a
        # 3x3), not for a stride-1 3x3 (nine taps, two loads each: 236 us fused vs 140 us in three passes, measured on
        # model0's 28 x 28 x 128 layer at batch 512)
        g = self.g
        if g.KH * g.KW > 2.5 * g.sh * g.sw:
            return None
        conv = bn.bias_grad_for
        dx = F.hip_backend().dw_dgrad_bn_bwd(dy, self.w_main, self.g, bn.x, bn.mean.reshape(-1), bn.invstd.reshape(-1),
                                             bn.gamma, bn.prev_act, bn.prev_alpha, bn.ggamma, bn.gbeta,
                                             conv.gbias if conv is not None else None)
        if dx is not None:
            bn.bwd_done = True
        return dx

    def release(self):
        self.x = self.y = self.x_affine = self.x_img = None


class BNOp(Op):
    """BatchNormalization (+ fused following activation; + fused act' of the preceding layer)."""
    has_params = True

    def __init__(self, spec: LayerSpec, channels: int, cphys: int, rank4: bool):
        self.spec, self.name, self.C, self.Cp, self.rank4 = spec, spec.name, channels, cphys, rank4
        self.momentum, self.eps = spec.cfg["momentum"], spec.cfg["epsilon"]
        self.post_act, self.post_alpha = None, 0.3
        self.prev_act, self.prev_alpha = None, 0.3
        self.mode = "batch"        # "batch" | "inference" 
        self.gamma = self.beta = self.ggamma = self.gbeta = None
        self.moving_mean = self.moving_var = None
        self.x = self.y = self.mean = self.invstd = None
        self.used_batch_stats = False
        self.bias_grad_for = None      # ConvOp (with a fused activation) whose bias gradient our backward emits
        self.pre_stats = None          # (partials, n_blocks) left by the producing conv kernel for the next fwd
        self.applied_by_producer = False   # inference forward: the producing conv already applied this BN
        self.groups = 1                # batch groups with their own batch statistics (set per forward pass)
        self.groups_used = 1
        self.defer_to = None           # ConvOp that applies this BN's affine on load (no output tensor of our own)
        self.fp32_handoff = False      # ... and does so in fp32 (depthwise tap sums): no bf16 rounding of our output
        self.pre_bwd = None            # (partials, blocks, sums) of our backward reduction, left by the kernel that made dy
        self.bwd_done = False          # the layer above already ran our backward (ConvOp._dw_bn_bwd): dy IS our dx
        self.defer_apply_to = None     # first-layer ConvOp ("up") whose weight gradient applies our backward on load
        self.below_needs_dx = False    # set per backward pass: somebody below that layer wants an input gradient

    def fwd(self, x, training):
        use_batch = training and self.mode == "batch"
        if not training and self.applied_by_producer:
            self.applied_by_producer = False
            return x
        if not training:            # pure inference: one fused launch, nothing kept
            return F.bn_apply_infer(x, self.gamma, self.beta, self.moving_mean, self.moving_var, self.eps,
                                    self.post_act, self.post_alpha, self.C)
        pre, self.pre_stats = self.pre_stats, None
        groups = self.groups if use_batch else 1
        if use_batch and pre is not None:
            scale, shift, mean, invstd = F.bn_train_affine_from_part(
                pre[0], pre[1], x.numel() // x.shape[-1], self.gamma, self.beta, self.moving_mean,
                self.moving_var, self.momentum, self.eps, self.rank4, True, groups)
        elif use_batch:
            scale, shift, mean, invstd = F.bn_train_affine(x, self.gamma, self.beta, self.moving_mean,
                                                           self.moving_var, self.momentum, self.eps,
                                                           self.rank4, True, groups)
        else:
            mean = self.moving_mean
            scale, shift, invstd = F.bn_infer_affine(self.gamma, self.beta, self.moving_mean, self.moving_var,
                                                     self.eps)
        if self.defer_to is not None and BN_ON_LOAD:
            self.defer_to.in_affine = (scale, shift, groups, self.C)
            y = x                       # the consumer reads x through (scale, shift); backward needs x only
        else:
            y = F.bn_apply(x, scale, shift, self.post_act, self.post_alpha, self.C, groups)
        if training:
            self.x, self.y, self.mean, self.invstd = x, y, mean, invstd
            self.used_batch_stats = use_batch
            self.groups_used = groups
        return y

    def fwd_stats(self, x):
        """Training forward WITHOUT the apply pass: batch statistics, moving-average update and the per-channel
        (scale, shift) a fused consumer applies on load (the GAN trainer's discriminator head).  Saves what
        `bwd_from_part` needs.  Device, batch-statistics mode only."""
        pre, self.pre_stats = self.pre_stats, None
        groups = self.groups
        if pre is not None:
            scale, shift, mean, invstd = F.bn_train_affine_from_part(
                pre[0], pre[1], x.numel() // x.shape[-1], self.gamma, self.beta, self.moving_mean,
                self.moving_var, self.momentum, self.eps, self.rank4, True, groups)
        else:
            scale, shift, mean, invstd = F.bn_train_affine(x, self.gamma, self.beta, self.moving_mean,
                                                           self.moving_var, self.momentum, self.eps,
                                                           self.rank4, True, groups)
        self.x, self.y, self.mean, self.invstd = x, None, mean, invstd
        self.used_batch_stats, self.groups_used = True, groups
        return scale, shift, mean, invstd

    def bwd_from_part(self, g, part, part_blocks, need_dw=True):
        """Backward when the consumer already produced g = dL/d(BN output before its activation) together with
        the column sums of (g, g * xhat): finalize + apply only."""
        hip = F.hip_backend()
        dx, _, _ = hip.bn_bwd(g, self.x, None, self.mean, self.invstd, self.gamma, None, 0.3, True,
                              self.prev_act, self.prev_alpha, self.ggamma if need_dw else None,
                              self.gbeta if need_dw else None, None, self.groups_used, part=part,
                              part_blocks=part_blocks)
        return dx

    def bwd(self, dy, need_dx=True, need_dw=True):
        if self.bwd_done:
            self.bwd_done = False
            return dy
        conv = self.bias_grad_for
        pre, self.pre_bwd = self.pre_bwd, None
        tgt = self.defer_apply_to
        if (tgt is not None and tgt.stream_used and BN_BWD_ON_LOAD and need_dw and not self.below_needs_dx
                and pre is None and dy.is_cuda
                and self.used_batch_stats and self.groups_used == 1 and self.post_act in (None, "linear")):
            # The layer below is the network's first (a Conv2DTranspose from the one-channel image): our dx has exactly
            # one consumer, that layer's weight gradient, which forms it on load from (dy, x) -- the apply pass and the
            # dx tensor (213 MB for model0 at batch 512) are not needed.  Reduction + finalize only here.
            c1, c2 = F.hip_backend().bn_bwd_sums(dy, self.x, self.mean.reshape(-1), self.invstd.reshape(-1), self.gamma,
                                                 self.prev_act, self.prev_alpha, self.ggamma, self.gbeta,
                                                 conv.gbias if conv is not None else None)
            tgt.bn_deferred = dict(x=self.x, mean=self.mean.reshape(-1), invstd=self.invstd.reshape(-1),
                                   gamma=self.gamma, c1=c1, c2=c2, act=self.prev_act, alpha=self.prev_alpha)
            return dy
        if pre is not None and pre[2] == (5 if (need_dw and conv is not None) else 2):
            dx, _, _ = F.hip_backend().bn_bwd(
                dy, self.x, None, self.mean, self.invstd, self.gamma, None, 0.3, self.used_batch_stats, self.prev_act,
                self.prev_alpha, self.ggamma if need_dw else None, self.gbeta if need_dw else None,
                conv.gbias if (need_dw and conv is not None) else None, self.groups_used, part=pre[0],
                part_blocks=pre[1])
            return dx
        no_y = self.x.is_cuda and (self.y is self.x or self.post_act in (None, "linear"))
        dx, _, _ = F.bn_bwd(dy, self.x, None if no_y else self.y, self.mean, self.invstd,
                            self.gamma, self.post_act,
                            self.post_alpha, self.used_batch_stats, self.prev_act, self.prev_alpha,
                            self.ggamma if need_dw else None, self.gbeta if need_dw else None,
                            conv.gbias if (need_dw and conv is not None) else None, self.groups_used)
        return dx

    def release(self):
        self.x = self.y = self.mean = self.invstd = self.pre_bwd = None


class ViewOp(Op):
    def __init__(self, spec, out_phys):
        self.spec, self.name, self.out_phys = spec, spec.name, tuple(out_phys)
        self.in_shape = None

    def fwd(self, x, training):
        self.in_shape = x.shape
        return x.reshape(x.shape[0], *self.out_phys)

    def bwd(self, dy, need_dx=True, need_dw=True):
        return dy.reshape(self.in_shape)


class PadOp(Op):
    def __init__(self, spec):
        self.spec, self.name = spec, spec.name
        (self.pt, self.pb), (self.pl, self.pr) = spec.cfg["padding"]

    def fwd(self, x, training):
        return F.spatial_pad(x, self.pt, self.pb, self.pl, self.pr)

    def bwd(self, dy, need_dx=True, need_dw=True):
        return F.spatial_pad(dy, -self.pt, -self.pb, -self.pl, -self.pr)


class ActOp(Op):
    def __init__(self, spec, act, alpha, c_true):
        self.spec, self.name, self.act, self.alpha, self.c_true = spec, spec.name, act, alpha, c_true
        self.y = None

    def fwd(self, x, training):
        y = F.act_fwd(x, self.act, self.alpha, self.c_true)
        if training:
            self.y = y
        return y

    def bwd(self, dy, need_dx=True, need_dw=True):
        return F.act_bwd(dy, self.y, self.act, self.alpha)

    def release(self):
        self.y = None


class AffineOp(Op):
    def __init__(self, spec, c_true):
        self.spec, self.name, self.c_true = spec, spec.name, c_true
        self.scale, self.shift = spec.cfg["scale"], spec.cfg["shift"]

    def fwd(self, x, training):
        return F.affine(x, self.scale, self.shift, self.c_true)

    def bwd(self, dy, need_dx=True, need_dw=True):
        return F.affine(dy, self.scale, 0.0, self.c_true)


# =============================================================================== network
class Network:
    def __init__(self, spec: ModelSpec, device="cpu", seed: int = 0, bn_mode: str = "batch"):
        self.spec = spec
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.cdtype = torch.bfloat16 if self.on_gpu else torch.float32
        self.cpad = 8 if self.on_gpu else 1
        self.bn_mode = bn_mode
        self.softmax_head = False
        self.branch_wgrad = False       # captured steps: wgrad chain as a parallel graph branch (SideBranch)
        # CPU oracle only: round every op's output (forward) and input gradient (backward) to bf16, the points at
        # which the device executor stores bf16 tensors -- the oracle then differs from the device by fp32
        # summation order alone and the end-to-end tests can use tight tolerances (tests/test_layerwise_gpu.py)
        self.emulate_bf16 = False
        self.ops: List[Op] = []
        self._params: List[_Param] = []
        self._build_ops()
        self._allocate()
        # BN (no activation of its own) straight in front of a single-output-channel transposed conv on the taps
        # path: the conv's kernels apply the BN affine as they load x, the BN writes no tensor of its own
        for a_, b_ in zip(self.ops, self.ops[1:]):
            if not (isinstance(a_, BNOp) and isinstance(b_, ConvOp) and a_.rank4
                    and a_.post_act in (None, "linear")):
                continue
            # in front of a depthwise layer the device applies the affine to fp32 tap sums: the BN output is never
            # rounded to bf16 there, and the bf16-emulating oracle must not round it either (Network.forward)
            a_.fp32_handoff = (BN_ON_LOAD and b_.onech is None and b_.g.depthwise and b_.kind != "conv2d_transpose"
                               and b_.g.KH == b_.g.KW <= 3 and b_.g.sh <= 3 and b_.g.sw <= 3)
            if not self.on_gpu:
                continue
            # ... or in front of any other transposed conv: the patch-resident kernel applies it while staging
            # ... or in front of a depthwise layer: the stencil kernels (forward, weight gradient) apply it per loaded pixel
            if (b_.onech == "out" or (BN_ON_LOAD_PCONV and b_.onech is None and b_.kind == "conv2d_transpose"
                                      and not b_.g.depthwise)
                    or (b_.onech is None and b_.g.depthwise and b_.kind != "conv2d_transpose")):
                a_.defer_to = b_
                b_.in_bn = a_
        self.init_weights(seed)
        if self.on_gpu:
            _BR.ensure_stream()

    def _phys(self, shape):
        return tuple(shape[:-1]) + (_round_up(shape[-1], self.cpad),)

    # ------------------------------------------------------------------ graph -> exec ops
    def _build_ops(self):
        spec = self.spec
        chain = spec.chain()
        shapes = spec.infer_shapes()
        i = 1
        cur = shapes[chain[0].name]          # logical shape (no batch dim)
        self.in_shape = cur
        flat_from = None                     # (H, W, C) if `cur` is a Flatten of a rank-3 tensor
        while i < len(chain):
            l = chain[i]
            nxt = chain[i + 1] if i + 1 < len(chain) else None
            k = l.kind
            consumed = 1
            new_flat_from = None
            if k in ("conv2d", "conv2d_transpose", "depthwise_conv2d", "dense"):
                kstruct = flat_from if (k == "dense" and flat_from is not None and self.cpad > 1) else None
                if k == "dense":
                    kin = cur[0] if kstruct is None else kstruct[0] * kstruct[1] * _round_up(kstruct[2], self.cpad)
                    g = dense_geom(kin, l.cfg["units"])
                else:
                    g = spec.conv_geom(l, cur)
                act, alpha = l.cfg.get("activation", "linear") or "linear", 0.3
                if act == "softmax":
                    if nxt is not None:
                        raise ValueError("softmax is only supported on the final Dense layer")
                    self.softmax_head, act = True, "linear"
                elif act == "linear" and nxt is not None and nxt.kind == "leaky_relu":
                    act, alpha, consumed = "leaky", nxt.cfg["alpha"], 2
                elif act == "tanh" and nxt is not None and nxt.kind == "lambda_affine" \
                        and nxt.cfg["scale"] == 0.5 and nxt.cfg["shift"] == 0.5:
                    act, consumed = "tanh01", 2
                self.ops.append(ConvOp(l, k, g, act, alpha, self.cpad, kstruct))
            elif k == "batch_norm":
                op = BNOp(l, cur[-1], _round_up(cur[-1], self.cpad), len(cur) == 3)
                op.mode = self.bn_mode
                prev = self.ops[-1] if self.ops else None
                if isinstance(prev, ConvOp):
                    prev.next_bn = op
                if isinstance(prev, ConvOp) and prev.act not in (None, "linear"):
                    op.prev_act, op.prev_alpha, prev.act_bwd_fused = prev.act, prev.alpha, True
                    op.bias_grad_for = prev        # dx of this BN is the conv's dY: its column sums come for free
                if nxt is not None and nxt.kind == "leaky_relu":
                    op.post_act, op.post_alpha, consumed = "leaky", nxt.cfg["alpha"], 2
                elif nxt is not None and nxt.kind == "activation" and nxt.cfg["activation"] != "softmax":
                    op.post_act, consumed = nxt.cfg["activation"], 2
                self.ops.append(op)
                new_flat_from = flat_from
            elif k == "flatten":
                if len(cur) == 3:
                    new_flat_from = cur
                    ph = self._phys(cur)
                    self.ops.append(ViewOp(l, (ph[0] * ph[1] * ph[2],)))
                else:
                    self.ops.append(ViewOp(l, self._phys(shapes[l.name])))
            elif k == "reshape":
                tgt = shapes[l.name]
                if self.cpad > 1 and (cur[-1] % self.cpad or tgt[-1] % self.cpad):
                    raise ValueError("device Reshape needs channel dims that are multiples of 8")
                self.ops.append(ViewOp(l, tgt))
            elif k == "zero_padding2d":
                self.ops.append(PadOp(l))
            elif k == "leaky_relu":
                self.ops.append(ActOp(l, "leaky", l.cfg["alpha"], cur[-1]))
            elif k == "activation":
                self.ops.append(ActOp(l, l.cfg["activation"], 0.3, cur[-1]))
            elif k == "lambda_affine":
                self.ops.append(AffineOp(l, cur[-1]))
            else:
                raise ValueError(f"cannot execute layer kind {k}")
            cur = shapes[chain[i + consumed - 1].name]
            flat_from = new_flat_from
            i += consumed
        self.out_shape = cur

    # ------------------------------------------------------------------ parameter storage
    def _allocate(self):
        train, state = [], []
        for op in self.ops:
            if isinstance(op, ConvOp):
                g = op.g
                phys, logical = op.shapes()
                cin, cout = op.in_channels(), op.out_channels()
                ks = {"conv2d": (g.KH, g.KW, cin, cout), "conv2d_transpose": (g.KH, g.KW, cout, cin),
                      "depthwise_conv2d": (g.KH, g.KW, cin, 1),
                      "dense": (int(np.prod(logical[1:])), cout)}[op.kind]
                kname = "depthwise_kernel" if op.kind == "depthwise_conv2d" else "kernel"
                train.append(_Param(op.name, kname, phys, logical, True, ks))
                train.append(_Param(op.name, "bias", (cout,), (cout,), True, (cout,)))
            elif isinstance(op, BNOp):
                train.append(_Param(op.name, "gamma", (op.Cp,), (op.C,), True, (op.C,)))
                train.append(_Param(op.name, "beta", (op.Cp,), (op.C,), True, (op.C,)))
                state.append(_Param(op.name, "moving_mean", (op.Cp,), (op.C,), False, (op.C,)))
                state.append(_Param(op.name, "moving_variance", (op.Cp,), (op.C,), False, (op.C,)))
        off = 0
        for p in train:
            p.offset = off
            off = _round_up(off + p.numel, ALIGN)
        self.n_train = max(off, ALIGN)
        off = 0
        for p in state:
            p.offset = off
            off = _round_up(off + p.numel, ALIGN)
        self.n_state = max(off, ALIGN)
        dev = self.device
        self.P = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.G = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.S = torch.zeros(self.n_state, dtype=torch.float32, device=dev)
        self._params = train + state
        self._pmap = {(p.layer, p.wname): p for p in self._params}
        # compute-dtype working copies
        self.Pc = torch.zeros(self.n_train, dtype=self.cdtype, device=dev) if self.on_gpu else self.P
        alt_off = 0
        self._alt: List[Tuple[ConvOp, int]] = []
        for op in self.ops:
            if isinstance(op, ConvOp) and op.kind != "depthwise_conv2d":
                self._alt.append((op, alt_off))
                alt_off = _round_up(alt_off + int(np.prod(op.alt_shape())), ALIGN)
        self.Palt = torch.zeros(max(alt_off, ALIGN), dtype=self.cdtype, device=dev) if self.on_gpu else None
        # one-channel convs (device): taps-as-channels weight copies gathered from the master buffer
        self._gather: List[Tuple] = []          # (src_off, dst_off, R1, C1, R2, C2, sr, sc)
        self._gviews = {}
        g_off = 0
        if self.on_gpu and ONECH:
            for op in self.ops:
                if isinstance(op, ConvOp) and op.kind == "conv2d_transpose" and op.g.O == 1 and op.g.I != 1:
                    op.onech = "up"             # Conv2DTranspose 1 -> N: streaming kernels on the canonical weights
                    # the network's first layer with a BatchNorm right behind it: that BatchNorm's backward apply step
                    # moves into this layer's weight gradient (BNOp.bwd)
                    if op is self.first_param_op() and op.next_bn is not None and op.next_bn.rank4:
                        op.next_bn.defer_apply_to = op
                    continue
                if not isinstance(op, ConvOp) or op.g.depthwise or op.g.I != 1:
                    continue
                if op.kind not in ("conv2d", "conv2d_transpose"):
                    continue
                g = op.g
                T, Tp = g.KH * g.KW, _round_up(g.KH * g.KW, 8)
                koff = self._pmap[(op.name, "kernel")].offset
                if op.kind == "conv2d":         # canonical [O][T][8]
                    Op = _round_up(g.O, 8)
                    op.onech = "in"
                    # small kernels (the random_layer search space): the streaming kernels of the "up" path
                    op.stream1 = STREAM1 and g.KH == g.KW <= 3
                    if op.stream1 and op is self.first_param_op() and op.next_bn is not None and op.next_bn.rank4:
                        op.next_bn.defer_apply_to = op
                    op.g1 = ConvGeom(g.OH, g.OW, T, g.OH, g.OW, g.O, 1, 1).check()
                    specs = [("w_tap", (g.O, 1, 1, Tp), (koff, g.O, T, g.O, Tp, T * 8, 8)),
                             ("w_tapT", (T, 1, 1, Op), (koff, T, g.O, T, Op, 8, T * 8))]
                else:                           # canonical [1][T][Cp]; 1x1 conv C -> T uses it as is
                    Cc, Cp = g.O, _round_up(g.O, 8)
                    op.onech = "out"
                    op.g1 = ConvGeom(g.OH, g.OW, Cc, g.OH, g.OW, T, 1, 1).check()
                    specs = [("w_tapT", (Cc, 1, 1, Tp), (koff, Cc, T, Cc, Tp, 1, Cp))]
                for attr, shape, (so, R1, C1, R2, C2, sr, sc) in specs:
                    self._gather.append((so, g_off, R1, C1, R2, C2, sr, sc))
                    self._gviews[(id(op), attr)] = (g_off, shape)
                    g_off = _round_up(g_off + R2 * C2, ALIGN)
        self.Pg = torch.zeros(max(g_off, ALIGN), dtype=self.cdtype, device=dev) if self.on_gpu else None
        self._bind_views()

    def _view(self, buf, p: _Param):
        return buf[p.offset:p.offset + p.numel].view(p.shape)

    def _bind_views(self):
        alt_off = {id(op): o for op, o in self._alt}
        for op in self.ops:
            if isinstance(op, ConvOp):
                kname = "depthwise_kernel" if op.kind == "depthwise_conv2d" else "kernel"
                pk, pb = self._pmap[(op.name, kname)], self._pmap[(op.name, "bias")]
                op.kernel, op.bias = self._view(self.P, pk), self._view(self.P, pb)   # bias consumed in fp32
                op.gkernel, op.gbias = self._view(self.G, pk), self._view(self.G, pb)
                op.w_main = self._view(self.Pc, pk)
                if op.kind == "dense" and op.kstruct is not None:
                    op.w_main = op.w_main.reshape(pk.shape[0], -1)
                    op.kernel2d = op.kernel.reshape(pk.shape[0], -1)
                if op.kind == "depthwise_conv2d":
                    op.w_alt = op.w_main
                elif self.on_gpu:
                    ashape = op.alt_shape()
                    o = alt_off[id(op)]
                    op.w_alt = self.Palt[o:o + int(np.prod(ashape))].view(ashape)
                else:
                    op.w_alt = op.kernel.t() if op.kind == "dense" else op.kernel.permute(3, 1, 2, 0)
                for attr in ("w_tap", "w_tapT"):
                    hit = self._gviews.get((id(op), attr))
                    if hit is not None:
                        o, shape = hit
                        setattr(op, attr, self.Pg[o:o + int(np.prod(shape))].view(shape))
            elif isinstance(op, BNOp):
                pg, pb = self._pmap[(op.name, "gamma")], self._pmap[(op.name, "beta")]
                op.gamma, op.beta = self._view(self.P, pg), self._view(self.P, pb)
                op.ggamma, op.gbeta = self._view(self.G, pg), self._view(self.G, pb)
                op.moving_mean = self._view(self.S, self._pmap[(op.name, "moving_mean")])
                op.moving_var = self._view(self.S, self._pmap[(op.name, "moving_variance")])

    def _scatter(self, buf: torch.Tensor, p: _Param, logical: torch.Tensor):
        v = buf[p.offset:p.offset + p.numel].view(p.shape)
        v.zero_()
        v[p.corner()] = logical.reshape(p.lshape)

    def init_weights(self, seed: int):
        """Glorot-uniform kernels, zero biases, BN gamma=1 beta=0 mean=0 var=1."""
        gen = torch.Generator().manual_seed(int(seed))
        P = torch.zeros(self.n_train, dtype=torch.float32)
        S = torch.zeros(self.n_state, dtype=torch.float32)
        for p in self._params:
            nl = int(np.prod(p.lshape))
            if p.wname in ("kernel", "depthwise_kernel"):
                ks = p.keras_shape
                rf = int(np.prod(ks[:-2])) if len(ks) > 2 else 1
                fan_in, fan_out = ks[-2] * rf, ks[-1] * rf
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                self._scatter(P, p, (torch.rand(nl, generator=gen) * 2 - 1) * lim)
            elif p.wname == "gamma":
                P[p.offset:p.offset + p.numel] = 1.0
            elif p.wname == "moving_variance":
                S[p.offset:p.offset + p.numel] = 1.0
        self.P.copy_(P)
        self.S.copy_(S)
        self.refresh_working_copies()

    def first_param_op(self):
        return next((op for op in self.ops if op.has_params), None)

    def set_no_input_grad(self, flag: bool = True):
        """Promise that `backward(need_input_grad=True)` is never called (a generator, a classifier): the first
        layer's adjoint-direction weight copy is then never read and drops out of the per-step relayout (the
        generator's Dense 128 -> 6272 is 3/4 of its relayout work)."""
        self.no_input_grad = bool(flag)
        self._relayout_tab = None

    def refresh_working_copies(self, cast: bool = True):
        """Re-derive the bf16 working copies from the fp32 master (device only).  The fused
        optimizer already writes `Pc`, in which case `cast=False` only redoes the relayouts."""
        if not self.on_gpu:
            return
        hip = F.hip_backend()
        if cast:
            hip.cast_f32_to_bf16(self.P, self.Pc)
        hip.relayout_all(self)

    # ------------------------------------------------------------------ execution
    def set_bn_mode(self, mode: str):
        self.bn_mode = mode
        for op in self.ops:
            if isinstance(op, BNOp):
                op.mode = mode

    def prepare_input(self, x: torch.Tensor, index: Optional[torch.Tensor] = None, scale: float = 1.0):
        """Logical input (fp32 / uint8 / bf16, [N, *in_shape] or a 1-channel image to be
        replicated) -> the device layout (bf16, padded channels).  `index` gathers samples on
        device (dataset resident in HBM)."""
        c_true = self.in_shape[-1]
        if self.on_gpu:
            x = x.to(self.device)
            if x.dim() == len(self.in_shape):          # [N,H,W] grayscale without channel axis
                x = x.unsqueeze(-1)
            if (x.dtype == torch.bfloat16 and x.shape[-1] == _round_up(c_true, 8) and index is None
                    and scale == 1.0 and (c_true % 8 == 0)):
                return x.contiguous()
            return F.hip_backend().pad_channels(x.contiguous(), _round_up(c_true, 8), index, scale, c_true)
        if x.dim() == len(self.in_shape):
            x = x.unsqueeze(-1)
        if index is not None:
            x = x[index]
        x = x.float() * scale if scale != 1.0 else x.float()
        if x.shape[-1] == 1 and c_true > 1:
            x = x.expand(*x.shape[:-1], c_true)
        return x.contiguous()

    def to_logical(self, y: torch.Tensor, c_true: Optional[int] = None) -> torch.Tensor:
        """Device layout -> fp32 logical tensor (padding stripped)."""
        c_true = self.out_shape[-1] if c_true is None else c_true
        if self.on_gpu:
            return F.hip_backend().strip_channels(y, c_true)
        return y.float()

    def forward(self, x, training: bool = False, start: int = 0, stop: Optional[int] = None, out=None,
                bn_groups: int = 1):
        """`out`: destination for the network's output when its last op can write in place of
        allocating (`can_forward_into()`); lets a caller assemble a batch without a concat copy.
        `bn_groups`: batch-statistics BatchNorm layers normalise that many equal contiguous parts of the
        batch independently (a discriminator batch [real ; fake] with per-half statistics)."""
        stop = len(self.ops) if stop is None else (stop if stop >= 0 else len(self.ops) + stop)
        for op in self.ops:
            if isinstance(op, BNOp):
                op.groups = bn_groups
        last = self.ops[-1] if self.ops else None
        if out is not None:
            if not (stop == len(self.ops) and self.can_forward_into()):
                raise ValueError("forward(out=...) needs a final single-channel transposed conv on device")
            last.out_buffer = out
        try:
            for op in self.ops[start:stop]:
                x = op.fwd(x, training)
                if self.emulate_bf16 and not self.on_gpu and not isinstance(op, ViewOp) \
                        and not (training and getattr(op, "fp32_handoff", False)):
                    x = x.bfloat16().float()
        finally:
            if out is not None:
                last.out_buffer = None
        return x

    def can_forward_into(self) -> bool:
        last = self.ops[-1] if self.ops else None
        return isinstance(last, ConvOp) and last.onech == "out"

    @staticmethod
    def act_owner(op):
        """(y, act, alpha) if `op` is a conv/dense layer whose activation backward is still a separate
        pass over dY (not already folded into a following BN), else None."""
        if isinstance(op, ConvOp) and op.act not in (None, "linear") and not op.act_bwd_fused and op.y is not None:
            return (op.y, op.act, op.alpha)
        return None

    def backward(self, dy, need_input_grad: bool = False, need_dw: bool = True,
                 start: Optional[int] = None, input_mul=None, dy_premultiplied: bool = False, after_op=None):
        """Back-propagate from op index `start` (exclusive upper bound, default: all ops).

        Activation backward fusion: when the op below a conv/dense layer (through reshapes) carries an
        unfused activation, its act'(y) is applied in the epilogue of this layer's input-gradient kernel
        instead of a separate elementwise pass.  `input_mul` does the same for the network INPUT gradient
        (y of the producer network's last layer: G -> D chains), `dy_premultiplied` tells this network that
        its incoming dY already carries the last layer's act'(y).

        `after_op(i)` is called once op i's backward kernels have been issued (weight-gradient branch joined): every
        gradient of ops i.. is then complete in stream order -- the hook of bucketed gradient all-reduces."""
        start = len(self.ops) if start is None else (start if start >= 0 else len(self.ops) + start)
        if need_input_grad and getattr(self, "no_input_grad", False):
            raise ValueError("this network was declared input-gradient-free (set_no_input_grad)")
        first_param = next((i for i, op in enumerate(self.ops) if op.has_params), len(self.ops))
        _BR.enabled = self.branch_wgrad
        for op in self.ops:
            if isinstance(op, ConvOp):
                op.dy_premul = False
            elif isinstance(op, BNOp):
                op.below_needs_dx = need_input_grad or not need_dw
        if dy_premultiplied and start > 0 and isinstance(self.ops[start - 1], ConvOp):
            self.ops[start - 1].dy_premul = True
        for i in range(start - 1, -1, -1):
            op = self.ops[i]
            need_dx = need_input_grad or i > first_param
            if not need_dx and not (need_dw and op.has_params):
                break
            if isinstance(op, ConvOp):
                mul = None
                if need_dx and FUSE_ACT_BWD:
                    j = i - 1
                    while j >= 0 and isinstance(self.ops[j], ViewOp):
                        j -= 1
                    if j >= 0:
                        mul = self.act_owner(self.ops[j])
                        if mul is not None:
                            self.ops[j].dy_premul = True
                    elif input_mul is not None:
                        mul = input_mul
                dy = op.bwd(dy, need_dx=need_dx, need_dw=need_dw, mul=mul)
            else:
                dy = op.bwd(dy, need_dx=need_dx, need_dw=need_dw)
            if self.emulate_bf16 and not self.on_gpu and dy is not None and not isinstance(op, ViewOp):
                dy = dy.bfloat16().float()
            if after_op is not None and op.has_params:
                _BR.join()
                after_op(i)
        _BR.join()
        return dy

    def grad_buckets(self, max_buckets: int = 2, min_tail_frac: float = 0.5):
        """Split of the flat gradient buffer for all-reduces that overlap the backward pass: [(lo, hi, ready_op)] in
        the order the buckets complete during `backward` -- bucket k may be reduced as soon as `after_op(ready_op)`
        fires.  The flat buffer is laid out in forward order and the backward pass walks it from the end, so a bucket
        is a contiguous TAIL [offset(op s), previous boundary): everything from conv-like op s upwards, BatchNorm
        parameters behind it included.  With two buckets the boundary is the latest layer (earliest ready) whose tail
        holds at least `min_tail_frac` of the parameters; if no layer above the first one does, the boundary sits
        right above the first parameter layer (everything but the last-finished gradient overlaps)."""
        pops = [i for i, op in enumerate(self.ops) if isinstance(op, ConvOp)]
        n = self.n_train
        if max_buckets < 2 or len(pops) < 2:
            return [(0, n, pops[0] if pops else 0)]
        offs = {}
        for i in pops:
            op = self.ops[i]
            kname = "depthwise_kernel" if op.kind == "depthwise_conv2d" else "kernel"
            offs[i] = self._pmap[(op.name, kname)].offset
        cuts = []
        hi = n
        cand = pops[1:]
        while len(cuts) < max_buckets - 1 and cand:
            want = min_tail_frac * hi
            good = [i for i in cand if hi - offs[i] >= want]
            s_ = max(good) if good else min(cand)
            if hi - offs[s_] <= 0:
                break
            cuts.append(s_)
            hi = offs[s_]
            cand = [i for i in cand if i < s_]
        out, hi = [], n
        for s_ in cuts:
            out.append((offs[s_], hi, s_))
            hi = offs[s_]
        out.append((0, hi, pops[0]))
        return out

    def release(self):
        for op in self.ops:
            op.release()

    def narrow_saved(self, lo: int, hi: int, group: int = 0):
        """Keep only batch rows [lo, hi) of the activations the last training forward saved (and, for BatchNorm
        layers that ran with batch groups, the statistics of `group`): the following backward then differentiates
        that part of the batch alone.  Used when one forward serves two purposes -- the DCGAN generator runs ONCE
        on [z1 ; z2] per step, the first half feeds the discriminator phase, only the second half is trained."""
        for op in self.ops:
            for attr in ("x", "y", "x_img"):
                t = getattr(op, attr, None)
                if isinstance(t, torch.Tensor) and t.dim() >= 1:
                    setattr(op, attr, t[lo:hi])
            if isinstance(op, BNOp) and op.mean is not None and op.groups_used > 1:
                op.mean, op.invstd = op.mean[group].contiguous(), op.invstd[group].contiguous()
                op.groups_used = 1
            if isinstance(op, ConvOp) and op.x_affine is not None and op.x_affine[2] > 1:
                sc, sh, _, ctrue = op.x_affine
                op.x_affine = (sc[group].contiguous(), sh[group].contiguous(), 1, ctrue)
            if isinstance(op, ViewOp) and op.in_shape is not None:
                op.in_shape = (hi - lo,) + tuple(op.in_shape[1:])

    def predict(self, x, batch_size: int = 4096, prepared: bool = False):
        """Inference forward; applies the softmax of a softmax head (Keras `model.predict`).
        Returns fp32 logical outputs."""
        outs = []
        n_out = self.out_shape[-1]
        for i in range(0, x.shape[0], batch_size):
            xb = x[i:i + batch_size]
            y = self.forward(xb if prepared else self.prepare_input(xb), training=False)
            y = F.softmax(y, n_out) if self.softmax_head else self.to_logical(y)
            outs.append(y.float())
        return torch.cat(outs, 0)

    # ------------------------------------------------------------------ weights I/O
    def param_table(self):
        return list(self._params)

    def _logical(self, p: _Param) -> np.ndarray:
        buf = self.P if p.trainable else self.S
        v = buf[p.offset:p.offset + p.numel].view(p.shape)[p.corner()]
        return v.detach().float().cpu().numpy()

    def _to_keras(self, p: _Param) -> np.ndarray:
        a = self._logical(p)
        if p.wname == "kernel":
            op = self._op(p.layer)
            if op.kind == "dense":
                a = a.reshape(a.shape[0], -1).T           # [N][K] -> (K, N)
            elif op.kind == "conv2d":
                a = a.transpose(1, 2, 3, 0)               # [O][KH][KW][I] -> (kh,kw,in,out)
            else:
                a = a.transpose(1, 2, 0, 3)               # [F][KH][KW][Cin] -> (kh,kw,filters,in)
        elif p.wname == "depthwise_kernel":
            a = a[..., None]                              # [KH][KW][C] -> (kh,kw,C,1)
        return np.ascontiguousarray(a, dtype=np.float32)

    def _from_keras(self, p: _Param, a: np.ndarray) -> torch.Tensor:
        a = np.asarray(a, dtype=np.float32)
        if tuple(a.shape) != p.keras_shape:
            raise ValueError(f"{p.layer}/{p.wname}: expected Keras shape {p.keras_shape}, got {a.shape}")
        if p.wname == "kernel":
            op = self._op(p.layer)
            if op.kind == "dense":
                a = a.T
            elif op.kind == "conv2d":
                a = a.transpose(3, 0, 1, 2)
            else:
                a = a.transpose(2, 0, 1, 3)
        elif p.wname == "depthwise_kernel":
            a = a[..., 0]
        return torch.from_numpy(np.ascontiguousarray(a)).reshape(p.lshape)

    def _op(self, name):
        for op in self.ops:
            if op.name == name:
                return op
        raise KeyError(name)

    def get_keras_weights(self) -> List[Tuple[str, str, np.ndarray]]:
        """[(layer_name, weight_name, array in Keras layout)] in Keras weight order."""
        out = []
        for op in self.ops:
            for p in [q for q in self._params if q.layer == op.name]:
                out.append((p.layer, p.wname, self._to_keras(p)))
        return out

    def set_keras_weights(self, weights: Dict[Tuple[str, str], np.ndarray], strict: bool = True):
        P, S = self.P.detach().cpu().clone(), self.S.detach().cpu().clone()
        for p in self._params:
            key = (p.layer, p.wname)
            if key not in weights:
                if strict:
                    raise KeyError(f"missing weight {key}")
                continue
            self._scatter(P if p.trainable else S, p, self._from_keras(p, weights[key]))
        self.P.copy_(P)
        self.S.copy_(S)
        self.refresh_working_copies()

    def logical_state(self) -> Dict[Tuple[str, str], torch.Tensor]:
        """Device-independent snapshot {(layer, weight): logical fp32 CPU tensor} -- what a
        transfer-learning child or another rank needs to rebuild this network."""
        out = {}
        for p in self._params:
            t = torch.from_numpy(self._logical(p).copy())
            if p.wname == "kernel" and self._op(p.layer).kind == "dense":
                t = t.reshape(t.shape[0], -1)          # structured-K Dense kernels travel as [N][K]
            out[(p.layer, p.wname)] = t
        return out

    def load_logical_state(self, state: Dict[Tuple[str, str], torch.Tensor], strict: bool = True):
        P, S = self.P.detach().cpu().clone(), self.S.detach().cpu().clone()
        for p in self._params:
            key = (p.layer, p.wname)
            if key not in state:
                if strict:
                    raise KeyError(f"missing weight {key}")
                continue
            t = state[key]
            if t.numel() == int(np.prod(p.lshape)) and p.wname == "kernel" and t.dim() == 2:
                t = t.reshape(p.lshape)
            if tuple(t.shape) != p.lshape:
                raise ValueError(f"{key}: shape {tuple(t.shape)} != {p.lshape}")
            self._scatter(P if p.trainable else S, p, t.float())
        self.P.copy_(P)
        self.S.copy_(S)
        self.refresh_working_copies()

    def num_params(self, trainable_only=False):
        return sum(int(np.prod(p.lshape)) for p in self._params if p.trainable or not trainable_only)
