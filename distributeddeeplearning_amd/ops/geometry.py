"""Keras-2.2.4 / TF-1.x convolution geometry, expressed as one *core correlation*.

Every conv-like layer of the reference (Conv2D, Conv2DTranspose, DepthwiseConv2D --
/root/reference/DDL_Normal4.py:19-53, /root/reference/GAN_distributed.py:15-72) is
described by a single `ConvGeom`: a plain correlation

    Y[b, oh, ow, o] = sum_{kh,kw,i} X[b, oh*sh + kh*dh - pt, ow*sw + kw*dw - pl, i] * W[o, kh, kw, i]

with out-of-range X pixels reading as zero.  Padding is *explicit and asymmetric*
(pt/pl only; bottom/right padding is implied by OH/OW plus the bounds check), which is what
TF's `same` rule needs and what PyTorch's symmetric `padding=` cannot
express.

* Conv2D forward            = fprop(core)        X = layer input,  Y = layer output
* Conv2DTranspose forward   = dgrad(core)        X = layer OUTPUT, Y = layer INPUT
* DepthwiseConv2D forward   = fprop(core) with one filter per channel

The HIP kernels (csrc/hip/igemm_conv.hip) and the CPU oracle (ops/cpu_ref.py) both consume
this struct, so geometry is decided in exactly one place.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Tuple


@dataclass(frozen=True)
class ConvGeom:
    H: int          # core input height  (X)
    W: int
    I: int          # core input channels
    OH: int         # core output height (Y)
    OW: int
    O: int          # core output channels
    KH: int
    KW: int
    sh: int = 1
    sw: int = 1
    dh: int = 1
    dw: int = 1
    pt: int = 0
    pl: int = 0
    depthwise: bool = False

    @property
    def taps(self) -> int:
        return self.KH * self.KW

    def check(self) -> "ConvGeom":
        for name in ("H", "W", "I", "OH", "OW", "O", "KH", "KW", "sh", "sw", "dh", "dw"):
            if getattr(self, name) <= 0:
                raise ValueError(f"ConvGeom.{name} must be positive, got {getattr(self, name)}")
        if self.pt < 0 or self.pl < 0:
            raise ValueError("negative padding")
        if self.depthwise and self.I != self.O:
            raise ValueError("depthwise core needs I == O (depth_multiplier 1)")
        return self

    def pad_bottom_right(self) -> Tuple[int, int]:
        """Implied bottom/right zero padding needed so every tap of the last output pixel
        is addressable (may be negative = trailing rows never read)."""
        pb = (self.OH - 1) * self.sh + (self.KH - 1) * self.dh - self.pt - (self.H - 1)
        pr = (self.OW - 1) * self.sw + (self.KW - 1) * self.dw - self.pl - (self.W - 1)
        return pb, pr

    def macs(self, batch: int) -> int:
        per_out = self.KH * self.KW * (1 if self.depthwise else self.I)
        return batch * self.OH * self.OW * self.O * per_out


def _eff(k: int, d: int) -> int:
    return k + (k - 1) * (d - 1)


def conv_out_len(n: int, k: int, s: int, d: int, padding: str) -> int:
    """keras.utils.conv_utils.conv_output_length."""
    ke = _eff(k, d)
    if padding == "same":
        out = n
    elif padding == "valid":
        out = n - ke + 1
    else:
        raise ValueError(f"unsupported padding {padding!r}")
    return (out + s - 1) // s


def same_pad_before(n: int, out: int, k: int, s: int, d: int) -> int:
    """TF `same`: total = max((out-1)*s + k_eff - n, 0); before = total // 2."""
    total = max((out - 1) * s + _eff(k, d) - n, 0)
    return total // 2


def deconv_out_len(n: int, k: int, s: int, d: int, padding: str) -> int:
    """keras.utils.conv_utils.deconv_length with output_padding=None."""
    ke = _eff(k, d)
    if padding == "valid":
        return n * s + max(ke - s, 0)
    if padding == "same":
        return n * s
    raise ValueError(f"unsupported padding {padding!r}")


def conv2d_geom(H, W, I, O, kernel, strides=(1, 1), dilation=(1, 1), padding="valid") -> ConvGeom:
    """Geometry of a Keras Conv2D (layer input HxWxI -> OHxOWxO)."""
    (kh, kw), (sh, sw), (dh, dw) = kernel, strides, dilation
    OH = conv_out_len(H, kh, sh, dh, padding)
    OW = conv_out_len(W, kw, sw, dw, padding)
    if OH <= 0 or OW <= 0:
        raise ValueError(
            f"Conv2D k={kernel} s={strides} d={dilation} {padding} does not fit input {H}x{W}")
    pt = same_pad_before(H, OH, kh, sh, dh) if padding == "same" else 0
    pl = same_pad_before(W, OW, kw, sw, dw) if padding == "same" else 0
    return ConvGeom(H, W, I, OH, OW, O, kh, kw, sh, sw, dh, dw, pt, pl).check()


def depthwise_geom(H, W, C, kernel, strides=(1, 1), dilation=(1, 1), padding="valid",
                   keras224_compat: bool = True) -> ConvGeom:
    """DepthwiseConv2D.  Keras 2.2.4's static shape inference ignores `dilation_rate`
    for this layer (model0 depthwise_conv2d_1 k3 s2 d2 maps 57 -> 28), and the
    Dense kernels stored in the shipped weights.h5 were sized from those shapes, so the
    compat mode treats the dilation as 1."""
    if keras224_compat:
        dilation = (1, 1)
    g = conv2d_geom(H, W, C, C, kernel, strides, dilation, padding)
    return replace(g, depthwise=True).check()


def conv2d_transpose_geom(H, W, Cin, filters, kernel, strides=(1, 1), dilation=(1, 1),
                          padding="valid") -> ConvGeom:
    """Geometry of a Keras Conv2DTranspose mapping HxWxCin -> OUTHxOUTWxfilters.

    Returned as the *core* correlation whose dgrad is the layer's forward:
    core X = layer output (OUTH x OUTW x filters), core Y = layer input (H x W x Cin).
    """
    (kh, kw), (sh, sw), (dh, dw) = kernel, strides, dilation
    OUTH = deconv_out_len(H, kh, sh, dh, padding)
    OUTW = deconv_out_len(W, kw, sw, dw, padding)
    pt = same_pad_before(OUTH, H, kh, sh, dh) if padding == "same" else 0
    pl = same_pad_before(OUTW, W, kw, sw, dw) if padding == "same" else 0
    return ConvGeom(OUTH, OUTW, filters, H, W, Cin, kh, kw, sh, sw, dh, dw, pt, pl).check()


def dense_geom(K: int, N: int) -> ConvGeom:
    """A Dense layer is a 1x1 correlation over a 1x1 image."""
    return ConvGeom(1, 1, K, 1, 1, N, 1, 1).check()
