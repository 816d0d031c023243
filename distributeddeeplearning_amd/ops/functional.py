"""Device dispatch for every framework op.

CPU tensors run the fp32 oracle (ops/cpu_ref.py); device tensors run the hand-written gfx950
kernels through ops/hip.py.  There is deliberately no third path: a device tensor with the HIP
extension missing raises instead of silently falling back to eager PyTorch.

Device tensors are bf16 with the channel (last) dimension physically padded to a multiple of 8
(zeros in the padding lanes), so every kernel access is a 16-byte vector and every reduction
dimension is MFMA-friendly; `c_true` arguments carry the logical channel count where an op must
keep the padding lanes at exactly zero.  CPU tensors are unpadded fp32.
"""
from __future__ import annotations

import torch

from . import cpu_ref

_hip = None


def hip_backend():
    global _hip
    if _hip is None:
        from . import hip as _h   # raises with a clear message if the .so is missing
        _hip = _h
    return _hip


def _b(t: torch.Tensor):
    return hip_backend() if t.is_cuda else cpu_ref


# conv family -------------------------------------------------------------------------------
def _cpu_mul(t, mul):
    if mul is None:
        return t
    y, act, alpha = mul
    return cpu_ref.act_bwd(t, y.reshape(t.shape), act, alpha)


def conv_fprop(x, wA, bias, g, act=None, alpha=0.3, mul=None, stats_out=None, bn_infer=None, in_affine=None,
               aff_sink=None):
    """`mul` = (y, act, alpha): result *= act'(y), the activation backward of the layer below (fused
    into the kernel epilogue on device).  `stats_out` (dict): on device, a kernel that can emit the
    output's per-channel partial sums for a following BatchNorm leaves them in stats_out["part"/"nb"];
    with `bn_infer` = (gamma, beta, moving_mean, moving_var, eps) a kernel that can fold that inference-phase
    BatchNorm into its epilogue does so and sets stats_out["affine"]."""
    if x.is_cuda:
        return hip_backend().conv_fprop(x, wA, bias, g, act, alpha, mul, stats_out, bn_infer, in_affine=in_affine,
                                        aff_sink=aff_sink)
    if in_affine is not None:
        raise ValueError("conv_fprop: in_affine exists on the device only")
    return _cpu_mul(cpu_ref.conv_fprop(x, wA, bias, g, act, alpha), mul)


def conv_dgrad(dy, wB, g, bias=None, act=None, alpha=0.3, mul=None, stats_out=None, bn_infer=None, in_affine=None,
               aff_sink=None):
    """`wB` is the core weight in [I][KH][KW][O] order (reduction channel contiguous);
    depthwise: [KH][KW][C].  `in_affine`: device only (a BatchNorm applied on load, ops/hip.py)."""
    if dy.is_cuda:
        return hip_backend().conv_dgrad(dy, wB, g, bias, act, alpha, mul, stats_out, bn_infer, in_affine=in_affine,
                                        aff_sink=aff_sink)
    if in_affine is not None:
        raise ValueError("conv_dgrad: in_affine exists on the device only")
    wA = wB if g.depthwise else wB.permute(3, 1, 2, 0)
    return _cpu_mul(cpu_ref.conv_dgrad(dy, wA, g, bias, act, alpha), mul)


def conv_wgrad(x, dy, g, layout="A", out=None, x_affine=None):
    """Weight gradient of the core correlation, written in layout A=[O][KH][KW][I] or
    B=[I][KH][KW][O] (depthwise: [KH][KW][C]), fp32.  `out` (a view into the flat gradient
    buffer) is overwritten."""
    if x.is_cuda:
        return hip_backend().conv_wgrad(x, dy, g, layout, out, x_affine=x_affine)
    if x_affine is not None:
        raise ValueError("conv_wgrad: x_affine exists on the device only")
    dw = cpu_ref.conv_wgrad(x, dy, g)
    if layout == "B" and not g.depthwise:
        dw = dw.permute(3, 1, 2, 0)
    if out is not None:
        out.copy_(dw.reshape(out.shape))
        return out
    return dw.contiguous()


def bias_grad(dy, out):
    if dy.is_cuda:
        return hip_backend().bias_grad(dy, out)
    out.copy_(cpu_ref.bias_grad(dy))
    return out


# activations -------------------------------------------------------------------------------
def act_fwd(x, kind, alpha=0.3, c_true=None):
    if x.is_cuda:
        return hip_backend().act_fwd(x, kind, alpha, c_true)
    return cpu_ref.act_fwd(x, kind, alpha)


def act_bwd(dy, y, kind, alpha=0.3):
    return _b(dy).act_bwd(dy, y, kind, alpha)


def affine(x, scale, shift, c_true=None):
    if x.is_cuda:
        return hip_backend().affine(x, scale, shift, c_true)
    return x * scale + shift


def spatial_pad(x, pt, pb, pl, pr):
    """ZeroPadding2D on NHWC (negative values crop: used for its backward)."""
    if x.is_cuda:
        return hip_backend().spatial_pad(x, pt, pb, pl, pr)
    return torch.nn.functional.pad(x, (0, 0, pl, pr, pt, pb))


# batch norm --------------------------------------------------------------------------------
def bn_train_affine(x, gamma, beta, mm, mv, momentum, eps, bessel, update_moving=True, groups=1):
    """Batch statistics of x -> (scale, shift, mean, invstd) with y = x*scale + shift, plus the
    Keras moving-average update  moving = moving*momentum + batch*(1-momentum)  (variance
    Bessel-corrected for 4-D inputs, as TF's fused batch norm does).

    `groups` > 1: the batch is `groups` equal contiguous parts, each normalised with its own
    statistics (results are [groups][C]); the moving statistics take one update per part, in order --
    what calling the layer once per part would do (a GAN discriminator on [real ; fake])."""
    if x.is_cuda:
        return hip_backend().bn_train_affine(x, gamma, beta, mm, mv, momentum, eps, bessel, update_moving, groups)
    outs = []
    for xs in x.chunk(groups, 0):
        mean, var = cpu_ref.bn_stats(xs)
        count = xs.numel() // xs.shape[-1]
        invstd = torch.rsqrt(var + eps)
        scale = gamma * invstd
        shift = beta - mean * scale
        if update_moving:
            mm.mul_(momentum).add_(mean, alpha=1 - momentum)
            v = var * (count / max(count - 1, 1)) if bessel else var
            mv.mul_(momentum).add_(v, alpha=1 - momentum)
        outs.append((scale, shift, mean, invstd))
    if groups == 1:
        return outs[0]
    return tuple(torch.stack([o[k] for o in outs], 0) for k in range(4))


def bn_train_affine_from_part(part, nb, rows, gamma, beta, mm, mv, momentum, eps, bessel, update_moving=True,
                              groups=1):
    """Device only: batch statistics from partial sums the producing conv kernel already wrote."""
    return hip_backend().bn_train_affine_from_part(part, nb, rows, gamma, beta, mm, mv, momentum, eps, bessel,
                                                   update_moving, groups)


def bn_infer_affine(gamma, beta, mm, mv, eps):
    """(scale, shift, invstd) from the moving statistics (inference-phase BN)."""
    if gamma.is_cuda:
        return hip_backend().bn_infer_affine(gamma, beta, mm, mv, eps)
    invstd = torch.rsqrt(mv + eps)
    scale = gamma * invstd
    return scale, beta - mm * scale, invstd


def bn_apply_infer(x, gamma, beta, mm, mv, eps, act=None, alpha=0.3, c_true=None):
    """y = act(BN(x)) with the moving statistics, nothing saved for a backward pass."""
    if x.is_cuda:
        return hip_backend().bn_apply_infer(x, gamma, beta, mm, mv, eps, act, alpha, c_true)
    scale, shift, _ = bn_infer_affine(gamma, beta, mm, mv, eps)
    return cpu_ref.bn_apply(x, scale, shift, act, alpha)


def bn_apply(x, scale, shift, act=None, alpha=0.3, c_true=None, groups=1):
    """`groups` > 1: scale / shift are [groups][C], one row per equal contiguous part of the batch."""
    if x.is_cuda:
        return hip_backend().bn_apply(x, scale, shift, act, alpha, c_true, groups)
    if groups == 1:
        return cpu_ref.bn_apply(x, scale, shift, act, alpha)
    return torch.cat([cpu_ref.bn_apply(xs, scale[i], shift[i], act, alpha)
                      for i, xs in enumerate(x.chunk(groups, 0))], 0)


def bn_bwd(dy, x, y, mean, invstd, gamma, act, alpha, train_stats, prev_act=None, prev_alpha=0.3,
           out_dgamma=None, out_dbeta=None, out_dbias=None, groups=1):
    """`out_dbias`: also write the bias gradient of the conv/dense layer feeding this BN (the column
    sums of dx) -- on device it comes out of the same reduction pass as dgamma/dbeta.
    `groups` > 1: per-part statistics (mean / invstd are [groups][C]); dgamma / dbeta / dbias sum over parts."""
    if dy.is_cuda:
        return hip_backend().bn_bwd(dy, x, y, mean, invstd, gamma, act, alpha, train_stats, prev_act,
                                    prev_alpha, out_dgamma, out_dbeta, out_dbias, groups)
    if groups == 1:
        dx, dgamma, dbeta = cpu_ref.bn_bwd(dy, x, y, mean, invstd, gamma, act, alpha, train_stats, prev_act,
                                           prev_alpha)
    else:
        ys = y.chunk(groups, 0) if y is not None else [None] * groups
        parts = [cpu_ref.bn_bwd(d, xx, yy, mean[i], invstd[i], gamma, act, alpha, train_stats, prev_act, prev_alpha)
                 for i, (d, xx, yy) in enumerate(zip(dy.chunk(groups, 0), x.chunk(groups, 0), ys))]
        dx = torch.cat([p[0] for p in parts], 0)
        dgamma, dbeta = sum(p[1] for p in parts), sum(p[2] for p in parts)
    if out_dgamma is not None:
        out_dgamma.copy_(dgamma)
        out_dbeta.copy_(dbeta)
    if out_dbias is not None:
        out_dbias.copy_(cpu_ref.bias_grad(dx)[:out_dbias.numel()])
    return dx, dgamma, dbeta


# losses ------------------------------------------------------------------------------------
def softmax(logits, n_true=None):
    if logits.is_cuda:
        return hip_backend().softmax(logits, n_true)
    return cpu_ref.softmax(logits)


def softmax_ce(logits, onehot, want_grad=True, index=None):
    """`index`: row r's target is onehot[index[r]] -- on device the gather happens inside the loss kernel."""
    if index is not None and not logits.is_cuda:
        onehot, index = onehot.index_select(0, index), None
    return _b(logits).softmax_ce(logits, onehot, want_grad, **({"index": index} if index is not None else {}))


def softmax_mse(logits, target, want_grad=True, index=None):
    if index is not None and not logits.is_cuda:
        target, index = target.index_select(0, index), None
    return _b(logits).softmax_mse(logits, target, want_grad, **({"index": index} if index is not None else {}))


def sigmoid_bce(logits, labels, want_grad=True):
    return _b(logits).sigmoid_bce(logits, labels, want_grad)


def dense1_bce_fwd_bwd(x, w, b, label_first, label_second, n_first, want_dx=True, want_dw=True,
                       out_dw=None, out_db=None, bn=None):
    """`bn`: device only -- the BatchNorm in front of the head fused into the kernel (ops/hip.py)."""
    if x.is_cuda:
        return hip_backend().dense1_bce_fwd_bwd(x, w, b, label_first, label_second, n_first, want_dx, want_dw,
                                                out_dw, out_db, bn)
    if bn is not None:
        raise ValueError("dense1_bce: the BatchNorm fusion exists on the device only")
    loss, dx, dw, db, logits = cpu_ref.dense1_bce_fwd_bwd(x, w, b, label_first, label_second, n_first,
                                                          want_dx, want_dw)
    if want_dw and out_dw is not None:
        out_dw.copy_(dw.reshape(out_dw.shape))
        out_db.copy_(db.reshape(out_db.shape))
    return loss, dx, dw, db, logits


def argmax_match_count(pred, onehot):
    return _b(pred).argmax_match_count(pred, onehot)
