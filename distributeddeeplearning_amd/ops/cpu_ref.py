"""Reference-semantics oracle: every op of the framework in plain fp32 PyTorch (CPU).

This is the numerical ground truth the HIP kernels are tested against (the reference's own
stack -- Keras 2.2.4 / TF 1.x -- cannot be installed here) and the
execution backend for CPU tensors (gloo plumbing config, unit tests).  Layout is NHWC
everywhere, weights are [O][KH][KW][I] (Dense: [N][K]; depthwise: [KH][KW][C]).

The same function names/signatures are implemented by ops/hip.py on device tensors;
ops/functional.py dispatches on `tensor.is_cuda`.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .geometry import ConvGeom

ACTIVATIONS = ("linear", "relu", "sigmoid", "softplus", "tanh", "softsign", "leaky", "tanh01")


# --------------------------------------------------------------------------- activations
def act_fwd(x: torch.Tensor, kind: str, alpha: float = 0.3) -> torch.Tensor:
    if kind in (None, "linear"):
        return x
    if kind == "relu":
        return torch.relu(x)
    if kind == "sigmoid":
        return torch.sigmoid(x)
    if kind == "softplus":
        return F.softplus(x)
    if kind == "tanh":
        return torch.tanh(x)
    if kind == "softsign":
        return x / (1 + x.abs())
    if kind == "leaky":
        return torch.where(x > 0, x, alpha * x)
    if kind == "tanh01":            # tanh followed by Lambda (x+1)/2, GAN_distributed.py:68-70
        return (torch.tanh(x) + 1) * 0.5
    raise ValueError(f"unknown activation {kind!r}")


def act_grad_from_y(y: torch.Tensor, kind: str, alpha: float = 0.3) -> torch.Tensor:
    """d act(x)/dx expressed through the *output* y = act(x) (all our activations are
    invertible enough for that), so the forward never has to keep pre-activations."""
    if kind in (None, "linear"):
        return torch.ones_like(y)
    if kind == "relu":
        return (y > 0).to(y.dtype)
    if kind == "sigmoid":
        return y * (1 - y)
    if kind == "softplus":
        return 1 - torch.exp(-y)
    if kind == "tanh":
        return 1 - y * y
    if kind == "softsign":
        return (1 - y.abs()) ** 2
    if kind == "leaky":
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, alpha))
    if kind == "tanh01":
        t = 2 * y - 1
        return (1 - t * t) * 0.5
    raise ValueError(f"unknown activation {kind!r}")


def act_bwd(dy: torch.Tensor, y: torch.Tensor, kind: str, alpha: float = 0.3) -> torch.Tensor:
    if kind in (None, "linear"):
        return dy
    return dy * act_grad_from_y(y, kind, alpha)


# --------------------------------------------------------------------------- conv core
def _corr(x_nhwc: torch.Tensor, w: torch.Tensor, g: ConvGeom) -> torch.Tensor:
    """Linear core correlation, NHWC in / NHWC out, fp32."""
    x = x_nhwc.permute(0, 3, 1, 2)
    pb, pr = g.pad_bottom_right()
    x = F.pad(x, (g.pl, pr, g.pt, pb))
    if g.depthwise:
        wt = w.permute(2, 0, 1).reshape(g.O, 1, g.KH, g.KW)     # canonical depthwise: [KH][KW][C]
        y = F.conv2d(x, wt, None, (g.sh, g.sw), 0, (g.dh, g.dw), groups=g.O)
    else:
        wt = w.permute(0, 3, 1, 2)
        y = F.conv2d(x, wt, None, (g.sh, g.sw), 0, (g.dh, g.dw))
    assert y.shape[2] == g.OH and y.shape[3] == g.OW, (y.shape, g)
    return y.permute(0, 2, 3, 1).contiguous()


def conv_fprop(x, w, bias, g: ConvGeom, act: Optional[str] = None, alpha: float = 0.3):
    y = _corr(x.float(), w.float(), g)
    if bias is not None:
        y = y + bias.float()
    return act_fwd(y, act, alpha)


def conv_dgrad(dy, w, g: ConvGeom, bias=None, act: Optional[str] = None, alpha: float = 0.3):
    """Adjoint of the core correlation w.r.t. X (optionally + bias + act: that is the
    forward pass of a Conv2DTranspose layer)."""
    B = dy.shape[0]
    with torch.enable_grad():
        x0 = torch.zeros(B, g.H, g.W, g.I, dtype=torch.float32, requires_grad=True)
        y = _corr(x0, w.float().detach(), g)
        (dx,) = torch.autograd.grad(y, x0, dy.float())
    if bias is not None:
        dx = dx + bias.float()
    return act_fwd(dx, act, alpha)


def conv_wgrad(x, dy, g: ConvGeom):
    """dW[O][KH][KW][I] (depthwise: [KH][KW][C]) = sum over pixels of dY x gathered X."""
    shape = (g.KH, g.KW, g.O) if g.depthwise else (g.O, g.KH, g.KW, g.I)
    with torch.enable_grad():
        w0 = torch.zeros(shape, dtype=torch.float32, requires_grad=True)
        y = _corr(x.float().detach(), w0, g)
        (dw,) = torch.autograd.grad(y, w0, dy.float())
    return dw


def bias_grad(dy):
    return dy.float().reshape(-1, dy.shape[-1]).sum(0)


# --------------------------------------------------------------------------- batch norm
def bn_stats(x):
    """Per-channel (last axis) mean and biased variance over all other axes."""
    xf = x.float().reshape(-1, x.shape[-1])
    mean = xf.mean(0)
    var = xf.var(0, unbiased=False)
    return mean, var


def bn_apply(x, scale, shift, act: Optional[str] = None, alpha: float = 0.3):
    """y = act(x * scale[c] + shift[c])"""
    return act_fwd(x.float() * scale + shift, act, alpha)


def bn_bwd(dy, x, y, mean, invstd, gamma, act: Optional[str], alpha: float, train_stats: bool,
           prev_act: Optional[str] = None, prev_alpha: float = 0.3):
    """Backward of y = act(gamma * (x-mean)*invstd + beta).

    Returns (dx, dgamma, dbeta).  With `train_stats` the batch statistics are functions of x
    (training-mode BN); otherwise mean/invstd are constants (Keras inference-phase BN, SURVEY
    A.2).  If `prev_act` is given, dx is additionally multiplied by prev_act'(x): x is then the
    post-activation output of the producing Conv/Dense layer (Keras order Conv(act) -> BN,
    SURVEY A.6) and the result is the gradient w.r.t. that layer's pre-activation.
    """
    C = x.shape[-1]
    xf = x.float().reshape(-1, C)
    g = act_bwd(dy.float(), y.float(), act, alpha).reshape(-1, C)
    xhat = (xf - mean) * invstd
    dbeta = g.sum(0)
    dgamma = (g * xhat).sum(0)
    if train_stats:
        M = xf.shape[0]
        dx = (gamma * invstd) * (g - dbeta / M - xhat * (dgamma / M))
    else:
        dx = g * (gamma * invstd)
    dx = dx.reshape(x.shape)
    if prev_act not in (None, "linear"):
        dx = dx * act_grad_from_y(x.float(), prev_act, prev_alpha)
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------- losses
def softmax(logits):
    return torch.softmax(logits.float(), dim=-1)


def softmax_ce(logits, onehot, want_grad: bool = True):
    """Keras categorical_crossentropy on softmax probabilities (clipped to [1e-7, 1-1e-7]),
    mean over the batch.  Returns (loss, dlogits, probs)."""
    p = torch.softmax(logits.float(), dim=-1)
    pc = p.clamp(1e-7, 1 - 1e-7)
    loss = -(onehot.float() * pc.log()).sum(-1).mean()
    dlogits = None
    if want_grad:
        # gradient of -sum t log softmax (clipping is inactive except at saturation)
        B = logits.shape[0]
        dlogits = (p * onehot.float().sum(-1, keepdim=True) - onehot.float()) / B
    return loss, dlogits, p


def softmax_mse(logits, target, want_grad: bool = True):
    """mean_squared_error on the softmax output (DDL_Normal4.py:67): mean over classes, then batch."""
    p = torch.softmax(logits.float(), dim=-1)
    diff = p - target.float()
    loss = (diff * diff).mean()
    dlogits = None
    if want_grad:
        gp = 2 * diff / diff.numel()
        dlogits = p * (gp - (gp * p).sum(-1, keepdim=True))
    return loss, dlogits, p


def sigmoid_bce(logits, labels, want_grad: bool = True):
    """tf.losses.sigmoid_cross_entropy: mean over batch of max(x,0) - x*z + log(1+exp(-|x|))."""
    x = logits.float().reshape(-1)
    z = labels.float().reshape(-1)
    loss = (x.clamp_min(0) - x * z + torch.log1p(torch.exp(-x.abs()))).mean()
    dl = None
    if want_grad:
        dl = ((torch.sigmoid(x) - z) / x.numel()).reshape(logits.shape)
    return loss, dl


def dense1_bce_fwd_bwd(x, w, b, label_first: float, label_second: float, n_first: int,
                       want_dx: bool = True, want_dw: bool = True):
    """Fused head of the DCGAN discriminator (GAN_distributed.py:39-40,97): Dense K->1 on the
    flattened features, sigmoid-BCE against labels that are `label_first` for the first
    `n_first` rows and `label_second` for the rest, and the backward of both.
    Returns (loss, dx, dw, db, logits)."""
    M = x.shape[0]
    xf = x.float().reshape(M, -1)
    wf = w.float().reshape(-1)
    logits = xf @ wf + b.float().reshape(())
    z = torch.full((M,), float(label_second))
    z[:n_first] = float(label_first)
    loss, dl = sigmoid_bce(logits, z, True)
    dl = dl.reshape(M)
    dx = (dl[:, None] * wf[None, :]).reshape(x.shape) if want_dx else None
    dw = (dl @ xf).reshape(w.shape) if want_dw else None
    db = dl.sum().reshape(b.shape) if want_dw else None
    return loss, dx, dw, db, logits


# --------------------------------------------------------------------------- metrics / data
def argmax_match_count(pred, onehot):
    return (pred.float().argmax(-1) == onehot.float().argmax(-1)).sum()


def uniform_noise(shape, gen: torch.Generator, lo: float = -1.0, hi: float = 1.0):
    return torch.rand(shape, generator=gen) * (hi - lo) + lo


# --------------------------------------------------------------------------- optimizers
def adam_step(p, g, m, v, step: int, lr: float, b1: float, b2: float, eps: float):
    """Keras / TF-1 Adam:  lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  p -= lr_t*m/(sqrt(v)+eps)."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    p.addcdiv_(m, v.sqrt().add_(eps), value=-lr_t)


def sgd_step(p, g, lr: float):
    p.add_(g, alpha=-lr)


def rmsprop_step(p, g, a, lr: float, rho: float, eps: float):
    """Keras RMSprop: a = rho*a + (1-rho) g^2 ; p -= lr * g / (sqrt(a) + eps)."""
    a.mul_(rho).addcmul_(g, g, value=1 - rho)
    p.addcdiv_(g, a.sqrt().add_(eps), value=-lr)


def adam_fresh_double_step(p, g, lr: float, b1: float, b2: float, eps: float):
    """The reference GAN's optimizer semantics (GAN_distributed.py:93-100): a brand
    new tf.train.AdamOptimizer every call, apply_gradients twice with the same gradients."""
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    adam_step(p, g, m, v, 1, lr, b1, b2, eps)
    adam_step(p, g, m, v, 2, lr, b1, b2, eps)
