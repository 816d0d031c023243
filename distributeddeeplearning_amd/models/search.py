"""Random architecture search space and transfer-model surgery, on the IR.

* `make_new_model`      -- /root/reference/DDL.py:58-84 (Conv2D*n -> Flatten -> Dense*m -> softmax,
                           activation inside the layer, BatchNorm after it, `same` padding) and the
                           v4 variant /root/reference/DDL_Normal4.py:82-95 built from `random_layer`.
* `random_layer`        -- /root/reference/DDL_Normal4.py:19-53: Conv2D / DepthwiseConv2D /
                           Conv2DTranspose with random k/s/d, `valid` padding, and the retry loop
                           that wraps the input in ZeroPadding2D until the layer fits.
* `make_transfer_model` -- /root/reference/DDL_Normal3.py:92-119, DDL_Ray3.py:97-123: clone incl.
                           weights, drop the last 1-2 layers, graft a new random Dense/BN head.
All randomness comes from an explicit numpy Generator (the reference used the unseeded global).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

from ..config import SearchSpace
from . import ir
from .optim import OptimConfig


@dataclass
class CompiledSpec:
    """What Keras' `model.compile(optimizer, loss)` attached to a model."""
    spec: ir.ModelSpec
    optimizer: OptimConfig
    loss: str


def _randint(rng: np.random.Generator, lohi) -> int:
    lo, hi = lohi
    return int(rng.integers(lo, hi)) if hi > lo else int(lo)


def sample_optimizer(space: SearchSpace, rng) -> OptimConfig:
    name = str(rng.choice(list(space.optimizers)))
    lr = 10.0 ** _randint(rng, space.lr_exponent)
    return OptimConfig.keras(name, lr)


def sample_loss(space: SearchSpace, rng) -> str:
    return str(rng.choice(list(space.losses)))


def random_layer(space: SearchSpace, rng, model: ir.ModelSpec, names: ir.NameScope) -> None:
    """Append one random conv-like layer (+ BN) to `model`, zero-padding the input until it fits."""
    kind = str(rng.choice(["conv2d", "depthwise_conv2d", "conv2d_transpose"]))
    filters = 2 ** _randint(rng, space.filters_log2)
    k = _randint(rng, space.rl_kernel)
    s = _randint(rng, space.rl_stride)
    d = _randint(rng, space.rl_dilation)
    act = str(rng.choice(list(space.activations)))
    cur = model.output_shape
    if kind == "conv2d":
        if rng.integers(0, 2) == 0:       # np.random.choice(['strides', 'dilation_rate']) = [1, 1]
            s = 1
        else:
            d = 1
        make = lambda n: ir.Conv2D(n, filters, k, s, d, "valid", act)
    elif kind == "depthwise_conv2d":
        make = lambda n: ir.DepthwiseConv2D(n, k, s, d, "valid", act)
    else:
        s = 1 if cur[0] > 28 else 2       # DDL_Normal4.py:39-43 ("keras bug" work-around)
        make = lambda n: ir.Conv2DTranspose(n, filters, k, s, d, "valid", act)
    lname = names(kind)
    while True:
        layer = make(lname)
        try:
            model.out_shape(layer, model.output_shape)
            break
        except ir.ShapeError:
            model.append(ir.ZeroPadding2D(names("zero_padding2d"), 1))
    model.append(layer)
    model.append(ir.BatchNorm(names("batch_norm")))


def make_new_model(space: SearchSpace, rng, input_shape, output_dim: int,
                   names: Optional[ir.NameScope] = None) -> CompiledSpec:
    names = names or ir.NameScope()
    m = ir.ModelSpec(names("model"), [ir.Input(names("input"), input_shape)])
    if space.layer_space == "random_layer":
        for _ in range(_randint(rng, space.layer_size)):
            random_layer(space, rng, m, names)
        m.append(ir.Flatten(names("flatten")))
    else:
        for _ in range(_randint(rng, space.conv2d_size)):
            filters = _randint(rng, space.filter_size)
            kernel = (_randint(rng, space.kernel_size), _randint(rng, space.kernel_size))
            act = str(rng.choice(list(space.activations)))
            m.append(ir.Conv2D(names("conv2d"), filters, kernel, 1, 1, "same", act))
            m.append(ir.BatchNorm(names("batch_norm")))
        m.append(ir.Flatten(names("flatten")))
        for _ in range(_randint(rng, space.dense_size)):
            units = _randint(rng, space.unit_size)
            act = str(rng.choice(list(space.activations)))
            m.append(ir.Dense(names("dense"), units, act))
            m.append(ir.BatchNorm(names("batch_norm")))
    m.append(ir.Dense(names("dense"), output_dim, "softmax"))
    return CompiledSpec(m, sample_optimizer(space, rng), sample_loss(space, rng))


def make_transfer_model(space: SearchSpace, rng, parent: ir.ModelSpec, output_dim: int) -> Tuple[CompiledSpec, set]:
    """Returns the child spec and the set of layer names inherited from the parent (their weights,
    BN moving statistics included, are copied by the caller; every layer stays trainable)."""
    child = parent.clone(parent.name + "_transfer")
    remove = _randint(rng, space.transfer_layer_remove_size)
    # `layers[-remove-1].output`: keep everything up to and including that layer
    keep = len(child.layers) - remove
    if keep < 1:
        keep = 1
    child.layers = child.layers[:keep]
    inherited = {l.name for l in child.layers}
    if len(child.output_shape) > 1:          # Keras: Flatten raises ValueError on a flat tensor -> skipped
        child.append(ir.Flatten(f"flatten_{rng.random():.17f}"))
    for _ in range(_randint(rng, space.dense_size)):
        units = _randint(rng, space.unit_size)
        act = str(rng.choice(list(space.activations)))
        child.append(ir.Dense(str(rng.random()), units, act))            # name=str(np.random.rand())
        child.append(ir.BatchNorm(str(rng.random())))
    child.append(ir.Dense(str(rng.random()), output_dim, "softmax"))
    return CompiledSpec(child, sample_optimizer(space, rng), sample_loss(space, rng)), inherited
