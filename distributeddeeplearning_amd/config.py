"""Configuration: every knob the reference hard-codes in its per-script class `C`.

The reference's "config system" is a class of static samplers and constants edited in place
(/root/reference/DDL.py:13-54, DDL_Ray.py:18-59, DDL_Ray2.py:53-59, DDL_Ray3.py:39-64,
DDL_Normal3.py:13-59, DDL_Normal4.py:14-78).  Here each script variant is a named
`SearchSpace` preset of one dataclass; any field can be overridden from the command line
(`--set field=value`) or the environment (`DDL_<FIELD>=value`), and there is an explicit seed
(the reference never seeds its RNG).
"""
from __future__ import annotations

import dataclasses
import json
import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

ACTIVATIONS = ("relu", "sigmoid", "softplus", "tanh", "softsign")
OPTIMIZERS = ("Adam", "SGD", "RMSprop")


@dataclass
class SearchSpace:
    """Half-open integer ranges follow np.random.randint(lo, hi) exactly as in the reference."""
    name: str = "ddl"
    # --- v1..v3 "Conv*n -> Flatten -> Dense*m -> softmax" space
    conv2d_size: Tuple[int, int] = (0, 10)
    dense_size: Tuple[int, int] = (0, 10)
    filter_size: Tuple[int, int] = (3, 20)
    kernel_size: Tuple[int, int] = (3, 30)           # each of (kh, kw) drawn independently
    unit_size: Tuple[int, int] = (3, 30)
    activations: Tuple[str, ...] = ACTIVATIONS
    optimizers: Tuple[str, ...] = OPTIMIZERS
    lr_exponent: Tuple[int, int] = (-3, -1)           # lr = 10 ** randint(lo, hi)
    losses: Tuple[str, ...] = ("categorical_crossentropy",)
    transfer_layer_remove_size: Tuple[int, int] = (1, 2)
    # --- v4 random_layer space (DDL_Normal4.py:16-53)
    layer_space: str = "conv_dense"                   # "conv_dense" | "random_layer"
    layer_size: Tuple[int, int] = (2, 5)
    filters_log2: Tuple[int, int] = (3, 8)            # filters = 2 ** randint
    rl_kernel: Tuple[int, int] = (1, 4)
    rl_stride: Tuple[int, int] = (1, 4)
    rl_dilation: Tuple[int, int] = (1, 4)
    # --- ensemble sizes / schedule
    random_model_size: int = 10
    transfer_model_size: int = 0
    predict_model_size: int = 5
    epochs: int = 10
    mini_epochs: int = 2
    early_stop_rate: float = 0.05
    data_train_size: int = 50000
    batch_size: int = 32                              # Keras fit() default
    channels: int = 3                                 # MNIST stacked x3 (DDL.py:119) or 1
    model_save_directory: str = "./models"
    seed: int = 0


def _preset(**kw) -> SearchSpace:
    return SearchSpace(**kw)


PRESETS: Dict[str, SearchSpace] = {
    # /root/reference/DDL.py:13-54
    "ddl": _preset(name="ddl"),
    # /root/reference/DDL_Ray.py:18-59 -- debug-shrunk: MLP only, 1000 samples, stop after round 1
    "ddl_ray": _preset(name="ddl_ray", conv2d_size=(0, 1), dense_size=(0, 7), unit_size=(3, 20),
                       random_model_size=10, predict_model_size=5, early_stop_rate=5.0, data_train_size=1000),
    # /root/reference/DDL_Ray2.py:18-59
    "ddl_ray2": _preset(name="ddl_ray2", conv2d_size=(0, 1), dense_size=(0, 7), unit_size=(3, 20),
                        random_model_size=5, predict_model_size=3, early_stop_rate=5.0, data_train_size=1000),
    # /root/reference/DDL_Ray3.py:18-64
    "ddl_ray3": _preset(name="ddl_ray3", conv2d_size=(0, 1), dense_size=(0, 7), unit_size=(3, 20),
                        random_model_size=5, transfer_model_size=3, predict_model_size=3,
                        transfer_layer_remove_size=(1, 2), early_stop_rate=5.0, data_train_size=1000),
    # /root/reference/DDL_Normal3.py:13-59
    "ddl_normal3": _preset(name="ddl_normal3", conv2d_size=(0, 5), dense_size=(0, 5), kernel_size=(2, 10),
                           transfer_layer_remove_size=(1, 3), random_model_size=3, transfer_model_size=2,
                           predict_model_size=3, channels=1),
    # /root/reference/DDL_Normal4.py:14-78
    "ddl_normal4": _preset(name="ddl_normal4", layer_space="random_layer", lr_exponent=(-4, -1),
                           losses=("categorical_crossentropy", "mean_squared_error"),
                           transfer_layer_remove_size=(1, 3), random_model_size=3, transfer_model_size=2,
                           predict_model_size=3, channels=1, dense_size=(0, 5)),
}


@dataclass
class GANSettings:
    """Literals of /root/reference/GAN_distributed.py:8-11,93,133,137."""
    noise_dimension: int = 128
    train_images: int = 1000
    learning_rate: float = 2e-4
    iterations: int = 10000
    log_period: int = 100
    sample_images: int = 10
    image_directory: str = "./images"
    size: int = 28
    reference_exact: bool = False
    seed: int = 0
    bn_policy: str = "split"      # split | inference | mixed (models/dcgan.py); reference_exact forces inference


def _coerce(old, text: str):
    if isinstance(old, bool):
        return text.lower() in ("1", "true", "yes", "on")
    if isinstance(old, int):
        return int(text)
    if isinstance(old, float):
        return float(text)
    if isinstance(old, tuple):
        v = json.loads(text) if text.strip().startswith("[") else text.split(",")
        proto = old[0] if old else ""
        return tuple(type(proto)(x) if not isinstance(proto, str) else str(x).strip() for x in v)
    return text


def apply_overrides(cfg, overrides: Sequence[str] = (), env_prefix: str = "DDL_"):
    """`field=value` strings and DDL_<FIELD> environment variables -> a new config object."""
    changes = {}
    fields = {f.name: f for f in dataclasses.fields(cfg)}
    for name in fields:
        ev = os.environ.get(env_prefix + name.upper())
        if ev is not None:
            changes[name] = _coerce(getattr(cfg, name), ev)
    for item in overrides:
        if "=" not in item:
            raise ValueError(f"override {item!r} is not field=value")
        k, v = item.split("=", 1)
        if k not in fields:
            raise KeyError(f"unknown config field {k!r}; known: {sorted(fields)}")
        changes[k] = _coerce(getattr(cfg, k), v)
    return dataclasses.replace(cfg, **changes)


def get_space(name: str, overrides: Sequence[str] = ()) -> SearchSpace:
    if name not in PRESETS:
        raise KeyError(f"unknown search space {name!r}; known: {sorted(PRESETS)}")
    return apply_overrides(dataclasses.replace(PRESETS[name]), overrides)
