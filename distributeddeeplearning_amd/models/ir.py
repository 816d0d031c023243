"""Layer-graph IR  <->  Keras 2.2.4 `architecture.json`, with Keras/TF shape rules.

The reference serialises models with `Model.to_json()` (/root/reference/DDL_Normal4.py:173-174);
five such files ship in /root/reference/models/model{0..4}/architecture.json.  This module is the
import/export side of that artifact and the single description every other
part of the framework (executor, search, transfer surgery, plotting, checkpointing) works on.

All reference networks are linear chains expressed through the functional API, so the IR keeps
`inbound` names for JSON fidelity but the executor only accepts chains (`ModelSpec.chain()`).
"""
from __future__ import annotations

import copy
import json
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from ..ops import geometry as G

KERAS_VERSION = "2.2.4"
KERAS_BACKEND = "tensorflow"

_CLASS_TO_KIND = {
    "InputLayer": "input", "Conv2D": "conv2d", "Conv2DTranspose": "conv2d_transpose",
    "DepthwiseConv2D": "depthwise_conv2d", "BatchNormalization": "batch_norm",
    "BatchNormalizationV1": "batch_norm", "Flatten": "flatten", "Dense": "dense",
    "ZeroPadding2D": "zero_padding2d", "LeakyReLU": "leaky_relu", "Reshape": "reshape",
    "Lambda": "lambda_affine", "Activation": "activation",
}
_KIND_TO_CLASS = {
    "input": "InputLayer", "conv2d": "Conv2D", "conv2d_transpose": "Conv2DTranspose",
    "depthwise_conv2d": "DepthwiseConv2D", "batch_norm": "BatchNormalization",
    "flatten": "Flatten", "dense": "Dense", "zero_padding2d": "ZeroPadding2D",
    "leaky_relu": "LeakyReLU", "reshape": "Reshape", "lambda_affine": "Lambda",
    "activation": "Activation",
}


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, int):
        return (v, v)
    return (int(v[0]), int(v[1]))


@dataclass
class LayerSpec:
    name: str
    kind: str
    cfg: Dict = field(default_factory=dict)      # normalised, kind-specific hyper-parameters
    inbound: List[str] = field(default_factory=list)
    raw: Optional[Dict] = None                    # original Keras config if imported

    def clone(self) -> "LayerSpec":
        return copy.deepcopy(self)


# ---- constructors (used by search.py / dcgan.py) -----------------------------------------
def Input(name, shape) -> LayerSpec:
    return LayerSpec(name, "input", {"shape": tuple(int(s) for s in shape)})


def Conv2D(name, filters, kernel, strides=1, dilation=1, padding="valid", activation="linear") -> LayerSpec:
    return LayerSpec(name, "conv2d", dict(filters=int(filters), kernel=_pair(kernel), strides=_pair(strides),
                                          dilation=_pair(dilation), padding=padding, activation=activation))


def Conv2DTranspose(name, filters, kernel, strides=1, dilation=1, padding="valid",
                    activation="linear") -> LayerSpec:
    return LayerSpec(name, "conv2d_transpose",
                     dict(filters=int(filters), kernel=_pair(kernel), strides=_pair(strides),
                          dilation=_pair(dilation), padding=padding, activation=activation))


def DepthwiseConv2D(name, kernel, strides=1, dilation=1, padding="valid", activation="linear") -> LayerSpec:
    return LayerSpec(name, "depthwise_conv2d",
                     dict(kernel=_pair(kernel), strides=_pair(strides), dilation=_pair(dilation),
                          padding=padding, activation=activation, depth_multiplier=1))


def BatchNorm(name, momentum=0.99, epsilon=1e-3) -> LayerSpec:
    return LayerSpec(name, "batch_norm", dict(momentum=float(momentum), epsilon=float(epsilon)))


def Flatten(name) -> LayerSpec:
    return LayerSpec(name, "flatten", {})


def Dense(name, units, activation="linear") -> LayerSpec:
    return LayerSpec(name, "dense", dict(units=int(units), activation=activation))


def ZeroPadding2D(name, padding=1) -> LayerSpec:
    p = padding
    if isinstance(p, int):
        p = ((p, p), (p, p))
    return LayerSpec(name, "zero_padding2d", dict(padding=tuple(tuple(int(a) for a in q) for q in p)))


def LeakyReLU(name, alpha=0.3) -> LayerSpec:
    return LayerSpec(name, "leaky_relu", dict(alpha=float(alpha)))


def Reshape(name, target_shape) -> LayerSpec:
    return LayerSpec(name, "reshape", dict(target_shape=tuple(int(s) for s in target_shape)))


def LambdaAffine(name, scale, shift) -> LayerSpec:
    """Lambda x: x*scale + shift  (the generator's `(x + 1) / 2`, GAN_distributed.py:70)."""
    return LayerSpec(name, "lambda_affine", dict(scale=float(scale), shift=float(shift)))


class ShapeError(ValueError):
    """Raised when a layer does not fit its input (the ValueError Keras raises, which
    /root/reference/DDL_Normal4.py:45-50 catches to retry with ZeroPadding2D)."""


@dataclass
class ModelSpec:
    name: str
    layers: List[LayerSpec]
    keras224_compat: bool = True

    # ---- structure ----------------------------------------------------------------------
    def chain(self) -> List[LayerSpec]:
        """Layers in execution order; verifies the graph is a single linear chain."""
        if not self.layers or self.layers[0].kind != "input":
            raise ValueError("model must start with an input layer")
        for prev, cur in zip(self.layers, self.layers[1:]):
            if cur.inbound and cur.inbound != [prev.name]:
                raise ValueError(f"only linear chains are executable: {cur.name} <- {cur.inbound}")
        return self.layers

    def layer(self, name: str) -> LayerSpec:
        for l in self.layers:
            if l.name == name:
                return l
        raise KeyError(name)

    def clone(self, name: Optional[str] = None) -> "ModelSpec":
        m = copy.deepcopy(self)
        if name:
            m.name = name
        return m

    def append(self, layer: LayerSpec) -> "ModelSpec":
        layer.inbound = [self.layers[-1].name] if self.layers else []
        self.layers.append(layer)
        return self

    # ---- shapes -------------------------------------------------------------------------
    def conv_geom(self, layer: LayerSpec, in_shape) -> G.ConvGeom:
        H, W, C = in_shape
        c = layer.cfg
        try:
            if layer.kind == "conv2d":
                return G.conv2d_geom(H, W, C, c["filters"], c["kernel"], c["strides"], c["dilation"], c["padding"])
            if layer.kind == "depthwise_conv2d":
                return G.depthwise_geom(H, W, C, c["kernel"], c["strides"], c["dilation"], c["padding"],
                                        self.keras224_compat)
            if layer.kind == "conv2d_transpose":
                return G.conv2d_transpose_geom(H, W, C, c["filters"], c["kernel"], c["strides"], c["dilation"],
                                               c["padding"])
        except ValueError as e:
            raise ShapeError(str(e)) from e
        raise ValueError(layer.kind)

    def out_shape(self, layer: LayerSpec, in_shape):
        k = layer.kind
        c = layer.cfg
        if k == "input":
            return tuple(c["shape"])
        if k in ("conv2d", "depthwise_conv2d"):
            if len(in_shape) != 3:
                raise ShapeError(f"{layer.name}: conv needs rank-3 (H,W,C) input, got {in_shape}")
            g = self.conv_geom(layer, in_shape)
            return (g.OH, g.OW, g.O)
        if k == "conv2d_transpose":
            if len(in_shape) != 3:
                raise ShapeError(f"{layer.name}: conv needs rank-3 (H,W,C) input, got {in_shape}")
            g = self.conv_geom(layer, in_shape)
            return (g.H, g.W, g.I)
        if k in ("batch_norm", "leaky_relu", "lambda_affine", "activation"):
            return tuple(in_shape)
        if k == "flatten":
            if len(in_shape) < 2:
                raise ShapeError("Flatten of rank-1 input")     # Keras: ValueError (DDL_Ray3.py:104-107)
            n = 1
            for s in in_shape:
                n *= s
            return (n,)
        if k == "dense":
            if len(in_shape) != 1:
                raise ShapeError(f"{layer.name}: Dense executor needs flat input, got {in_shape}")
            return (c["units"],)
        if k == "zero_padding2d":
            (pt, pb), (pl, pr) = c["padding"]
            H, W, C = in_shape
            return (H + pt + pb, W + pl + pr, C)
        if k == "reshape":
            n = 1
            for s in in_shape:
                n *= s
            t = tuple(c["target_shape"])
            m = 1
            for s in t:
                m *= s
            if m != n:
                raise ShapeError(f"reshape {in_shape} -> {t}")
            return t
        raise ValueError(f"unknown layer kind {k}")

    def infer_shapes(self) -> Dict[str, Tuple[int, ...]]:
        shapes: Dict[str, Tuple[int, ...]] = {}
        cur = None
        for l in self.chain():
            cur = self.out_shape(l, cur)
            shapes[l.name] = cur
        return shapes

    @property
    def input_shape(self):
        return tuple(self.layers[0].cfg["shape"])

    @property
    def output_shape(self):
        return self.infer_shapes()[self.layers[-1].name]

    # ---- parameters ---------------------------------------------------------------------
    def param_specs(self):
        """[(layer_name, weight_name, keras_shape, trainable)] in Keras weight order."""
        out = []
        cur = None
        for l in self.chain():
            c = l.cfg
            if l.kind == "conv2d":
                kh, kw = c["kernel"]
                out.append((l.name, "kernel", (kh, kw, cur[2], c["filters"]), True))
                out.append((l.name, "bias", (c["filters"],), True))
            elif l.kind == "conv2d_transpose":
                kh, kw = c["kernel"]
                out.append((l.name, "kernel", (kh, kw, c["filters"], cur[2]), True))
                out.append((l.name, "bias", (c["filters"],), True))
            elif l.kind == "depthwise_conv2d":
                kh, kw = c["kernel"]
                out.append((l.name, "depthwise_kernel", (kh, kw, cur[2], 1), True))
                out.append((l.name, "bias", (cur[2],), True))
            elif l.kind == "dense":
                out.append((l.name, "kernel", (cur[0], c["units"]), True))
                out.append((l.name, "bias", (c["units"],), True))
            elif l.kind == "batch_norm":
                ch = cur[-1]
                out += [(l.name, "gamma", (ch,), True), (l.name, "beta", (ch,), True),
                        (l.name, "moving_mean", (ch,), False), (l.name, "moving_variance", (ch,), False)]
            cur = self.out_shape(l, cur)
        return out

    def count_params(self, trainable_only: bool = False) -> int:
        n = 0
        for _, _, shp, tr in self.param_specs():
            if trainable_only and not tr:
                continue
            k = 1
            for s in shp:
                k *= s
            n += k
        return n

    # ---- Keras JSON ---------------------------------------------------------------------
    @staticmethod
    def from_keras_json(text_or_dict, keras224_compat: bool = True) -> "ModelSpec":
        d = json.loads(text_or_dict) if isinstance(text_or_dict, (str, bytes)) else text_or_dict
        if d.get("class_name") not in ("Model", "Sequential"):
            raise ValueError(f"unsupported top-level class {d.get('class_name')}")
        cfg = d["config"]
        layer_dicts = cfg["layers"] if isinstance(cfg, dict) else cfg
        name = cfg.get("name", "model") if isinstance(cfg, dict) else "sequential"
        layers: List[LayerSpec] = []
        for ld in layer_dicts:
            cls = ld["class_name"]
            if cls not in _CLASS_TO_KIND:
                raise ValueError(f"unsupported Keras layer class {cls}")
            kind = _CLASS_TO_KIND[cls]
            c = ld["config"]
            lname = ld.get("name", c.get("name"))
            inbound = [n[0] for n in ld["inbound_nodes"][0]] if ld.get("inbound_nodes") else []
            if kind == "input":
                n = dict(shape=tuple(c["batch_input_shape"][1:]))
            elif kind in ("conv2d", "conv2d_transpose"):
                n = dict(filters=c["filters"], kernel=_pair(c["kernel_size"]), strides=_pair(c["strides"]),
                         dilation=_pair(c.get("dilation_rate", 1)), padding=c["padding"],
                         activation=c.get("activation") or "linear")
            elif kind == "depthwise_conv2d":
                if c.get("depth_multiplier", 1) != 1:
                    raise ValueError("depth_multiplier != 1 unsupported")
                n = dict(kernel=_pair(c["kernel_size"]), strides=_pair(c["strides"]),
                         dilation=_pair(c.get("dilation_rate", 1)), padding=c["padding"],
                         activation=c.get("activation") or "linear", depth_multiplier=1)
            elif kind == "batch_norm":
                ax = c.get("axis", -1)
                if isinstance(ax, list):
                    ax = ax[0]
                n = dict(momentum=c.get("momentum", 0.99), epsilon=c.get("epsilon", 1e-3), axis=ax)
            elif kind == "dense":
                n = dict(units=c["units"], activation=c.get("activation") or "linear")
            elif kind == "zero_padding2d":
                n = ZeroPadding2D(lname, c.get("padding", 1)).cfg
            elif kind == "leaky_relu":
                n = dict(alpha=float(c.get("alpha", 0.3)))
            elif kind == "reshape":
                n = dict(target_shape=tuple(c["target_shape"]))
            elif kind == "lambda_affine":
                n = dict(scale=float(c.get("scale", 1.0)), shift=float(c.get("shift", 0.0)))
            elif kind == "activation":
                n = dict(activation=c["activation"])
            else:
                n = {}
            layers.append(LayerSpec(lname, kind, n, inbound, raw=copy.deepcopy(c)))
        return ModelSpec(name, layers, keras224_compat)

    def to_keras_dict(self) -> Dict:
        shapes = self.infer_shapes()
        layers = []
        prev = None
        for l in self.chain():
            cfg = _export_config(l, shapes[prev.name] if prev is not None else None)
            inbound = [[[prev.name, 0, 0, {}]]] if prev is not None else []
            layers.append({"name": l.name, "class_name": _KIND_TO_CLASS[l.kind], "config": cfg,
                           "inbound_nodes": inbound})
            prev = l
        return {"class_name": "Model",
                "config": {"name": self.name, "layers": layers,
                           "input_layers": [[self.layers[0].name, 0, 0]],
                           "output_layers": [[self.layers[-1].name, 0, 0]]},
                "keras_version": KERAS_VERSION, "backend": KERAS_BACKEND}

    def to_keras_json(self) -> str:
        return json.dumps(self.to_keras_dict())

    # ---- description --------------------------------------------------------------------
    def summary(self) -> str:
        shapes = self.infer_shapes()
        rows = [f"model {self.name}: {self.count_params()} params ({self.count_params(True)} trainable)"]
        for l in self.layers:
            extra = {k: v for k, v in l.cfg.items() if k not in ("shape",)}
            rows.append(f"  {l.name:28s} {l.kind:18s} -> {shapes[l.name]}  {extra}")
        return "\n".join(rows)


_GLOROT = {"class_name": "VarianceScaling",
           "config": {"scale": 1.0, "mode": "fan_avg", "distribution": "uniform", "seed": None}}
_ZEROS = {"class_name": "Zeros", "config": {}}
_ONES = {"class_name": "Ones", "config": {}}


def _export_config(l: LayerSpec, in_shape) -> Dict:
    """Emit the same key set Keras 2.2.4 writes (cf. /root/reference/models/model4/architecture.json)."""
    c = l.cfg
    k = l.kind
    if k == "input":
        return {"batch_input_shape": [None] + list(c["shape"]), "dtype": "float32", "sparse": False,
                "name": l.name}
    if k in ("conv2d", "conv2d_transpose"):
        out = {"name": l.name, "trainable": True, "filters": c["filters"], "kernel_size": list(c["kernel"]),
               "strides": list(c["strides"]), "padding": c["padding"], "data_format": "channels_last",
               "dilation_rate": list(c["dilation"]), "activation": c["activation"], "use_bias": True,
               "kernel_initializer": copy.deepcopy(_GLOROT), "bias_initializer": copy.deepcopy(_ZEROS),
               "kernel_regularizer": None, "bias_regularizer": None, "activity_regularizer": None,
               "kernel_constraint": None, "bias_constraint": None}
        if k == "conv2d_transpose":
            out["output_padding"] = None
        return out
    if k == "depthwise_conv2d":
        return {"name": l.name, "trainable": True, "kernel_size": list(c["kernel"]), "strides": list(c["strides"]),
                "padding": c["padding"], "data_format": "channels_last", "dilation_rate": list(c["dilation"]),
                "activation": c["activation"], "use_bias": True, "bias_initializer": copy.deepcopy(_ZEROS),
                "bias_regularizer": None, "activity_regularizer": None, "bias_constraint": None,
                "depth_multiplier": 1, "depthwise_initializer": copy.deepcopy(_GLOROT),
                "depthwise_regularizer": None, "depthwise_constraint": None}
    if k == "batch_norm":
        return {"name": l.name, "trainable": True, "axis": -1, "momentum": c["momentum"], "epsilon": c["epsilon"],
                "center": True, "scale": True, "beta_initializer": copy.deepcopy(_ZEROS),
                "gamma_initializer": copy.deepcopy(_ONES), "moving_mean_initializer": copy.deepcopy(_ZEROS),
                "moving_variance_initializer": copy.deepcopy(_ONES), "beta_regularizer": None,
                "gamma_regularizer": None, "beta_constraint": None, "gamma_constraint": None}
    if k == "flatten":
        return {"name": l.name, "trainable": True, "data_format": "channels_last"}
    if k == "dense":
        return {"name": l.name, "trainable": True, "units": c["units"], "activation": c["activation"],
                "use_bias": True, "kernel_initializer": copy.deepcopy(_GLOROT),
                "bias_initializer": copy.deepcopy(_ZEROS), "kernel_regularizer": None, "bias_regularizer": None,
                "activity_regularizer": None, "kernel_constraint": None, "bias_constraint": None}
    if k == "zero_padding2d":
        return {"name": l.name, "trainable": True, "padding": [list(p) for p in c["padding"]],
                "data_format": "channels_last"}
    if k == "leaky_relu":
        return {"name": l.name, "trainable": True, "alpha": c["alpha"]}
    if k == "reshape":
        return {"name": l.name, "trainable": True, "target_shape": list(c["target_shape"])}
    if k == "lambda_affine":
        # Keras marshals the Python lambda; we record the affine map it computed instead.
        return {"name": l.name, "trainable": True, "function": "affine", "scale": c["scale"], "shift": c["shift"]}
    if k == "activation":
        return {"name": l.name, "trainable": True, "activation": c["activation"]}
    raise ValueError(k)


class NameScope:
    """Keras-style global layer-name counters (`conv2d_1`, `batch_normalization_18`, ...;
    SURVEY section 2.6.3: counters are global across the five exported models)."""

    _PREFIX = {"input": "input", "conv2d": "conv2d", "conv2d_transpose": "conv2d_transpose",
               "depthwise_conv2d": "depthwise_conv2d", "batch_norm": "batch_normalization",
               "flatten": "flatten", "dense": "dense", "zero_padding2d": "zero_padding2d",
               "leaky_relu": "leaky_re_lu", "reshape": "reshape", "lambda_affine": "lambda",
               "activation": "activation", "model": "model"}

    def __init__(self):
        self.counts: Dict[str, int] = {}

    def __call__(self, kind: str) -> str:
        p = self._PREFIX[kind]
        self.counts[p] = self.counts.get(p, 0) + 1
        return f"{p}_{self.counts[p]}"
