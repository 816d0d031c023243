"""Per-replica checkpoint directory `models/model<N>/{architecture.json, weights.h5, graph.png}`.

Export side = /root/reference/DDL_Normal4.py:166-177.  The reference has no loader at all
(SURVEY 5.4); `load_model` adds it: architecture.json -> IR, weights.h5 -> flat parameter buffer
(Keras HWIO / (kh,kw,out,in) / (in,out) layouts converted to the kernels' K-contiguous layouts,
channel padding applied on device).  `save_training_state` / `load_training_state` add the
resume capability the reference lacks (optimizer moments + step + RNG in one .pt file).
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Optional

import torch

from ..models.ir import ModelSpec
from ..models.network import Network
from . import h5, plot


def save_model(net: Network, directory, graph: bool = True) -> Path:
    d = Path(directory)
    d.mkdir(parents=True, exist_ok=True)          # the reference ignores FileExistsError
    spec = net.spec
    layer_names = [l.name for l in spec.layers]
    h5.write_keras_weights(d / "weights.h5", layer_names, net.get_keras_weights())
    (d / "architecture.json").write_text(spec.to_keras_json())
    if graph:
        plot.plot_model(spec, d / "graph.png", show_shapes=True)
    return d


def load_model(directory, device="cpu", keras224_compat: bool = True, bn_mode: str = "batch",
               strict: bool = True) -> Network:
    d = Path(directory)
    spec = ModelSpec.from_keras_json((d / "architecture.json").read_text(), keras224_compat)
    net = Network(spec, device, seed=0, bn_mode=bn_mode)
    _, weights, _ = h5.read_keras_weights(d / "weights.h5")
    net.set_keras_weights(weights, strict=strict)
    return net


def save_training_state(path, net: Network, opt=None, extra: Optional[dict] = None, write: bool = True):
    sd = {"logical": net.logical_state(), "arch": net.spec.to_keras_json(), "extra": extra or {}}
    if opt is not None:
        # state_dict() reads the DEVICE step counter (graph replays never touch the host mirror) and, for a
        # sharded optimizer, all-gathers the shards: every rank calls this, `write=False` on all but one
        o = opt.state_dict()
        sd["opt"] = {"t": o["t"], "cfg": o["cfg"],
                     "m": None if o["m"] is None else o["m"].detach().cpu(),
                     "v": None if o["v"] is None else o["v"].detach().cpu(),
                     "padded_layout": net.cpad}
    if not write:
        return
    tmp = str(path) + ".tmp"
    torch.save(sd, tmp)
    os.replace(tmp, path)             # atomic: a crash never leaves a torn checkpoint


def load_training_state(path, net: Network, opt=None) -> dict:
    sd = torch.load(path, map_location="cpu", weights_only=True)
    net.load_logical_state(sd["logical"])
    if opt is not None and "opt" in sd:
        o = sd["opt"]
        if o["padded_layout"] != net.cpad:
            raise ValueError("optimizer state was saved from a different device layout "
                             "(cpu <-> gpu); weights were restored, moments were not")
        opt.load_state_dict(o)
    return sd.get("extra", {})
