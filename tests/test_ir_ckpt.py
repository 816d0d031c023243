"""Keras-2.2.4 artifact compatibility: architecture.json <-> IR, weights.h5 <-> flat buffers
(SURVEY 2.6.3, 2.7, A.4, A.5), checked against the five replicas the reference ships."""
import json
import os

import numpy as np
import pytest
import torch

from distributeddeeplearning_amd.models.ir import ModelSpec
from distributeddeeplearning_amd.utils import ckpt, h5

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "reference_models")

# Flatten sizes / last-conv shapes printed in the reference's graph.png files
EXPECT = {0: ((26, 26, 128), 86528, 878602), 1: ((9, 9, 32), 2592, None), 2: ((13, 13, 32), 5408, 54826),
          3: ((34, 34, 32), 36992, 399186), 4: ((5, 5, 128), 3200, 54586)}


def _spec(i):
    return ModelSpec.from_keras_json(open(f"{REF}/model{i}/architecture.json").read())


@pytest.mark.parametrize("i", range(5))
def test_shapes_match_reference_graphs(i):
    spec = _spec(i)
    shapes = spec.infer_shapes()
    flat = [l for l in spec.layers if l.kind == "flatten"][0]
    before = spec.layers[spec.layers.index(flat) - 1]
    last_shape, flat_n, nparams = EXPECT[i]
    assert shapes[before.name] == last_shape
    assert shapes[flat.name] == (flat_n,)
    assert spec.output_shape == (10,)
    if nparams is not None:
        assert spec.count_params() == nparams


def test_intermediate_shapes_model0_and_model3():
    s0 = _spec(0).infer_shapes()
    assert [s0[n] for n in ("conv2d_transpose_1", "conv2d_transpose_2", "depthwise_conv2d_1",
                            "depthwise_conv2d_2")] == [(57, 57, 64), (57, 57, 128), (28, 28, 128), (26, 26, 128)]
    s3 = _spec(3).infer_shapes()      # Keras `valid` transposed conv: in*s + max(k_eff - s, 0)  (A.4)
    assert s3["conv2d_transpose_4"] == (16, 16, 128) and s3["conv2d_transpose_5"] == (34, 34, 32)


def test_strict_mode_accounts_for_depthwise_dilation():
    spec = ModelSpec.from_keras_json(open(f"{REF}/model0/architecture.json").read(), keras224_compat=False)
    assert spec.infer_shapes()["depthwise_conv2d_1"] == (27, 27, 128)      # A.5: Keras 2.2.4 said 28


@pytest.mark.parametrize("i", range(5))
def test_json_roundtrip(i):
    orig = json.load(open(f"{REF}/model{i}/architecture.json"))
    spec = ModelSpec.from_keras_json(orig)
    again = spec.to_keras_dict()
    assert again["keras_version"] == "2.2.4" and again["backend"] == "tensorflow"
    assert [l["name"] for l in again["config"]["layers"]] == [l["name"] for l in orig["config"]["layers"]]
    for lo, ln in zip(orig["config"]["layers"], again["config"]["layers"]):
        assert lo["class_name"] == ln["class_name"]
        assert lo["inbound_nodes"] == ln["inbound_nodes"]
        for k, v in lo["config"].items():            # every key Keras wrote is reproduced
            assert k in ln["config"], (lo["name"], k)
            if k in ("filters", "kernel_size", "strides", "padding", "dilation_rate", "activation", "units",
                     "momentum", "epsilon", "batch_input_shape", "depth_multiplier", "axis"):
                assert ln["config"][k] == v, (lo["name"], k)
    assert ModelSpec.from_keras_json(spec.to_keras_json()).infer_shapes() == spec.infer_shapes()


@pytest.mark.parametrize("i", range(5))
def test_h5_matches_architecture(i):
    spec = _spec(i)
    layers, weights, meta = h5.read_keras_weights(f"{REF}/model{i}/weights.h5")
    assert layers == [l.name for l in spec.layers]
    assert meta["keras_version"] == "2.2.4" and meta["backend"] == "tensorflow"
    want = {(l, w): shp for l, w, shp, _ in spec.param_specs()}
    assert set(weights) == set(want)
    for k, shp in want.items():
        assert weights[k].shape == tuple(shp) and weights[k].dtype == np.float32
    # untrained Keras init: BN gamma 1 / beta 0 / mean 0 / var 1, biases 0 (A.10)
    for (l, w), a in weights.items():
        if w in ("gamma", "moving_variance"):
            assert np.all(a == 1)
        if w in ("beta", "moving_mean", "bias"):
            assert np.all(a == 0)


def test_h5_write_read_bit_exact(tmp_path):
    layers, weights, _ = h5.read_keras_weights(f"{REF}/model2/weights.h5")
    spec = _spec(2)
    ordered = [(l, w, weights[(l, w)]) for l, w, _, _ in spec.param_specs()]
    h5.write_keras_weights(tmp_path / "w.h5", layers, ordered)
    layers2, weights2, meta2 = h5.read_keras_weights(tmp_path / "w.h5")
    assert layers2 == layers and meta2["keras_version"] == "2.2.4"
    for k in weights:
        assert np.array_equal(weights[k], weights2[k])


def test_h5_rejects_garbage(tmp_path):
    p = tmp_path / "bad.h5"
    p.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(h5.H5Error):
        h5.H5File(p)


@pytest.mark.parametrize("i", [1, 2, 4])
def test_load_save_load_predictions_identical(i, tmp_path):
    net = ckpt.load_model(f"{REF}/model{i}", "cpu")
    torch.manual_seed(0)
    x = torch.randint(0, 256, (8, 28, 28, 1)).float()
    p0 = net.predict(x)
    assert p0.shape == (8, 10) and torch.allclose(p0.sum(1), torch.ones(8), atol=1e-5)
    d = ckpt.save_model(net, tmp_path / "m")
    assert {f.name for f in d.iterdir()} >= {"architecture.json", "weights.h5", "graph.png", "graph.dot"}
    p1 = ckpt.load_model(d, "cpu").predict(x)
    assert torch.equal(p0, p1)
    # exported weights are byte-identical to the reference file's datasets
    _, w0, _ = h5.read_keras_weights(f"{REF}/model{i}/weights.h5")
    _, w1, _ = h5.read_keras_weights(d / "weights.h5")
    for k in w0:
        assert np.array_equal(w0[k], w1[k]), k


def test_training_state_resume(tmp_path):
    from distributeddeeplearning_amd.models.optim import FlatOptimizer, OptimConfig
    net = ckpt.load_model(f"{REF}/model2", "cpu")
    opt = FlatOptimizer(net, OptimConfig.keras("adam", 1e-3))
    net.G.normal_(generator=torch.Generator().manual_seed(0))
    opt.step()
    ckpt.save_training_state(tmp_path / "s.pt", net, opt, {"epoch": 3})
    net2 = ckpt.load_model(f"{REF}/model2", "cpu")
    opt2 = FlatOptimizer(net2, OptimConfig.keras("adam", 1e-3))
    extra = ckpt.load_training_state(tmp_path / "s.pt", net2, opt2)
    assert extra == {"epoch": 3} and opt2.t == 1
    def same(a, b):
        sa, sb = a.logical_state(), b.logical_state()
        return all(torch.equal(sa[k], sb[k]) for k in sa)
    assert same(net, net2) and torch.equal(opt.m, opt2.m) and torch.equal(opt.v, opt2.v)
    net.G.normal_(generator=torch.Generator().manual_seed(1))
    net2.G.copy_(net.G)
    opt.step(); opt2.step()
    assert same(net, net2)          # the resumed optimizer continues bit-identically
