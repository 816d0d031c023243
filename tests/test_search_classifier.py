"""Search space `C`, random_layer retry, transfer surgery, RandomModel training semantics."""
import numpy as np
import pytest
import torch

from distributeddeeplearning_amd import config
from distributeddeeplearning_amd.models import ir, search
from distributeddeeplearning_amd.models.classifier import RandomModel
from distributeddeeplearning_amd.utils.data import Data, one_hot


def test_presets_mirror_reference_constants():
    c = config.get_space("ddl")
    assert (c.random_model_size, c.predict_model_size, c.epochs, c.mini_epochs) == (10, 5, 10, 2)
    assert c.early_stop_rate == 0.05 and c.data_train_size == 50000 and c.channels == 3
    r = config.get_space("ddl_ray")
    assert r.conv2d_size == (0, 1) and r.early_stop_rate == 5.0 and r.data_train_size == 1000
    r3 = config.get_space("ddl_ray3")
    assert (r3.random_model_size, r3.transfer_model_size, r3.predict_model_size) == (5, 3, 3)
    n4 = config.get_space("ddl_normal4")
    assert n4.layer_space == "random_layer" and "mean_squared_error" in n4.losses and n4.channels == 1


def test_overrides_cli_and_env(monkeypatch):
    monkeypatch.setenv("DDL_EPOCHS", "4")
    c = config.get_space("ddl", ["random_model_size=3", "early_stop_rate=0.5", "kernel_size=2,5",
                                 "activations=relu,tanh"])
    assert c.epochs == 4 and c.random_model_size == 3 and c.early_stop_rate == 0.5
    assert c.kernel_size == (2, 5) and c.activations == ("relu", "tanh")
    with pytest.raises(KeyError):
        config.get_space("ddl", ["nope=1"])


def test_conv_dense_space_structure_and_ranges():
    sp = config.get_space("ddl")
    rng = np.random.default_rng(0)
    for _ in range(20):
        cs = search.make_new_model(sp, rng, (28, 28, 3), 10)
        kinds = [l.kind for l in cs.spec.layers]
        assert kinds[0] == "input" and kinds[-1] == "dense" and cs.spec.layers[-1].cfg["activation"] == "softmax"
        f = kinds.index("flatten")
        assert all(k in ("conv2d", "batch_norm") for k in kinds[1:f])
        for l in cs.spec.layers:
            if l.kind == "conv2d":
                assert 3 <= l.cfg["filters"] < 20 and all(3 <= k < 30 for k in l.cfg["kernel"])
                assert l.cfg["padding"] == "same"
        assert cs.optimizer.kind in ("adam", "sgd", "rmsprop") and cs.optimizer.lr in (1e-3, 1e-2)
        assert cs.spec.output_shape == (10,)
    # same seed -> same architectures (the reference was unseeded; we are reproducible)
    a = search.make_new_model(sp, np.random.default_rng(5), (28, 28, 3), 10).spec.to_keras_json()
    b = search.make_new_model(sp, np.random.default_rng(5), (28, 28, 3), 10).spec.to_keras_json()
    assert a == b


def test_random_layer_retries_with_zero_padding():
    sp = config.get_space("ddl_normal4")
    rng = np.random.default_rng(1)
    seen_pad = False
    for _ in range(60):
        m = ir.ModelSpec("m", [ir.Input("in", (2, 2, 4))])           # tiny input forces the retry loop
        search.random_layer(sp, rng, m, ir.NameScope())
        kinds = [l.kind for l in m.layers]
        assert kinds[-1] == "batch_norm" and kinds[-2] in ("conv2d", "depthwise_conv2d", "conv2d_transpose")
        seen_pad |= "zero_padding2d" in kinds
        conv = m.layers[-2]
        if conv.kind == "conv2d":
            assert conv.cfg["strides"] == (1, 1) or conv.cfg["dilation"] == (1, 1)
            assert conv.cfg["filters"] in (8, 16, 32, 64, 128)
        m.infer_shapes()
    assert seen_pad


def test_transfer_surgery_inherits_weights_and_keeps_names():
    sp = config.get_space("ddl_normal3")
    rng = np.random.default_rng(2)
    data = Data("cpu", channels=1, n_train=200, n_test=40, synthetic=True)
    parent_c = search.make_new_model(sp, rng, data.input_shape, 10)
    parent = RandomModel(data, data.get_random_sample(160, seed=0), sp, parent_c, "cpu", 0, seed=3, verbose=False)
    compiled, state = parent.get_model()
    child_c, inherited = search.make_transfer_model(sp, rng, compiled.spec, 10)
    n_removed = len(compiled.spec.layers) - len(inherited)
    assert n_removed in (1, 2) and child_c.spec.layers[-1].cfg["activation"] == "softmax"
    assert child_c.spec.output_shape == (10,)
    child = RandomModel(data, data.get_random_sample(160, seed=1), sp, child_c, "cpu", 1, seed=99,
                        parent_state=state, inherited=inherited, verbose=False)
    cs = child.net.logical_state()
    for (layer, w), v in state.items():
        if layer in inherited and (layer, w) in cs:
            assert torch.equal(cs[(layer, w)], v), (layer, w)
    new_layers = [l.name for l in child_c.spec.layers if l.name not in inherited]
    assert new_layers and all(n not in {l.name for l in compiled.spec.layers} for n in new_layers)


def test_one_hot_and_random_sample_split():
    y = one_hot(torch.tensor([[3], [0], [9]]), 10)
    assert y.shape == (3, 10) and y[0, 3] == 1 and y.sum() == 3
    data = Data("cpu", channels=3, n_train=300, n_test=50, synthetic=True)
    assert data.train_x.dtype == torch.uint8 and data.train_x.shape == (300, 28, 28, 1)     # raw, unnormalised
    assert data.input_shape == (28, 28, 3)
    s = data.get_random_sample(250, seed=4)
    assert s.train_idx.numel() == 250 and s.val_idx.numel() == 50
    assert torch.equal(torch.cat([s.train_idx, s.val_idx]).sort().values, torch.arange(300))
    s2 = data.get_random_sample(250, seed=4)
    assert torch.equal(s.train_idx, s2.train_idx)


def _mlp_member(space_over=(), n=600, seed=0):
    sp = config.get_space("ddl_ray", [f"data_train_size={n - 100}", *space_over])
    data = Data("cpu", channels=3, n_train=n, n_test=100, synthetic=True)
    spec = ir.ModelSpec("m", [ir.Input("in", data.input_shape)])
    for l in (ir.Flatten("f"), ir.Dense("d1", 24, "relu"), ir.BatchNorm("b1"), ir.Dense("o", 10, "softmax")):
        spec.append(l)
    from distributeddeeplearning_amd.models.optim import OptimConfig
    cs = search.CompiledSpec(spec, OptimConfig.keras("adam", 1e-3), "categorical_crossentropy")
    return RandomModel(data, data.get_random_sample(n - 100, seed=seed), sp, cs, "cpu", 0, seed=seed, verbose=False)


def test_early_stopping_rule():
    m = _mlp_member(["early_stop_rate=5.0"])            # DDL_Ray debug value: always stop after round 1
    m.train()
    assert len(m.history) == 1 and len(m.history[0]) == 2
    m = _mlp_member(["early_stop_rate=-100.0", "epochs=6"])      # never stop: epochs/mini_epochs rounds
    m.train()
    assert len(m.history) == 3
    assert m.history[-1][-1] < m.history[0][0]


def test_member_learns_and_accuracy_api():
    m = _mlp_member(["early_stop_rate=-1.0", "epochs=8"], n=1200)
    m.train()
    acc_val = m.get_accuracy()
    acc_test = m.get_accuracy((m.data.test_x, m.data.test_y))
    assert acc_val > 0.5 and acc_test > 0.5                 # synthetic digits are easily separable
    p = m.predict(m.data.test_x)
    assert p.shape == (100, 10) and torch.allclose(p.sum(1), torch.ones(100), atol=1e-4)


def test_fit_keeps_partial_last_batch_and_reports_sample_weighted_mean():
    m = _mlp_member(n=170)                                   # 70 training samples, batch 32 -> 32+32+6
    seen = []
    orig = m.train_step
    m.train_step = lambda idx: (seen.append(idx.numel()), orig(idx))[1]
    losses = m.fit(1)
    assert seen == [32, 32, 6] and len(losses) == 1 and np.isfinite(losses[0])
