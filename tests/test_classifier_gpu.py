"""Classifier worker on the device: imported reference replicas vs the CPU oracle, and a short
training run through the hipGraph-replayed step."""
import os

import numpy as np
import pytest
import torch

from distributeddeeplearning_amd.config import get_space
from distributeddeeplearning_amd.models import search
from distributeddeeplearning_amd.models.classifier import RandomModel
from distributeddeeplearning_amd.utils import ckpt
from distributeddeeplearning_amd.utils.data import Data

pytestmark = pytest.mark.gpu
REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "reference_models")


@pytest.mark.parametrize("i", range(5))
def test_reference_replica_forward_matches_oracle(i):
    if not os.path.isdir(f"{REF}/model{i}"):
        pytest.skip("reference checkpoints not present on this box")
    c = ckpt.load_model(f"{REF}/model{i}", "cpu")
    g = ckpt.load_model(f"{REF}/model{i}", "cuda")
    torch.manual_seed(i)
    x = torch.randint(0, 256, (16, 28, 28, 1), dtype=torch.uint8)
    pc = c.predict(x)
    pg = g.predict(x.cuda()).cpu()
    assert pg.shape == pc.shape == (16, 10)
    assert float((pg - pc).abs().max()) < 5e-2


def test_mlp_member_trains_on_device():
    sp = get_space("ddl_ray", ["data_train_size=1600", "early_stop_rate=0.0", "epochs=4"])
    data = Data("cuda", channels=3, n_train=2000, n_test=400, synthetic=True)
    rng = np.random.default_rng(3)
    cs = search.make_new_model(sp, rng, data.input_shape, 10)
    cs.optimizer.kind, cs.optimizer.lr = "adam", 1e-3
    m = RandomModel(data, data.get_random_sample(1600, seed=1), sp, cs, "cuda", number=0, seed=0, verbose=False)
    m.train()
    first, last = m.history[0][0], m.history[-1][-1]
    assert np.isfinite(last) and last < first
    acc = m.get_accuracy((data.test_x, data.test_y))
    assert 0.0 <= acc <= 1.0


def test_cnn_member_step_matches_oracle_direction():
    sp = get_space("ddl_normal4", ["data_train_size=256", "epochs=2"])
    data = Data("cuda", channels=1, n_train=320, n_test=64, synthetic=True)
    rng = np.random.default_rng(5)
    for k in range(3):
        cs = search.make_new_model(sp, rng, data.input_shape, 10)
        m = RandomModel(data, data.get_random_sample(256, seed=k), sp, cs, "cuda", number=k, seed=k, verbose=False)
        m.train()
        assert all(np.isfinite(h).all() for h in m.history), cs.spec.summary()


def test_concurrent_chunked_training_equals_serial_single_step_graphs(monkeypatch):
    """Batch-32 regime (DDL.py:92): members trained side by side on their own streams with 32-step graphs must end
    with exactly the bits of members trained one after another with one graph per step."""
    import numpy as np
    from distributeddeeplearning_amd import config
    from distributeddeeplearning_amd.models import classifier, search
    from distributeddeeplearning_amd.utils.data import Data
    monkeypatch.setenv("DDL_MLP_PERSISTENT", "0")        # this test is about the step-graph path (bf16, multi-kernel)
    sp = config.get_space("ddl_normal3", ["data_train_size=1100", "epochs=4", "early_stop_rate=0.05", "seed=3",
                                          "kernel_size=2,6", "conv2d_size=0,3"])
    data = Data("cuda", channels=1, n_train=1300, n_test=64, synthetic=True, seed=1)

    def build():
        rng = np.random.default_rng(5)
        ms = []
        for i in range(4):
            c = search.make_new_model(sp, rng, data.input_shape, 10)
            ms.append(classifier.RandomModel(data, data.get_random_sample(1100, seed=10 + i), sp, c, "cuda", i, i,
                                             verbose=False))
        return ms

    a = build()
    classifier.train_concurrently(a)
    assert any(getattr(m, "_chunk_S", 0) >= 2 for m in a)
    monkeypatch.setattr(classifier.RandomModel, "CHUNK", 0)
    b = build()
    for m in b:
        m.train()
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert x.history == y.history
        assert torch.equal(x.net.P, y.net.P) and torch.equal(x.net.S, y.net.S)
    accs = [m.get_accuracy() for m in a]
    assert max(accs) > 0.5          # the synthetic digits are learnable in a few epochs


@pytest.mark.parametrize("opt_name,n_train", [("SGD", 80), ("Adam", 80), ("RMSprop", 80)])
def test_persistent_mlp_kernel_matches_cpu_oracle(opt_name, n_train):
    """Whole-network persistent kernel (one workgroup per member, an epoch per launch, csrc/hip/mlp_persistent.hip)
    vs the fp32 CPU oracle executor: same init, same shuffles, ragged last batch (80 = 2 x 32 + 16 samples), 12
    optimizer steps.  (Measured: after 2 steps the two agree to 1e-7; on raw 0..255 pixels most sigmoid / tanh units
    saturate and BatchNorm divides by the tiny spread that is left, so rounding differences between two correct fp32
    implementations grow step after step -- 44 steps already differ by 2 % in the epoch loss.  Hence the short run.)"""
    import numpy as np
    from distributeddeeplearning_amd import config
    from distributeddeeplearning_amd.models import classifier, mlp_fast, search
    sp = config.get_space("ddl_ray3", [f"data_train_size={n_train}", "epochs=4", "early_stop_rate=-1.0", "seed=2",
                                       f"optimizers={opt_name}", "lr_exponent=-3,-2"])
    dg = Data("cuda", channels=3, n_train=400, n_test=64, synthetic=True, seed=4)
    dc = Data("cpu", channels=3, n_train=400, n_test=64, synthetic=True, seed=4)
    rng = np.random.default_rng(8)
    specs = [search.make_new_model(sp, rng, dg.input_shape, 10) for _ in range(4)]
    assert len({len(c.spec.layers) for c in specs}) > 1           # different depths in one launch
    gm = [classifier.RandomModel(dg, dg.get_random_sample(n_train, seed=20 + i), sp, c, "cuda", i, i, verbose=False)
          for i, c in enumerate(specs)]
    cm = [classifier.RandomModel(dc, dc.get_random_sample(n_train, seed=20 + i), sp, c, "cpu", i, i, verbose=False)
          for i, c in enumerate(specs)]
    assert all(mlp_fast.eligible(m) for m in gm)
    classifier.train_concurrently(gm)            # -> mlp_fast group: 4 members, one launch per epoch
    # the oracle's own sensitivity: the same run from parameters perturbed by 1e-6 (relative).  Deep members on raw
    # 0..255 pixels are chaotic (saturated units behind BatchNorm): 1e-7 already moves the epoch losses by 0.1 after
    # 12 steps, so agreement is demanded to the extent the oracle agrees with itself.
    sens = []
    for i, (c, m) in enumerate(zip(specs, cm)):
        p = classifier.RandomModel(dc, dc.get_random_sample(n_train, seed=20 + i), sp, c, "cpu", i, i, verbose=False)
        p.net.P.mul_(1.0 + 1e-6)
        p.train()
        m.train()
        sens.append(float(np.abs(np.array(p.history) - np.array(m.history)).max()))
    assert min(sens) < 1e-4                        # at least one well-conditioned member pins the kernel tightly
    for g, c, sn in zip(gm, cm, sens):
        assert len(g.history) == len(c.history) == 2
        hg, hc = np.array(g.history), np.array(c.history)
        if sn > 1e-4:           # chaotic member: comparable only up to the oracle's own spread; must stay sane
            assert np.isfinite(hg).all() and np.abs(hg - hc).max() < 0.05 + 5 * sn, (hg, hc, sn)
            continue
        assert np.allclose(hg, hc, rtol=2e-3, atol=2e-3), (hg, hc, sn)
        sg, sc = g.net.logical_state(), c.net.logical_state()
        for k in sc:
            ref = sc[k].float()
            err = (sg[k].float() - ref).abs().max().item()
            assert err <= 2e-3 * max(1.0, ref.abs().max().item()), (k, err)
        assert g.opt.steps_taken() == c.opt.t == 4 * (-(-n_train // 32))
        # the framework sees an ordinary trained member: bf16 working copies refreshed, predict works
        pg = g.predict(dg.test_x).cpu()
        pc = c.predict(dc.test_x)
        assert (pg - pc).abs().max().item() < 5e-2


def test_persistent_mlp_kernel_covers_the_deepest_ddl_member():
    """/root/reference/DDL.py:19-20,31-32: `dense_size` draws up to 9 hidden Dense layers, plus the softmax head = 10
    Dense layers, each up to 29 units: such a member must run in the persistent kernel (MLP_MAXL = 11) and agree with
    the fp32 CPU oracle after a few steps."""
    import numpy as np
    from distributeddeeplearning_amd import config
    from distributeddeeplearning_amd.models import classifier, mlp_fast, search
    sp = config.get_space("ddl", ["data_train_size=48", "epochs=2", "early_stop_rate=-1.0", "seed=2", "conv2d_size=0,1",
                                  "dense_size=9,10", "optimizers=Adam", "lr_exponent=-3,-2", "activations=relu,softplus"])
    dg = Data("cuda", channels=3, n_train=200, n_test=64, synthetic=True, seed=4)
    dc = Data("cpu", channels=3, n_train=200, n_test=64, synthetic=True, seed=4)
    c = search.make_new_model(sp, np.random.default_rng(1), dg.input_shape, 10)
    g = classifier.RandomModel(dg, dg.get_random_sample(48, seed=3), sp, c, "cuda", 0, 0, verbose=False)
    o = classifier.RandomModel(dc, dc.get_random_sample(48, seed=3), sp, c, "cpu", 0, 0, verbose=False)
    chain = mlp_fast._chain(g.net)
    assert chain is not None and len(chain) == 10 and mlp_fast.limits()[0] >= 10
    assert mlp_fast.eligible(g)
    classifier.train_concurrently([g])
    o.train()
    hg, hc = np.array(g.history), np.array(o.history)
    assert hg.shape == hc.shape == (1, 2) and np.isfinite(hg).all()
    assert np.allclose(hg, hc, rtol=5e-3, atol=5e-3), (hg, hc)
    assert g.opt.steps_taken() == o.opt.t == 4


def test_member_streams_places_longest_first_on_least_loaded_stream():
    """Concurrent ensemble members share `k` streams (one per hardware queue): sorted by measured step time, each goes
    to the stream with the least work so far -- no stream ends up with the three slowest members."""
    from distributeddeeplearning_amd.models.classifier import member_streams

    class _M:
        def __init__(self, ms):
            self.chunk_ms = ms
    costs = [9.0, 1.0, 8.0, 2.0, 7.0, 3.0, 6.0, 1.5, 0.5, 4.0]
    members = [_M(c) for c in costs]
    assign, pool = member_streams(members, torch.device("cuda"), 4)
    assert len(pool) == 4 and len(assign) == len(members) and all(s in pool for s in assign)
    load = {id(s): 0.0 for s in pool}
    for m, s in zip(members, assign):
        load[id(s)] += m.chunk_ms
    assert max(load.values()) <= 1.15 * (sum(costs) / 4)           # longest-first: near the balanced optimum
    assert len({id(assign[i]) for i in (0, 2, 4, 6)}) == 4           # the four slowest on four different streams
    one_each, pool1 = member_streams(members, torch.device("cuda"), 0)
    assert len(pool1) == len(members)
