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
