"""GPU vs CPU-oracle inference outputs of random DDL-space conv classifiers after a few training steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distributeddeeplearning_amd.config import get_space
from distributeddeeplearning_amd.models import search
from distributeddeeplearning_amd.models.classifier import RandomModel
from distributeddeeplearning_amd.models.network import Network
from distributeddeeplearning_amd.utils.data import Data

space = get_space("ddl", ["data_train_size=2000", "batch_size=128", "conv2d_size=1,4", "dense_size=0,3"])
data = Data("cuda", channels=3, n_train=2400, n_test=400, synthetic=True, seed=0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for k in range(4):
    compiled = search.make_new_model(space, rng, data.input_shape, 10)
    sample = data.get_random_sample(2000, seed=k)
    m = RandomModel(data, sample, space, compiled, "cuda", number=k, seed=k, verbose=False, use_graph=False)
    losses = m.fit(1)
    acc_val = m.get_accuracy()
    c = Network(compiled.spec, "cpu", 0)
    c.load_logical_state(m.net.logical_state())
    x = data.test_x[:200]
    yg = m.net.predict(x).cpu()
    yc = c.predict(x.cpu())
    rel = float((yg - yc).norm() / (yc.norm() + 1e-6))
    agree = float((yg.argmax(1) == yc.argmax(1)).float().mean())
    ytrain_mode = m.net.forward(m.net.prepare_input(x), training=True)
    print(f"model {k}: loss {losses[-1]:.3f} val_acc {acc_val:.3f} | infer rel err GPU vs CPU {rel:.3f}, argmax agree {agree:.2f}",
          [ (l.kind, l.cfg.get('activation'), l.cfg.get('filters'), l.cfg.get('kernel_size')) for l in compiled.spec.layers if l.kind in ('conv2d','dense')][:4], flush=True)
