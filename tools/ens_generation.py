#!/usr/bin/env python3
"""Wall-clock of ONE ensemble generation in the reference's regime on one GPU: N random members of a preset,
50,000 training images, batch 32, <= 5 rounds x 2 epochs with the reference's early-stop rule (DDL.py:90-96).
Only `make_and_train_new_models` is timed (dataset synthesis, imports and evaluation are outside).

    python tools/ens_generation.py --preset ddl_ray3 [--members 10] [--train 50000]
Knobs (environment): DDL_MLP_PERSISTENT=0 (no whole-network kernel), DDL_SERIAL_MEMBERS=1 (one member at a time),
DDL_CHUNK=1 (one graph per step) -- all three together are round 1's serial loop.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd import config  # noqa: E402
from distributeddeeplearning_amd.parallel.ensemble import EnsembleModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="ddl_ray3")
ap.add_argument("--members", type=int, default=10)
ap.add_argument("--train", type=int, default=50000)
ap.add_argument("--early-stop", type=float, default=0.05)
a = ap.parse_args()
sp = config.get_space(a.preset, [f"data_train_size={a.train}", f"random_model_size={a.members}",
                                 f"early_stop_rate={a.early_stop}", "seed=0"])
ens = EnsembleModel(sp, "local", device="cuda", data_kw={"synthetic": True, "n_train": a.train + 10000, "n_test": 2000},
                    verbose=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
ens.make_and_train_new_models()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rounds = [len(e[1]._a.model.history) for e in ens.accuracy_with_models]
steps = sum(r * sp.mini_epochs * -(-a.train // sp.batch_size) for r in rounds)
imgs = sum(r * sp.mini_epochs * a.train for r in rounds)
knobs = {k: os.environ.get(k) for k in ("DDL_MLP_PERSISTENT", "DDL_SERIAL_MEMBERS", "DDL_CHUNK") if os.environ.get(k)}
print(f"preset {a.preset} members {a.members} knobs {knobs}: generation {dt:.2f} s, rounds per member {rounds}, "
      f"{steps} optimizer steps, {imgs / dt:,.0f} images/s, {dt / steps * 1e6:.1f} us per member-step, "
      f"val acc {[round(float(e[0]), 3) for e in ens.accuracy_with_models]}", flush=True)
