"""Installation smoke checks -- the reference's Test.py (one Keras fit step on zeros) and
RayTest.py (three 10-second tasks must take ~10 s, not 30 s, on a working cluster).

    python -m distributeddeeplearning_amd.cli.selftest [--device cpu|cuda] [--sleep 1.0]
"""
from __future__ import annotations

import argparse
import sys
import time

import torch


def _sleep_task(seconds):
    time.sleep(seconds)
    return seconds


def fit_one_step(device="cpu"):
    """Test.py: Conv2D(1 filter, 3x3) -> Flatten -> Dense(10), mean_squared_error, one batch of zeros."""
    from ..models import ir
    from ..models.network import Network
    from ..models.optim import FlatOptimizer, OptimConfig
    from ..ops import functional as F
    m = ir.ModelSpec("selftest", [ir.Input("input_1", (28, 28, 1))])
    m.append(ir.Conv2D("conv2d_1", 1, 3, 1, 1, "valid"))
    m.append(ir.Flatten("flatten_1"))
    m.append(ir.Dense("dense_1", 10, "softmax"))
    net = Network(m, device, seed=0)
    opt = FlatOptimizer(net, OptimConfig.keras("sgd", 1e-2))
    x = net.prepare_input(torch.zeros(1, 28, 28, 1, device=device))
    y = torch.zeros(1, 10, device=device)
    logits = net.forward(x, training=True)
    loss, dl, _ = F.softmax_mse(logits, y)
    net.backward(dl)
    opt.step()
    return float(loss)


def parallel_tasks(workers=3, seconds=1.0):
    from ..parallel import runtime
    with runtime.Runtime(workers, ["cpu"] * workers) as rt:
        f = rt.remote(_sleep_task)
        rt.get([f.remote(0.0) for _ in range(workers)])      # first call pays the worker's imports
        t0 = time.time()
        rt.get([f.remote(seconds) for _ in range(workers)])
        return time.time() - t0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--sleep", type=float, default=1.0)
    a = ap.parse_args(argv)
    print("fit step loss:", fit_one_step(a.device))
    dt = parallel_tasks(3, a.sleep)
    print(f"3 parallel {a.sleep:.1f}s tasks took {dt:.2f}s", "(parallel: ok)" if dt < 2 * a.sleep else "(SERIAL!)")
    return 0 if dt < 2 * a.sleep else 1


if __name__ == "__main__":
    sys.exit(main())
