"""Ensemble-classifier entry point covering every DDL*.py script of the reference.

    python -m distributeddeeplearning_amd.cli.ddl --preset ddl            # DDL.py
    python -m distributeddeeplearning_amd.cli.ddl --preset ddl_ray        # DDL_Ray.py   (barrier actors)
    python -m distributeddeeplearning_amd.cli.ddl --preset ddl_ray2       # DDL_Ray2.py  (async futures)
    python -m distributeddeeplearning_amd.cli.ddl --preset ddl_ray3       # DDL_Ray3.py  (+ transfer generation)
    python -m distributeddeeplearning_amd.cli.ddl --preset ddl_normal3    # DDL_Normal3.py
    python -m distributeddeeplearning_amd.cli.ddl --preset ddl_normal4    # DDL_Normal4.py main2(): export 5 models
    torchrun --nproc-per-node 8 -m distributeddeeplearning_amd.cli.ddl --preset ddl_ray3 --mode spmd

The reference reads a Ray/Redis cluster address from stdin (/root/reference/DDL_Ray.py:5-7); here
the worker pool is local to the node: `--workers N` processes, one per GPU.  Every `C` constant is
overridable: `--set random_model_size=20 --set epochs=4` or `DDL_EPOCHS=4` in the environment.
"""
from __future__ import annotations

import argparse
import sys
import time

import torch

from ..config import get_space

_DEFAULT_MODE = {"ddl": "local", "ddl_ray": "barrier", "ddl_ray2": "async", "ddl_ray3": "async",
                 "ddl_normal3": "local", "ddl_normal4": "local"}


def export_untrained(space, device, n: int, directory=None, verbose=True):
    """DDL_Normal4.py main2() (:271-276): build n random models and save() them, no training."""
    import numpy as np
    from ..models import search
    from ..models.classifier import RandomModel
    from ..models.ir import NameScope
    from ..utils.data import Data
    data = Data(device, channels=space.channels, synthetic=None)
    rng, names = np.random.default_rng(space.seed), NameScope()
    out = []
    for i in range(n):
        compiled = search.make_new_model(space, rng, data.input_shape, 10, names)
        sample = data.get_random_sample_dataset(space.data_train_size, seed=space.seed + i)
        m = RandomModel(data, sample, space, compiled, device, number=i, seed=space.seed + i, verbose=verbose)
        out.append(str(m.save(directory)))
        if verbose:
            print(compiled.spec.summary(), flush=True)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--preset", default="ddl", choices=sorted(_DEFAULT_MODE))
    ap.add_argument("--mode", default=None, choices=["local", "barrier", "async", "spmd"])
    ap.add_argument("--workers", type=int, default=None, help="actor worker processes (default: one per GPU)")
    ap.add_argument("--device", default=None, help="cuda | cpu (default: cuda if available)")
    ap.add_argument("--set", action="append", default=[], metavar="FIELD=VALUE")
    ap.add_argument("--synthetic", action="store_true", help="force the synthetic MNIST stand-in")
    ap.add_argument("--train-size", type=int, default=None, help="shrink the resident dataset (smoke runs)")
    ap.add_argument("--train", action="store_true", help="ddl_normal4: run the training main() instead of main2()")
    ap.add_argument("--save", action="store_true", help="export every member to models/modelN")
    ap.add_argument("--no-plot", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--log-jsonl", default=None,
                    help="append one JSON record per phase (timings, member accuracies, loss histories) to this file")
    a = ap.parse_args(argv)

    space = get_space(a.preset, a.set)
    mode = a.mode or _DEFAULT_MODE[a.preset]
    device = a.device or ("cuda" if torch.cuda.is_available() else "cpu")
    data_kw = {}
    if a.synthetic:
        data_kw["synthetic"] = True
    if a.train_size:
        data_kw["n_train"] = a.train_size
        data_kw["n_test"] = max(100, a.train_size // 6)
    verbose = not a.quiet
    t0 = time.time()

    if a.preset == "ddl_normal4" and not a.train:
        for d in export_untrained(space, device, 5, verbose=verbose):
            print("saved", d)
        return 0

    rt = None
    if mode == "spmd":
        from ..parallel import ddp
        from ..parallel.ensemble import SpmdEnsemble
        local = int(__import__("os").environ.get("LOCAL_RANK", "0"))
        dev = torch.device(f"cuda:{local}") if device == "cuda" else torch.device("cpu")
        if dev.type == "cuda":
            torch.cuda.set_device(dev)
        rank, world, _ = ddp.init_process_group(device=dev)
        ens = SpmdEnsemble(space, dev, data_kw=data_kw, verbose=verbose and rank == 0)
        say = print if rank == 0 else (lambda *x, **k: None)
    else:
        from ..parallel.ensemble import EnsembleModel
        if mode != "local":
            from ..parallel import runtime
            ngpu = torch.cuda.device_count() if device == "cuda" else 0
            n = a.workers or (ngpu if ngpu else 2)
            rt = runtime.init(n, [f"cuda:{i % ngpu}" if ngpu else "cpu" for i in range(n)])
        ens = EnsembleModel(space, mode=mode, runtime=rt, device=device, data_kw=data_kw, verbose=verbose)
        say = print

    from ..utils import trace
    log = trace.JsonlLogger(a.log_jsonl if say is print else None)
    log.log(event="start", preset=a.preset, mode=mode, device=device, members=space.random_model_size,
            transfer=space.transfer_model_size, data_train_size=space.data_train_size, batch=space.batch_size)

    def histories():
        """Per-member loss histories where the members live in this process (local / spmd modes)."""
        out = []
        for e in getattr(ens, "accuracy_with_models", []):
            actor = getattr(e[1], "_a", None)
            if actor is not None:
                out.append(actor.model.history)
        for m in getattr(ens, "members", []):
            if m.get("actor") is not None:
                out.append(m["actor"].model.history)
        return out

    try:
        t1 = time.time()
        with trace.range("generation_random"):
            ens.make_and_train_new_models(space.random_model_size)
        log.log(event="generation", kind="random", seconds=time.time() - t1, loss_histories=histories())
        if space.transfer_model_size > 0:
            t1 = time.time()
            with trace.range("generation_transfer"):
                ens.make_and_train_transfer_models(space.transfer_model_size)
            log.log(event="generation", kind="transfer", seconds=time.time() - t1, loss_histories=histories())
        with trace.range("evaluate"):
            accs = [float(x) for x in ens.member_test_accuracies()]
            for acc in accs:
                say("model accuracy : ", acc, flush=True)                   # DDL.py:171-172
            vote = float(ens.get_accuracy())
        say("ensemble accuracy : ", vote, flush=True)                       # DDL.py:174
        # ranking order (descending validation accuracy on each member's own hold-out, DDL.py:144-147)
        if hasattr(ens, "accuracy_with_models"):
            val = [float(e[0]) for e in ens.accuracy_with_models]
            numbers = [getattr(getattr(e[1], "_a", None), "model", None) and e[1]._a.model.number
                       for e in ens.accuracy_with_models]
        else:
            val, numbers = [float(m["acc"]) for m in ens.members], [m["number"] for m in ens.members]
        log.log(event="result", member_val_accuracy=val, member_numbers=numbers, member_test_accuracy=accs,
                top_k=space.predict_model_size, ensemble_accuracy=vote, seconds=time.time() - t0)
        if a.save:
            ens.save_models()
        if not a.no_plot and hasattr(ens, "plot_models"):
            ens.plot_models()
        say(f"done in {time.time() - t0:.1f}s", flush=True)
    finally:
        log.close()
        if rt is not None:
            from ..parallel import runtime
            runtime.shutdown()
        if mode == "spmd" and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
