"""EnsembleModel: local / barrier / async back-ends agree, vote semantics, transfer generation,
artifacts, failed-member policy."""
import numpy as np
import pytest
import torch

from distributeddeeplearning_amd import config
from distributeddeeplearning_amd.parallel import runtime
from distributeddeeplearning_amd.parallel.ensemble import EnsembleModel, SpmdEnsemble

DATA_KW = {"synthetic": True, "n_train": 700, "n_test": 120}
OVER = ["data_train_size=600", "random_model_size=3", "transfer_model_size=2", "predict_model_size=2",
        "early_stop_rate=-1.0", "epochs=4", "seed=11"]


def _run(mode, rt=None, preset="ddl_ray3"):
    sp = config.get_space(preset, OVER)
    ens = EnsembleModel(sp, mode=mode, runtime=rt, device="cpu", data_kw=DATA_KW, verbose=False)
    ens.make_and_train_new_models()
    ens.make_and_train_transfer_models()
    ens.sort_models()
    accs = [a for a, _, _ in ens.accuracy_with_models]
    return ens, accs, ens.member_test_accuracies(), ens.get_accuracy()


@pytest.fixture(scope="module")
def rt():
    r = runtime.Runtime(2, ["cpu"] * 2)
    yield r
    r.shutdown()


def test_local_generation_ranking_and_vote():
    ens, accs, test_accs, vote = _run("local")
    assert len(accs) == 5 and accs == sorted(accs, reverse=True)
    assert all(0 <= a <= 1 for a in accs + test_accs) and 0 <= vote <= 1
    # the vote is the argmax of the SUM of the top-k softmax matrices (DDL.py:149-165)
    top = ens.accuracy_with_models[:2]
    s = sum(torch.from_numpy(h.call("predict_test").get()) for _, h, _ in top)
    manual = float((s.argmax(1) == ens.data.test_labels.cpu()).float().mean())
    assert abs(manual - vote) < 1e-6
    assert max(accs) > 0.3                       # members really learn on the synthetic digits


def test_actor_backends_reproduce_local_results(rt):
    _, accs_l, test_l, vote_l = _run("local")
    _, accs_b, test_b, vote_b = _run("barrier", rt)
    _, accs_a, test_a, vote_a = _run("async", rt)
    # same seeds -> same architectures, splits and initial weights in every back-end; worker processes
    # run with a different OpenMP thread count, so allow a few borderline samples to flip
    tol = dict(atol=0.03, rtol=0)
    assert np.allclose(accs_b, accs_a, **tol) and np.allclose(accs_l, accs_b, **tol)
    assert np.allclose(test_b, test_a, **tol) and np.allclose(test_l, test_b, **tol)
    assert abs(vote_l - vote_b) <= 0.03 and abs(vote_b - vote_a) <= 0.03


def test_async_mode_defers_resolution(rt):
    sp = config.get_space("ddl_ray2", OVER)
    ens = EnsembleModel(sp, mode="async", runtime=rt, device="cpu", data_kw=DATA_KW, verbose=False)
    ens.make_and_train_new_models(2)
    assert all(hasattr(e[0], "done") for e in ens.accuracy_with_models)      # still futures
    ens.sort_models()
    assert all(isinstance(e[0], float) for e in ens.accuracy_with_models)


def test_artifacts(tmp_path):
    sp = config.get_space("ddl_normal4", OVER + [f"model_save_directory={tmp_path}/models", "epochs=2"])
    ens = EnsembleModel(sp, mode="local", device="cpu", data_kw=DATA_KW, verbose=False)
    ens.make_and_train_new_models(2)
    dirs = ens.save_models()
    assert len(dirs) == 2
    for d in dirs:
        assert {"architecture.json", "weights.h5", "graph.png"} <= {p.name for p in __import__("pathlib").Path(d).iterdir()}
    plots = ens.plot_models()
    assert all("accuracy" in p and p.endswith(".png") for p in plots)


def test_dead_member_is_dropped():
    import time
    with runtime.Runtime(2, ["cpu"] * 2) as r:
        sp = config.get_space("ddl_ray2", OVER)
        ens = EnsembleModel(sp, mode="async", runtime=r, device="cpu", data_kw=DATA_KW, verbose=False)
        ens.make_and_train_new_models(4)                   # two members per worker
        ens.sort_models()
        victim = ens.accuracy_with_models[0][1]._h._worker
        r._procs[victim].kill()                            # fault injection: exact child PID
        t0 = time.time()
        while victim in r.alive_workers() and time.time() - t0 < 20:
            time.sleep(0.05)
        assert victim not in r.alive_workers()
        vote = ens.get_accuracy(k=2)                       # dead members are dropped, vote refilled
        assert 0 <= vote <= 1
        assert len(ens.member_test_accuracies()) == 2     # touching every member finds all the dead
        assert len(ens.accuracy_with_models) == 2
        assert all(e[1]._h._worker != victim for e in ens.accuracy_with_models)
        strict = EnsembleModel(sp, mode="async", runtime=r, device="cpu", data_kw=DATA_KW, verbose=False,
                               drop_failed=False)
        strict.make_and_train_new_models(1)                # lands on the surviving worker
        assert len(strict.member_test_accuracies()) == 1


def test_spmd_single_rank_matches_local_semantics():
    sp = config.get_space("ddl_ray3", OVER)
    ens = SpmdEnsemble(sp, "cpu", data_kw=DATA_KW)
    ens.make_and_train_new_models()
    ens.make_and_train_transfer_models()
    ens.sort_models()
    accs = [m["acc"] for m in ens.members]
    assert len(accs) == 5 and accs == sorted(accs, reverse=True)
    assert 0 <= ens.get_accuracy() <= 1 and len(ens.member_test_accuracies()) == 5


def test_spmd_placement_balances_cost_and_is_rank_independent():
    """SPMD members were i % world.  Placement is now longest-processing-time-first on an estimated cost,
    computed identically by every rank: same owners from any rank's point of view, no rank idle, and the heaviest
    rank within one member of the mean."""
    from distributeddeeplearning_amd.models import search
    sp = config.get_space("ddl_normal3", OVER)
    owners = []
    for rank in (0, 3):
        e = SpmdEnsemble(sp, "cpu", data_kw=DATA_KW)
        e.world, e.rank = 4, rank
        new = [e._add(search.make_new_model(sp, e.rng, e.data.input_shape, 10)) for _ in range(10)]
        e._place(new)
        owners.append([m["owner"] for m in new])
        assert all((m["actor"] is not None) == (m["owner"] == rank) for m in new)
        costs = [SpmdEnsemble.member_cost(m["compiled"]) for m in new]
        load = [sum(c for c, m in zip(costs, new) if m["owner"] == r) for r in range(4)]
        assert min(load) > 0
        assert max(load) <= sum(costs) / 4 + max(costs)
    assert owners[0] == owners[1]
