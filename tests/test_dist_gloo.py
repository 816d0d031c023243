"""Multi-process plumbing on the gloo backend, world_size 2 (BASELINE.json config 1: runs without a
GPU).  The same FlatDataParallel / PeriodicAverager / SpmdEnsemble code runs over RCCL on device."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dist_worker as W  # noqa: E402


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, out, *extra, world=2):
    mp.spawn(fn, args=(world, _port(), str(out), *extra), nprocs=world, join=True)


def test_flat_collectives(tmp_path):
    _spawn(W.collectives, tmp_path)
    assert all(int(np.load(tmp_path / f"ok{r}.npy")[0]) == 1 for r in range(2))


def test_sync_dp_gradient_equals_full_batch(tmp_path):
    _spawn(W.dp_grad_equals_full_batch, tmp_path)
    g0, g1, ref = (np.load(tmp_path / n) for n in ("grad0.npy", "grad1.npy", "grad_ref.npy"))
    assert np.array_equal(g0, g1)
    assert np.allclose(g0, ref, rtol=1e-4, atol=1e-6)
    assert np.array_equal(np.load(tmp_path / "P0.npy"), np.load(tmp_path / "P1.npy"))


@pytest.mark.parametrize("overlap", [False, True])
def test_local_sgd_periodic_averaging(tmp_path, overlap):
    _spawn(W.local_sgd, tmp_path, overlap)
    t0, t1 = np.load(tmp_path / "trace0.npy"), np.load(tmp_path / "trace1.npy")
    f0, f1 = np.load(tmp_path / "final0.npy"), np.load(tmp_path / "final1.npy")
    assert np.array_equal(f0, f1)                              # finish() leaves one common model
    assert not np.array_equal(t0[0], t1[0])                    # between averages replicas drift apart
    if not overlap:
        assert np.allclose(t0[2], t1[2]) and np.allclose(t0[5], t1[5])      # steps 3 and 6 are sync points
        assert not np.allclose(t0[3], t1[3])
        assert int(np.load(tmp_path / "navg0.npy")[0]) == 3    # steps 3, 6 + finish
    else:
        # delayed correction: replicas are pulled together 2 steps after each launch but not equal
        d_before = np.abs(t0[3] - t1[3]).max()
        d_after = np.abs(t0[4] - t1[4]).max()
        assert d_after < d_before
        assert int(np.load(tmp_path / "navg0.npy")[0]) >= 3


def test_spmd_ensemble_two_ranks_agree(tmp_path):
    _spawn(W.spmd_ensemble, tmp_path)
    a, b = np.load(tmp_path / "ens0.npy"), np.load(tmp_path / "ens1.npy")
    assert np.array_equal(a, b)                                # every rank ends with the same table
    accs, owners = a[:5], a[-5:]
    assert list(accs) == sorted(accs, reverse=True) and set(owners) == {0.0, 1.0}
    assert 0 <= a[10] <= 1


def test_sharded_optimizer_matches_replicated(tmp_path):
    _spawn(W.sharded_adam, tmp_path)
    for r in range(2):
        assert np.load(tmp_path / f"zero{r}.npy").max() < 1e-6


@pytest.mark.parametrize("sharded", [False, True])
def test_gan_data_parallel_replicas_stay_identical(tmp_path, sharded):
    _spawn(W.gan_dp, tmp_path, sharded)
    a, b = np.load(tmp_path / "gan0.npy"), np.load(tmp_path / "gan1.npy")
    assert np.array_equal(a[:-2], b[:-2])                      # weights identical after 2 DP steps
    assert np.isfinite(a).all() and not np.array_equal(a[-2:], b[-2:])     # local losses differ (own shard)


@pytest.mark.parametrize("world,rank_order", [(2, False), (4, True), (4, False)])
def test_gan_bucketed_allreduce_equals_one_collective_per_network(tmp_path, world, rank_order):
    """Gradient buckets reduced from inside the backward pass (models/dcgan.py `_bucket_hook`): the bucketed trainer
    issues exactly [tail, rest] per network per step and ends on the parameters of the unbucketed one, BIT FOR BIT,
    whenever an element's summation order does not depend on the message it travels in: two ranks (a two-term sum is
    commutative) or a rank-ordered sum (what the xGMI kernel does; here all-gather + ordered add over gloo).  gloo's
    own ring at four ranks owns chunks by position in the message, so the order differs between the two message
    shapes: agreement is then to fp32 rounding (amplified by three Adam steps), and the replicas of all ranks still
    hold the same bits.  tests/test_p2p_gpu.py pins bit-equality on the device with the real kernel."""
    _spawn(W.gan_dp_buckets, tmp_path, rank_order, world=world)
    got = [np.load(tmp_path / f"bk{r}.npy") for r in range(world)]
    exact = world == 2 or rank_order
    for g in got:
        assert g[0] == 1.0                              # the collectives issued were the planned buckets, in order
        assert g[1] <= (0.0 if exact else 1e-4)
        assert g[2] == (1.0 if exact else g[2])
    for g in got[1:]:
        assert np.array_equal(got[0][3:], g[3:])        # replicas identical across ranks


@pytest.mark.parametrize("world", [2, 4])
def test_replica_group_step_equals_serial_order(tmp_path, world):
    _spawn(W.replica_group_step, tmp_path, world=world)
    rs = [np.load(tmp_path / f"rg{r}.npy") for r in range(world)]
    assert all(r[0] == 0.0 for r in rs)                        # overlapped schedule == serial schedule, exactly
    assert all(np.array_equal(rs[0][1:], r[1:]) for r in rs)   # replicas identical across ranks


def test_dp_early_stop_is_a_collective_decision(tmp_path):
    _spawn(W.dp_early_stop, tmp_path)
    a, b = np.load(tmp_path / "es0.npy"), np.load(tmp_path / "es1.npy")
    assert np.array_equal(a, b) and a[0] >= 1                  # same rounds, same (rank-averaged) loss history
    assert np.array_equal(np.load(tmp_path / "esS0.npy"), np.load(tmp_path / "esS1.npy"))   # BN stats averaged


def test_sharded_optimizer_checkpoint_round_trip(tmp_path):
    _spawn(W.sharded_checkpoint, tmp_path)
    assert all(float(np.load(tmp_path / f"sck{r}.npy")[0]) == 0.0 for r in range(2))


def test_selfcheck_ranks_seen_gloo(tmp_path):
    _spawn(W.selfcheck_cpu, tmp_path, world=4)
    for r in range(4):
        assert np.load(tmp_path / f"sc{r}.npy").tolist() == [0, 1, 2, 3, 0]
