"""xGMI direct all-reduce kernel (csrc/comm/p2p_allreduce.hip) with 2 and 4 PROCESSES sharing the one GPU of the
test box: IPC handles are exchanged over gloo, every process maps the others' buffers, and the kernels signal each
other through flags exactly as they would across xGMI.  Checked bit for bit against a host sum in rank order:
one-shot and two-shot sizes, repeated calls (double-buffer parity), hipGraph replay, and the fused
all-reduce + optimizer epilogue against all-reduce followed by the stand-alone optimizer kernel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import p2p_worker as W  # noqa: E402

pytestmark = pytest.mark.gpu


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_allreduce_processes_sharing_one_gpu(tmp_path, world):
    mp.spawn(W.run, args=(world, _port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"p2p{r}.npy")
        assert got.tolist() == [1] * len(got), open(tmp_path / f"p2p{r}.txt").read()


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_fused_bucketed_graphed_gan_step_equals_unfused_flat_eager(tmp_path, world):
    """Whole-step hipGraph with the fused all-reduce + update launched per gradient BUCKET from inside the backward
    pass == bucketed all-reduce + optimizer kernel, eager == one collective per network after the backward, eager:
    bit for bit (the kernel sums in rank order whatever the message length), replicas identical on every rank."""
    mp.spawn(W.gan, args=(world, _port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(tmp_path / f"gan{r}.npy").tolist() == [1, 1, 1]


def _flags(tmp_path, stem, world):
    for r in range(world):
        got = np.load(tmp_path / f"{stem}{r}.npy")
        assert got.tolist() == [1] * len(got), open(tmp_path / f"{stem}{r}.txt").read()


def test_replica_group_step_device_modes_two_processes(tmp_path):
    """BASELINE config 2's step (ReplicaGroupStep: replica i's all-reduce on the comm stream while replica i+1
    computes) on the device with two ranks: eager == graph segments == one full graph == the serial order."""
    mp.spawn(W.replica_group_gpu, args=(2, _port(), str(tmp_path)), nprocs=2, join=True)
    _flags(tmp_path, "rg", 2)


def test_local_sgd_overlapped_averaging_device_two_processes(tmp_path):
    """BASELINE config 5 on the device: side-stream all-reduce + `avg_delta` `delay` steps later, `scale_emit` in
    finish(), against the averager's CPU branch over gloo on the same trajectories; trained members end common."""
    mp.spawn(W.local_sgd_gpu, args=(2, _port(), str(tmp_path)), nprocs=2, join=True)
    _flags(tmp_path, "ls", 2)


def test_spmd_ensemble_device_two_processes_with_transfer_broadcast(tmp_path):
    mp.spawn(W.spmd_ensemble_gpu, args=(2, _port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "ens0.npy"), np.load(tmp_path / "ens1.npy")
    assert np.array_equal(a, b)                     # every rank holds the same accuracies, vote and placement
    n = 6
    accs, test_accs, vote, owners = a[:n], a[n:2 * n], a[2 * n], a[2 * n + 1:3 * n + 1]
    assert np.isfinite(accs).all() and np.isfinite(test_accs).all() and 0.0 <= vote <= 1.0
    assert set(owners.tolist()) == {0.0, 1.0}      # the generation really was spread over both ranks
    assert a[-1] == 1.0                            # kernel healthy
    assert vote >= max(test_accs[:3]) - 0.1        # top-3 vote is no worse than its best member (small test set)


def test_dp_step_tuner_two_processes_agree_and_restore_state(tmp_path):
    mp.spawn(W.tune_gan, args=(2, _port(), str(tmp_path)), nprocs=2, join=True)
    _flags(tmp_path, "tn", 2)
