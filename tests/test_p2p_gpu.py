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
