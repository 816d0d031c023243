"""Native RCCL communicator on the one GPU of the test box (world = 1: every collective is the
identity, which still exercises bootstrap, dtype mapping, stream ordering and graph capture)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rccl_world1_collectives_and_graph_capture():
    from distributeddeeplearning_amd.parallel.rccl import RcclCommunicator
    comm = RcclCommunicator.from_env("cuda:0")
    assert (comm.world, comm.rank) == (1, 0) and comm.healthy()
    t = torch.arange(1024, dtype=torch.float32, device="cuda")
    ref = t.clone()
    comm.all_reduce_sum_(t)
    comm.broadcast_(t)
    assert torch.equal(t, ref)
    b = torch.randn(64, device="cuda").bfloat16()
    rb = b.clone()
    comm.all_reduce_mean_(b)
    assert torch.equal(b, rb)
    assert torch.equal(comm.all_gather_scalars(t[:4]), ref[:4][None])
    assert torch.equal(comm.reduce_scatter_sum(t), ref)
    out = torch.empty_like(t)
    comm.all_gather_flat(t, out)
    assert torch.equal(out, ref)
    comm.barrier()
    # a collective between two kernels inside one hipGraph
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        t.mul_(1.0)
        comm.all_reduce_sum_(t)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        t.add_(1.0)
        comm.all_reduce_sum_(t)
        t.mul_(2.0)
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(t, (ref + 1) * 2 * 2 + 2)          # ((x+1)*2 + 1)*2
    comm.close()


def test_rccl_bf16_wire_dtype():
    from distributeddeeplearning_amd.parallel.rccl import RcclCommunicator
    comm = RcclCommunicator.from_env("cuda:0", comm_dtype=torch.bfloat16)
    t = torch.full((256,), 1.5, device="cuda")
    comm.all_reduce_sum_(t)
    assert torch.equal(t, torch.full((256,), 1.5, device="cuda"))
    with pytest.raises(ValueError):
        comm.all_reduce_sum_(torch.zeros(4))               # CPU tensors are rejected loudly
    comm.close()


@pytest.mark.parametrize("sharded", [False, True])
def test_graphed_dp_step_with_one_rank_group(sharded):
    """The segmented-graph DP step (all-reduce of D grads overlapped with G's forward on a comm
    stream) on a 1-rank RCCL group: same losses as the plain eager trainer."""
    import os
    import torch.distributed as dist
    from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer, GraphedGANStep
    from distributeddeeplearning_amd.parallel import ddp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29677")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        dp = ddp.FlatDataParallel()
        real = torch.rand(64, 28, 28, 1, generator=torch.Generator().manual_seed(0))
        a = GANTrainer(GANConfig(batch=64, seed=5), "cuda:0", dp, 0, shard_optimizer=sharded)
        b = GANTrainer(GANConfig(batch=64, seed=5), "cuda:0", None, 0)
        rp = a.D.prepare_input(real)
        step = GraphedGANStep(a, rp, warmup=2)
        for _ in range(2):
            b.train_step(rp, prepared=True)
        for _ in range(3):
            la = step.step()
            lb = b.train_step(rp, prepared=True)
        torch.cuda.synchronize()
        assert abs(float(la[0]) - float(lb[0])) < 2e-2 * max(1.0, abs(float(lb[0])))
        assert abs(float(la[1]) - float(lb[1])) < 2e-2 * max(1.0, abs(float(lb[1])))
    finally:
        dist.destroy_process_group()


def test_captured_collective_selftest_and_replica_group_modes():
    """The start-up self-test that picks `--graph full` (parallel/selfcheck.py) on a 1-rank RCCL group, for both
    communicator back-ends, and the three ReplicaGroupStep modes producing identical parameters."""
    import os
    import torch.distributed as dist
    from distributeddeeplearning_amd.models.classifier import ReplicaGroupStep
    from distributeddeeplearning_amd.parallel import ddp, selfcheck
    from distributeddeeplearning_amd.parallel.rccl import RcclCommunicator
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29678")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        dp = ddp.FlatDataParallel()
        assert selfcheck.ranks_seen(dp, "cuda:0") == [0]
        assert selfcheck.captured_collective_ok(dp, "cuda:0", verbose=True) is True
        native = RcclCommunicator.from_env("cuda:0")
        assert selfcheck.captured_collective_ok(native, "cuda:0", verbose=True) is True
        native.close()

        class Broken:                       # a communicator whose captured replay silently does nothing
            rank, world = 0, 1

            def all_reduce_sum_(self, t):
                if torch.cuda.is_current_stream_capturing():
                    return
                t.mul_(2.0)
        assert selfcheck.captured_collective_ok(Broken(), "cuda:0") is False

        from bench_helpers import small_members
        results = {}
        for mode in ("eager", "segments", "full"):
            members, idxs = small_members(dp)
            group = ReplicaGroupStep(members, dp, idxs[0], mode)
            for i in range(4):
                group.step(idxs[i % len(idxs)])
            torch.cuda.synchronize()
            results[mode] = torch.cat([m.net.P for m in members]).clone()
        # graph modes run one warm-up step more than eager: compare them with each other, eager after 5 steps
        members, idxs = small_members(dp)
        group = ReplicaGroupStep(members, dp, idxs[0], "eager")
        group.step(idxs[0])
        for i in range(4):
            group.step(idxs[i % len(idxs)])
        torch.cuda.synchronize()
        eager5 = torch.cat([m.net.P for m in members])
        assert torch.equal(results["segments"], results["full"])
        assert torch.equal(results["full"], eager5)
    finally:
        dist.destroy_process_group()
