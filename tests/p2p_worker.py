"""Rank body for tests/test_p2p_gpu.py: several processes share ONE GPU and all-reduce through IPC peer mappings."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distributeddeeplearning_amd.models.dcgan import discriminator_spec
    from distributeddeeplearning_amd.models.network import Network
    from distributeddeeplearning_amd.models.optim import FlatOptimizer, OptimConfig
    from distributeddeeplearning_amd.parallel import ddp
    from distributeddeeplearning_amd.parallel.p2p import P2PAllReduce
    base = ddp.FlatDataParallel()
    cap = 1 << 21
    dp = P2PAllReduce(base, dev, cap, max_blocks=8, timeout_s=30.0)
    res = {}

    def host_sum(x):
        """bit-exact reference: gather every rank's input on the host, add in rank order in fp32"""
        parts = [torch.empty_like(x.cpu()) for _ in range(world)]
        dist.all_gather(parts, x.cpu())
        acc = parts[0].clone()
        for p in parts[1:]:
            acc += p
        return acc

    gen = torch.Generator().manual_seed(100 + rank)
    ok = True
    # one-shot (<= 256 KB) and two-shot sizes, odd tails, repeated (both parities), interleaved
    for it, n in enumerate([4, 1000, 65536, 65540, 266752, 1066180, 4096, 1 << 21, 266752]):
        x = torch.randn(n, generator=gen).to(dev)
        ref = host_sum(x)
        y = x.clone()
        dp.all_reduce_sum_(y)
        torch.cuda.synchronize()
        ok = ok and torch.equal(y.cpu(), ref)
        z = x.clone()
        dp.all_reduce_mean_(z)
        torch.cuda.synchronize()
        ok = ok and torch.equal(z.cpu(), ref * (1.0 / world))
    res["allreduce_exact"] = ok
    res["validate"] = dp.validate(1 << 20)

    # captured into a hipGraph and replayed on new contents
    buf = torch.zeros(266752, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dp.all_reduce_sum_(buf)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    base.barrier()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf.mul_(2.0)
        dp.all_reduce_sum_(buf)
        dp.all_reduce_sum_(buf)
    ok = True
    for k in range(3):
        x = torch.randn(266752, generator=gen).to(dev)
        buf.copy_(x)
        ref = host_sum(x * 2.0)
        g.replay()
        torch.cuda.synchronize()
        # second all-reduce sums W identical copies in rank order
        acc = ref.clone()
        for _ in range(world - 1):
            acc += ref
        ok = ok and torch.equal(buf.cpu(), acc)
    res["graph_replay_exact"] = ok

    # fused all-reduce + Adam == all-reduce, then the optimizer kernel (bitwise)
    ok = True
    for kind in ("adam", "sgd", "rmsprop", "adam_fresh_double"):
        a, b = Network(discriminator_spec(), dev, 3), Network(discriminator_spec(), dev, 3)
        cfg = OptimConfig(kind=kind, lr=1e-3, eps=1e-8)
        oa, ob = FlatOptimizer(a, cfg), FlatOptimizer(b, cfg)
        for _ in range(3):
            gr = torch.randn(a.G.numel(), generator=gen).to(dev) * 0.01
            a.G.copy_(gr)
            b.G.copy_(gr)
            dp.all_reduce_sum_(a.G)
            oa.step(dp.grad_scale)
            ob.step_allreduce_fused(dp)
        torch.cuda.synchronize()
        ok = ok and torch.equal(a.P, b.P) and torch.equal(a.Pc, b.Pc) and torch.equal(a.Palt, b.Palt)
        if oa.m is not None:
            ok = ok and torch.equal(oa.m, ob.m)
        if oa.v is not None:
            ok = ok and torch.equal(oa.v, ob.v)
        # replicas identical across ranks
        ref = host_sum(b.P) * (1.0 / world)
        ok = ok and torch.allclose(b.P.cpu(), ref, rtol=0, atol=0)
    res["fused_update_exact"] = ok
    res["healthy"] = dp.healthy()
    base.barrier()
    dp.close()
    np.save(f"{out}/p2p{rank}.npy", np.array([int(res[k]) for k in sorted(res)]))
    with open(f"{out}/p2p{rank}.txt", "w") as f:
        f.write(repr(res))
    dist.destroy_process_group()


def gan(rank, world, port, out):
    """Data-parallel DCGAN steps over the xGMI kernel: fused + whole-step hipGraph == unfused eager, bit for bit,
    and the replicas of all ranks stay identical."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer, GraphedGANStep
    from distributeddeeplearning_amd.parallel import ddp
    from distributeddeeplearning_amd.parallel.p2p import P2PAllReduce

    class HostBroadcast(ddp.FlatDataParallel):
        """gloo bootstrap with device tensors: broadcasts bounce through the host (test plumbing only)."""

        def broadcast_(self, *tensors, src=0):
            for t in tensors:
                h = t.cpu()
                dist.broadcast(h, src=src)
                t.copy_(h)

    base = HostBroadcast()
    dp = P2PAllReduce(base, dev, 1 << 21, max_blocks=8, timeout_s=30.0)
    real = torch.rand(64, 28, 28, 1, generator=torch.Generator().manual_seed(rank))
    # Kernel plans first: every rank tunes on its own clock and may pick different (equally correct) plans, and
    # GraphedGANStep makes all ranks adopt rank 0's table between its warm-up steps and the capture -- a bit-for-bit
    # comparison of two trainers is only meaningful once the plans can no longer change under them.
    from distributeddeeplearning_amd.ops import functional as F
    w = GANTrainer(GANConfig(batch=64, seed=5), dev, dp, rank)
    rp = w.D.prepare_input(real)
    w.train_step(rp, prepared=True)
    torch.cuda.synchronize()
    F.hip_backend().sync_choices(base)
    del w

    a = GANTrainer(GANConfig(batch=64, seed=5), dev, dp, rank)
    assert a.fused and a._bucketed(a.D) and a._bucketed(a.G)     # fused update per gradient bucket, under the backward
    step = GraphedGANStep(a, rp, warmup=2)
    assert step.full
    for _ in range(4):
        step.step()
    torch.cuda.synchronize()

    dp.supports_fused_update = False                 # same communicator, all-reduce then the optimizer kernel
    b = GANTrainer(GANConfig(batch=64, seed=5), dev, dp, rank)
    assert not b.fused and b._bucketed(b.D)
    c = GANTrainer(GANConfig(batch=64, seed=5), dev, dp, rank)
    c.buckets = False                                # ... and one collective per network, after the whole backward
    for _ in range(6):
        b.train_step(rp, prepared=True)
        c.train_step(rp, prepared=True)
    torch.cuda.synchronize()
    same = True
    for o in (b, c):
        same = same and torch.equal(a.G.P, o.G.P) and torch.equal(a.D.P, o.D.P) and torch.equal(a.optG.m, o.optG.m) \
            and torch.equal(a.optD.v, o.optD.v) and torch.equal(a.G.S, o.G.S)
    parts = [torch.empty_like(a.G.P.cpu()) for _ in range(world)]
    dist.all_gather(parts, a.G.P.cpu())
    replicas = all(torch.equal(parts[0], p) for p in parts)
    healthy = dp.healthy()
    base.barrier()
    dp.close()
    np.save(f"{out}/gan{rank}.npy", np.array([int(same), int(replicas), int(healthy)]))
    dist.destroy_process_group()
