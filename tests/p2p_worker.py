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


# ------------------------------------------------------------------------------------------------------------------
# Multi-rank DEVICE paths that RCCL cannot rehearse on a one-GPU box (it refuses duplicate devices): the same code
# over the xGMI kernel with a gloo bootstrap, several processes sharing the GPU.
def _device_comm(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distributeddeeplearning_amd.parallel import ddp
    from distributeddeeplearning_amd.parallel.p2p import P2PAllReduce

    class HostBounce(ddp.FlatDataParallel):
        """gloo bootstrap with device tensors: whatever the xGMI kernel does not take (broadcasts, tiny or odd-sized
        all-reduces, gathers) bounces through the host (test plumbing only)."""

        def broadcast_(self, *tensors, src=0):
            for t in tensors:
                h = t.cpu()
                dist.broadcast(h, src=src)
                t.copy_(h)

        def all_reduce_sum_(self, t, async_op=False):
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)

        def all_gather_scalars(self, values):
            return super().all_gather_scalars(values.cpu()).to(values.device)

    base = HostBounce()
    return dev, base, P2PAllReduce(base, dev, 1 << 21, max_blocks=8, timeout_s=30.0)


def _cnn_members(dev, dp, rank, n_members=3):
    """Small conv members (conv -> BN -> depthwise -> BN -> dense) on the device, each rank its own data."""
    from distributeddeeplearning_amd import config
    from distributeddeeplearning_amd.models import ir
    from distributeddeeplearning_amd.models.classifier import RandomModel
    from distributeddeeplearning_amd.models.optim import OptimConfig
    from distributeddeeplearning_amd.models.search import CompiledSpec
    from distributeddeeplearning_amd.utils.data import Data
    data = Data(dev, channels=1, n_train=160, n_test=16, synthetic=True, seed=rank)
    space = config.get_space("ddl_normal4", ["data_train_size=128", "batch_size=32", "epochs=2"])
    members = []
    for i in range(n_members):
        spec = ir.ModelSpec(f"m{i}", [ir.Input("in", (28, 28, 1))])
        for l in (ir.Conv2D("c1", 8 << i, 3, 2, 1, "valid", "relu"), ir.BatchNorm("b1"),
                  ir.DepthwiseConv2D("dw", 3, 1, 1, "valid", "tanh"), ir.BatchNorm("b2"),
                  ir.Flatten("f"), ir.Dense("o", 10, "softmax")):
            spec.append(l)
        comp = CompiledSpec(spec, OptimConfig.keras("adam", 1e-3), "categorical_crossentropy")
        m = RandomModel(data, data.get_random_sample(128, seed=5 + rank), space, comp, dev, number=i, seed=i,
                        dp=dp, use_graph=False, verbose=False)
        dp.broadcast_(m.net.P, m.net.S)
        m.net.refresh_working_copies()
        members.append(m)
    return data, members


def replica_group_gpu(rank, world, port, out):
    """ReplicaGroupStep on the device in all three modes (eager / graph segments around eager collectives / one
    graph with the collectives captured), fused and unfused, == the serial order fwd/bwd, all-reduce, update per
    network: bit for bit, replicas identical across ranks."""
    dev, base, dp = _device_comm(rank, world, port)
    from distributeddeeplearning_amd.models.classifier import ReplicaGroupStep
    from distributeddeeplearning_amd.ops import functional as F
    gen = torch.Generator().manual_seed(50 + rank)
    idxs = [torch.randint(0, 160, (32,), generator=gen).to(dev) for _ in range(4)]
    _, w = _cnn_members(dev, dp, rank)                   # plans: tuned once, then rank 0's on every rank
    for m in w:
        m._step_body(idxs[0])
    torch.cuda.synchronize()
    F.hip_backend().sync_choices(base)
    res = {}
    for fused in (True, False):
        dp.supports_fused_update = fused
        _, ref = _cnn_members(dev, dp, rank)
        assert ref[0]._fused_dp == fused
        for idx in idxs:                                  # 1 warm-up step inside ReplicaGroupStep + 3 more below
            for m in ref:
                m._step_body(idx)
        torch.cuda.synchronize()
        for mode in ("eager", "segments", "full"):
            _, ms = _cnn_members(dev, dp, rank)
            if mode == "eager":
                group = ReplicaGroupStep(ms, dp, idxs[0], mode)
                group.step(idxs[0])
            else:                                         # the constructor's warm-up step trains on the index batch
                group = ReplicaGroupStep(ms, dp, idxs[0], mode)
                assert group.mode == mode and group.graphs is not None
            for idx in idxs[1:]:
                group.step(idx)
            torch.cuda.synchronize()
            res[f"{'fused' if fused else 'unfused'}_{mode}"] = all(
                torch.equal(x.net.P, y.net.P) and torch.equal(x.net.S, y.net.S) and torch.equal(x.net.Pc, y.net.Pc)
                for x, y in zip(ms, ref))
        flat = torch.cat([m.net.P for m in ref]).cpu()
        parts = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat)
        res[f"{'fused' if fused else 'unfused'}_replicas"] = all(torch.equal(parts[0], p) for p in parts)
    res["healthy"] = dp.healthy()
    base.barrier()
    dp.close()
    with open(f"{out}/rg{rank}.txt", "w") as f:
        f.write(repr(res))
    np.save(f"{out}/rg{rank}.npy", np.array([int(res[k]) for k in sorted(res)]))
    dist.destroy_process_group()


def local_sgd_gpu(rank, world, port, out):
    """PeriodicAverager(overlap=True) on the device (side stream, events, `avg_delta`; `scale_emit` in finish()):
    (1) against its own CPU branch on the same parameter trajectories (a CPU twin of every rank's buffer gets the
    same local updates and is averaged over gloo), (2) end to end: members trained with local SGD end on a common
    model on every rank."""
    dev, base, dp = _device_comm(rank, world, port)
    from distributeddeeplearning_amd.models import ir
    from distributeddeeplearning_amd.models.network import Network
    from distributeddeeplearning_amd.parallel import ddp
    from distributeddeeplearning_amd.parallel.averaging import PeriodicAverager
    res = {}
    spec = ir.ModelSpec("m", [ir.Input("in", (28, 28, 1))])
    for l in (ir.Conv2D("c1", 16, 3, 2, 1, "valid", "relu"), ir.BatchNorm("b1"), ir.Flatten("f"),
              ir.Dense("o", 10, "softmax")):
        spec.append(l)
    gnet, cnet = Network(spec, dev, 1), Network(spec, "cpu", 1)
    n = min(gnet.P.numel(), cnet.P.numel())
    # twins: device buffers are channel-padded, so the CPU twin simply mirrors the device's flat buffers
    cP = gnet.P.cpu().clone()
    cS = gnet.S.cpu().clone()

    class Twin:                          # what PeriodicAverager needs of a network, on the CPU
        on_gpu, device = False, torch.device("cpu")
        P, S = cP, cS
    twin = Twin()
    cpu_dp = ddp.FlatDataParallel()      # plain gloo on CPU tensors
    ga = PeriodicAverager(dp, every=3, overlap=True, delay=2)
    ca = PeriodicAverager(cpu_dp, every=3, overlap=True, delay=2)
    ga.attach(gnet)
    ca.attach(twin)
    gen = torch.Generator().manual_seed(9 + rank)
    ok_traj, ok_pc = True, True
    for step in range(11):
        d = torch.randn(gnet.P.numel(), generator=gen) * 0.01      # this rank's "local SGD progress"
        ds = torch.randn(gnet.S.numel(), generator=gen) * 0.01
        gnet.P.add_(d.to(dev)); gnet.S.add_(ds.to(dev))
        twin.P.add_(d); twin.S.add_(ds)
        ga.after_step()
        ca.after_step()
        torch.cuda.synchronize()
        ok_traj = ok_traj and torch.allclose(gnet.P.cpu(), twin.P, rtol=0, atol=2e-6) \
            and torch.allclose(gnet.S.cpu(), twin.S, rtol=0, atol=2e-6)
    ga.finish()
    ca.finish()
    torch.cuda.synchronize()
    ok_traj = ok_traj and torch.allclose(gnet.P.cpu(), twin.P, rtol=0, atol=2e-6)
    ok_pc = torch.equal(gnet.Pc.cpu(), gnet.P.cpu().bfloat16())         # bf16 working copy re-emitted by the kernels
    parts = [torch.empty_like(twin.P) for _ in range(world)]
    dist.all_gather(parts, gnet.P.cpu())
    res["trajectory_matches_cpu_branch"] = ok_traj
    res["bf16_copy_reemitted"] = ok_pc
    res["averages_done"] = ga.n_averages == ca.n_averages and ga.n_averages >= 4
    res["common_model_after_finish"] = all(torch.equal(parts[0], p) for p in parts)

    # end to end: RandomModel members with the averager, graphed steps
    from distributeddeeplearning_amd.models.classifier import RandomModel
    data, ms = _cnn_members(dev, dp, rank, n_members=1)
    m0 = ms[0]
    avg = PeriodicAverager(dp, every=2, overlap=True, delay=1)
    m = RandomModel(data, m0.sample, m0.space, m0.compiled, dev, number=0, seed=0, dp=dp, averager=avg,
                    use_graph=True, verbose=False)
    dp.broadcast_(m.net.P, m.net.S)
    m.net.refresh_working_copies()
    m.train()
    torch.cuda.synchronize()
    flat = m.net.P.cpu()
    parts = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat)
    res["trained_members_common"] = all(torch.equal(parts[0], p) for p in parts) and avg.n_averages >= 2 \
        and bool(np.isfinite(np.array(m.history)).all())
    res["healthy"] = dp.healthy()
    base.barrier()
    dp.close()
    with open(f"{out}/ls{rank}.txt", "w") as f:
        f.write(repr(res))
    np.save(f"{out}/ls{rank}.npy", np.array([int(res[k]) for k in sorted(res)]))
    dist.destroy_process_group()


def spmd_ensemble_gpu(rank, world, port, out):
    """SpmdEnsemble on the device: a generation spread over two ranks, transfer children built from a parent that
    lives on the OTHER rank (weights broadcast as one flat buffer), votes all-reduced."""
    dev, base, dp = _device_comm(rank, world, port)
    from distributeddeeplearning_amd import config
    from distributeddeeplearning_amd.parallel.ensemble import SpmdEnsemble
    sp = config.get_space("ddl_ray3", ["data_train_size=600", "random_model_size=4", "transfer_model_size=2",
                                       "predict_model_size=3", "early_stop_rate=-1.0", "epochs=4", "seed=11"])
    ens = SpmdEnsemble(sp, dev, comm=dp, data_kw={"synthetic": True, "n_train": 700, "n_test": 128})
    ens.make_and_train_new_models()
    parents = [(m["number"], m["owner"]) for m in sorted(ens.members, key=lambda m: -m["acc"])[:2]]
    ens.make_and_train_transfer_models()
    ens.sort_models()
    children = [m for m in ens.members if m["number"] >= 4]
    # at least one child must sit on another rank than its parent, or the broadcast proved nothing
    crossed = any(c["owner"] != p[1] for c, p in zip(sorted(children, key=lambda m: m["number"]), parents))
    accs = [m["acc"] for m in ens.members]
    test_accs = ens.member_test_accuracies()
    vote = ens.get_accuracy()
    owners = [m["owner"] for m in ens.members]
    healthy = dp.healthy()
    base.barrier()
    dp.close()
    np.save(f"{out}/ens{rank}.npy", np.array(accs + test_accs + [vote] + owners + [float(crossed), float(healthy)],
                                             dtype=np.float64))
    dist.destroy_process_group()


def tune_gan(rank, world, port, out):
    """parallel/dp_tune.tune_gan_step with two ranks on the device: every candidate builds and is timed, all ranks
    agree on the winner, the training state is exactly what it was before tuning, and the tuned step trains replicas
    that stay identical."""
    dev, base, dp = _device_comm(rank, world, port)
    from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer
    from distributeddeeplearning_amd.parallel import dp_tune
    t = GANTrainer(GANConfig(batch=64, seed=5), dev, dp, rank)
    real = torch.rand(64, 28, 28, 1, generator=torch.Generator().manual_seed(rank))
    rp = t.D.prepare_input(real)
    before = [x.clone() for x in (t.G.P, t.D.P, t.G.S, t.D.S, t.optG.m, t.optD.v, t.optG.step_dev, t.rng.state)]
    step, rep = dp_tune.tune_gan_step(t, rp, {"xgmi": dp}, {"xgmi": True}, steps=5, warmup=1)
    after = [t.G.P, t.D.P, t.G.S, t.D.S, t.optG.m, t.optD.v, t.optG.step_dev, t.rng.state]
    # the final capture's warm-up step is a real training step: undo nothing, but the state BEFORE it must have been
    # the saved one -- checked through the step counter: exactly `warmup` steps were taken since `before`
    res = {"candidates": len(rep["dp_candidates_us"]) == 4 and rep["dp_candidates_dropped"] == [],
           "fields": all(k in rep for k in ("comm_chosen", "dp_variant", "allreduce_us_D", "allreduce_us_G",
                                            "exposed_comm_us_per_step")),
           "one_step_since_restore": int(t.optG.step_dev[0]) == int(before[6][0]) + 1}
    choice = torch.tensor([float(rep["dp_variant"]["merged_generator_pass"]),
                           float(rep["dp_variant"]["bucketed_allreduce"])])
    got = [torch.empty_like(choice) for _ in range(world)]
    dist.all_gather(got, choice)
    res["same_choice_everywhere"] = all(torch.equal(got[0], g) for g in got)
    for _ in range(3):
        step.step()
    torch.cuda.synchronize()
    parts = [torch.empty_like(t.G.P.cpu()) for _ in range(world)]
    dist.all_gather(parts, t.G.P.cpu())
    res["replicas_identical"] = all(torch.equal(parts[0], p) for p in parts)
    res["moved"] = not torch.equal(before[0], t.G.P)
    res["healthy"] = dp.healthy()
    base.barrier()
    dp.close()
    with open(f"{out}/tn{rank}.txt", "w") as f:
        f.write(repr(res) + repr(rep))
    np.save(f"{out}/tn{rank}.npy", np.array([int(res[k]) for k in sorted(res)]))
    dist.destroy_process_group()
