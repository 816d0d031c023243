"""Rank body for tests/test_dist_gloo.py (spawned with torch.multiprocessing, gloo backend)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from distributeddeeplearning_amd.parallel import ddp
    r, w, _ = ddp.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    return ddp.FlatDataParallel()


def _mlp_spec():
    from distributeddeeplearning_amd.models import ir
    m = ir.ModelSpec("m", [ir.Input("in", (6, 6, 1))])
    for l in (ir.Flatten("f"), ir.Dense("d1", 16, "tanh"), ir.Dense("o", 10, "softmax")):
        m.append(l)
    return m


def collectives(rank, world, port, out):
    dp = _init(rank, world, port)
    t = torch.arange(8, dtype=torch.float32) + rank
    dp.all_reduce_sum_(t)
    ok = torch.equal(t, torch.arange(8, dtype=torch.float32) * world + sum(range(world)))
    b = torch.full((4,), float(rank))
    dp.broadcast_(b, src=1)
    ok &= bool((b == 1).all())
    g = dp.all_gather_scalars(torch.tensor([rank * 1.0, 7.0]))
    ok &= g.shape == (world, 2) and g[:, 0].tolist() == list(map(float, range(world)))
    sh = dp.reduce_scatter_sum(torch.arange(4 * world, dtype=torch.float32))
    ok &= torch.equal(sh, torch.arange(4 * world, dtype=torch.float32)[rank * 4:(rank + 1) * 4] * world)
    full = torch.empty(4 * world)
    dp.all_gather_flat(sh, full)
    ok &= torch.equal(full, torch.arange(4 * world, dtype=torch.float32) * world)
    if rank == 0:
        dp.send(torch.tensor([3.5]), 1)
    elif rank == 1:
        r = torch.zeros(1)
        dp.recv(r, 0)
        ok &= float(r) == 3.5
    m = torch.tensor([float(rank)])
    dp.all_reduce_mean_(m)
    ok &= abs(float(m) - (world - 1) / 2) < 1e-6
    dp.barrier()
    np.save(f"{out}/ok{rank}.npy", np.array([int(ok)]))
    dist.destroy_process_group()


def dp_grad_equals_full_batch(rank, world, port, out):
    """Sync DP: mean of per-rank gradients on half batches == single-process full-batch gradient,
    and after the fused update every replica holds identical weights."""
    dp = _init(rank, world, port)
    from distributeddeeplearning_amd.models.network import Network
    from distributeddeeplearning_amd.models.optim import FlatOptimizer, OptimConfig
    from distributeddeeplearning_amd.ops import functional as F
    g = torch.Generator().manual_seed(0)
    X = torch.rand(16, 6, 6, 1, generator=g)
    Y = torch.nn.functional.one_hot(torch.randint(0, 10, (16,), generator=g), 10).float()
    net = Network(_mlp_spec(), "cpu", seed=rank)             # deliberately different init per rank...
    dp.broadcast_(net.P, net.S)                              # ...made identical by the broadcast
    opt = FlatOptimizer(net, OptimConfig.keras("sgd", 0.1))
    n = 16 // world
    xs, ys = X[rank * n:(rank + 1) * n], Y[rank * n:(rank + 1) * n]
    loss, dl, _ = F.softmax_ce(net.forward(net.prepare_input(xs), training=True), ys)
    net.backward(dl)
    dp.all_reduce_sum_(net.G)
    np.save(f"{out}/grad{rank}.npy", (net.G * dp.grad_scale).numpy())
    opt.step(dp.grad_scale)
    np.save(f"{out}/P{rank}.npy", net.P.numpy())
    if rank == 0:
        ref = Network(_mlp_spec(), "cpu", seed=0)
        loss, dl, _ = F.softmax_ce(ref.forward(ref.prepare_input(X), training=True), Y)
        ref.backward(dl)
        np.save(f"{out}/grad_ref.npy", ref.G.numpy())
    dist.destroy_process_group()


def local_sgd(rank, world, port, out, overlap):
    dp = _init(rank, world, port)
    from distributeddeeplearning_amd.models.network import Network
    from distributeddeeplearning_amd.models.optim import FlatOptimizer, OptimConfig
    from distributeddeeplearning_amd.ops import functional as F
    from distributeddeeplearning_amd.parallel.averaging import PeriodicAverager
    net = Network(_mlp_spec(), "cpu", seed=0)
    dp.broadcast_(net.P, net.S)
    opt = FlatOptimizer(net, OptimConfig.keras("sgd", 0.05))
    avg = PeriodicAverager(dp, every=3, overlap=overlap, delay=2)
    avg.attach(net)
    g = torch.Generator().manual_seed(100 + rank)            # each rank sees different data
    trace = []
    for step in range(7):
        x = torch.rand(8, 6, 6, 1, generator=g)
        y = torch.nn.functional.one_hot(torch.randint(0, 10, (8,), generator=g), 10).float()
        loss, dl, _ = F.softmax_ce(net.forward(net.prepare_input(x), training=True), y)
        net.backward(dl)
        opt.step()
        net.release()
        avg.after_step()
        trace.append(net.P.clone())
    avg.finish()
    np.save(f"{out}/trace{rank}.npy", torch.stack(trace).numpy())
    np.save(f"{out}/final{rank}.npy", net.P.numpy())
    np.save(f"{out}/navg{rank}.npy", np.array([avg.n_averages]))
    dist.destroy_process_group()


def spmd_ensemble(rank, world, port, out):
    _init(rank, world, port)
    from distributeddeeplearning_amd import config
    from distributeddeeplearning_amd.parallel.ensemble import SpmdEnsemble
    sp = config.get_space("ddl_ray3", ["data_train_size=600", "random_model_size=3", "transfer_model_size=2",
                                       "predict_model_size=2", "early_stop_rate=-1.0", "epochs=4", "seed=11"])
    ens = SpmdEnsemble(sp, "cpu", data_kw={"synthetic": True, "n_train": 700, "n_test": 120})
    ens.make_and_train_new_models()
    ens.make_and_train_transfer_models()
    ens.sort_models()
    res = [m["acc"] for m in ens.members] + ens.member_test_accuracies() + [ens.get_accuracy()]
    owners = [m["owner"] for m in ens.members]
    np.save(f"{out}/ens{rank}.npy", np.array(res + owners, dtype=np.float64))
    dist.destroy_process_group()


def gan_dp(rank, world, port, out, sharded=False):
    dp = _init(rank, world, port)
    from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer
    t = GANTrainer(GANConfig(batch=4, seed=3), "cpu", dp, rank, shard_optimizer=sharded)
    real = torch.rand(4, 28, 28, 1, generator=torch.Generator().manual_seed(rank))
    for _ in range(2):
        dl, gl = t.train_step(real)
    np.save(f"{out}/gan{rank}.npy", np.concatenate([t.G.P.numpy(), t.D.P.numpy(), [float(dl), float(gl)]]))
    dist.destroy_process_group()


def gan_dp_buckets(rank, world, port, out, rank_order):
    """Bucketed gradient all-reduce (tail bucket sent from inside the backward pass) vs one collective per network:
    same parameters; every collective the bucketed trainer issued is recorded to prove the hook really fired.
    `rank_order`: sum by all-gather + addition in rank order (what the xGMI kernel does) instead of gloo's ring."""
    dp = _init(rank, world, port)
    from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer
    if rank_order:
        def ordered(t, *a, **k):
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t.contiguous())
            acc = parts[0].clone()
            for p in parts[1:]:
                acc += p
            t.copy_(acc)
        dp.all_reduce_sum_ = ordered
    calls = []
    orig = dp.all_reduce_sum_

    def spy(t, *a, **k):
        calls.append(int(t.numel()))
        return orig(t, *a, **k)
    real = torch.rand(4, 28, 28, 1, generator=torch.Generator().manual_seed(rank))
    a = GANTrainer(GANConfig(batch=4, seed=3), "cpu", dp, rank)
    assert a.buckets and len(a._bk[id(a.D)]) == 2 and len(a._bk[id(a.G)]) == 2
    b = GANTrainer(GANConfig(batch=4, seed=3), "cpu", dp, rank)
    b.buckets = False
    for _ in range(3):
        dp.all_reduce_sum_ = spy
        a.train_step(real)
        dp.all_reduce_sum_ = orig
        b.train_step(real)
    nD, nG = a.D.G.numel(), a.G.G.numel()
    sizes = [hi - lo for lo, hi, _ in a._bk[id(a.D)]] + [hi - lo for lo, hi, _ in a._bk[id(a.G)]]
    ok_calls = calls == sizes * 3 and sum(sizes[:2]) == nD and sum(sizes[2:]) == nG
    d = max(float((a.G.P - b.G.P).abs().max()), float((a.D.P - b.D.P).abs().max()))
    exact = torch.equal(a.G.P, b.G.P) and torch.equal(a.D.P, b.D.P)
    np.save(f"{out}/bk{rank}.npy", np.concatenate([[float(ok_calls), d, float(exact)], a.G.P.numpy(), a.D.P.numpy()]))
    dist.destroy_process_group()


def sharded_adam(rank, world, port, out):
    """ZeRO-1 Adam == replicated Adam on all-reduced gradients, step for step."""
    dp = _init(rank, world, port)
    from distributeddeeplearning_amd.models import ir
    from distributeddeeplearning_amd.models.network import Network
    from distributeddeeplearning_amd.models.optim import FlatOptimizer, OptimConfig, ShardedFlatOptimizer
    spec = ir.ModelSpec("m", [ir.Input("in", (6, 6, 1))])
    for l in (ir.Flatten("f"), ir.Dense("d1", 7, "relu"), ir.Dense("d2", 10, "softmax")):
        spec.append(l)
    res = []
    for kind in ("adam", "rmsprop", "sgd"):
        a, b = Network(spec, "cpu", 5), Network(spec, "cpu", 5)
        oa = FlatOptimizer(a, OptimConfig.keras(kind, 1e-2))
        ob = ShardedFlatOptimizer(b, OptimConfig.keras(kind, 1e-2), dp)
        gen = torch.Generator().manual_seed(100 + rank)
        for _ in range(3):
            g = torch.randn(a.G.shape, generator=gen)
            a.G.copy_(g)
            b.G.copy_(g)
            dp.all_reduce_sum_(a.G)
            oa.step(dp.grad_scale)
            ob.step()
        res.append(float((a.P - b.P).abs().max()))
    np.save(f"{out}/zero{rank}.npy", np.array(res))
    dist.destroy_process_group()


def _members(dp, rank, n_members=3, lr=0.05):
    """Small MLP ensemble members on a shared synthetic pool (each rank its own data)."""
    from distributeddeeplearning_amd import config
    from distributeddeeplearning_amd.models.classifier import RandomModel
    from distributeddeeplearning_amd.models.optim import OptimConfig
    from distributeddeeplearning_amd.models.search import CompiledSpec
    from distributeddeeplearning_amd.models import ir
    from distributeddeeplearning_amd.utils.data import Data
    data = Data("cpu", channels=1, n_train=96, n_test=16, synthetic=True, seed=rank)
    space = config.get_space("ddl_normal4", ["data_train_size=64", "batch_size=16", "epochs=6", "early_stop_rate=0.05"])
    members = []
    for i in range(n_members):
        spec = ir.ModelSpec(f"m{i}", [ir.Input("in", (28, 28, 1))])
        for l in (ir.Flatten("f"), ir.Dense("d1", 8 + 4 * i, "tanh"), ir.BatchNorm("bn"), ir.Dense("o", 10, "softmax")):
            spec.append(l)
        comp = CompiledSpec(spec, OptimConfig.keras("sgd", lr), "categorical_crossentropy")
        m = RandomModel(data, data.get_random_sample(64, seed=5 + rank), space, comp, "cpu", number=i, seed=i,
                        dp=dp, use_graph=False, verbose=False)
        dp.broadcast_(m.net.P, m.net.S)
        members.append(m)
    return data, members


def replica_group_step(rank, world, port, out):
    """ReplicaGroupStep (all-reduces forked while the next network computes, updates at the end) ==
    the serial order fwd/bwd, all-reduce, update per network."""
    dp = _init(rank, world, port)
    from distributeddeeplearning_amd.models.classifier import ReplicaGroupStep
    _, a = _members(dp, rank)
    _, b = _members(dp, rank)
    gen = torch.Generator().manual_seed(50 + rank)
    group = ReplicaGroupStep(a, dp, torch.zeros(16, dtype=torch.int64), "eager")
    for _ in range(3):
        idx = torch.randint(0, 96, (16,), generator=gen)
        group.step(idx)
        for m in b:
            m._step_body(idx)
    diff = max(float((x.net.P - y.net.P).abs().max()) for x, y in zip(a, b))
    np.save(f"{out}/rg{rank}.npy", np.concatenate([np.array([diff]), torch.cat([m.net.P for m in a]).numpy()]))
    dist.destroy_process_group()


def dp_early_stop(rank, world, port, out):
    """Ranks whose LOCAL losses would stop at different rounds must still leave `train()` together."""
    dp = _init(rank, world, port)
    _, ms = _members(dp, rank, n_members=1, lr=0.2 if rank == 0 else 0.2)
    m = ms[0]
    # make the ranks' data differ in difficulty: rank 1 trains on shuffled labels (its local loss plateaus at once)
    if rank == 1:
        perm = torch.randperm(m.data.train_y.shape[0], generator=torch.Generator().manual_seed(1))
        m.data.train_y = m.data.train_y[perm]
    m.train()
    m.sync_bn_stats()
    np.save(f"{out}/es{rank}.npy", np.array([len(m.history)] + [h for r in m.history for h in r]))
    np.save(f"{out}/esS{rank}.npy", m.net.S.numpy())
    dist.destroy_process_group()


def sharded_checkpoint(rank, world, port, out):
    """ZeRO-1 optimizer: save (all ranks, shards gathered) -> load into a fresh trainer -> same continuation."""
    dp = _init(rank, world, port)
    from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer
    real = torch.rand(4, 28, 28, 1, generator=torch.Generator().manual_seed(rank))
    a = GANTrainer(GANConfig(batch=4, seed=3, base=8), "cpu", dp, rank, shard_optimizer=True)
    for _ in range(2):
        a.train_step(real)
    path = f"{out}/gan.ckpt"
    a.save_checkpoint(path, 2, write=rank == 0)
    dp.barrier()
    b = GANTrainer(GANConfig(batch=4, seed=77, base=8), "cpu", dp, rank, shard_optimizer=True)
    assert b.load_checkpoint(path) == 2
    for _ in range(2):
        a.train_step(real)
        b.train_step(real)
    d = max(float((a.G.P - b.G.P).abs().max()), float((a.D.P - b.D.P).abs().max()),
            float((a.G.S - b.G.S).abs().max()))
    np.save(f"{out}/sck{rank}.npy", np.array([d]))
    dist.destroy_process_group()


def selfcheck_cpu(rank, world, port, out):
    dp = _init(rank, world, port)
    from distributeddeeplearning_amd.parallel import selfcheck
    seen = selfcheck.ranks_seen(dp, "cpu")
    ok = selfcheck.captured_collective_ok(dp, "cpu")           # no device: the verdict is "no", without any collective
    np.save(f"{out}/sc{rank}.npy", np.array(seen + [int(ok)]))
    dist.destroy_process_group()
