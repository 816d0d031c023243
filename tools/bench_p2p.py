#!/usr/bin/env python3
"""Latency of the xGMI direct all-reduce kernel for the framework's message sizes, W processes on the visible GPUs
(all on device 0 when fewer GPUs than ranks: IPC mappings of the same device, i.e. no xGMI hop -- that measures the
kernel's protocol + memory overhead, not the links).

    python tools/bench_p2p.py --world 1 2 4 [--blocks 128]
"""
import argparse
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def body(rank, world, port, blocks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    ngpu = torch.cuda.device_count()
    dev = torch.device(f"cuda:{rank % ngpu}")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distributeddeeplearning_amd.models.dcgan import discriminator_spec, generator_spec
    from distributeddeeplearning_amd.models.network import Network
    from distributeddeeplearning_amd.models.optim import FlatOptimizer, OptimConfig
    from distributeddeeplearning_amd.parallel import ddp
    from distributeddeeplearning_amd.parallel.p2p import P2PAllReduce
    base = ddp.FlatDataParallel()
    dp = P2PAllReduce(base, dev, 1 << 22, max_blocks=blocks, timeout_s=30.0)

    def timeit(fn, n=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        base.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    rows = []
    for n in (1024, 65536, 266752, 1066432, 1 << 22):
        x = torch.randn(n, device=dev)
        rows.append((n * 4 / 1e6, timeit(lambda: dp.all_reduce_sum_(x))))
    for name, spec in (("D", discriminator_spec()), ("G", generator_spec())):
        net = Network(spec, dev, 0)
        opt = FlatOptimizer(net, OptimConfig.tf_adam())
        t_sep = timeit(lambda: (dp.all_reduce_sum_(net.G), opt.step(dp.grad_scale)))
        t_fused = timeit(lambda: opt.step_allreduce_fused(dp))
        t_opt = timeit(lambda: opt.step(1.0))
        rows.append((name, net.G.numel() * 4 / 1e6, t_sep, t_fused, t_opt))
    healthy = dp.healthy()
    if rank == 0:
        print(f"world {world} ({ngpu} GPU(s) visible), max_blocks {blocks}, healthy {healthy}")
        for r in rows:
            if len(r) == 2:
                print(f"  all-reduce {r[0]:8.3f} MB : {r[1]:7.1f} us")
            else:
                print(f"  {r[0]} ({r[1]:.2f} MB): all-reduce + optimizer + relayouts {r[2]:6.1f} us | fused {r[3]:6.1f} us | "
                      f"optimizer + relayouts alone {r[4]:6.1f} us")
    base.barrier()
    dp.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--blocks", type=int, default=128)
    a = ap.parse_args()
    for w in a.world:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(body, args=(w, port, a.blocks), nprocs=w, join=True)
