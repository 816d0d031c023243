"""Child-process probe of the xGMI all-reduce kernel (launched by parallel/dp_tune.probe_p2p, one per rank).

Joins a gloo group of the probes (RANK / WORLD_SIZE / MASTER_* from the parent), opens the SAME kind of IPC peer
mappings on the parent's device and checks, against host sums in rank order: one-shot and two-shot all-reduces
(both buffer parities), the fused all-reduce + Adam epilogue against all-reduce followed by the stand-alone optimizer
kernel, a bucketed (ranged) fused update, and the kernel's health word.  Exit code 0 = all of it held on this rank.
Anything else -- an exception, a fault that kills the process, a time-out -- is the parent's signal to stay on RCCL.
"""
import os
import sys


def main() -> int:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    idx = int(os.environ.get("DDL_PROBE_DEVICE", "0"))
    cap = int(os.environ.get("DDL_PROBE_CAPACITY", str(1 << 21)))
    torch.cuda.set_device(idx)
    dev = torch.device(f"cuda:{idx}")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distributeddeeplearning_amd.parallel import ddp
    from distributeddeeplearning_amd.parallel.p2p import P2PAllReduce
    base = ddp.FlatDataParallel()
    dp = P2PAllReduce(base, dev, cap, timeout_s=20.0)

    def host_sum(x):
        parts = [torch.empty_like(x.cpu()) for _ in range(world)]
        dist.all_gather(parts, x.cpu())
        acc = parts[0].clone()
        for p in parts[1:]:
            acc += p
        return acc

    gen = torch.Generator().manual_seed(7 + rank)
    ok = True
    for n in (1024, 52224, 207240, min(cap, 1066184) // 8 * 8, 1024):
        x = torch.randn(n, generator=gen).to(dev)
        ref = host_sum(x)
        y = x.clone()
        dp.all_reduce_sum_(y)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(y.cpu(), ref))
    # fused update, whole buffer and as two ranges, against all-reduce + optimizer kernel
    from distributeddeeplearning_amd.models.dcgan import discriminator_spec
    from distributeddeeplearning_amd.models.network import Network
    from distributeddeeplearning_amd.models.optim import FlatOptimizer, OptimConfig
    nets = [Network(discriminator_spec(), dev, 3) for _ in range(3)]
    opts = [FlatOptimizer(n, OptimConfig.tf_adam(2e-4)) for n in nets]
    (lo, hi, _), _rest = nets[0].grad_buckets(2)
    for _ in range(2):
        g = torch.randn(nets[0].G.numel(), generator=gen).to(dev) * 0.01
        for n in nets:
            n.G.copy_(g)
        dp.all_reduce_sum_(nets[0].G)
        opts[0].step(dp.grad_scale)
        opts[1].step_allreduce_fused(dp)
        opts[2].step_allreduce_fused(dp, lo, hi, True, False)
        opts[2].step_allreduce_fused(dp, 0, lo, False, True)
    torch.cuda.synchronize()
    for n in nets[1:]:
        ok = ok and bool(torch.equal(nets[0].P, n.P)) and bool(torch.equal(nets[0].Pc, n.Pc))
    ok = ok and dp.healthy()
    f = torch.tensor([0.0 if ok else 1.0])
    dist.all_reduce(f, op=dist.ReduceOp.MAX)
    base.barrier()
    dp.close()
    dist.destroy_process_group()
    return 0 if float(f.item()) == 0.0 else 3


if __name__ == "__main__":
    sys.exit(main())
