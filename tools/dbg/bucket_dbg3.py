import os, sys, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def run(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0); dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer
    from distributeddeeplearning_amd.parallel import ddp
    from distributeddeeplearning_amd.parallel.p2p import P2PAllReduce
    class HB(ddp.FlatDataParallel):
        def broadcast_(self, *tensors, src=0):
            for t in tensors:
                h = t.cpu(); dist.broadcast(h, src=src); t.copy_(h)
    base = HB()
    dp = P2PAllReduce(base, dev, 1 << 21, max_blocks=int(os.environ.get("MB", "8")), timeout_s=30.0)
    real = torch.rand(64, 28, 28, 1, generator=torch.Generator().manual_seed(rank))
    N = int(os.environ.get("NSTEP", "6"))
    def go(fused, buckets, inline):
        dp.supports_fused_update = fused
        t = GANTrainer(GANConfig(batch=64, seed=5), dev, dp, rank)
        t.buckets = buckets
        if inline:
            t._on_comm = lambda fn: fn()
            t._join_comm = lambda: None
        rp = t.D.prepare_input(real)
        for _ in range(N):
            t.train_step(rp, prepared=True)
        torch.cuda.synchronize(); base.barrier()
        return t
    first = go(True, True, False) if os.environ.get("FIRST") == "fb" else None
    ref = go(False, False, False)
    if first is not None and rank == 0:
        dD = (first.D.P - ref.D.P).abs(); dG = (first.G.P - ref.G.P).abs()
        print(f"rep -1 FIRST fused+buckets: D n={int((dD>0).sum())} max={float(dD.max()):.2e}  G n={int((dG>0).sum())} max={float(dG.max()):.2e}", flush=True)
    for rep in range(4):
        for tag, args in (("fused+buckets", (True, True, False)), ("fused+buckets inline", (True, True, True)),
                          ("unfused+buckets", (False, True, False)), ("fused flat", (True, False, False))):
            t = go(*args)
            dD = (t.D.P - ref.D.P).abs(); dG = (t.G.P - ref.G.P).abs()
            if rank == 0:
                print(f"rep {rep} {tag}: D n={int((dD>0).sum())} max={float(dD.max()):.2e}  G n={int((dG>0).sum())} max={float(dG.max()):.2e}", flush=True)
    base.barrier(); dp.close(); dist.destroy_process_group()

if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(run, args=(2, port), nprocs=2, join=True)
