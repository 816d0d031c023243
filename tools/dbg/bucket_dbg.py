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
    dp = P2PAllReduce(base, dev, 1 << 21, max_blocks=8, timeout_s=30.0)
    real = torch.rand(64, 28, 28, 1, generator=torch.Generator().manual_seed(rank))
    res = {}
    for fused in (True, False):
        for buckets in (True, False):
            dp.supports_fused_update = fused
            t = GANTrainer(GANConfig(batch=64, seed=5), dev, dp, rank)
            t.buckets = buckets
            rp = t.D.prepare_input(real)
            for _ in range(int(os.environ.get("NSTEP", "1"))):
                t.train_step(rp, prepared=True)
            torch.cuda.synchronize()
            res[(fused, buckets)] = (t.D.P.clone(), t.G.P.clone(), t)
            if fused:
                from distributeddeeplearning_amd.models.dcgan import GraphedGANStep
                t = GANTrainer(GANConfig(batch=64, seed=5), dev, dp, rank)
                t.buckets = buckets
                st = GraphedGANStep(t, rp, warmup=2)
                for _ in range(int(os.environ.get("NSTEP", "1")) - 2):
                    st.step()
                torch.cuda.synchronize()
                res[("graph", buckets)] = (t.D.P.clone(), t.G.P.clone(), t)
    ref = res[(False, False)]
    for k, v in res.items():
        dD = (v[0] - ref[0]).abs(); dG = (v[1] - ref[1]).abs()
        msg = f"rank {rank} fused={k[0]} buckets={k[1]}: D maxdiff {float(dD.max()):.3e} n={int((dD>0).sum())} G maxdiff {float(dG.max()):.3e} n={int((dG>0).sum())}"
        if dD.max() > 0:
            idx = torch.nonzero(dD > 0).flatten(); msg += f" D idx range {int(idx.min())}..{int(idx.max())}"
        if dG.max() > 0:
            idx = torch.nonzero(dG > 0).flatten(); msg += f" G idx range {int(idx.min())}..{int(idx.max())}"
        print(msg, flush=True)
    if rank == 0:
        t = ref[2]; print("buckets D", t._bk[id(t.D)], "G", t._bk[id(t.G)], flush=True)
    base.barrier(); dp.close(); dist.destroy_process_group()

if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(run, args=(2, port), nprocs=2, join=True)
