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
    for rep in range(3):
        dp.supports_fused_update = True
        A = GANTrainer(GANConfig(batch=64, seed=5), dev, dp, rank)
        dp.supports_fused_update = False
        R = GANTrainer(GANConfig(batch=64, seed=5), dev, dp, rank)
        R.buckets = False
        rp = A.D.prepare_input(real)
        def rng(net, d):
            idx = torch.nonzero(d > 0).flatten()
            names = sorted({(p.layer, p.wname) for p in net._params if p.trainable and ((idx >= p.offset) & (idx < p.offset + p.numel)).any()})
            return f"n={idx.numel()} max={float(d.max()):.2e} in {names}"
        for step in range(6):
            A.train_step(rp, prepared=True); torch.cuda.synchronize(); base.barrier()
            R.train_step(rp, prepared=True); torch.cuda.synchronize(); base.barrier()
            for name, x, y, net in (("D.P", A.D.P, R.D.P, A.D), ("D.m", A.optD.m, R.optD.m, A.D), ("D.v", A.optD.v, R.optD.v, A.D), ("D.Pc", A.D.Pc.float(), R.D.Pc.float(), A.D),
                                    ("G.P", A.G.P, R.G.P, A.G), ("G.m", A.optG.m, R.optG.m, A.G), ("G.v", A.optG.v, R.optG.v, A.G)):
                d = (x - y).abs()
                if d.max() > 0 and rank == 0:
                    print(f"rep {rep} step {step} {name}: {rng(net, d)}", flush=True)
            if rank == 0:
                print(f"rep {rep} step {step} counters A {A.optD.step_dev.tolist()} {A.optG.step_dev.tolist()} R {R.optD.step_dev.tolist()} {R.optG.step_dev.tolist()}", flush=True)
    base.barrier(); dp.close(); dist.destroy_process_group()

if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(run, args=(2, port), nprocs=2, join=True)
