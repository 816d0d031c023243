import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "64"))
t = GANTrainer(GANConfig(batch=B, seed=5), dev)
real = torch.rand(B, 28, 28, 1, generator=torch.Generator().manual_seed(0))
rp = t.D.prepare_input(real)
for step in range(3):
    snaps = {}
    for net, name in ((t.D, "D"), (t.G, "G")):
        bk = net.grad_buckets(2)
        lo, hi, ready = bk[0]
        def hook(i, net=net, name=name, lo=lo, hi=hi, ready=ready):
            if i == ready:
                snaps[name] = net.G[lo:hi].clone()
        net._dbg_hook = hook
    t._bucket_hook = lambda net, opt: net._dbg_hook
    t.phase_a(rp)
    torch.cuda.synchronize()
    lo, hi, _ = t.D.grad_buckets(2)[0]
    d = (t.D.G[lo:hi] - snaps["D"]).abs()
    print(f"step {step} D tail late-written elements: {int((d > 0).sum())} max {float(d.max()):.3e}", end="")
    if d.max() > 0:
        idx = torch.nonzero(d > 0).flatten(); print(f" range {int(idx.min()) + lo}..{int(idx.max()) + lo}", end="")
    print()
    t.phase_b(B)
    torch.cuda.synchronize()
    lo, hi, _ = t.G.grad_buckets(2)[0]
    d = (t.G.G[lo:hi] - snaps["G"]).abs()
    print(f"step {step} G tail late-written elements: {int((d > 0).sum())} max {float(d.max()):.3e}", end="")
    if d.max() > 0:
        idx = torch.nonzero(d > 0).flatten(); print(f" range {int(idx.min()) + lo}..{int(idx.max()) + lo}", end="")
    print()
    t.phase_c()
print({(p.layer, p.wname): p.offset for p in t.D._params if p.trainable})
print({(p.layer, p.wname): p.offset for p in t.G._params if p.trainable})
