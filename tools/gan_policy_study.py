"""CPU study of DCGAN BatchNorm / optimizer policies with plain PyTorch autograd (tool only: never on the
product path).  Same graphs, padding geometry, initialisers and step order as models/dcgan.py /
GAN_distributed.py:15-117; small minibatches so 2000 iterations run in minutes on 8 CPU threads.

    python tools/gan_policy_study.py --policy split --iters 2000 --batch 128 --out gpurun_out/study

Policies
  mixed      D phase: D(train BN) on [real;fake] as ONE batch, fake from G in inference BN;
             G phase: G(train BN), D(train BN) on the fake-only batch       (round-1 default: BN leak)
  split      D phase: batch statistics per half (real / fake separately);  G phase as above
  split_gtrain   as split, and the D-phase fakes also come from G in training-mode BN
  inference  BN in inference phase everywhere (moving stats stay 0/1), persistent Adam
  refexact   inference BN + fresh-Adam-applied-twice (= 2 lr sign(g)), the reference's literal semantics
Metric: mean nearest-neighbour L2 distance of 256 generated samples to the training images (lower = closer
to the data; real-vs-real hold-out distance and uniform-noise distance are printed for scale).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.utils.data import synthetic_mnist  # noqa: E402


def glorot(shape, fan_in, fan_out, gen):
    lim = (6.0 / (fan_in + fan_out)) ** 0.5
    return ((torch.rand(shape, generator=gen) * 2 - 1) * lim).requires_grad_()


class BN:
    def __init__(self, c):
        self.g = torch.ones(c, requires_grad=True)
        self.b = torch.zeros(c, requires_grad=True)
        self.mm, self.mv = torch.zeros(c), torch.ones(c)

    def __call__(self, x, train, groups=1):
        if not train:
            return F.batch_norm(x, self.mm, self.mv, self.g, self.b, False, 0.0, 1e-3)
        outs = []
        for xs in x.chunk(groups, 0):
            outs.append(F.batch_norm(xs, self.mm, self.mv, self.g, self.b, True, 0.01, 1e-3))
        return torch.cat(outs, 0)

    def params(self):
        return [self.g, self.b]


class Disc:
    def __init__(self, gen):
        self.w1, self.b1 = glorot((32, 1, 5, 5), 25, 25 * 32, gen), torch.zeros(32, requires_grad=True)
        self.w2, self.b2 = glorot((64, 32, 5, 5), 25 * 32, 25 * 64, gen), torch.zeros(64, requires_grad=True)
        self.w3, self.b3 = glorot((128, 64, 5, 5), 25 * 64, 25 * 128, gen), torch.zeros(128, requires_grad=True)
        self.wd, self.bd = glorot((1, 2048), 2048, 1, gen), torch.zeros(1, requires_grad=True)
        self.bn2, self.bn3 = BN(64), BN(128)

    def params(self):
        return [self.w1, self.b1, self.w2, self.b2, self.w3, self.b3, self.wd, self.bd] + self.bn2.params() + \
            self.bn3.params()

    def __call__(self, x, train, groups=1):
        x = F.leaky_relu(F.conv2d(F.pad(x, (1, 2, 1, 2)), self.w1, self.b1, 2), 0.3)
        x = F.leaky_relu(self.bn2(F.conv2d(F.pad(x, (1, 2, 1, 2)), self.w2, self.b2, 2), train, groups), 0.3)
        x = F.leaky_relu(self.bn3(F.conv2d(F.pad(x, (2, 2, 2, 2)), self.w3, self.b3, 2), train, groups), 0.3)
        return x.flatten(1) @ self.wd.t() + self.bd


class Gen:
    def __init__(self, gen):
        self.wd, self.bd = glorot((6272, 128), 128, 6272, gen), torch.zeros(6272, requires_grad=True)
        self.w1, self.b1 = glorot((128, 64, 5, 5), 25 * 128, 25 * 64, gen), torch.zeros(64, requires_grad=True)
        self.w2, self.b2 = glorot((64, 32, 5, 5), 25 * 64, 25 * 32, gen), torch.zeros(32, requires_grad=True)
        self.w3, self.b3 = glorot((32, 1, 5, 5), 25 * 32, 25, gen), torch.zeros(1, requires_grad=True)
        self.bn1, self.bn2 = BN(64), BN(32)

    def params(self):
        return [self.wd, self.bd, self.w1, self.b1, self.w2, self.b2, self.w3, self.b3] + self.bn1.params() + \
            self.bn2.params()

    def __call__(self, z, train):
        x = (z @ self.wd.t() + self.bd).reshape(-1, 128, 7, 7)
        # TF 'same' transposed conv, k5 s2: the adjoint of pad (1,2) forward == padding 2 with output_padding 1
        # shifted by one pixel; equivalent up to a relabelling of kernel taps, irrelevant for training dynamics
        x = self.bn1(F.relu(F.conv_transpose2d(x, self.w1, self.b1, 2, 2, 1)), train)
        x = self.bn2(F.relu(F.conv_transpose2d(x, self.w2, self.b2, 2, 2, 1)), train)
        x = torch.tanh(F.conv_transpose2d(x, self.w3, self.b3, 1, 2))
        return (x + 1) * 0.5


class Adam:
    def __init__(self, params, lr, fresh_double):
        self.p, self.lr, self.fd, self.t = params, lr, fresh_double, 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    def step(self, grads):
        if self.fd:
            with torch.no_grad():
                for p, g in zip(self.p, grads):
                    m, v = torch.zeros_like(p), torch.zeros_like(p)
                    for t in (1, 2):
                        m.mul_(0.9).add_(g, alpha=0.1)
                        v.mul_(0.999).addcmul_(g, g, value=0.001)
                        p.addcdiv_(m, v.sqrt() + 1e-8, value=-self.lr * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t))
            return
        self.t += 1
        lr_t = self.lr * (1 - 0.999 ** self.t) ** 0.5 / (1 - 0.9 ** self.t)
        with torch.no_grad():
            for p, g, m, v in zip(self.p, grads, self.m, self.v):
                m.mul_(0.9).add_(g, alpha=0.1)
                v.mul_(0.999).addcmul_(g, g, value=0.001)
                p.addcdiv_(m, v.sqrt() + 1e-8, value=-lr_t)


def nn_dist(a, b):
    """mean over rows of a of the min L2 distance (per pixel rms) to rows of b"""
    d = torch.cdist(a.flatten(1), b.flatten(1))
    return float(d.min(1).values.mean()) / 28.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--policy", default="split")
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--out", default="gpurun_out/study")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    os.makedirs(a.out, exist_ok=True)
    gen = torch.Generator().manual_seed(0)
    (tx, _), (vx, _) = synthetic_mnist(1000, 256, 0)
    real_all = torch.from_numpy(tx).float().div(255).unsqueeze(1)
    hold = torch.from_numpy(vx).float().div(255).unsqueeze(1)
    D, G = Disc(gen), Gen(gen)
    pol = a.policy
    inf = pol in ("inference", "refexact")
    optD, optG = Adam(D.params(), 2e-4, pol == "refexact"), Adam(G.params(), 2e-4, pol == "refexact")
    B = a.batch
    log = open(os.path.join(a.out, f"{pol}.jsonl"), "w")
    print(f"[{pol}] real-vs-real nn {nn_dist(hold, real_all):.4f}  noise nn "
          f"{nn_dist(torch.rand(256, 1, 28, 28), real_all):.4f}", flush=True)
    t0 = time.time()
    for it in range(a.iters):
        real = real_all[torch.randint(0, 1000, (B,), generator=gen)]
        z = torch.rand(B, 128, generator=gen) * 2 - 1
        with torch.no_grad():
            fake = G(z, train=(pol == "split_gtrain"))
        xs = torch.cat([real, fake], 0)
        logits = D(xs, train=not inf, groups=2 if pol.startswith("split") else 1)
        ys = torch.cat([torch.ones(B, 1), torch.zeros(B, 1)], 0)
        dl = F.binary_cross_entropy_with_logits(logits, ys)
        optD.step(torch.autograd.grad(dl, D.params()))
        z = torch.rand(B, 128, generator=gen) * 2 - 1
        lg = D(G(z, train=not inf), train=not inf)
        gl = F.binary_cross_entropy_with_logits(lg, torch.ones(B, 1))
        optG.step(torch.autograd.grad(gl, G.params()))
        if it % 50 == 0 or it == a.iters - 1:
            with torch.no_grad():
                s = G(torch.rand(256, 128, generator=gen) * 2 - 1, train=False)
                s2 = G(torch.rand(256, 128, generator=gen) * 2 - 1, train=not inf)
            rec = dict(it=it, d_loss=float(dl), g_loss=float(gl), nn_eval=nn_dist(s, real_all),
                       nn_train=nn_dist(s2, real_all), logit_real=float(logits[:B].mean()),
                       logit_fake=float(logits[B:].mean()), logit_gphase=float(lg.mean()),
                       sec=time.time() - t0)
            log.write(json.dumps(rec) + "\n")
            log.flush()
            if it % 250 == 0 or it == a.iters - 1:
                print(f"[{pol}] " + " ".join(f"{k}={v:.4f}" if isinstance(v, float) else f"{k}={v}"
                                               for k, v in rec.items()), flush=True)
    from PIL import Image
    with torch.no_grad():
        s = G(torch.rand(32, 128, generator=gen) * 2 - 1, train=False)
    grid = s.reshape(4, 8, 28, 28).permute(0, 2, 1, 3).reshape(4 * 28, 8 * 28).clamp(0, 1).numpy()
    Image.fromarray((grid * 255).astype(np.uint8)).save(os.path.join(a.out, f"{pol}.png"))


if __name__ == "__main__":
    main()
