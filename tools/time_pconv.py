"""Device time of one patch-resident convolution with a fixed plan (no autotuning): the generator's two stride-2
transposed convolutions and the discriminator's conv2 input gradient.  usage: time_pconv.py <ct1|ct2|dc2> [B]"""
import sys
import torch
from distributeddeeplearning_amd.ops import hip
from distributeddeeplearning_amd.ops.geometry import conv2d_geom

which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cfg = {"ct1": (14, 64, 128, dict(budget=160 * 1024, img=5, fr=4)),      # core conv 14x14x64 -> 7x7x128, run as dgrad
       "ct2": (28, 32, 64, dict(budget=80 * 1024, img=1, fr=4)),
       "dc2": (14, 32, 64, dict(budget=80 * 1024, img=4, fr=4)),
       "ct1w2": (14, 64, 128, dict(budget=160 * 1024, img=5, fr=2, nw=8)),
       "ct1w4": (14, 64, 128, dict(budget=160 * 1024, img=5, fr=4, nw=8)),
       "ct1v2": (14, 64, 128, dict(budget=80 * 1024, img=2, fr=2, nw=8)),
       "ct2w2": (28, 32, 64, dict(budget=80 * 1024, img=1, fr=2, nw=8)),
       "ct2w4": (28, 32, 64, dict(budget=80 * 1024, img=1, fr=4, nw=8)),
       "dc2w2": (14, 32, 64, dict(budget=80 * 1024, img=4, fr=2, nw=8)),
       "dc2w4": (14, 32, 64, dict(budget=80 * 1024, img=4, fr=4, nw=8))}[which]
H, I, O, kw = cfg
g = conv2d_geom(H, H, I, O, (5, 5), (2, 2), (1, 1), "same")
dev = torch.device("cuda")
dy = torch.randn(B, g.OH, g.OW, O, device=dev).bfloat16()
w = (torch.randn(I, 5, 5, O, device=dev) * 0.05).bfloat16()
dx = torch.empty(B, H, H, I, device=dev, dtype=torch.bfloat16)
pl = hip._pconv_plan(g, "dgrad", B, O, I, (1, 1), dev, min_rows=64, **kw)
stats = torch.empty(2 * I * hip._pconv_units(pl, B), dtype=torch.float32, device=dev) if "--stats" in sys.argv else None
f = lambda: hip._run_pconv(pl, dy, w, None, dx, g, B, g.OH, g.OW, O, H, H, I, I, (1, 1), (g.sh, g.sw), "relu", 0.0,
                           stats=stats)
print(which, B, f"IMG={pl['IMG']} BR={pl['BR']} FR={pl['FR']} NW={pl['NW']} BN={pl['BN']} nst={pl['nst']}", "us", round(hip._time_us(f, reps=6, inner=8), 2))
