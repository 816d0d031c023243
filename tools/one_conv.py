#!/usr/bin/env python3
"""Run ONE conv case a few times (for rocprofv3 --pmc):  python tools/one_conv.py G4 [impl] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.ops import geometry as G  # noqa: E402
from distributeddeeplearning_amd.ops import hip  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "G4"
hip.CONV_IMPL = sys.argv[2] if len(sys.argv) > 2 else "pconv"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
CASES = {
    "G4": (G.conv2d_transpose_geom(14, 14, 64, 32, (5, 5), (2, 2), (1, 1), "same"), "dgrad"),
    "G4b": (G.conv2d_transpose_geom(14, 14, 64, 32, (5, 5), (2, 2), (1, 1), "same"), "fprop"),
    "G6": (G.conv2d_transpose_geom(28, 28, 32, 1, (5, 5), (1, 1), (1, 1), "same"), "dgrad"),
    "G2": (G.conv2d_transpose_geom(7, 7, 128, 64, (5, 5), (2, 2), (1, 1), "same"), "dgrad"),
    "D3": (G.conv2d_geom(14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same"), "fprop"),
}
g, mode = CASES[which]
Ip, Op = hip.pad8(g.I), hip.pad8(g.O)
r = lambda *s: (torch.randn(*s, device="cuda") * 0.1).bfloat16()
if mode == "fprop":
    x, w = r(B, g.H, g.W, Ip), r(g.O, g.KH, g.KW, Ip)
    fn = lambda: hip.conv_fprop(x, w, None, g)
else:
    x, w = r(B, g.OH, g.OW, Op), r(g.I, g.KH, g.KW, Op)
    fn = lambda: hip.conv_dgrad(x, w, g)
for _ in range(5):
    fn()
torch.cuda.synchronize()
