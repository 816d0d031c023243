#!/usr/bin/env python3
"""Weight-gradient micro-benchmark on the DCGAN shapes: wgrad2 vs wgrad (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.ops import geometry as G  # noqa: E402
from distributeddeeplearning_amd.ops import hip  # noqa: E402
from tools.bench_conv import t  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
r = lambda *s: (torch.randn(*s, device="cuda") * 0.1).bfloat16()
cases = [
    ("D3 32->64 s2", G.conv2d_geom(14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same"), 2 * B, "A"),
    ("D6 64->128 s2", G.conv2d_geom(7, 7, 64, 128, (5, 5), (2, 2), (1, 1), "same"), 2 * B, "A"),
    ("G2 convT 128->64", G.conv2d_transpose_geom(7, 7, 128, 64, (5, 5), (2, 2), (1, 1), "same"), B, "B"),
    ("G4 convT 64->32", G.conv2d_transpose_geom(14, 14, 64, 32, (5, 5), (2, 2), (1, 1), "same"), B, "B"),
    ("G1 dense 128->6272", G.dense_geom(128, 6272), B, "A"),
]
for name, g, b, layout in cases:
    x, dy = r(b, g.H, g.W, hip.pad8(g.I)), r(b, g.OH, g.OW, hip.pad8(g.O))
    row = []
    for impl in ("wgrad2", "wgrad"):
        hip.WGRAD_IMPL = impl
        us = t(lambda: hip.conv_wgrad(x, dy, g, layout))
        macs = b * g.OH * g.OW * g.O * g.KH * g.KW * g.I
        row.append(f"{impl} {us:8.1f} us ({2 * macs / us / 1e6:6.1f} TF)")
    print(f"{name:22s} {row[0]}   {row[1]}", flush=True)
