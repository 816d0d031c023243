#!/usr/bin/env python3
"""Every wgrad2 candidate plan on the four DCGAN28 5x5 layers, timed alone (run on the GPU box):
`python tools/time_wgrad2_plans.py [batch]`."""
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
    ("D.conv2 32->64", G.conv2d_geom(14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same"), 2 * B, "A"),
    ("D.conv3 64->128", G.conv2d_geom(7, 7, 64, 128, (5, 5), (2, 2), (1, 1), "same"), 2 * B, "A"),
    ("G.convT1 128->64", G.conv2d_transpose_geom(7, 7, 128, 64, (5, 5), (2, 2), (1, 1), "same"), B, "B"),
    ("G.convT2 64->32", G.conv2d_transpose_geom(14, 14, 64, 32, (5, 5), (2, 2), (1, 1), "same"), B, "B"),
]
hip.WGRAD_IMPL = "wgrad2"
for name, g, b, layout in cases:
    Ip, Op, T = hip.pad8(g.I), hip.pad8(g.O), g.KH * g.KW
    x, dy = r(b, g.H, g.W, Ip), r(b, g.OH, g.OW, Op)
    if layout == "A":
        box = (g.O, Ip, T * Ip, Ip, 1)
    else:
        box = (Op, g.I, 1, Op, T * Op)
    plans = hip._wgrad2_plans(g, b, Ip, Op, box[0], box[1])
    key = ("wgrad", g, b, Ip, Op) + box
    rows = []
    for i, pl in enumerate(plans):
        hip._CONV_CHOICE[key] = i
        us = t(lambda: hip.conv_wgrad(x, dy, g, layout))
        rows.append((us, i, pl))
    rows.sort(key=lambda z: z[0])
    print(f"== {name} B={b}", flush=True)
    for us, i, pl in rows[:8]:
        print(f"  {us:7.1f} us  plan {i:2d}  " +
              " ".join(f"{k}={pl[k]}" for k in ("TN", "TC", "TPW", "NW", "IMG", "BR", "ngroups", "nsplit", "budget")), flush=True)
