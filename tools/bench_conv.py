#!/usr/bin/env python3
"""Per-layer conv micro-benchmark on the DCGAN shapes: pconv vs igemm (run on the GPU box).
    python tools/bench_conv.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.ops import geometry as G  # noqa: E402
from distributeddeeplearning_amd.ops import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = "cuda"


def t(fn, n=20):
    """GPU time per call: the n calls are captured into one hipGraph so host launch cost is excluded."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.1).bfloat16()


def main():
    cases = []
    # (name, geom, mode, batch)
    cases.append(("D1 fprop 1->32 s2", G.conv2d_geom(28, 28, 1, 32, (5, 5), (2, 2), (1, 1), "same"), "fprop", 2 * B))
    cases.append(("D3 fprop 32->64 s2", G.conv2d_geom(14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same"), "fprop", 2 * B))
    cases.append(("D6 fprop 64->128 s2", G.conv2d_geom(7, 7, 64, 128, (5, 5), (2, 2), (1, 1), "same"), "fprop", 2 * B))
    cases.append(("D6 dgrad", G.conv2d_geom(7, 7, 64, 128, (5, 5), (2, 2), (1, 1), "same"), "dgrad", 2 * B))
    cases.append(("D3 dgrad", G.conv2d_geom(14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same"), "dgrad", 2 * B))
    cases.append(("D1 dgrad", G.conv2d_geom(28, 28, 1, 32, (5, 5), (2, 2), (1, 1), "same"), "dgrad", B))
    cases.append(("G1 dense 128->6272", G.dense_geom(128, 6272), "fprop", B))
    cases.append(("G2 convT 128->64 s2 (dgrad form)", G.conv2d_transpose_geom(7, 7, 128, 64, (5, 5), (2, 2), (1, 1), "same"), "dgrad", B))
    cases.append(("G4 convT 64->32 s2 (dgrad form)", G.conv2d_transpose_geom(14, 14, 64, 32, (5, 5), (2, 2), (1, 1), "same"), "dgrad", B))
    cases.append(("G6 convT 32->1 s1 (dgrad form)", G.conv2d_transpose_geom(28, 28, 32, 1, (5, 5), (1, 1), (1, 1), "same"), "dgrad", B))
    cases.append(("G6 bwd (fprop form)", G.conv2d_transpose_geom(28, 28, 32, 1, (5, 5), (1, 1), (1, 1), "same"), "fprop", B))
    cases.append(("G4 bwd (fprop form)", G.conv2d_transpose_geom(14, 14, 64, 32, (5, 5), (2, 2), (1, 1), "same"), "fprop", B))
    cases.append(("G2 bwd (fprop form)", G.conv2d_transpose_geom(7, 7, 128, 64, (5, 5), (2, 2), (1, 1), "same"), "fprop", B))

    tot = {"pconv": 0.0, "igemm": 0.0}
    print(f"B={B}")
    for name, g, mode, b in cases:
        Ip, Op = hip.pad8(g.I), hip.pad8(g.O)
        if mode == "fprop":
            x = rnd(b, g.H, g.W, Ip)
            w = rnd(g.O, g.KH, g.KW, Ip)
            fn = lambda: hip.conv_fprop(x, w, None, g)
            macs = b * g.OH * g.OW * g.O * g.KH * g.KW * g.I
        else:
            x = rnd(b, g.OH, g.OW, Op)
            w = rnd(g.I, g.KH, g.KW, Op)
            fn = lambda: hip.conv_dgrad(x, w, g)
            macs = b * g.OH * g.OW * g.O * g.KH * g.KW * g.I
        row = []
        for impl in ("pconv", "igemm"):
            hip.CONV_IMPL = impl
            us = t(fn)
            tot[impl] += us
            row.append(f"{impl} {us:8.1f} us ({2 * macs / us / 1e6:7.1f} TF)")
        print(f"{name:36s} {row[0]}   {row[1]}", flush=True)
    print("total", tot)


if __name__ == "__main__":
    main()
