"""Time the one-channel transposed-conv kernels (csrc/hip/upconv_1ch.hip) against the generic path on model0's first
layer (28x28x1 -> 57x57x64, k3 s2, softplus, B = 512): python tools/time_upconv.py [B]"""
import sys
import torch
sys.path.insert(0, ".")
from distributeddeeplearning_amd.ops import geometry as G, hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = G.conv2d_transpose_geom(28, 28, 1, 64, (3, 3), (2, 2), (1, 1), "valid")
x = hip.pad_channels(torch.rand(B, 28, 28, 1).cuda(), 8)
w = torch.zeros(64, 3, 3, 8)
w[..., 0] = torch.randn(64, 3, 3) / 3
w = w.bfloat16().cuda()
b = torch.randn(64).cuda()
dy = torch.randn(B, g.H, g.W, 64, device="cuda").bfloat16()
out = torch.empty(64, 3, 3, 8, device="cuda")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


mb = B * g.H * g.W * 64 * 2 / 1e6
for act in ("softplus", "relu"):
    us = t(lambda: hip.upconv_1ch(x, w, g, b, act, 0.3, stats_out={}))
    print(f"upconv fwd {act} + stats: {us:.1f} us ({mb / us * 1e-3 * 1e3:.0f} GB/s of stores)")
    us = t(lambda: hip.upconv_1ch(x, w, g, b, act, 0.3))
    print(f"upconv fwd {act}: {us:.1f} us")
us = t(lambda: hip.conv_dgrad(x, w, g, b, "softplus", 0.3, stats_out={}))
print(f"generic fwd + stats: {us:.1f} us")
us = t(lambda: hip.upconv_1ch_wgrad(x, dy, w, g, out))
print(f"upconv wgrad (+finish): {us:.1f} us ({mb / us:.0f} GB/s of dY)")
us = t(lambda: hip.conv_wgrad(dy, x, g, "B", out=out))
print(f"generic wgrad: {us:.1f} us")
