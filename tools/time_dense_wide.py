"""Device time of the generator's Dense 128 -> 6272 (gemm_sk wide path) at 2000 rows; DDL_SK_WIDE_BLOCKS sweeps
the grid cap.  usage: python tools/time_dense_wide.py [rows]"""
import sys
import torch
from distributeddeeplearning_amd.ops import hip
from distributeddeeplearning_amd.ops.geometry import dense_geom

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
x = torch.randn(M, 1, 1, 128, device="cuda").bfloat16()
w = torch.randn(6272, 1, 1, 128, device="cuda").bfloat16()
b = torch.randn(6272, device="cuda")
g = dense_geom(128, 6272)
f = lambda: hip.conv_fprop(x, w, b, g, None, 0.0)
y = f()
ref = x.reshape(M, 128).float() @ w.reshape(6272, 128).float().t() + b
print("max err", (y.reshape(M, -1).float() - ref).abs().max().item(), "us", round(hip._time_us(f, reps=6, inner=8), 2))
