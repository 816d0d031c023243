"""A few launches of the memory-bound BN kernels on DCGAN-sized tensors (target of rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributeddeeplearning_amd.ops import hip

rows, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (784000, 32)
x = torch.rand(rows, C, device="cuda").bfloat16()
dy = torch.randn(rows, C, device="cuda").bfloat16()
gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
mm = torch.zeros(C, device="cuda"); mv = torch.ones(C, device="cuda")
dbias = torch.zeros(C, device="cuda")
for _ in range(3):
    sc, sh, mean, invstd = hip.bn_train_affine(x, gamma, beta, mm, mv, 0.99, 1e-3, True, True)
    y = hip.bn_apply(x, sc, sh, None, 0.3, C)
    hip.bn_bwd(dy, x, y, mean, invstd, gamma, None, 0.3, True, "relu", 0.3, None, None, dbias)
    hip.bn_bwd(dy, x, y, mean, invstd, gamma, "leaky", 0.3, True, None, 0.3)
    hip.bias_grad(dy, dbias)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("stats", lambda: hip.bn_train_affine(x, gamma, beta, mm, mv, 0.99, 1e-3, True, True)),
                 ("apply", lambda: hip.bn_apply(x, sc, sh, None, 0.3, C)),
                 ("bwd3", lambda: hip.bn_bwd(dy, x, y, mean, invstd, gamma, None, 0.3, True, "relu", 0.3, None, None, dbias)),
                 ("bwd1", lambda: hip.bn_bwd(dy, x, y, mean, invstd, gamma, "leaky", 0.3, True, None, 0.3)),
                 ("bias", lambda: hip.bias_grad(dy, dbias))):
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); e1.synchronize()
    print(f"{name:6s} {e0.elapsed_time(e1) * 100:.1f} us  ({rows}x{C}, {rows * C * 2 / 1e6:.0f} MB/tensor)", flush=True)
