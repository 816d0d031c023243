"""Times MIOpen (through torch.nn.functional, bf16 channels_last) on the DCGAN28 convolution shapes, forward /
input gradient / weight gradient, so the hand-written kernels' times in profiles/ have a vendor yardstick.
Not part of the framework: torch's conv ops are only ever called here."""
import sys
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = "cuda"


def t_us(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


shapes = [  # name, transposed, Cin, Cout, Hin, stride
    ("D.c2 14x14x32->7x7x64", False, 32, 64, 14, 2),
    ("D.c3 7x7x64->4x4x128", False, 64, 128, 7, 2),
    ("G.ct1 7x7x128->14x14x64", True, 128, 64, 7, 2),
    ("G.ct2 14x14x64->28x28x32", True, 64, 32, 14, 2),
]
print(f"B={B}")
for name, tr, ci, co, h, s in shapes:
    x = torch.randn(B, ci, h, h, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if tr:
        w = torch.randn(ci, co, 5, 5, device=dev, dtype=torch.bfloat16)
        f = lambda: F.conv_transpose2d(x, w, stride=s, padding=2, output_padding=1)
    else:
        w = torch.randn(co, ci, 5, 5, device=dev, dtype=torch.bfloat16)
        f = lambda: F.conv2d(x, w, stride=s, padding=2)
    y = f()
    dy = torch.randn_like(y)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    if tr:
        g = lambda: F.conv_transpose2d(xr, wr, stride=s, padding=2, output_padding=1)
    else:
        g = lambda: F.conv2d(xr, wr, stride=s, padding=2)
    yy = g()
    dgrad = lambda: torch.autograd.grad(yy, xr, dy, retain_graph=True)
    wgrad = lambda: torch.autograd.grad(yy, wr, dy, retain_graph=True)
    print(f"{name:30s} fwd {t_us(f):7.1f} us   dgrad {t_us(dgrad):7.1f} us   wgrad {t_us(wgrad):7.1f} us", flush=True)
