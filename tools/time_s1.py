"""Times hip.stream1_wgrad alone on the two DCGAN shapes (D.conv1: B=2000, G.convT3: B=1000 with affine)."""
import torch
from distributeddeeplearning_amd.ops import hip, geometry as G

g = G.conv2d_geom(28, 28, 1, 32, (5, 5), (2, 2), (1, 1), "same")
for B, aff in ((2000, False), (1000, True)):
    img = torch.randn(B, 28, 28, 8, device="cuda").bfloat16()
    t = torch.randn(B, 14, 14, 32, device="cuda").bfloat16()
    out = torch.empty(32 * 25 * 8, device="cuda")
    bias = torch.empty(32, device="cuda")
    a = (torch.rand(32, device="cuda") + 0.5, torch.randn(32, device="cuda"), 32) if aff else None
    def run():
        if aff:
            return hip.stream1_wgrad(img, t, g, out, 1, 32, affine=a, bias=bias, bias_mode=2)
        return hip.stream1_wgrad(img, t, g, out, 200, 8, zfill=8, n_out=32, bias=bias, bias_mode=1)
    assert run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} affine={aff}: {e0.elapsed_time(e1) / 200 * 1000:.1f} us per call (kernel + finish)")
