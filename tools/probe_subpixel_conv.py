"""Feasibility probe: a stride-2 5x5 transposed conv as ONE stride-1 3x3 conv with classes x channels outputs."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributeddeeplearning_amd.ops import hip, geometry as G

def t(fn):
    return hip._time_us(fn, reps=3, inner=2)

for (H, Cin, Cout, B) in ((32, 64, 32, 8192), (16, 128, 64, 16384), (8, 128, 64, 32768), (16, 64, 32, 32768)):
    # the equivalent sub-pixel problem: HxWxCin -> HxWx(4*Cout), 3x3 stride 1 same
    g = G.conv2d_geom(H, H, Cin, 4 * Cout, (3, 3), (1, 1), (1, 1), "same")
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = (torch.randn(4 * Cout, 3, 3, Cin, device="cuda") / 20).bfloat16()
    y = torch.empty(B, H, H, 4 * Cout, device="cuda", dtype=torch.bfloat16)
    geo = (B, H, H, Cin, H, H, 4 * Cout, 4 * Cout, (1, 1), (1, 1))
    res = {}
    for tile in (33, 34, 38):
        try:
            res[tile] = round(t(lambda: hip._run_igemm(x, w, None, y, g, "fprop", *geo, "linear", 0.3, tile=tile)), 1)
        except Exception as e:
            res[tile] = str(e)[:40]
    useful = B * (2 * H) ** 2 * Cout * Cin * 6.25 * 2
    print(res, flush=True); best = min(v for v in res.values() if isinstance(v, float))
    print(f"{H}x{H}x{Cin}->x{4*Cout} B={B}: {res}  useful {useful/1e12:.2f} TF -> {useful/best/1e6:.0f} TF/s effective", flush=True)
    del x, w, y
