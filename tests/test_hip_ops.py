"""HIP kernels vs the fp32 CPU oracle (ops/cpu_ref.py) on bf16-rounded inputs.

Inputs are rounded to bf16 first, so the only differences left are fp32 accumulation order and
the final bf16 rounding of the kernel output (rel ~ 2^-8)."""
import math

import pytest
import torch

from distributeddeeplearning_amd.ops import cpu_ref
from distributeddeeplearning_amd.ops import geometry as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from distributeddeeplearning_amd.ops import hip as h
    return h


def r16(t):
    return t.bfloat16().float()


def dev(hip, x):
    """logical fp32 [.., C] -> device bf16 padded"""
    return hip.pad_channels(x.contiguous().cuda(), hip.pad8(x.shape[-1]))


def host(hip, y, c):
    return hip.strip_channels(y, c).cpu()


def padw(hip, w, lead_true=True):
    """[R][KH][KW][C] logical -> bf16 device with C padded"""
    Cp = hip.pad8(w.shape[-1])
    out = torch.zeros(*w.shape[:-1], Cp)
    out[..., :w.shape[-1]] = w
    return out.bfloat16().cuda().contiguous()


def close(a, b, rtol=2e-2, atol=None):
    scale = b.abs().max().item() + 1e-6
    atol = atol if atol is not None else 1e-2 * scale
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    bad = err > (atol + rtol * b.abs())
    assert not bad.any(), f"max err {err.max().item():.4g} (scale {scale:.4g}), {int(bad.sum())} bad"


CONV_CASES = [
    # H, W, I, O, k, s, d, padding, act
    (28, 28, 1, 32, (5, 5), (2, 2), (1, 1), "same", "leaky"),        # D1
    (14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same", "linear"),      # D3
    (7, 7, 64, 128, (5, 5), (2, 2), (1, 1), "same", "linear"),       # D6
    (28, 28, 3, 19, (29, 27), (1, 1), (1, 1), "same", "softsign"),   # DDL.py kernel larger than image
    (28, 28, 1, 32, (1, 1), (1, 1), (2, 2), "valid", "sigmoid"),     # model1 conv2d_1
    (28, 28, 32, 128, (1, 1), (3, 3), (1, 1), "valid", "relu"),      # model1 conv2d_2
    (26, 26, 128, 8, (3, 3), (3, 3), (1, 1), "valid", "sigmoid"),    # model3 conv2d_5
    (13, 11, 16, 24, (3, 2), (1, 2), (2, 1), "valid", "tanh"),
    (9, 9, 40, 72, (3, 3), (1, 1), (1, 1), "same", "softplus"),
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("B", [3, 70])
def test_conv_fprop_dgrad_wgrad(hip, case, B):
    H, W, I, O, k, s, d, pad, act = case
    g = G.conv2d_geom(H, W, I, O, k, s, d, pad)
    torch.manual_seed(1)
    x = r16(torch.randn(B, H, W, I))
    w = r16(torch.randn(O, k[0], k[1], I) / math.sqrt(k[0] * k[1] * I))
    b = torch.randn(O)
    y_ref = cpu_ref.conv_fprop(x, w, b, g, act)
    y = hip.conv_fprop(dev(hip, x), padw(hip, w), b.cuda(), g, act)
    assert y.shape[-1] == hip.pad8(O)
    if hip.pad8(O) != O:
        assert float(y[..., O:].float().abs().max()) == 0.0
    close(host(hip, y, O), y_ref)
    # dgrad (weights in layout B = [I][KH][KW][Op])
    dy = r16(torch.randn(B, g.OH, g.OW, O))
    dx_ref = cpu_ref.conv_dgrad(dy, w, g)
    wB = padw(hip, w.permute(3, 1, 2, 0).contiguous())
    dx = hip.conv_dgrad(dev(hip, dy), wB, g)
    close(host(hip, dx, I), dx_ref)
    # wgrad, both layouts
    dw_ref = cpu_ref.conv_wgrad(x, dy, g)
    dwA = hip.conv_wgrad(dev(hip, x), dev(hip, dy), g, "A").cpu()
    close(dwA[..., :I], dw_ref, rtol=1e-2)
    if hip.pad8(I) != I:
        assert float(dwA[..., I:].abs().max()) == 0.0
    dwB = hip.conv_wgrad(dev(hip, x), dev(hip, dy), g, "B").cpu()
    close(dwB[..., :O], dw_ref.permute(3, 1, 2, 0), rtol=1e-2)


@pytest.mark.parametrize("B", [3, 500])
def test_wgrad_bias_tail_from_ones_column(hip, B):
    """One-channel input conv: im2col_1ch(ones=True) puts 1.0 in the first padding tap, and the streaming weight
    gradient routes that column of dW to out[off + n] -- the bias gradient sum_m dY[m][n] from the same launch.
    The weight part must equal the plain call bit for bit; the bias part is compared with the fp32 column sum."""
    g = G.conv2d_geom(28, 28, 1, 32, (5, 5), (2, 2), (1, 1), "same")
    T, Tp = 25, 32
    g1 = G.ConvGeom(g.OH, g.OW, T, g.OH, g.OW, g.O, 1, 1).check()
    torch.manual_seed(21)
    img = torch.zeros(B, 28, 28, 8)
    img[..., 0] = torch.randn(B, 28, 28)
    img = img.bfloat16().cuda().contiguous()
    dy = torch.randn(B, g.OH, g.OW, 32).bfloat16().cuda().contiguous()
    numel = g.O * T * 8
    plain = torch.zeros(numel, device="cuda")
    hip.conv_wgrad(hip.im2col_1ch(img, g), dy, g1, "A", out=plain, box=(g.O, T, T * 8, 0, 8, numel))
    xc1 = hip.im2col_1ch(img, g, ones=True)
    assert float(xc1[..., T].float().min()) == 1.0 and float(xc1[..., T + 1:].float().abs().max()) == 0.0
    off = numel + 64                                    # an alignment gap between the two gradient slots
    flat = torch.zeros(off + g.O, device="cuda")
    hip.conv_wgrad(xc1, dy, g1, "A", out=flat, box=(g.O, T, T * 8, 0, 8, flat.numel()), bias_tail=(T, off))
    assert torch.equal(flat[:numel], plain)
    assert float(flat[numel:off].abs().max()) == 0.0
    ref = dy.float().sum(dim=(0, 1, 2))
    close(flat[off:].cpu(), ref.cpu(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()) + 1e-3)
    with pytest.raises(ValueError):
        hip.conv_wgrad(xc1, dy, g1, "A", out=flat, box=(g.O, T, T * 8, 0, 8, flat.numel()), bias_tail=(T, flat.numel()))


@pytest.mark.parametrize("B", [3, 500])
@pytest.mark.parametrize("affine", [False, True])
def test_wgrad_bias_tail_from_a_row_sum_of_the_dy_operand(hip, B, affine):
    """One-channel transposed conv: the weight gradient's dY operand is the taps tensor of the image gradient, and
    its centre-tap row summed over all pixels (one more MFMA against a fragment of ones) is the scalar bias gradient,
    written at out[off].  The weight part must equal the plain call bit for bit (also with the on-load affine)."""
    g = G.conv2d_geom(28, 28, 1, 32, (5, 5), (1, 1), (1, 1), "same")
    T, Tp, C = 25, 32, 32
    g1 = G.ConvGeom(g.OH, g.OW, C, g.OH, g.OW, T, 1, 1).check()
    torch.manual_seed(22)
    img = torch.zeros(B, 28, 28, 8)
    img[..., 0] = torch.randn(B, 28, 28)
    img = img.bfloat16().cuda().contiguous()
    x = torch.randn(B, 28, 28, C).bfloat16().cuda().contiguous()
    xa = (torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, C) if affine else None
    d = hip.im2col_1ch(img, g)
    plain = hip.conv_wgrad(x, d, g1, "A", x_affine=xa).reshape(-1)
    numel = T * C
    assert plain.numel() == numel
    off = numel + 8
    flat = torch.zeros(off + 4, device="cuda")
    hip.conv_wgrad(x, d, g1, "A", out=flat, x_affine=xa, box=(T, C, C, C, 1, flat.numel()), bias_tail=("row", 12, off))
    assert torch.equal(flat[:numel], plain)
    assert float(flat[numel:off].abs().max()) == 0.0 and float(flat[off + 1:].abs().max()) == 0.0
    ref = float(img[..., 0].float().sum())
    assert abs(float(flat[off]) - ref) <= 1e-3 * float(img[..., 0].float().abs().sum()) / 30 + 1e-3


@pytest.mark.parametrize("shape", [(1000, 128, 6272), (37, 200, 10), (512, 64, 72)])
def test_wgrad_bias_tail_all_rows_dense(hip, shape):
    """Dense layer: dW and the bias gradient (column sums of dY = row sums of the kernel's dY operand) from one
    launch; the weight part must equal the plain call bit for bit, the tail the fp32 column sums."""
    M, K, N = shape
    g = G.ConvGeom(1, 1, K, 1, 1, N, 1, 1).check()
    Kp, Np = hip.pad8(K), hip.pad8(N)
    torch.manual_seed(23)
    x = torch.zeros(M, 1, 1, Kp)
    x[..., :K] = torch.randn(M, 1, 1, K)
    dy = torch.zeros(M, 1, 1, Np)
    dy[..., :N] = torch.randn(M, 1, 1, N)
    x, dy = x.bfloat16().cuda().contiguous(), dy.bfloat16().cuda().contiguous()
    plain = hip.conv_wgrad(x, dy, g, "A").reshape(-1)
    numel = N * Kp
    assert plain.numel() == numel
    off = numel + 8
    total = (off + N + 3) // 4 * 4
    flat = torch.zeros(total, device="cuda")
    hip.conv_wgrad(x, dy, g, "A", out=flat, box=(N, Kp, Kp, Kp, 1, total), bias_tail=("rows", off))
    assert torch.equal(flat[:numel], plain)
    assert float(flat[numel:off].abs().max()) == 0.0
    ref = dy.float().sum(dim=(0, 1, 2))[:N]
    close(flat[off:off + N].cpu(), ref.cpu(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()) + 1e-3)


@pytest.mark.parametrize("impl", ["pconv", "igemm"])
@pytest.mark.parametrize("prev", ["relu", "leaky", "linear", "tanh"])
def test_dgrad_fused_act_grad_is_bit_identical_to_the_separate_pass(hip, impl, prev, monkeypatch):
    """dX * act'(y) of the layer below rides in the data-gradient kernel's epilogue for linear / relu / leaky (pconv
    EPI = 3, the gather kernel's pre-store pass); it must give the bits of the unfused kernel + the elementwise
    multiply (bf16 round, multiply, bf16 round), and other activations must still take the separate pass."""
    g = G.conv2d_geom(14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same")
    B = 37
    torch.manual_seed(5)
    w = r16(torch.randn(64, 5, 5, 32) / math.sqrt(5 * 5 * 32))
    wB = padw(hip, w.permute(3, 1, 2, 0).contiguous())
    dy = dev(hip, r16(torch.randn(B, g.OH, g.OW, 64)))
    y = torch.randn(B, 14, 14, 32).bfloat16().cuda()
    if prev == "tanh":
        y = torch.tanh(y.float()).bfloat16()
    monkeypatch.setattr(hip, "CONV_IMPL", impl)
    monkeypatch.setattr(hip, "AUTOTUNE", False)
    monkeypatch.setattr(hip, "_CONV_CHOICE", {})
    monkeypatch.setattr(hip, "_CHOICE_SPEC", {})
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(hip, "PCONV_MULY", fused)
        monkeypatch.setattr(hip, "IGEMM_MULY", fused)
        outs.append(hip.conv_dgrad(dy, wB, g, mul=(y, prev, 0.2)))
    assert torch.equal(outs[0], outs[1])
    plain = hip.conv_dgrad(dy, wB, g).float()
    ref = (plain * cpu_ref.act_grad_from_y(y.float().cpu(), prev, 0.2).cuda()).cpu()
    close(outs[0].float().cpu(), ref, atol=1e-2)


DMA_CASES = [
    # H, W, I, O, k, s, padding          B
    ((14, 14, 32, 64, (5, 5), (2, 2), "same"), 37),      # D3 shape: asymmetric `same` padding, rows beyond M
    ((8, 8, 128, 256, (5, 5), (2, 2), "same"), 9),       # wide N: two channel tiles per pixel tile
    ((9, 9, 64, 72, (3, 3), (1, 1), "same"), 5),         # N tail beyond the weight rows
    ((6, 7, 32, 16, (3, 2), (1, 1), "valid"), 11),       # narrow N (weight pieces past BN fetch zeros)
]


@pytest.mark.parametrize("tile", [33, 34, 35, 36, 37, 38, 39, 40, 41])
@pytest.mark.parametrize("case", DMA_CASES)
def test_igemm_lds_dma_tiles(hip, case, tile, monkeypatch):
    """The LDS-DMA staged implicit-GEMM variants (buffer_load ... lds: source-side swizzle, zero fill through
    out-of-range offsets, counted vmcnt across one raw barrier per K step) against the register-staged kernel of the
    same reduction order (bit for bit) and the fp32 oracle, for forward, data gradient (parity classes) and the fused
    act'(y) epilogue."""
    (H, W, I, O, k, s, pad), B = case
    g = G.conv2d_geom(H, W, I, O, k, s, (1, 1), pad)
    torch.manual_seed(7)
    x = r16(torch.randn(B, H, W, I))
    w = r16(torch.randn(O, k[0], k[1], I) / math.sqrt(k[0] * k[1] * I))
    b = torch.randn(O)
    dy = r16(torch.randn(B, g.OH, g.OW, O))
    wB = padw(hip, w.permute(3, 1, 2, 0).contiguous())
    monkeypatch.setattr(hip, "CONV_IMPL", "igemm")
    monkeypatch.setattr(hip, "AUTOTUNE", False)
    ypre = torch.randn(B, H, W, hip.pad8(I)).bfloat16().cuda()
    outs = {}
    for t in (22, tile):
        monkeypatch.setattr(hip, "IGEMM_TILE", t)
        y = hip.conv_fprop(dev(hip, x), padw(hip, w), b.cuda(), g, "leaky", 0.3)
        outs[t] = [y]
        if hip.pad8(O) % 32 == 0:           # the data gradient reduces over the output channels
            dx = hip.conv_dgrad(dev(hip, dy), wB, g)
            outs[t] += [dx, hip.conv_dgrad(dev(hip, dy), wB, g, mul=(ypre, "leaky", 0.2))]
    close(host(hip, outs[tile][0], O), cpu_ref.conv_fprop(x, w, b, g, "leaky", 0.3))
    if len(outs[tile]) > 1:
        close(host(hip, outs[tile][1], I), cpu_ref.conv_dgrad(dy, w, g))
    for a, r in zip(outs[tile], outs[22]):
        assert torch.equal(a, r)


CONVT_CASES = [
    # H, W, Cin, filters, k, s, d, padding, act   (layer view)
    (7, 7, 128, 64, (5, 5), (2, 2), (1, 1), "same", "relu"),         # G2
    (14, 14, 64, 32, (5, 5), (2, 2), (1, 1), "same", "relu"),        # G4
    (28, 28, 32, 1, (5, 5), (1, 1), (1, 1), "same", "tanh01"),       # G6
    (28, 28, 1, 64, (3, 3), (2, 2), (1, 1), "valid", "softplus"),    # model0 conv2d_transpose_1
    (8, 8, 8, 128, (1, 1), (2, 2), (2, 2), "valid", "relu"),         # model3 conv2d_transpose_4 (empty parity classes)
    (16, 16, 128, 32, (2, 2), (2, 2), (3, 3), "valid", "softplus"),  # model3 conv2d_transpose_5
    (3, 3, 128, 32, (3, 3), (2, 2), (2, 2), "valid", "softsign"),    # model1 conv2d_transpose_3
    (5, 6, 24, 12, (3, 2), (3, 1), (1, 1), "valid", "sigmoid"),
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transpose(hip, case):
    H, W, Cin, Fo, k, s, d, pad, act = case
    g = G.conv2d_transpose_geom(H, W, Cin, Fo, k, s, d, pad)
    B = 5
    torch.manual_seed(2)
    x = r16(torch.randn(B, H, W, Cin))
    w = r16(torch.randn(Fo, k[0], k[1], Cin) / math.sqrt(k[0] * k[1] * Cin))   # canonical [filters][KH][KW][in]
    b = torch.randn(Fo)
    wA = w.permute(3, 1, 2, 0).contiguous()                                       # core [O=in][T][I=filters]
    y_ref = cpu_ref.conv_dgrad(x, wA, g, b, act)
    y = hip.conv_dgrad(dev(hip, x), padw(hip, w), g, b.cuda(), act)
    assert tuple(y.shape[1:3]) == (g.H, g.W)
    close(host(hip, y, Fo), y_ref)
    if hip.pad8(Fo) != Fo:
        assert float(y[..., Fo:].float().abs().max()) == 0.0
    dout = r16(torch.randn(B, g.H, g.W, Fo))
    dx_ref = cpu_ref.conv_fprop(dout, wA, None, g)
    dx = hip.conv_fprop(dev(hip, dout), padw(hip, wA), None, g)
    close(host(hip, dx, Cin), dx_ref)
    dw_ref = cpu_ref.conv_wgrad(dout, x, g).permute(3, 1, 2, 0)                  # -> canonical
    dw = hip.conv_wgrad(dev(hip, dout), dev(hip, x), g, "B").cpu()
    close(dw[..., :Cin], dw_ref, rtol=1e-2)


@pytest.mark.parametrize("case", [(57, 57, 128, (3, 3), (2, 2)), (28, 28, 12, (3, 3), (1, 1)),
                                  (10, 10, 128, (2, 2), (3, 3)), (18, 18, 16, (3, 2), (1, 2)),
                                  (9, 11, 8, (1, 1), (2, 2)), (16, 16, 24, (3, 3), (3, 3))])
def test_depthwise(hip, case):
    H, W, Cc, k, s = case
    g = G.depthwise_geom(H, W, Cc, k, s, (2, 2), "valid", keras224_compat=True)
    B = 3 if H > 30 else 41
    torch.manual_seed(3)
    x = r16(torch.randn(B, H, W, Cc))
    w = r16(torch.randn(k[0], k[1], Cc))
    b = torch.randn(Cc)
    y_ref = cpu_ref.conv_fprop(x, w, b, g, "tanh")
    y = hip.conv_fprop(dev(hip, x), padw(hip, w), b.cuda(), g, "tanh")
    close(host(hip, y, Cc), y_ref)
    dy = r16(torch.randn(B, g.OH, g.OW, Cc))
    close(host(hip, hip.conv_dgrad(dev(hip, dy), padw(hip, w), g), Cc), cpu_ref.conv_dgrad(dy, w, g))
    dw = hip.conv_wgrad(dev(hip, x), dev(hip, dy), g).cpu()
    close(dw[..., :Cc], cpu_ref.conv_wgrad(x, dy, g), rtol=1e-2)


@pytest.mark.parametrize("case", [(57, 57, 128, (3, 3), (2, 2), "valid"), (28, 28, 24, (3, 3), (1, 1), "valid"),
                                  (10, 10, 128, (2, 2), (3, 3), "valid"), (13, 13, 40, (3, 3), (1, 1), "same"),
                                  (9, 11, 8, (1, 1), (2, 2), "valid"), (12, 12, 20, (3, 3), (2, 2), "same")])
def test_depthwise_input_affine(hip, case):
    """A per-channel affine (a BatchNorm) in front of a depthwise layer, applied by the stencil kernels to their tap
    sums: forward and weight gradient equal the layer run on the fp32 affine output, zero padding applied AFTER the
    affine (padded cases run the per-pixel live-tap variant, channel counts that do not divide 256 the generic grid)."""
    H, W, Cc, k, s, pad = case
    g = G.depthwise_geom(H, W, Cc, k, s, (1, 1), pad)
    B = 3 if H > 30 else 21
    torch.manual_seed(7)
    x = r16(torch.randn(B, H, W, Cc))
    w = r16(torch.randn(k[0], k[1], Cc))
    b = torch.randn(Cc)
    sc, sh = torch.rand(Cc) + 0.5, torch.randn(Cc)
    xa = x * sc + sh                                            # fp32, not rounded: what the kernels emulate
    Cp = hip.pad8(Cc)
    scp, shp = torch.zeros(Cp), torch.zeros(Cp)
    scp[:Cc], shp[:Cc] = sc, sh
    aff = (scp.cuda(), shp.cuda(), 1, Cc)
    y = hip.conv_fprop(dev(hip, x), padw(hip, w), b.cuda(), g, "tanh", in_affine=aff)
    close(host(hip, y, Cc), cpu_ref.conv_fprop(xa, w, b, g, "tanh"))
    dy = r16(torch.randn(B, g.OH, g.OW, Cc))
    dw = hip.conv_wgrad(dev(hip, x), dev(hip, dy), g, x_affine=(aff[0], aff[1], Cc)).cpu()
    close(dw[..., :Cc], cpu_ref.conv_wgrad(xa, dy, g), rtol=1e-2)
    if Cp != Cc:
        assert float(dw[..., Cc:].abs().max()) == 0.0


@pytest.mark.parametrize("case", [(14, 14, 64, 128, 5, 2), (9, 11, 24, 40, 3, 1)])
@pytest.mark.parametrize("layout", ["A", "B"])
def test_wgrad2_slab_orders_agree(hip, case, layout, monkeypatch):
    """wgrad2's split-K slabs in fragment order (default) and in the gradient's own layout are summed over the
    splits in the same order: the two paths must agree bit for bit (and with the fp32 oracle to tolerance)."""
    H, W, I, O, k, s = case
    g = G.conv2d_geom(H, W, I, O, (k, k), (s, s), (1, 1), "same")
    B = 96
    torch.manual_seed(11)
    x = r16(torch.randn(B, H, W, I))
    dy = r16(torch.randn(B, g.OH, g.OW, O))
    xd, dyd = dev(hip, x), dev(hip, dy)
    monkeypatch.setattr(hip, "WGRAD_IMPL", "wgrad2")
    monkeypatch.setattr(hip, "AUTOTUNE", False)          # heuristic plan 0 on both calls
    monkeypatch.setattr(hip, "W2_FRAG_SLAB", True)
    a = hip.conv_wgrad(xd, dyd, g, layout).cpu()
    monkeypatch.setattr(hip, "W2_FRAG_SLAB", False)
    b = hip.conv_wgrad(xd, dyd, g, layout).cpu()
    assert torch.equal(a, b)
    ref = cpu_ref.conv_wgrad(x, dy, g)
    if layout == "A":
        close(a[..., :I], ref, rtol=1e-2)


@pytest.mark.parametrize("case", [(14, 14, 64, 128, 5, 2, 37), (7, 7, 64, 128, 5, 2, 50), (9, 11, 24, 40, 3, 1, 20),
                                  (14, 14, 64, 128, 5, 2, 700)])
@pytest.mark.parametrize("frag", [True, False])
def test_wgrad2_every_candidate_plan(hip, case, frag, monkeypatch):
    """Every launch plan the autotuner may pick (tile, taps per wave, four- and eight-wave blocks, LDS budget, register
    or LDS-DMA staging -- the latter double-buffered: at B = 700 a block walks several units) gives the oracle's
    gradient, in both slab orders."""
    H, W, I, O, k, s, B = case
    g = G.conv2d_geom(H, W, I, O, (k, k), (s, s), (1, 1), "same")
    torch.manual_seed(12)
    x = r16(torch.randn(B, H, W, I))
    dy = r16(torch.randn(B, g.OH, g.OW, O))
    xd, dyd = dev(hip, x), dev(hip, dy)
    ref = cpu_ref.conv_wgrad(x, dy, g)
    monkeypatch.setattr(hip, "WGRAD_IMPL", "wgrad2")
    monkeypatch.setattr(hip, "W2_FRAG_SLAB", frag)
    Ip, Op, T = hip.pad8(I), hip.pad8(O), k * k
    plans = hip._wgrad2_plans(g, B, Ip, Op, O, Ip)
    assert any(pl["NW"] == 8 for pl in plans) and any(pl["NW"] == 4 for pl in plans)
    assert any(pl.get("DMA") for pl in plans) and not plans[0].get("DMA")
    key = ("wgrad", g, B, Ip, Op, O, Ip, T * Ip, Ip, 1)
    for i, pl in enumerate(plans):
        monkeypatch.setitem(hip._CONV_CHOICE, key, i)
        got = hip.conv_wgrad(xd, dyd, g, "A").cpu()
        close(got[..., :I], ref, rtol=1e-2)


def test_dense_large_k(hip):
    g = G.dense_geom(86528, 10)           # model0 dense_1
    B = 32
    torch.manual_seed(4)
    x = r16(torch.randn(B, 1, 1, 86528))
    w = r16(torch.randn(10, 1, 1, 86528) / 300)
    y = hip.conv_fprop(dev(hip, x), padw(hip, w), None, g)
    close(host(hip, y, 10), cpu_ref.conv_fprop(x, w, None, g))
    dy = r16(torch.randn(B, 1, 1, 10))
    dw = hip.conv_wgrad(dev(hip, x), dev(hip, dy), g, "A").cpu()
    close(dw, cpu_ref.conv_wgrad(x, dy, g), rtol=1e-2)


@pytest.mark.parametrize("shape", [(2000, 7, 7, 64), (64, 13), (33, 5, 5, 19), (300, 28, 28, 32)])
@pytest.mark.parametrize("post,prev", [(None, "relu"), ("leaky", None), (None, "softsign")])
@pytest.mark.parametrize("train", [True, False])
def test_batchnorm(hip, shape, post, prev, train):
    Cc = shape[-1]
    Cp = hip.pad8(Cc)
    torch.manual_seed(5)
    x = r16(torch.rand(*shape) * 3 + 0.1) if prev else r16(torch.randn(*shape) * 2 + 1)
    if prev:
        x = r16(cpu_ref.act_fwd(x, prev))
    gamma, beta = torch.rand(Cc) + 0.5, torch.randn(Cc)
    mm, mv = torch.randn(Cc) * 0.1, torch.rand(Cc) + 0.5

    def p(t, fill=0.0):
        o = torch.full((Cp,), fill)
        o[:Cc] = t
        return o.cuda()

    gd, bd, mmd, mvd = p(gamma, 1.0), p(beta), p(mm), p(mv, 1.0)
    xd = dev(hip, x)
    bessel = len(shape) == 4
    if train:
        mean, var = cpu_ref.bn_stats(x)
        invstd = torch.rsqrt(var + 1e-3)
        scale, shift, mean_d, invstd_d = hip.bn_train_affine(xd, gd, bd, mmd, mvd, 0.99, 1e-3, bessel, True)
        close(mean_d[:Cc].cpu(), mean, rtol=1e-4, atol=1e-4)
        close(invstd_d[:Cc].cpu(), invstd, rtol=1e-3, atol=1e-4)
        n = x.numel() // Cc
        v = var * n / (n - 1) if bessel else var
        close(mmd[:Cc].cpu(), mm * 0.99 + mean * 0.01, rtol=1e-4, atol=1e-5)
        close(mvd[:Cc].cpu(), mv * 0.99 + v * 0.01, rtol=1e-4, atol=1e-5)
    else:
        mean, invstd = mm, torch.rsqrt(mv + 1e-3)
        scale, shift, invstd_d = hip.bn_infer_affine(gd, bd, mmd, mvd, 1e-3)
        mean_d = mmd
    sc = gamma * invstd
    y_ref = cpu_ref.bn_apply(x, sc, beta - mean * sc, post)
    y = hip.bn_apply(xd, scale, shift, post, 0.3, Cc)
    close(host(hip, y, Cc), y_ref)
    if not train:       # the single-launch inference form computes the same affine inside the kernel
        y1 = hip.bn_apply_infer(xd, gd, bd, mmd, mvd, 1e-3, post, 0.3, Cc)
        close(host(hip, y1, Cc), y_ref)
        assert float(y1[..., Cc:].float().abs().max()) == 0.0 if Cp > Cc else True
    dy = r16(torch.randn(*shape))
    dx_ref, dg_ref, db_ref = cpu_ref.bn_bwd(dy, x, r16(y_ref), mean, invstd, gamma, post, 0.3, train, prev)
    dx, dg, db = hip.bn_bwd(dev(hip, dy), xd, y, mean_d, invstd_d, gd, post, 0.3, train, prev)
    close(host(hip, dx, Cc), dx_ref, rtol=3e-2)
    close(dg[:Cc].cpu(), dg_ref, rtol=2e-2)
    close(db[:Cc].cpu(), db_ref, rtol=2e-2)
    if prev:
        # bias gradient of the conv in front (column sums of dx) out of the same reduction pass
        dbias = torch.full((Cc,), 7.0, device="cuda")
        dx2, dg2, db2 = hip.bn_bwd(dev(hip, dy), xd, y, mean_d, invstd_d, gd, post, 0.3, train, prev,
                                   out_dbias=dbias)
        assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)
        ref = cpu_ref.bias_grad(dx_ref)
        close(dbias.cpu(), ref, rtol=2e-2, atol=2e-3 * float(dx_ref.abs().sum(tuple(range(dx_ref.dim() - 1))).max()))


@pytest.mark.parametrize("shape", [(2000, 7, 7, 64), (2000, 4, 4, 128), (66, 5, 5, 19), (48, 24)])
@pytest.mark.parametrize("post,prev", [("leaky", None), (None, "relu")])
def test_batchnorm_batch_groups(hip, shape, post, prev):
    """Per-half batch statistics (the DCGAN discriminator on [real ; fake], models/dcgan.py "split"): one grouped
    launch == the layer applied to each half on its own (bitwise), and both match the fp32 oracle."""
    Cc, Cp = shape[-1], hip.pad8(shape[-1])
    torch.manual_seed(11)
    x = r16(torch.randn(*shape) * 2 + 1)
    x[shape[0] // 2:] = r16(x[shape[0] // 2:] * 0.3 - 2.0)          # the halves differ in distribution
    if prev:
        x = r16(cpu_ref.act_fwd(x, prev))
    gamma, beta = torch.rand(Cc) + 0.5, torch.randn(Cc)

    def p(t, fill=0.0):
        o = torch.full((Cp,), fill)
        o[:Cc] = t
        return o.cuda()

    gd, bd = p(gamma, 1.0), p(beta)
    xd = dev(hip, x)
    bessel = len(shape) == 4
    mm2, mv2 = p(torch.zeros(Cc)), p(torch.ones(Cc), 1.0)
    sc2, sh2, mean2, is2 = hip.bn_train_affine(xd, gd, bd, mm2, mv2, 0.99, 1e-3, bessel, True, groups=2)
    assert sc2.shape == (2, Cp)
    y2 = hip.bn_apply(xd, sc2, sh2, post, 0.3, Cc, groups=2)
    dy = r16(torch.randn(*shape))
    dyd = dev(hip, dy)
    dbias = torch.zeros(Cc, device="cuda") if prev else None
    dx2, dg2, db2 = hip.bn_bwd(dyd, xd, y2, mean2, is2, gd, post, 0.3, True, prev, out_dbias=dbias, groups=2)
    # the same thing one half at a time
    mm1, mv1 = p(torch.zeros(Cc)), p(torch.ones(Cc), 1.0)
    h = shape[0] // 2
    ys, dxs, dgs, dbs, dbi = [], [], [], [], []
    y_ref, dx_ref, dg_ref, db_ref = [], [], 0, 0
    for k in range(2):
        xs, ds = xd[k * h:(k + 1) * h], dyd[k * h:(k + 1) * h]
        sc, sh, mean, inv = hip.bn_train_affine(xs, gd, bd, mm1, mv1, 0.99, 1e-3, bessel, True)
        assert torch.equal(sc, sc2[k]) and torch.equal(sh, sh2[k]) and torch.equal(mean, mean2[k])
        y = hip.bn_apply(xs, sc, sh, post, 0.3, Cc)
        ys.append(y)
        b1 = torch.zeros(Cc, device="cuda") if prev else None
        dx, dg, db = hip.bn_bwd(ds, xs, y, mean, inv, gd, post, 0.3, True, prev, out_dbias=b1)
        dxs.append(dx), dgs.append(dg.double()), dbs.append(db.double()), dbi.append(b1)
        xc, dc = x[k * h:(k + 1) * h], dy[k * h:(k + 1) * h]
        m_, v_ = cpu_ref.bn_stats(xc)
        i_ = torch.rsqrt(v_ + 1e-3)
        yr = cpu_ref.bn_apply(xc, gamma * i_, beta - m_ * gamma * i_, post)
        y_ref.append(yr)
        a, b, c = cpu_ref.bn_bwd(dc, xc, r16(yr), m_, i_, gamma, post, 0.3, True, prev)
        dx_ref.append(a)
        dg_ref, db_ref = dg_ref + b, db_ref + c
    assert torch.equal(y2, torch.cat(ys, 0)) and torch.equal(dx2, torch.cat(dxs, 0))
    assert torch.equal(mm2, mm1) and torch.equal(mv2, mv1)          # two momentum updates, in group order
    close(dg2.cpu().double(), (dgs[0] + dgs[1]).cpu(), rtol=1e-5, atol=1e-4 * float(dg2.abs().max()) + 1e-6)
    close(db2.cpu().double(), (dbs[0] + dbs[1]).cpu(), rtol=1e-5, atol=1e-4 * float(db2.abs().max()) + 1e-6)
    close(host(hip, y2, Cc), torch.cat(y_ref, 0))
    close(host(hip, dx2, Cc), torch.cat(dx_ref, 0), rtol=3e-2)
    close(dg2[:Cc].cpu(), dg_ref, rtol=2e-2)
    close(db2[:Cc].cpu(), db_ref, rtol=2e-2)
    if prev:
        close(dbias.cpu().double(), (dbi[0] + dbi[1]).cpu().double(), rtol=1e-4,
              atol=1e-4 * float(dbias.abs().max()) + 1e-5)


@pytest.mark.parametrize("kind", ["relu", "sigmoid", "softplus", "tanh", "softsign", "leaky", "tanh01"])
def test_activations(hip, kind):
    torch.manual_seed(6)
    x = r16(torch.randn(37, 9, 11) * 3)
    y_ref = cpu_ref.act_fwd(x, kind)
    y = hip.act_fwd(dev(hip, x), kind, 0.3, 11)
    close(host(hip, y, 11), y_ref, atol=1e-2)
    assert float(y[..., 11:].float().abs().max()) == 0.0
    dy = r16(torch.randn_like(x))
    dx = hip.act_bwd(dev(hip, dy), y, kind)
    close(host(hip, dx, 11), cpu_ref.act_bwd(dy, host(hip, y, 11), kind), atol=2e-2)


@pytest.mark.parametrize("rows,C", [(1000, 6272), (37, 1024), (8192, 1032)])
def test_bias_grad_few_rows_many_channels(hip, rows, C):
    """The single-pass column sum (few rows, >= 1024 channels: a wide Dense layer's bias gradient)."""
    torch.manual_seed(8)
    dy = r16(torch.randn(rows, C))
    out = torch.zeros(C - 3, device="cuda")
    hip.bias_grad(dy.bfloat16().cuda(), out)
    close(out.cpu(), dy.sum(0)[:C - 3], rtol=1e-3, atol=1e-2)


def test_bias_grad_and_pad(hip):
    torch.manual_seed(7)
    dy = r16(torch.randn(500, 7, 7, 19))
    out = torch.zeros(19, device="cuda")
    hip.bias_grad(dev(hip, dy), out)
    close(out.cpu(), cpu_ref.bias_grad(dy), rtol=1e-3)
    x = r16(torch.randn(4, 5, 6, 8))
    y = hip.spatial_pad(dev(hip, x), 1, 2, 3, 0)
    ref = torch.nn.functional.pad(x, (0, 0, 3, 0, 1, 2))
    assert torch.equal(host(hip, y, 8), ref)
    z = hip.spatial_pad(y, -1, -2, -3, 0)
    assert torch.equal(host(hip, z, 8), x)
    # uint8 gather + 1->3 channel replicate
    img = torch.randint(0, 256, (10, 28, 28, 1), dtype=torch.uint8)
    idx = torch.tensor([3, 3, 9, 0])
    o = hip.pad_channels(img.cuda(), 8, idx.cuda(), 1.0, 3)
    ref = img[idx].float().expand(-1, -1, -1, 3)
    assert torch.equal(host(hip, o, 3), ref)
    assert float(o[..., 3:].float().abs().max()) == 0.0


def test_softmax_losses(hip):
    torch.manual_seed(8)
    B, N = 300, 10
    logits = r16(torch.randn(B, N) * 4)
    onehot = torch.nn.functional.one_hot(torch.randint(0, N, (B,)), N).float()
    ld = dev(hip, logits)
    loss, dl, p = hip.softmax_ce(ld, onehot.cuda())
    lr, dlr, pr = cpu_ref.softmax_ce(logits, onehot)
    assert abs(float(loss) - float(lr)) < 1e-4 * max(1, abs(float(lr)))
    close(p.cpu(), pr, rtol=1e-4, atol=1e-6)
    close(host(hip, dl, N), dlr, rtol=1e-2, atol=1e-5)
    loss, dl, p = hip.softmax_mse(ld, onehot.cuda())
    lr, dlr, pr = cpu_ref.softmax_mse(logits, onehot)
    assert abs(float(loss) - float(lr)) < 1e-5
    close(host(hip, dl, N), dlr, rtol=1e-2, atol=1e-6)
    close(hip.softmax(ld, N).cpu(), pr, rtol=1e-4, atol=1e-6)
    cnt = hip.argmax_match_count(p, onehot.cuda())
    assert int(cnt) == int(cpu_ref.argmax_match_count(pr, onehot))
    x = torch.randn(77, 1) * 3
    z = (torch.rand(77, 1) > 0.5).float()
    l1, d1 = hip.sigmoid_bce(x.cuda(), z.cuda())
    l2, d2 = cpu_ref.sigmoid_bce(x, z)
    assert abs(float(l1) - float(l2)) < 1e-5
    close(d1.cpu(), d2, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("M,nf", [(2000, 1000), (1000, 1000), (7, 3)])
def test_dense1_bce(hip, M, nf):
    torch.manual_seed(9)
    K = 2048
    x = r16(torch.randn(M, 4, 4, 128))
    w = r16(torch.randn(1, K) * 0.05)
    b = torch.tensor([0.3])
    lr, dxr, dwr, dbr, lgr = cpu_ref.dense1_bce_fwd_bwd(x, w, b, 1.0, 0.0, nf)
    loss, dx, dw, db, lg = hip.dense1_bce_fwd_bwd(dev(hip, x), w.bfloat16().cuda(), b.cuda(), 1.0, 0.0, nf)
    close(lg.cpu(), lgr, rtol=1e-3, atol=1e-3)
    assert abs(float(loss) - float(lr)) < 1e-4
    close(host(hip, dx, 128), dxr, rtol=1e-2)
    close(dw.cpu().reshape(-1), dwr.reshape(-1), rtol=1e-3)
    close(db.cpu(), dbr, rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("kind", ["adam", "sgd", "rmsprop", "adam_fresh_double"])
def test_optimizers_match_oracle(hip, kind):
    from distributeddeeplearning_amd.models import ir
    from distributeddeeplearning_amd.models.network import Network
    from distributeddeeplearning_amd.models.optim import FlatOptimizer, OptimConfig
    n = ir.NameScope()
    spec = ir.ModelSpec("m", [ir.Input(n("input"), (16,))]).append(ir.Dense(n("dense"), 24, "relu")).append(
        ir.Dense(n("dense"), 8))
    cpu, gpu = Network(spec, "cpu", 3), Network(spec, "cuda", 3)
    assert torch.allclose(gpu.P.cpu(), cpu.P)
    cfg = OptimConfig(kind=kind, lr=1e-2, eps=1e-7)
    oc, og = FlatOptimizer(cpu, cfg), FlatOptimizer(gpu, cfg)
    for it in range(4):
        g = torch.randn_like(cpu.P)
        cpu.G.copy_(g)
        gpu.G.copy_(g)
        oc.step()
        og.step()
        assert torch.allclose(gpu.P.cpu(), cpu.P, rtol=1e-5, atol=1e-6), (kind, it)
        assert torch.allclose(gpu.Pc.float().cpu(), cpu.P, rtol=1e-2, atol=1e-3)
    # alternate layout copy follows the master
    op = gpu.ops[0]
    assert torch.allclose(op.w_alt.float().cpu(), op.kernel.t().cpu(), rtol=1e-2, atol=1e-3)


def test_philox(hip):
    st = torch.zeros(3, dtype=torch.int64, device="cuda")
    st[0] = 1234
    a = hip.philox_uniform((1000, 128), torch.bfloat16, st).float()
    b = hip.philox_uniform((1000, 128), torch.bfloat16, st).float()
    assert int(st[1]) == 2
    assert a.min() >= -1 and a.max() <= 1 and not torch.equal(a, b)
    assert abs(float(a.mean())) < 0.01 and abs(float(a.var()) - 1 / 3) < 0.01


ONECH_CASES = [
    # H, W, k, s, d, padding   (forward geometry of a Conv2D 1->N; its adjoint is the Conv2DTranspose N->1 fold)
    (28, 28, (5, 5), (2, 2), (1, 1), "same"),        # D1 / its dgrad
    (28, 28, (5, 5), (1, 1), (1, 1), "same"),        # G6
    (28, 28, (3, 2), (1, 2), (2, 1), "valid"),
    (64, 64, (5, 5), (2, 2), (1, 1), "same"),        # dcgan64
    (9, 11, (1, 1), (3, 3), (1, 1), "valid"),
]


@pytest.mark.parametrize("case", ONECH_CASES)
@pytest.mark.parametrize("B", [1, 37])
def test_onech_im2col_col2im(hip, case, B):
    """Single-channel path: im2col is pure data movement (bit-exact vs an index gather), col2im is its
    adjoint with fused bias + activation."""
    H, W, k, s, d, pad = case
    g = G.conv2d_geom(H, W, 1, 8, k, s, d, pad)
    T, Tp = g.KH * g.KW, hip.pad8(g.KH * g.KW)
    torch.manual_seed(11)
    img = r16(torch.randn(B, H, W, 1))
    ref = torch.zeros(B, g.OH, g.OW, Tp)
    for kh in range(g.KH):
        for kw in range(g.KW):
            for oh in range(g.OH):
                ih = oh * g.sh + kh * g.dh - g.pt
                if not 0 <= ih < H:
                    continue
                for ow in range(g.OW):
                    iw = ow * g.sw + kw * g.dw - g.pl
                    if 0 <= iw < W:
                        ref[:, oh, ow, kh * g.KW + kw] = img[:, ih, iw, 0]
    col = hip.im2col_1ch(dev(hip, img), g)
    assert col.shape == (B, g.OH, g.OW, Tp)
    assert torch.equal(col.float().cpu(), ref)
    # fold: <col2im(c), img> == <c, im2col(img)> for every c  =>  check against the explicit scatter-add
    c = r16(torch.randn(B, g.OH, g.OW, Tp) * 0.5)
    c[..., T:] = 0
    fold = torch.zeros(B, H, W)
    for kh in range(g.KH):
        for kw in range(g.KW):
            for oh in range(g.OH):
                ih = oh * g.sh + kh * g.dh - g.pt
                if not 0 <= ih < H:
                    continue
                for ow in range(g.OW):
                    iw = ow * g.sw + kw * g.dw - g.pl
                    if 0 <= iw < W:
                        fold[:, ih, iw] += c[:, oh, ow, kh * g.KW + kw]
    bias = torch.tensor([0.25], device="cuda")
    out = hip.col2im_1ch(c.bfloat16().cuda().contiguous(), g, bias, "tanh")
    close(out[..., 0].float().cpu(), torch.tanh(fold + 0.25), atol=2e-2)
    assert float(out[..., 1:].float().abs().max()) == 0.0
    plain = hip.col2im_1ch(c.bfloat16().cuda().contiguous(), g)
    close(plain[..., 0].float().cpu(), fold, rtol=2e-2)


@pytest.mark.parametrize("case", ONECH_CASES)
@pytest.mark.parametrize("C", [32, 24, 64])
def test_tapfold_fused_equals_gemm_plus_col2im(hip, case, C):
    """Fused taps-GEMM + fold must reproduce the two-kernel path bit for bit (same MFMA order, same bf16
    rounding of the taps tensor, same fold order), with bias/activation and the fused act' multiply."""
    H, W, k, s, d, pad = case
    g = G.conv2d_geom(H, W, 1, C, k, s, d, pad)
    T, Tp = g.KH * g.KW, hip.pad8(g.KH * g.KW)
    if T > 32:
        pytest.skip("outside the fused kernel's range")
    B = 19
    torch.manual_seed(12)
    x = (torch.randn(B, g.OH, g.OW, C) * 0.5).bfloat16().cuda().contiguous()
    w = (torch.randn(T, C) * 0.2).bfloat16().cuda().contiguous()
    bias = torch.tensor([0.1], device="cuda")
    ymul = torch.rand(B, H, W, 8, device="cuda").bfloat16()
    g1 = G.ConvGeom(g.OH, g.OW, C, g.OH, g.OW, T, 1, 1).check()
    z = hip.conv_fprop(x, w.reshape(T, 1, 1, C), None, g1)
    for kw in (dict(), dict(bias=bias, act="tanh01"), dict(mul=(ymul, "tanh01", 0.3))):
        ref = hip.col2im_1ch(z, g, **kw)
        got = hip.tapfold_1ch(x, w, g, **kw)
        assert got is not None
        assert torch.equal(got, ref), kw.keys()


@pytest.mark.parametrize("case", ONECH_CASES)
@pytest.mark.parametrize("N", [32, 20, 64])
def test_tapconv_fused_equals_im2col_plus_gemm(hip, case, N):
    """Direct convolution from the single-channel image must reproduce im2col_1ch + the 1x1 GEMM bit for bit (same
    MFMA k order over the taps, same bias / activation epilogue), padded output channels staying zero."""
    H, W, k, s, d, pad = case
    g = G.conv2d_geom(H, W, 1, N, k, s, d, pad)
    T, Tp = g.KH * g.KW, hip.pad8(g.KH * g.KW)
    if T > 32:
        pytest.skip("outside the fused kernel's range")
    B = 21
    torch.manual_seed(16)
    img = torch.zeros(B, H, W, 8)
    img[..., 0] = torch.randn(B, H, W)
    img = img.bfloat16().cuda().contiguous()
    w = torch.zeros(N, 1, 1, Tp)
    w[..., :T] = torch.randn(N, 1, 1, T) * 0.3
    w = w.bfloat16().cuda().contiguous()
    bias = torch.randn(N, device="cuda")
    g1 = G.ConvGeom(g.OH, g.OW, T, g.OH, g.OW, N, 1, 1).check()
    xc = hip.im2col_1ch(img, g)
    for kw in (dict(bias=None, act=None), dict(bias=bias, act="leaky"), dict(bias=bias, act="tanh")):
        ref = hip.conv_fprop(xc, w, kw["bias"], g1, kw["act"], 0.3)
        got = hip.tapconv_1ch(img, w, g, kw["bias"], kw["act"], 0.3)
        assert got is not None
        assert torch.equal(got, ref), kw
        if N % 8:
            assert float(got[..., N:].float().abs().max()) == 0.0


@pytest.mark.parametrize("C,ctrue,groups", [(32, 32, 1), (32, 32, 2), (24, 20, 1), (64, 64, 3)])
def test_tapfold_input_affine_equals_materialised_bn(hip, C, ctrue, groups):
    """BatchNorm applied on load: tapfold(x, in_affine) == tapfold(bn_apply(x)) bit for bit (same bf16 rounding of
    the normalised activation), per batch group, padded channels staying zero."""
    g = G.conv2d_geom(28, 28, 1, C, (5, 5), (1, 1), (1, 1), "same")
    T = g.KH * g.KW
    B = 6 * groups
    torch.manual_seed(14)
    x = (torch.randn(B, g.OH, g.OW, C) * 0.7).bfloat16().cuda().contiguous()
    w = (torch.randn(T, C) * 0.2).bfloat16().cuda().contiguous()
    scale = (torch.rand(groups, C, device="cuda") + 0.5).contiguous()
    shift = torch.randn(groups, C, device="cuda").contiguous()
    a = hip.bn_apply(x, scale, shift, None, 0.3, ctrue, groups)
    ref = hip.tapfold_1ch(a, w, g, None, "tanh01")
    got = hip.tapfold_1ch(x, w, g, None, "tanh01", in_affine=(scale, shift, groups, ctrue))
    assert got is not None and torch.equal(got, ref)
    assert not torch.equal(got, hip.tapfold_1ch(x, w, g, None, "tanh01"))


@pytest.mark.parametrize("C,ctrue,T", [(32, 32, 25), (24, 20, 9), (64, 64, 25)])
def test_wgrad_input_affine_equals_materialised_bn(hip, C, ctrue, T):
    """conv_wgrad(x, x_affine) == conv_wgrad(bn_apply(x)) bit for bit on the streaming kernel (the 1x1 problem
    the single-channel transposed conv's weight gradient is)."""
    B, H = 7, 12
    Tp = hip.pad8(T)
    g1 = G.ConvGeom(H, H, C, H, H, T, 1, 1).check()
    torch.manual_seed(15)
    x = (torch.randn(B, H, H, C) * 0.7).bfloat16().cuda().contiguous()
    dy = (torch.randn(B, H, H, Tp) * 0.3).bfloat16().cuda().contiguous()
    dy[..., T:] = 0
    scale = (torch.rand(C, device="cuda") + 0.5).contiguous()
    shift = torch.randn(C, device="cuda").contiguous()
    a = hip.bn_apply(x, scale, shift, None, 0.3, ctrue, 1)
    ref = hip.conv_wgrad(a, dy, g1, "A")
    got = hip.conv_wgrad(x, dy, g1, "A", x_affine=(scale, shift, ctrue))
    assert torch.equal(got, ref)
    # a padded 5x5 problem (one the persistent kernel would take): with an affine the streaming kernel runs instead,
    # the zero padding applying to the BatchNorm's OUTPUT
    g5 = G.conv2d_geom(H, H, C, 16, (5, 5), (1, 1), (1, 1), "same")
    dy5 = (torch.randn(B, H, H, 16) * 0.3).bfloat16().cuda().contiguous()
    got5 = hip.conv_wgrad(x, dy5, g5, "A", x_affine=(scale, shift, ctrue))
    ref5 = hip.conv_wgrad(a, dy5, g5, "A")
    close(got5.cpu(), ref5.cpu(), rtol=1e-2)


@pytest.mark.parametrize("K,N,B", [(128, 6272, 1000), (96, 300, 37), (40, 136, 70)])
def test_dense_wide_output_skinny_k(hip, K, N, B):
    """Dense with K <= 128 and a wide output (G1: 128 -> 6272) runs on gemm_sk's column tiles."""
    g = G.dense_geom(K, N)
    torch.manual_seed(13)
    x = r16(torch.randn(B, 1, 1, K))
    w = r16(torch.randn(N, 1, 1, K) / 8)
    b = torch.randn(N)
    y = hip.conv_fprop(dev(hip, x), padw(hip, w), b.cuda(), g, "relu")
    close(host(hip, y, N), cpu_ref.conv_fprop(x, w, b, g, "relu"))
    assert y.shape[-1] == hip.pad8(N) and (N % 8 == 0 or float(y[..., N:].float().abs().max()) == 0.0)


@pytest.mark.parametrize("shape", [(14, 14, 32, 64, "same", 2, "linear"), (7, 7, 40, 24, "same", 1, "relu")])
@pytest.mark.parametrize("B", [5, 130, 1300])
def test_pconv_emits_batchnorm_statistics(hip, shape, B):
    """pconv's STATS epilogue: per-block partial sums of the stored (bf16) output == a column reduction of it."""
    H, W, I, O, pad, s, act = shape
    g = G.conv2d_geom(H, W, I, O, (5, 5), (s, s), (1, 1), pad)
    torch.manual_seed(21)
    x = dev(hip, r16(torch.randn(B, H, W, I)))
    w = padw(hip, r16(torch.randn(O, 5, 5, I) / 20))
    bias = torch.randn(O).cuda()
    Ip, Op = hip.pad8(I), hip.pad8(O)
    # B = 1300 with one image per unit: > 1024 partials per channel -> the wide combine kernel
    pl = hip._pconv_plan(g, "fprop", B, Ip, Op, (g.sh, g.sw), x.device, min_rows=16, img=1 if B > 1000 else None)
    assert pl is not None
    nb = hip._pconv_units(pl, B)
    assert B < 1000 or nb > 1024
    y = torch.empty(B, g.OH, g.OW, Op, dtype=torch.bfloat16, device="cuda")
    part = torch.full((2 * Op * nb,), float("nan"), device="cuda")
    hip._run_pconv(pl, x, w, bias, y, g, B, g.H, g.W, Ip, g.OH, g.OW, Op, O, (g.sh, g.sw), (1, 1), act, 0.3,
                   stats=part)
    y2 = hip.conv_fprop(x, w, bias, g, act)
    assert torch.equal(y, y2) or torch.allclose(y.float(), y2.float(), rtol=2e-2, atol=2e-2)
    sums = part.view(2, Op, nb).double().sum(-1).cpu()
    yf = y.float().reshape(-1, Op).double().cpu()
    assert torch.allclose(sums[0], yf.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(sums[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-2)
    gamma, beta = torch.ones(Op).cuda(), torch.zeros(Op).cuda()
    mm1, mv1, mm2, mv2 = (torch.zeros(Op).cuda(), torch.ones(Op).cuda(), torch.zeros(Op).cuda(), torch.ones(Op).cuda())
    a = hip.bn_train_affine_from_part(part, nb, B * g.OH * g.OW, gamma, beta, mm1, mv1, 0.99, 1e-3, True)
    b = hip.bn_train_affine(y, gamma, beta, mm2, mv2, 0.99, 1e-3, True)
    for u, v in zip(a, b):
        assert torch.allclose(u, v, rtol=1e-4, atol=1e-5)
    assert torch.allclose(mm1, mm2, rtol=1e-4, atol=1e-6) and torch.allclose(mv1, mv2, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tile", [6, 2, 22, 20, 33, 34, 35, 36, 37, 38])
@pytest.mark.parametrize("mode", ["fprop", "dgrad"])
def test_igemm_emits_batchnorm_statistics(hip, tile, mode):
    """The statistics epilogue of the implicit-GEMM kernels (register-staged deep / shallow, LDS-DMA): per-block
    partial sums of the stored (bf16) output, one column per (pixel tile, class), == a column reduction of it; the
    output itself is untouched by the epilogue.  dgrad: four parity classes, rows past a class's end."""
    g = G.conv2d_geom(14, 14, 64, 64, (5, 5), (2, 2), (1, 1), "same")
    B = 37
    torch.manual_seed(22)
    w = r16(torch.randn(64, 5, 5, 64) / 30)
    bias = torch.randn(64).cuda()
    if mode == "fprop":
        src, wgt = dev(hip, r16(torch.randn(B, 14, 14, 64))), padw(hip, w)
        geo = (B, 14, 14, 64, 7, 7, 64, 64, (2, 2), (1, 1))
        shape = (B, 7, 7, 64)
    else:
        src, wgt = dev(hip, r16(torch.randn(B, 7, 7, 64))), padw(hip, w.permute(3, 1, 2, 0).contiguous())
        geo = (B, 7, 7, 64, 14, 14, 64, 64, (1, 1), (2, 2))
        shape = (B, 14, 14, 64)
    cols = hip._igemm_stats_cols(tile, g, mode, B, src.device)
    assert cols is not None
    nb = cols[0]
    y = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
    y0 = torch.empty_like(y)
    part = torch.full((2 * 64 * (nb + 3),), float("nan"), device="cuda")
    hip._run_igemm(src, wgt, bias, y, g, mode, *geo, "leaky", 0.3, tile=tile, stats=(part, nb + 3, 2))
    hip._run_igemm(src, wgt, bias, y0, g, mode, *geo, "leaky", 0.3, tile=tile)
    assert torch.equal(y, y0)
    pv = part.view(2, 64, nb + 3)
    assert bool(torch.isnan(pv[:, :, :2]).all()) and bool(torch.isnan(pv[:, :, -1]).all())   # only its own columns
    sums = pv[:, :, 2:2 + nb].double().sum(-1).cpu()
    yf = y.float().reshape(-1, 64).double().cpu()
    assert torch.allclose(sums[0], yf.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(sums[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("choice", ["pconv", 22, 34])
def test_batch_slices_keep_statistics_and_on_load_batchnorm(hip, choice, monkeypatch):
    """A layer split into batch slices (operands beyond one launch's offsets) keeps its fusions: every slice's
    statistics epilogue writes into one partial array (two batch groups, slices never cross the group boundary) and
    a BatchNorm in front of a transposed conv stays applied on load when every slice runs the patch-resident kernel
    -- same output bits as the unsliced layer, same batch statistics."""
    torch.manual_seed(4)
    g = G.conv2d_geom(14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same")      # transposed: 7x7x64 -> 14x14x32
    B, groups = 48, 2
    dy = dev(hip, r16(torch.randn(B, 7, 7, 64)))
    wB = padw(hip, r16(torch.randn(64, 5, 5, 32) * 0.05).permute(3, 1, 2, 0).contiguous())
    scale = (torch.rand(groups, 64) + 0.5).cuda().reshape(-1)
    shift = torch.randn(groups, 64).cuda().reshape(-1)
    monkeypatch.setattr(hip, "AUTOTUNE", False)
    if choice != "pconv":
        monkeypatch.setattr(hip, "_conv_choice", lambda *a, **k: ("igemm", choice))
    else:
        def pick(src, wgt, bias, dst, g_, mode, B_, Hs, Ws, Cs, Hd, Wd, Nd, Nw, ss, *rest, **kw):
            pl = hip._pconv_plan(g_, mode, B_, Cs, Nd, ss, src.device, min_rows=16)
            assert pl is not None
            return ("pconv", pl)
        monkeypatch.setattr(hip, "_conv_choice", pick)

    def run():
        so, sink = {"groups": groups}, {}
        out = hip.conv_dgrad(dy, wB, g, None, "relu", 0.3, stats_out=so, in_affine=(scale, shift, groups, 64),
                             aff_sink=sink)
        assert "part" in so
        gamma, beta = torch.ones(32).cuda(), torch.zeros(32).cuda()
        mm, mv = torch.zeros(32).cuda(), torch.ones(32).cuda()
        if B % (so["img"] * groups) == 0:       # what a consumer with per-group statistics checks (network.py)
            st = hip.bn_train_affine_from_part(so["part"], so["nb"], B * 14 * 14, gamma, beta, mm, mv, 0.99, 1e-3, True,
                                               True, groups)
        else:                                   # one launch, pixel tiles straddling images: the plain pass
            assert not sliced
            st = hip.bn_train_affine(out, gamma, beta, mm, mv, 0.99, 1e-3, True, True, groups)
        return out, st, sink

    sliced = False
    ref, ref_st, ref_sink = run()
    assert (choice == "pconv") == (not ref_sink)            # on load only in the patch-resident kernel
    monkeypatch.setattr(hip, "ELEM_LIMIT", 11 * 14 * 14 * 32)          # 11 images per launch -> 8-image slices
    assert hip._batch_ranges(B, 14 * 14 * 32, B // groups) == [(0, 8), (8, 16), (16, 24), (24, 32), (32, 40), (40, 48)]
    monkeypatch.setattr(hip, "ELEM_LIMIT", 13 * 14 * 14 * 32)          # 8-image slices, 24-image groups: (16, 24) ...
    assert (16, 24) in hip._batch_ranges(B, 14 * 14 * 32, B // groups)
    monkeypatch.setattr(hip, "ELEM_LIMIT", 21 * 14 * 14 * 32)          # 16-image slices: (0,16) (16,24) | (24,40) (40,48)
    assert hip._batch_ranges(B, 14 * 14 * 32, B // groups) == [(0, 16), (16, 24), (24, 40), (40, 48)]
    sliced = True
    got, got_st, sink = run()
    assert torch.equal(got, ref)
    assert (choice == "pconv") == (not sink)
    for u, v in zip(got_st, ref_st):
        assert torch.allclose(u, v, rtol=1e-4, atol=1e-5)
    # against the plain pass over the output
    gamma, beta = torch.ones(32).cuda(), torch.zeros(32).cuda()
    plain = hip.bn_train_affine(got, gamma, beta, torch.zeros(32).cuda(), torch.ones(32).cuda(), 0.99, 1e-3, True, True,
                                groups)
    for u, v in zip(got_st, plain):
        assert torch.allclose(u, v, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("groups", [1, 2])
def test_wide_partials_are_folded_before_the_finalize(hip, groups, monkeypatch):
    """BatchNorm statistics from very many partial columns (one per image and row band at HBM-sized batches): the
    64-fold pre-reduction gives the statistics of the direct finalize."""
    torch.manual_seed(8)
    Cp, nb, rows = 32, 2 * 9001, 2 * 9001 * 7
    part = torch.randn(2, Cp, nb, device="cuda")
    part[1] = part[1].abs() * 7 + 40                 # sums of squares: keep the variances positive
    outs = []
    for lim in (1 << 30, 1000):
        monkeypatch.setattr(hip, "FOLD_PARTIALS_MIN", lim)
        gamma, beta = torch.rand(Cp).cuda() + 0.5, torch.randn(Cp).cuda()
        torch.manual_seed(9)
        mm, mv = torch.zeros(Cp).cuda(), torch.ones(Cp).cuda()
        st = hip.bn_train_affine_from_part(part.reshape(-1), nb, rows, torch.ones(Cp).cuda(), torch.zeros(Cp).cuda(), mm,
                                           mv, 0.9, 1e-3, True, True, groups)
        outs.append([t.clone() for t in st] + [mm, mv])
    for u, v in zip(*outs):
        assert torch.allclose(u, v, rtol=1e-5, atol=1e-6)


def test_batch_split_launches_for_tensors_beyond_32bit_addressing(hip, monkeypatch):
    """Tensors of >= 2^31 elements are split along the batch into several launches (BASELINE config 4: batch sized to
    HBM).  Exercised here with an artificially small limit: fprop / dgrad / one-channel ops give the same bits as one
    launch, wgrad the same sums up to fp32 addition order."""
    torch.manual_seed(2)
    g = G.conv2d_geom(14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same")
    B = 48
    x = dev(hip, r16(torch.randn(B, 14, 14, 32)))
    w = r16(torch.randn(64, 5, 5, 32) * 0.05)
    wA = padw(hip, w)
    wB = padw(hip, w.permute(3, 1, 2, 0).contiguous())
    bias = torch.randn(64).cuda()
    dy = dev(hip, r16(torch.randn(B, 7, 7, 64)))
    ref_y = hip.conv_fprop(x, wA, bias, g, "leaky", 0.3)
    ref_dx = hip.conv_dgrad(dy, wB, g)
    ref_dw = hip.conv_wgrad(x, dy, g, "A")
    g1 = G.conv2d_geom(28, 28, 1, 32, (5, 5), (2, 2), (1, 1), "same")
    img = dev(hip, r16(torch.rand(B, 28, 28, 1)))
    ref_col = hip.im2col_1ch(img, g1)
    ref_img = hip.col2im_1ch(ref_col, g1, None, "tanh01", 0.3)
    monkeypatch.setattr(hip, "ELEM_LIMIT", 11 * 14 * 14 * 32)          # 11 images per launch -> 8-image slices
    assert len(hip._batch_ranges(B, 14 * 14 * 32)) == 6
    assert torch.equal(hip.conv_fprop(x, wA, bias, g, "leaky", 0.3), ref_y)
    assert torch.equal(hip.conv_dgrad(dy, wB, g), ref_dx)
    dw = hip.conv_wgrad(x, dy, g, "A")
    close(dw.cpu(), ref_dw.cpu(), rtol=1e-4, atol=1e-4 * float(ref_dw.abs().max()))
    assert torch.equal(hip.im2col_1ch(img, g1), ref_col)
    assert torch.equal(hip.col2im_1ch(ref_col, g1, None, "tanh01", 0.3), ref_img)


def test_autotune_choices_export_import_and_persist(hip, tmp_path, monkeypatch):
    """Tuned plans are plain data: another rank / a later process rebuilds the same kernels from the table."""
    torch.manual_seed(3)
    g = G.conv2d_geom(14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same")
    x = dev(hip, r16(torch.randn(40, 14, 14, 32)))
    w = padw(hip, r16(torch.randn(64, 5, 5, 32) * 0.05))
    dy = dev(hip, r16(torch.randn(40, 7, 7, 64)))
    monkeypatch.setenv("DDL_AUTOTUNE_CACHE", str(tmp_path / "tune.json"))
    monkeypatch.setattr(hip, "AUTOTUNE", True)          # (the suite also runs with DDL_AUTOTUNE=0)
    y0 = hip.conv_fprop(x, w, None, g)
    dw0 = hip.conv_wgrad(x, dy, g, "A")
    table = hip.export_choices()
    assert any(k.startswith("conv|") for k in table) and any(k.startswith("wgrad|") for k in table)
    hip._CHOICE_DIRTY = True
    hip.save_choice_cache()
    assert (tmp_path / "tune.json").exists()
    # a "new process": nothing resolved, tuning forbidden -> the plans must come from the table
    hip._CONV_CHOICE.clear()
    hip._CHOICE_SPEC.clear()
    hip._load_choice_cache()
    assert hip.export_choices() == table
    monkeypatch.setattr(hip, "_time_us", lambda *a, **k: (_ for _ in ()).throw(AssertionError("re-tuned")))
    assert torch.equal(hip.conv_fprop(x, w, None, g), y0)
    assert torch.equal(hip.conv_wgrad(x, dy, g, "A"), dw0)
    hip._CONV_CHOICE.clear()
    hip.import_choices(table)                     # what sync_choices does on ranks != 0
    assert torch.equal(hip.conv_fprop(x, w, None, g), y0)


# ------------------------------------------------------------------------------ local-SGD kernels (BASELINE config 5)
@pytest.mark.parametrize("n", [8, 1000, 259848, 1066184])
def test_avg_delta_and_scale_emit_vs_fp32_torch(hip, n):
    """`avg_delta` (P += sum * 1/W - snapshot, bf16 copy re-emitted) and `scale_emit` (P *= 1/W, bf16 copy) are the
    two kernels of parallel/averaging.py; checked against the same arithmetic in fp32 torch, bit for bit (one fused
    multiply-add free expression per element: `p + (s * w - o)` evaluated left to right as the kernel does)."""
    g = torch.Generator().manual_seed(n)
    P = torch.randn(n, generator=g).cuda()
    total = (torch.randn(n, generator=g) * 4).cuda()
    snap = torch.randn(n, generator=g).cuda()
    Pc = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    for inv in (0.5, 0.125, 1.0 / 3.0):
        p0 = P.clone()
        hip.avg_delta(P, Pc, total, snap, inv)
        ref = p0.cpu() + (total.cpu() * torch.tensor(inv, dtype=torch.float32) - snap.cpu())
        # the compiler may contract s * w - o into one fma: allow one ulp of the product term
        assert torch.allclose(P.cpu(), ref, rtol=0, atol=float(total.abs().max()) * inv * 2 ** -22)
        assert torch.equal(Pc.cpu(), P.cpu().bfloat16())
        p0 = P.clone()
        hip.scale_emit(P, Pc, inv)
        assert torch.equal(P.cpu(), p0.cpu() * torch.tensor(inv, dtype=torch.float32))
        assert torch.equal(Pc.cpu(), P.cpu().bfloat16())
    # no bf16 copy requested: only the master buffer moves
    p0 = P.clone()
    hip.scale_emit(P, None, 0.25)
    assert torch.equal(P.cpu(), p0.cpu() * 0.25)


def test_averaging_kernels_equal_the_cpu_form_of_the_averager(hip):
    """The device branch of PeriodicAverager._apply / _average_now against its own CPU branch on the same numbers:
    W replicas' parameters, a snapshot taken `delay` steps ago, local progress since then."""
    W, n = 4, 4096
    g = torch.Generator().manual_seed(1)
    snaps = [torch.randn(n, generator=g) for _ in range(W)]            # every rank's snapshot
    prog = torch.randn(n, generator=g) * 0.01                          # this rank's local progress since the snapshot
    total = torch.stack(snaps).sum(0)
    cpu = snaps[0] + prog
    cpu = cpu + (total * (1.0 / W) - snaps[0])                         # averaging.py CPU branch
    P = (snaps[0] + prog).cuda()
    Pc = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    hip.avg_delta(P, Pc, total.cuda(), snaps[0].cuda(), 1.0 / W)
    assert torch.allclose(P.cpu(), cpu, rtol=0, atol=1e-6)
    # the average keeps the local progress and replaces the snapshot by the mean
    assert torch.allclose(P.cpu(), total / W + prog, rtol=0, atol=1e-5)


def test_softmax_loss_label_gather_by_index_equals_gathered_targets(hip):
    """`softmax_ce(..., index=)` / `softmax_mse(..., index=)`: the minibatch's label rows are read through the index
    vector inside the loss kernel -- bit-identical to gathering them first."""
    g = torch.Generator().manual_seed(3)
    logits = hip.pad_channels(torch.randn(96, 10, generator=g).cuda(), 16)
    pool = torch.nn.functional.one_hot(torch.randint(0, 10, (500,), generator=g), 10).float().cuda()
    idx = torch.randint(0, 500, (96,), generator=g).cuda()
    for fn in (hip.softmax_ce, hip.softmax_mse):
        l0, d0, p0 = fn(logits, pool.index_select(0, idx).contiguous())
        l1, d1, p1 = fn(logits, pool, index=idx)
        assert torch.equal(l0, l1) and torch.equal(d0, d1) and torch.equal(p0, p1)


@pytest.mark.parametrize("case", [(57, 57, 128, 3, 2, 2, "softplus"), (28, 28, 128, 3, 1, 2, "relu"),
                                  (18, 18, 16, 3, 1, 3, "linear"), (10, 10, 64, 2, 3, 2, "sigmoid"),
                                  (13, 12, 32, 1, 2, 1, "tanh")])
def test_depthwise_dgrad_fused_with_batchnorm_backward(hip, case):
    """`dw_dgrad_bn_bwd`: depthwise input gradient + backward of the BatchNorm in front of the layer as two passes that
    recompute the BatchNorm's dY instead of storing it.  Against the three-pass form on the same operands (dw dgrad ->
    bn_bwd with the bias-gradient sums): dx agrees to bf16 rounding (the column sums are taken in another order), and
    dgamma / dbeta / dbias to fp32 summation; and against the fp32 oracle."""
    H, W, Cc, k, s, d, prev_act = case
    g = G.depthwise_geom(H, W, Cc, (k, k), (s, s), (d, d), "valid", keras224_compat=True)
    B = 4 if H > 30 else 19
    torch.manual_seed(5)
    x = r16(torch.randn(B, H, W, Cc) * 1.5 + 0.3)            # the BatchNorm's input (the conv output, post-activation)
    if prev_act in ("relu", "softplus", "sigmoid"):
        x = r16(x.abs() * (0.5 if prev_act == "sigmoid" else 1.0)).clamp(max=0.99 if prev_act == "sigmoid" else 1e9)
    w = r16(torch.randn(k, k, Cc))
    dy = r16(torch.randn(B, g.OH, g.OW, Cc))
    gamma = torch.rand(Cc) + 0.5
    xf = x.reshape(-1, Cc)
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    xd, wd, dyd = dev(hip, x), padw(hip, w), dev(hip, dy)
    md, isd, gd = mean.cuda(), invstd.cuda(), gamma.cuda()
    outs = {}
    for fused in (True, False):
        dgam, dbet, dbias = (torch.zeros(Cc, device="cuda") for _ in range(3))
        if fused:
            dx = hip.dw_dgrad_bn_bwd(dyd, wd, g, xd, md, isd, gd, prev_act, 0.3, dgam, dbet, dbias)
            assert dx is not None
        else:
            gfull = hip.conv_dgrad(dyd, wd, g)
            dx, _, _ = hip.bn_bwd(gfull, xd, None, md, isd, gd, None, 0.3, True, prev_act, 0.3, dgam, dbet, dbias)
        torch.cuda.synchronize()
        outs[fused] = (host(hip, dx, Cc), dgam.cpu(), dbet.cpu(), dbias.cpu())
    a, b = outs[True], outs[False]
    close(a[0], b[0], rtol=1e-2)
    # sums of many signed terms (the bias gradient of a conv in front of a BatchNorm cancels to ~0 where act' == 1):
    # tolerances are set by the magnitude of the terms, not of the result
    mag_g = float(cpu_ref.conv_dgrad(dy, w, g).abs().reshape(-1, Cc).sum(0).max())
    mag_dx = float(b[0].abs().reshape(-1, Cc).sum(0).max())
    close(a[1], b[1], rtol=1e-3, atol=1e-4 * mag_g * 3)
    close(a[2], b[2], rtol=1e-3, atol=1e-4 * mag_g)
    close(a[3], b[3], rtol=1e-3, atol=1e-3 * mag_dx)
    # fp32 oracle of the same chain on the bf16-rounded g
    gref = r16(cpu_ref.conv_dgrad(dy, w, g))
    xhat = (x - mean) * invstd
    rows = B * H * W
    c1 = gref.reshape(-1, Cc).sum(0) / rows
    c2 = (gref * xhat).reshape(-1, Cc).sum(0) / rows
    dxr = gamma * invstd * (gref - c1 - xhat * c2)
    ad = {"linear": torch.ones_like(x), "relu": (x > 0).float(), "sigmoid": x * (1 - x), "tanh": 1 - x * x,
          "softplus": 1 - torch.exp(-x)}[prev_act]
    dxr = dxr * ad
    close(a[0], dxr, rtol=3e-2)
    close(a[1], (gref * xhat).reshape(-1, Cc).sum(0), rtol=1e-2)
    close(a[2], gref.reshape(-1, Cc).sum(0), rtol=1e-2, atol=1e-2 * float(gref.reshape(-1, Cc).sum(0).abs().max()) + 1e-3)
    close(a[3], dxr.reshape(-1, Cc).sum(0), rtol=2e-2, atol=2e-3 * float(dxr.abs().reshape(-1, Cc).sum(0).max()))


UPCONV_CASES = [
    # H, W, filters, k, s, d, padding, act
    (28, 28, 64, 3, 2, 1, "valid", "softplus"),     # model0 conv2d_transpose_1 (57 x 57 x 64)
    (28, 28, 16, 2, 2, 3, "valid", "relu"),         # k_eff 4 > s
    (11, 9, 128, 3, 1, 2, "valid", "tanh"),         # stride 1: one class
    (8, 8, 8, 1, 2, 2, "valid", "sigmoid"),         # k 1 s 2: three of four classes are bias only
    (7, 7, 32, 3, 3, 1, "valid", "softsign"),       # nine classes, one tap each
    (14, 14, 64, 3, 2, 1, "same", "leaky"),
]


@pytest.mark.parametrize("case", UPCONV_CASES)
@pytest.mark.parametrize("B", [3, 40])
def test_upconv_1ch_forward_statistics_and_weight_gradient(hip, case, B):
    """Streaming kernels for a Conv2DTranspose fed by a one-channel image (csrc/hip/upconv_1ch.hip) against the fp32
    oracle of the same op; the statistics partials against a column reduction of the stored output."""
    H, W, Fo, k, s, d, pad, act = case
    g = G.conv2d_transpose_geom(H, W, 1, Fo, (k, k), (s, s), (d, d), pad)
    torch.manual_seed(31)
    x = r16(torch.randn(B, H, W, 1))
    w = r16(torch.randn(Fo, k, k, 1) / k)                       # canonical [filters][KH][KW][in]
    b = torch.randn(Fo)
    wA = w.permute(3, 1, 2, 0).contiguous()
    y_ref = cpu_ref.conv_dgrad(x, wA, g, b, act)
    so = {}
    xd, wd = dev(hip, x), padw(hip, w)
    y = hip.upconv_1ch(xd, wd, g, b.cuda(), act, 0.3, stats_out=so)
    assert y is not None and tuple(y.shape) == (B, g.H, g.W, hip.pad8(Fo))
    close(host(hip, y, Fo), y_ref)
    y2 = hip.conv_dgrad(xd, wd, g, b.cuda(), act)               # the generic kernels agree to rounding
    assert torch.allclose(y.float(), y2.float(), rtol=2e-2, atol=2e-2)
    Np, nb = hip.pad8(Fo), so["nb"]
    sums = so["part"].view(2, Np, nb).double().sum(-1).cpu()
    yf = y.float().reshape(-1, Np).double().cpu()
    assert torch.allclose(sums[0], yf.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(sums[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-2)
    dout = r16(torch.randn(B, g.H, g.W, Fo))
    dw_ref = cpu_ref.conv_wgrad(dout, x, g).permute(3, 1, 2, 0)                  # -> canonical
    out = torch.full(tuple(wd.shape), float("nan"), device="cuda")
    assert hip.upconv_1ch_wgrad(xd, dev(hip, dout), wd, g, out)
    close(out[..., :1].cpu(), dw_ref, rtol=1e-2)
    assert float(out[..., 1:].abs().max()) == 0.0
    out2 = torch.empty_like(out)
    assert hip.upconv_1ch_wgrad(xd, dev(hip, dout), wd, g, out2) and torch.equal(out, out2)      # deterministic


def test_network_with_a_transposed_first_layer_uses_the_streaming_kernels(hip, monkeypatch):
    """model0 (ConvT 1 -> 64 first): one training step with the streaming path equals the generic path to rounding."""
    from pathlib import Path
    from distributeddeeplearning_amd.models import network
    from distributeddeeplearning_amd.models.ir import ModelSpec
    from distributeddeeplearning_amd.ops import hip as H_
    arch = Path(__file__).parent / "fixtures" / "reference_models" / "model0" / "architecture.json"
    spec = ModelSpec.from_keras_json(arch.read_text())
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(H_, "UPCONV", flag)
        net = network.Network(spec, "cuda", 5)
        torch.manual_seed(9)
        x = torch.rand(16, 28, 28, 1) * 255
        xd = net.prepare_input(x.cuda())
        y = net.forward(xd, training=True)
        dy = torch.ones_like(y) / y.numel()
        net.backward(dy)
        first = net.first_param_op()
        assert first.onech == "up"
        res[flag] = (y.float().cpu(), first.gkernel.clone().cpu(), net.G.clone().cpu())
    assert torch.allclose(res[True][0], res[False][0], rtol=2e-2, atol=2e-3)
    gk_a, gk_b = res[True][1], res[False][1]
    assert torch.allclose(gk_a, gk_b, rtol=5e-2, atol=2e-2 * float(gk_b.abs().max()) + 1e-12)


@pytest.mark.parametrize("case", [(40, 21, 21, 64, 128, "softplus"), (3, 9, 11, 24, 40, "relu"), (70, 16, 16, 128, 16, "linear"),
                                  (5, 7, 7, 8, 64, "tanh")])
def test_gemm_sk_applies_batchnorm_on_load_and_emits_statistics(hip, case, monkeypatch):
    """A 1x1 transposed conv behind a BatchNorm (model0's second layer): the skinny-K GEMM reads x through the affine
    (bit-identical to the materialised BatchNorm output) and its store loop leaves the statistics of the next one."""
    B, H, W, Cin, Fo, act = case
    monkeypatch.setattr(hip, "SK_STATS_MIN_ROWS", 0)
    g = G.conv2d_transpose_geom(H, W, Cin, Fo, (1, 1), (1, 1), (1, 1), "valid")
    torch.manual_seed(41)
    x = dev(hip, r16(torch.randn(B, H, W, Cin)))
    w = padw(hip, r16(torch.randn(Fo, 1, 1, Cin) / math.sqrt(Cin)))
    b = torch.randn(Fo).cuda()
    Cp, Np = hip.pad8(Cin), hip.pad8(Fo)
    sc, sh = (torch.rand(1, Cp) + 0.5).cuda(), torch.randn(1, Cp).cuda()
    xm = hip.bn_apply(x, sc, sh, None, 0.3, Cin, 1)
    y_ref = hip.conv_dgrad(xm, w, g, b, act)
    so = {"groups": 1}
    y = hip.conv_dgrad(x, w, g, b, act, in_affine=(sc, sh, 1, Cin), stats_out=so)
    assert torch.equal(y, y_ref)
    assert "part" in so
    nb = so["nb"]
    sums = so["part"].view(2, Np, nb).double().sum(-1).cpu()
    yf = y.float().reshape(-1, Np).double().cpu()
    assert torch.allclose(sums[0], yf.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(sums[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-2)
    # the weight gradient of the same layer through the streaming kernel with the affine on its X operand
    dout = dev(hip, r16(torch.randn(B, g.H, g.W, Fo)))
    dw_ref = hip.conv_wgrad(dout, xm, g, "B")
    gs = G.ConvGeom(g.OH, g.OW, g.O, g.H, g.W, g.I, 1, 1).check()
    dw = hip.conv_wgrad(x, dout, gs, "A", x_affine=(sc.reshape(-1), sh.reshape(-1), Cin))
    close(dw.cpu(), dw_ref.cpu(), rtol=1e-2)


@pytest.mark.parametrize("case", [(57, 57, 128, 3, 2, "relu"), (28, 28, 128, 3, 1, "tanh"), (10, 10, 128, 2, 3, "sigmoid"),
                                  (18, 18, 16, 3, 1, "softsign"), (5, 4, 24, 3, 1, "linear")])
def test_depthwise_ring_forward_emits_batchnorm_statistics(hip, case):
    """The row-ring depthwise forward leaves the partial sums of its stored output (one column per image): equal to a
    column reduction of the tensor, and the output equal to the plain launch."""
    H, W, Cc, k, s, act = case
    g = G.depthwise_geom(H, W, Cc, (k, k), (s, s), (1, 1), "valid")
    B = 6
    torch.manual_seed(61)
    x = dev(hip, r16(torch.randn(B, H, W, Cc)))
    w = padw(hip, r16(torch.randn(k, k, Cc) / k))
    b = torch.randn(Cc).cuda()
    y0 = hip.conv_fprop(x, w, b, g, act)
    so = {"groups": 1}
    y = hip.conv_fprop(x, w, b, g, act, stats_out=so)
    assert torch.equal(y, y0)
    if "part" not in so:            # channel chunking outside the epilogue's range (e.g. 24 channels): plain path
        assert Cc % 16
        return
    Cp, nb = hip.pad8(Cc), so["nb"]
    sums = so["part"].view(2, Cp, nb).double().sum(-1).cpu()
    yf = y.float().reshape(-1, Cp).double().cpu()
    assert torch.allclose(sums[0], yf.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(sums[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("act", ["softplus", "relu", "linear"])
def test_upconv_wgrad_applies_the_batchnorm_backward_on_load(hip, act):
    """`upconv_1ch_wgrad(bn=...)`: reduction + finalize (`bn_bwd_sums`), then the weight gradient forms the gradient at
    the BatchNorm's input on load -- against bn_bwd (which writes that tensor) followed by the plain weight gradient."""
    H, Fo, k, s, B = 28, 64, 3, 2, 9
    g = G.conv2d_transpose_geom(H, H, 1, Fo, (k, k), (s, s), (1, 1), "valid")
    torch.manual_seed(71)
    x = dev(hip, r16(torch.rand(B, H, H, 1)))
    w = padw(hip, r16(torch.randn(Fo, k, k, 1) / k))
    y = hip.upconv_1ch(x, w, g, torch.randn(Fo).cuda(), act, 0.3)              # the BatchNorm's input
    dy = dev(hip, r16(torch.randn(B, g.H, g.W, Fo)))
    yf = y.float().reshape(-1, Fo)
    mean, invstd = yf.mean(0).contiguous(), (1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-3)).contiguous()
    gamma = (torch.rand(Fo) + 0.5).cuda()
    ref_g = [torch.zeros(Fo, device="cuda") for _ in range(3)]
    dx, _, _ = hip.bn_bwd(dy, y, None, mean, invstd, gamma, None, 0.3, True, act, 0.3, *ref_g)
    ref = torch.empty(tuple(w.shape), device="cuda")
    assert hip.upconv_1ch_wgrad(x, dx, w, g, ref)
    got_g = [torch.zeros(Fo, device="cuda") for _ in range(3)]
    c1, c2 = hip.bn_bwd_sums(dy, y, mean, invstd, gamma, act, 0.3, *got_g)
    got = torch.empty(tuple(w.shape), device="cuda")
    assert hip.upconv_1ch_wgrad(x, dy, w, g, got, bn=dict(x=y, mean=mean, invstd=invstd, gamma=gamma, c1=c1, c2=c2,
                                                          act=act, alpha=0.3))
    for a, b in zip(got_g, ref_g):
        assert torch.equal(a, b)
    close(got[..., :1].cpu(), ref[..., :1].cpu(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()))
    assert float(got[..., 1:].abs().max()) == 0.0


@pytest.mark.parametrize("case", [(28, 28, 128, 3, 1, 1, "valid", "softplus"), (28, 28, 32, 2, 1, 1, "valid", "softplus"),
                                  (28, 28, 128, 2, 3, 1, "valid", "softsign"), (28, 28, 32, 1, 1, 2, "valid", "sigmoid"),
                                  (28, 28, 16, 3, 1, 1, "same", "relu"), (13, 11, 64, 3, 2, 2, "same", "tanh")])
@pytest.mark.parametrize("B", [2, 33])
def test_stream1_forward_conv_from_one_channel(hip, case, B):
    """The streaming one-channel kernels in the forward direction (Conv2D 1 -> N, k <= 3: the first layers of replicas
    1-4): output, BatchNorm statistics partials and the deterministic weight gradient against the fp32 oracle."""
    H, W, Fo, k, s, d, pad, act = case
    g = G.conv2d_geom(H, W, 1, Fo, (k, k), (s, s), (d, d), pad)
    torch.manual_seed(81)
    x = r16(torch.randn(B, H, W, 1))
    w = r16(torch.randn(Fo, k, k, 1) / k)
    b = torch.randn(Fo)
    y_ref = cpu_ref.conv_fprop(x, w, b, g, act)
    so = {}
    xd, wd = dev(hip, x), padw(hip, w)
    y = hip.upconv_1ch(xd, wd, g, b.cuda(), act, 0.3, stats_out=so, fwdconv=True)
    assert y is not None and tuple(y.shape) == (B, g.OH, g.OW, hip.pad8(Fo))
    close(host(hip, y, Fo), y_ref)
    Np, nb = hip.pad8(Fo), so["nb"]
    sums = so["part"].view(2, Np, nb).double().sum(-1).cpu()
    yf = y.float().reshape(-1, Np).double().cpu()
    assert torch.allclose(sums[0], yf.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(sums[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-2)
    dout = r16(torch.randn(B, g.OH, g.OW, Fo))
    dw_ref = cpu_ref.conv_wgrad(x, dout, g)
    out = torch.full(tuple(wd.shape), float("nan"), device="cuda")
    assert hip.upconv_1ch_wgrad(xd, dev(hip, dout), wd, g, out, fwdconv=True)
    close(out[..., :1].cpu(), dw_ref, rtol=1e-2)
    assert float(out[..., 1:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(28, 28, 32, 5, 2, "same"), (28, 28, 64, 4, 1, "valid"), (17, 13, 16, 5, 1, "same"),
                                  (64, 64, 32, 5, 2, "same")])
@pytest.mark.parametrize("B", [3, 130])
def test_stream1_wgrad_of_a_5x5_layer_at_a_one_channel_image(hip, case, B):
    """The streaming weight gradient of the DCGAN pair's one-channel ends (k = 4, 5): as the discriminator's first
    Conv2D (canonical [O][KH][KW][8] layout, bias = column sums of dY) and as the generator's last Conv2DTranspose
    ([KH][KW][Cp] layout, BatchNorm applied to x on load, scalar bias = sum of the dY image), against the fp32 oracle."""
    H, W, Fo, k, s, pad = case
    g = G.conv2d_geom(H, W, 1, Fo, (k, k), (s, s), (1, 1), pad)
    torch.manual_seed(83)
    T, Np = k * k, hip.pad8(Fo)
    img = r16(torch.randn(B, H, W, 1))
    t = r16(torch.randn(B, g.OH, g.OW, Fo))
    ref = cpu_ref.conv_wgrad(img, t, g)                     # [Fo][k][k][1]
    imgd, td = dev(hip, img), dev(hip, t)
    out = torch.full((Fo, k, k, 8), float("nan"), device="cuda")
    bias = torch.full((Fo,), float("nan"), device="cuda")
    assert hip.stream1_wgrad(imgd, td, g, out.view(-1), T * 8, 8, zfill=8, n_out=Fo, bias=bias, bias_mode=1)
    close(out[..., :1].cpu(), ref, rtol=1e-2)
    assert float(out[..., 1:].abs().max()) == 0.0
    assert torch.allclose(bias.cpu(), t.reshape(-1, Fo).sum(0), rtol=1e-3, atol=1e-2)
    out2 = torch.empty_like(out)
    assert hip.stream1_wgrad(imgd, td, g, out2.view(-1), T * 8, 8, zfill=8, n_out=Fo)
    assert torch.equal(out, out2)                           # deterministic, and the bias output is optional
    # transposed-layer form: t is the layer's input in front of a BatchNorm folded into the load
    ctrue = Fo - 3
    sc, sh = torch.rand(Np) + 0.5, torch.randn(Np) * 0.3
    tn = r16(t * sc[:Fo] + sh[:Fo])
    tn[..., ctrue:] = 0
    ref2 = cpu_ref.conv_wgrad(img, tn, g)[..., 0].reshape(Fo, T).t()       # [T][Fo]
    out3 = torch.full((T, Np), float("nan"), device="cuda")
    b1 = torch.full((1,), float("nan"), device="cuda")
    assert hip.stream1_wgrad(imgd, td, g, out3.view(-1), 1, Np, affine=(sc.cuda(), sh.cuda(), ctrue), bias=b1, bias_mode=2)
    close(out3[:, :Fo].cpu(), ref2, rtol=1e-2)
    assert float(out3[:, ctrue:].abs().max()) == 0.0
    assert torch.allclose(b1.cpu(), img.sum().reshape(1), rtol=1e-3, atol=1e-2)


@pytest.mark.gpu
def test_stream1_wgrad_in_batch_slices_and_over_many_image_groups(hip, monkeypatch):
    """Tensors of 2^31 elements and more reach the streaming weight gradient in batch slices whose slabs are summed by
    one finish launch, and a block walks several image groups when the batch exceeds one round of blocks: both against
    the unsliced result and the fp32 oracle."""
    g = G.conv2d_geom(28, 28, 1, 32, (5, 5), (2, 2), (1, 1), "same")
    torch.manual_seed(84)
    B, T = 2100, 25
    img = r16(torch.randn(B, 28, 28, 1))
    t = r16(torch.randn(B, g.OH, g.OW, 32))
    ref = cpu_ref.conv_wgrad(img, t, g)
    imgd, td = dev(hip, img), dev(hip, t)
    out = torch.full((32, 5, 5, 8), float("nan"), device="cuda")
    bias = torch.full((32,), float("nan"), device="cuda")
    assert hip.stream1_wgrad(imgd, td, g, out.view(-1), T * 8, 8, zfill=8, n_out=32, bias=bias, bias_mode=1)
    close(out[..., :1].cpu(), ref, rtol=1e-2)
    monkeypatch.setattr(hip, "ELEM_LIMIT", 700 * g.OH * g.OW * 32)          # three slices of 700 images
    out2 = torch.full_like(out, float("nan"))
    bias2 = torch.full_like(bias, float("nan"))
    assert hip.stream1_wgrad(imgd, td, g, out2.view(-1), T * 8, 8, zfill=8, n_out=32, bias=bias2, bias_mode=1)
    assert torch.allclose(out2, out, rtol=1e-4, atol=1e-3)
    assert torch.allclose(bias2, bias, rtol=1e-4, atol=1e-3)
    assert torch.allclose(bias.cpu(), t.reshape(-1, 32).sum(0), rtol=1e-3, atol=2e-2)
    # 64 x 64 images: two per block fit the LDS budget, 1100 images = 550 groups over one round of 512 blocks
    monkeypatch.undo()
    g2 = G.conv2d_geom(64, 64, 1, 8, (5, 5), (1, 1), (1, 1), "same")
    img = r16(torch.randn(1100, 64, 64, 1))
    t = r16(torch.randn(1100, 64, 64, 8))
    ref = cpu_ref.conv_wgrad(img, t, g2)
    out3 = torch.full((8, 5, 5, 8), float("nan"), device="cuda")
    assert hip.stream1_wgrad(dev(hip, img), dev(hip, t), g2, out3.view(-1), T * 8, 8, zfill=8, n_out=8)
    close(out3[..., :1].cpu(), ref, rtol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(14, 14, 32, 64, 2, 2), (7, 7, 40, 24, 1, 4), (14, 14, 64, 32, 2, 4)])
@pytest.mark.parametrize("B", [3, 260])
def test_pconv_eight_wave_blocks_match_four_wave_blocks(hip, shape, B):
    """The eight-wave instantiations of the patch-resident kernel (same patch, same weight stages, twice the waves per
    CU): output bit-identical to the four-wave plan of the same problem (the K order per output value is the same),
    BatchNorm statistics equal to a column reduction of the stored output, and the oracle within tolerance."""
    H, W, I, O, s, fr = shape
    g = G.conv2d_geom(H, W, I, O, (5, 5), (s, s), (1, 1), "same")
    torch.manual_seed(85)
    xh, wh = r16(torch.randn(B, H, W, I)), r16(torch.randn(O, 5, 5, I) / 20)
    x, w = dev(hip, xh), padw(hip, wh)
    bias = torch.randn(O)
    Ip, Op = hip.pad8(I), hip.pad8(O)
    outs = []
    for nw in (4, 8):
        pl = hip._pconv_plan(g, "fprop", B, Ip, Op, (g.sh, g.sw), x.device, min_rows=16, fr=fr, nw=nw)
        assert pl is not None and pl["NW"] == nw
        nb = hip._pconv_units(pl, B)
        y = torch.full((B, g.OH, g.OW, Op), float("nan"), dtype=torch.bfloat16, device="cuda")
        part = torch.full((2 * Op * nb,), float("nan"), device="cuda")
        hip._run_pconv(pl, x, w, bias.cuda(), y, g, B, g.H, g.W, Ip, g.OH, g.OW, Op, O, (g.sh, g.sw), (1, 1), "leaky", 0.3,
                       stats=part)
        sums = part.view(2, Op, nb).double().sum(-1).cpu()
        yf = y.float().reshape(-1, Op).double().cpu()
        assert torch.allclose(sums[0], yf.sum(0), rtol=1e-4, atol=1e-2)
        assert torch.allclose(sums[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-2)
        y0 = torch.full_like(y, float("nan"))
        hip._run_pconv(pl, x, w, bias.cuda(), y0, g, B, g.H, g.W, Ip, g.OH, g.OW, Op, O, (g.sh, g.sw), (1, 1), "tanh", 0.3)
        outs.append((y, y0))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    close(host(hip, outs[1][0], O), cpu_ref.conv_fprop(xh, wh, bias, g, "leaky", 0.3))
    close(host(hip, outs[1][1], O), cpu_ref.conv_fprop(xh, wh, bias, g, "tanh"))


@pytest.mark.gpu
@pytest.mark.parametrize("B", [6, 300])
def test_wgrad2_applies_a_batchnorm_to_the_dy_operand_on_load(hip, B):
    """conv_wgrad(dy_affine=...): the persistent weight-gradient kernel stages its dY operand as bf16(dY * scale + shift)
    (channels >= c_true zero) -- the weight gradient of a transposed convolution behind a BatchNorm applied on load.
    Bit-identical to the gradient taken from the materialised operand with the same plan set."""
    g = G.conv2d_geom(28, 28, 32, 64, (5, 5), (2, 2), (1, 1), "same")      # core geometry of convT 14x14x64 -> 28x28x32
    torch.manual_seed(86)
    xd = dev(hip, r16(torch.randn(B, 28, 28, 32)))                          # core X: the layer's output gradient
    yd = dev(hip, r16(torch.randn(B, 14, 14, 64)))                          # core dY: the layer's input, before its BatchNorm
    ctrue = 61
    sc, sh = (torch.rand(64) + 0.5).cuda(), (torch.randn(64) * 0.3).cuda()
    ym = hip.bn_apply(yd, sc, sh, None, 0.3, ctrue, 1)
    ref = hip.conv_wgrad(xd, ym, g, "B")
    got = hip.conv_wgrad(xd, yd, g, "B", dy_affine=(sc, sh, ctrue))
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    # and against the fp32 oracle
    ymh = host(hip, ym, 64)
    dw = cpu_ref.conv_wgrad(host(hip, xd, 32), ymh, g).permute(3, 1, 2, 0)
    close(got[..., :64].cpu(), dw, rtol=2e-2)
