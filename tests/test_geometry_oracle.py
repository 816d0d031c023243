"""Keras/TF convolution geometry and the CPU oracle's adjoint consistency."""
import pytest
import torch

from distributeddeeplearning_amd.ops import cpu_ref
from distributeddeeplearning_amd.ops import geometry as G


def test_tf_same_padding_is_asymmetric_for_stride2_k5():
    g = G.conv2d_geom(28, 28, 1, 32, (5, 5), (2, 2), (1, 1), "same")
    assert (g.OH, g.OW, g.pt, g.pl) == (14, 14, 1, 1) and g.pad_bottom_right() == (2, 2)
    g = G.conv2d_geom(14, 14, 32, 64, (5, 5), (2, 2), (1, 1), "same")
    assert (g.OH, g.pt) == (7, 1) and g.pad_bottom_right() == (2, 2)
    g = G.conv2d_geom(7, 7, 64, 128, (5, 5), (2, 2), (1, 1), "same")
    assert (g.OH, g.pt) == (4, 2) and g.pad_bottom_right() == (2, 2)


def test_same_conv_with_kernel_larger_than_image():
    g = G.conv2d_geom(28, 28, 3, 7, (29, 17), (1, 1), (1, 1), "same")       # DDL.py kernel range 3..29
    assert (g.OH, g.OW) == (28, 28) and (g.pt, g.pl) == (14, 8)
    x, w = torch.randn(2, 28, 28, 3), torch.randn(7, 29, 17, 3)
    assert cpu_ref.conv_fprop(x, w, None, g).shape == (2, 28, 28, 7)


@pytest.mark.parametrize("n,k,s,d,want", [(28, 3, 2, 1, 57), (8, 1, 2, 2, 16), (16, 2, 2, 3, 34), (3, 3, 2, 2, 9),
                                           (57, 1, 1, 2, 57)])
def test_keras_valid_deconv_length(n, k, s, d, want):
    assert G.deconv_out_len(n, k, s, d, "valid") == want


def test_valid_conv_raises_when_it_does_not_fit():
    with pytest.raises(ValueError):
        G.conv2d_geom(2, 2, 1, 8, (3, 3), (1, 1), (2, 2), "valid")


@pytest.mark.parametrize("geom", [
    G.conv2d_geom(9, 11, 3, 5, (5, 5), (2, 2), (1, 1), "same"),
    G.conv2d_geom(10, 10, 4, 6, (3, 2), (1, 1), (2, 3), "valid"),
    G.conv2d_transpose_geom(5, 6, 4, 3, (3, 3), (2, 2), (2, 2), "valid"),
    G.conv2d_transpose_geom(7, 7, 8, 4, (5, 5), (2, 2), (1, 1), "same"),
    G.depthwise_geom(9, 9, 6, (3, 3), (2, 2), (1, 1), "valid"),
])
def test_oracle_adjoints(geom):
    """<Y, corr(X, W)> has dgrad as its X-adjoint and wgrad as its W-adjoint."""
    g = geom
    torch.manual_seed(0)
    x = torch.randn(2, g.H, g.W, g.I, dtype=torch.float64)
    wshape = (g.KH, g.KW, g.O) if g.depthwise else (g.O, g.KH, g.KW, g.I)
    w = torch.randn(wshape, dtype=torch.float64)
    y = cpu_ref._corr(x.float(), w.float(), g).double()
    r = torch.randn_like(y)
    lhs = (y * r).sum()
    dx = cpu_ref.conv_dgrad(r.float(), w.float(), g).double()
    dw = cpu_ref.conv_wgrad(x.float(), r.float(), g).double()
    assert torch.allclose((dx * x).sum(), lhs, rtol=1e-4)
    assert torch.allclose((dw * w).sum(), lhs, rtol=1e-4)


def test_transpose_conv_valid_extra_rows_are_bias_only():
    # k=1, s=2: out = in*2; odd rows/cols receive no input contribution (A.4)
    g = G.conv2d_transpose_geom(4, 4, 2, 3, (1, 1), (2, 2), (1, 1), "valid")
    x, w, b = torch.randn(1, 4, 4, 2), torch.randn(2, 1, 1, 3), torch.tensor([1.0, 2.0, 3.0])
    y = cpu_ref.conv_dgrad(x, w, g, b)
    assert y.shape == (1, 8, 8, 3)
    assert torch.allclose(y[0, 1::2, :, :], b.expand(4, 8, 3)) and torch.allclose(y[0, :, 1::2, :], b.expand(8, 4, 3))


def test_losses_and_optimizers_closed_form():
    logits = torch.tensor([[2.0, 0.0, -1.0], [0.0, 0.0, 0.0]])
    onehot = torch.tensor([[1.0, 0, 0], [0, 0, 1.0]])
    loss, dl, p = cpu_ref.softmax_ce(logits, onehot)
    ref = torch.nn.functional.cross_entropy(logits, torch.tensor([0, 2]))
    assert torch.allclose(loss, ref, atol=1e-6)
    assert torch.allclose(dl, (p - onehot) / 2, atol=1e-6)
    l2, d2 = cpu_ref.sigmoid_bce(torch.tensor([0.3, -2.0]), torch.tensor([1.0, 0.0]))
    ref2 = torch.nn.functional.binary_cross_entropy_with_logits(torch.tensor([0.3, -2.0]), torch.tensor([1.0, 0.0]))
    assert torch.allclose(l2, ref2, atol=1e-6)
    # the reference GAN's "fresh Adam applied twice" == 2*lr*sign(g)
    p0 = torch.zeros(4)
    g = torch.tensor([0.5, -3.0, 1e-3, -20.0])
    cpu_ref.adam_fresh_double_step(p0, g, 2e-4, 0.9, 0.999, 1e-8)
    assert torch.allclose(p0, -4e-4 * torch.sign(g), rtol=2e-3)
