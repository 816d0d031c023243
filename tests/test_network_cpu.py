"""The static executor's hand-written backward vs finite differences (CPU oracle path), for every
layer family of the reference, plus DCGAN step semantics."""
import numpy as np
import pytest
import torch

from distributeddeeplearning_amd.models import ir
from distributeddeeplearning_amd.models.dcgan import (GANConfig, GANTrainer, discriminator_spec, generator_spec)
from distributeddeeplearning_amd.models.network import Network
from distributeddeeplearning_amd.ops import functional as F


def _directional_check(spec, x, loss_fn, bn_mode="batch", seed=0, eps=5e-3, tol=0.12):
    net = Network(spec, "cpu", seed, bn_mode)
    # make BN statistics / affine non-trivial
    g = torch.Generator().manual_seed(seed + 1)
    net.P.add_(torch.randn(net.P.shape, generator=g) * 0.05 * (net.P != 0))
    xin = net.prepare_input(x)
    y = net.forward(xin, training=True)
    loss, dy = loss_fn(y)
    net.G.zero_()
    net.backward(dy)
    P0 = net.P.clone()
    G0 = net.G.clone()

    def L(p):
        net.P.copy_(p)
        out = net.forward(xin, training=True)
        net.release()
        return float(loss_fn(out)[0])

    # one random direction per parameter tensor (a single whole-network direction lets the
    # per-tensor contributions cancel and turns the check into noise)
    for p in net.param_table():
        if not p.trainable:
            continue
        d = torch.zeros_like(P0)
        d[p.offset:p.offset + p.numel] = torch.randn(p.numel, generator=g)
        d *= (G0 != 0)                                   # only real (non-padding) slots
        if float(d.norm()) == 0:
            continue
        d /= d.norm()
        analytic = float((G0 * d).sum())
        fd = (L(P0 + eps * d) - L(P0 - eps * d)) / (2 * eps)
        scale = max(abs(fd), abs(analytic))
        assert abs(fd - analytic) <= tol * scale + 5e-3 * float(G0.norm()), (p.layer, p.wname, fd, analytic)


def _ce(n_classes, B, seed=0):
    t = torch.nn.functional.one_hot(torch.randint(0, n_classes, (B,), generator=torch.Generator().manual_seed(seed)),
                                    n_classes).float()
    return lambda y: F.softmax_ce(y, t)[:2]


def _chain(layers, shape):
    m = ir.ModelSpec("m", [ir.Input("in", shape)])
    for l in layers:
        m.append(l)
    return m


def test_mlp_with_bn_every_activation():
    layers = []
    for i, a in enumerate(["relu", "sigmoid", "softplus", "tanh", "softsign"]):
        layers += [ir.Dense(f"d{i}", 9 + i, a), ir.BatchNorm(f"b{i}")]
    layers.append(ir.Dense("out", 10, "softmax"))
    spec = _chain([ir.Flatten("f")] + layers, (6, 6, 3))
    x = torch.rand(16, 6, 6, 3, generator=torch.Generator().manual_seed(0))
    _directional_check(spec, x, _ce(10, 16))


def test_ddl_style_cnn_same_padding_rect_kernels():
    spec = _chain([ir.Conv2D("c1", 5, (7, 3), 1, 1, "same", "tanh"), ir.BatchNorm("b1"),
                   ir.Conv2D("c2", 4, (13, 15), 1, 1, "same", "softplus"), ir.BatchNorm("b2"),   # kernel > image
                   ir.Flatten("f"), ir.Dense("d", 7, "sigmoid"), ir.BatchNorm("b3"), ir.Dense("o", 10, "softmax")],
                  (10, 10, 3))
    x = torch.rand(8, 10, 10, 3, generator=torch.Generator().manual_seed(1))
    _directional_check(spec, x, _ce(10, 8))


def test_random_layer_family_valid_padding_stride_dilation():
    spec = _chain([ir.Conv2D("c", 8, 3, 1, 2, "valid", "softsign"), ir.BatchNorm("b1"),
                   ir.ZeroPadding2D("zp", 1),
                   ir.DepthwiseConv2D("dw", 3, 2, 1, "valid", "tanh"), ir.BatchNorm("b2"),
                   ir.Conv2DTranspose("ct", 6, 2, 2, 3, "valid", "sigmoid"), ir.BatchNorm("b3"),
                   ir.Flatten("f"), ir.Dense("o", 10, "softmax")], (12, 12, 1))
    x = torch.rand(6, 12, 12, 1, generator=torch.Generator().manual_seed(2))
    for loss in (_ce(10, 6), lambda y: F.softmax_mse(y, torch.eye(10)[:6])[:2]):
        _directional_check(spec, x, loss)


@pytest.mark.parametrize("bn_mode", ["batch", "inference"])
def test_dcgan_discriminator_and_generator_backward(bn_mode):
    r = torch.randn(6, 1, generator=torch.Generator().manual_seed(3))
    _directional_check(discriminator_spec(), torch.rand(6, 28, 28, 1), lambda y: ((y * r).sum(), r), bn_mode)
    rg = torch.randn(4, 28, 28, 1, generator=torch.Generator().manual_seed(4))
    _directional_check(generator_spec(), torch.rand(4, 128) * 2 - 1, lambda y: ((y * rg).sum(), rg), bn_mode)


def test_dcgan_shapes_and_param_counts():
    d, g = discriminator_spec(), generator_spec()
    assert d.count_params(True) == 259457 and d.count_params() - d.count_params(True) == 384
    assert g.count_params(True) == 1066177 and g.count_params() - g.count_params(True) == 192
    sd, sg = d.infer_shapes(), g.infer_shapes()
    assert [sd[n] for n in ("conv2d_1", "conv2d_2", "conv2d_3")] == [(14, 14, 32), (7, 7, 64), (4, 4, 128)]
    assert [sg[n] for n in ("dense_1", "conv2d_transpose_1", "conv2d_transpose_2", "conv2d_transpose_3")] == \
        [(6272,), (14, 14, 64), (28, 28, 32), (28, 28, 1)]
    assert generator_spec(64).output_shape == (64, 64, 1) and discriminator_spec(64).output_shape == (1,)


def test_gan_reference_exact_step_is_sign_sgd():
    """A.1: fresh Adam applied twice == theta -= 2*lr*sign(g); A.2: BN uses (never updated) moving stats."""
    t = GANTrainer(GANConfig(batch=8, reference_exact=True, seed=1), "cpu")
    real = t.D.prepare_input(torch.rand(8, 28, 28, 1))
    P0, S0 = t.D.P.clone(), t.D.S.clone()
    t.phase_a(real)
    G = t.D.G.clone()
    t.optD.step()
    moved = G.abs() > 1e-4            # below that, eps=1e-8 vs sqrt(v)~0.03|g| is visible
    assert moved.sum() > 1000
    delta = (t.D.P - P0)[moved]
    assert torch.allclose(delta, -2 * 2e-4 * torch.sign(G[moved]), rtol=5e-2, atol=1e-7)
    assert torch.equal(t.D.S, S0)


def test_gan_default_mode_steps_and_generates_in_unit_range(tmp_path):
    """Plumbing of one trainer (phases touch the right networks, outputs in range, checkpoint round trip).
    Whether the game is actually adversarial is pinned in tests/test_gan_learning.py."""
    t = GANTrainer(GANConfig(batch=16, seed=0), "cpu")
    real = (torch.rand(16, 28, 28, 1, generator=torch.Generator().manual_seed(0)) > 0.7).float()
    losses = [tuple(float(v) for v in t.train_step(real)) for _ in range(6)]
    assert all(np.isfinite(l).all() for l in losses)
    img = t.generate(3)
    assert img.shape == (3, 28, 28, 1) and float(img.min()) >= 0 and float(img.max()) <= 1
    # G phase must not touch D's weights, D phase must not touch G's
    PD, PG = t.D.P.clone(), t.G.P.clone()
    t.phase_a(t.D.prepare_input(real))
    assert torch.equal(t.G.P, PG)
    t.optD.step(); PD = t.D.P.clone()
    t.g_backward(16); t.optG.step()
    assert torch.equal(t.D.P, PD) and not torch.equal(t.G.P, PG)
    # checkpoint -> resume continues identically
    t.save_checkpoint(tmp_path / "gan.ck", 7)
    t2 = GANTrainer(GANConfig(batch=16, seed=0), "cpu")
    assert t2.load_checkpoint(tmp_path / "gan.ck") == 7
    assert all(torch.equal(a, b) for a, b in zip(t.G.logical_state().values(), t2.G.logical_state().values()))
    # ... including the noise stream and the optimizer step: the next steps are identical
    la, lb = t.train_step(real), t2.train_step(real)
    assert float(la[0]) == float(lb[0]) and float(la[1]) == float(lb[1])
    assert torch.equal(t.G.P, t2.G.P) and torch.equal(t.D.P, t2.D.P)


def test_fused_activation_backward_matches_separate_pass_cpu(monkeypatch):
    """Network.backward applies act'(y) of the layer below inside the input-gradient op (and across the
    G -> D boundary via input_mul / dy_premultiplied): same parameters as the separate act_bwd pass."""
    import torch
    from distributeddeeplearning_amd.models import network
    from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer
    real = torch.rand(12, 28, 28, 1, generator=torch.Generator().manual_seed(2))
    a = GANTrainer(GANConfig(batch=12, seed=4), "cpu")
    for _ in range(2):
        a.train_step(real)
    monkeypatch.setattr(network, "FUSE_ACT_BWD", False)
    b = GANTrainer(GANConfig(batch=12, seed=4), "cpu")
    for _ in range(2):
        b.train_step(real)
    assert torch.allclose(a.D.P, b.D.P, rtol=1e-5, atol=1e-7) and torch.allclose(a.G.P, b.G.P, rtol=1e-5, atol=1e-7)
