"""Whole-network GPU execution vs the CPU oracle executor (same logical weights)."""
import pytest
import torch

from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer, discriminator_spec, generator_spec
from distributeddeeplearning_amd.models.network import Network

pytestmark = pytest.mark.gpu


def _pair(spec, seed=0, bn="batch"):
    c, g = Network(spec, "cpu", seed, bn), Network(spec, "cuda", seed, bn)
    # identical, bf16-representable weights on both sides
    st = {k: v.bfloat16().float() for k, v in c.logical_state().items()}
    c.load_logical_state(st)
    g.load_logical_state(st)
    c.emulate_bf16 = True       # the oracle rounds to bf16 where the device stores bf16 (tests/test_layerwise_gpu.py)
    return c, g


def _rel(a, b):
    """relative L2 error (chained bf16 layers: every intermediate is rounded to bf16)"""
    # the floor keeps mathematically-zero gradients (conv bias in front of a batch-stat BN) from
    # turning bf16 rounding noise into a huge relative error
    return float((a - b).norm() / (b.norm() + 1e-2))


@pytest.mark.parametrize("bn", ["batch", "inference"])
def test_discriminator_fwd_bwd(bn):
    c, g = _pair(discriminator_spec(), 1, bn)
    torch.manual_seed(0)
    x = torch.rand(64, 28, 28, 1).bfloat16().float()
    yc = c.forward(c.prepare_input(x), training=True)
    yg = g.forward(g.prepare_input(x), training=True)
    assert _rel(g.to_logical(yg).cpu(), yc) < 1e-2
    dy = torch.randn_like(yc).abs().bfloat16().float() / 64
    dxc = c.backward(dy, need_input_grad=True)
    import distributeddeeplearning_amd.ops.hip as hip
    dxg = g.backward(hip.pad_channels(dy.cuda(), 8), need_input_grad=True)
    assert _rel(g.to_logical(dxg, 1).cpu(), dxc) < 3e-2
    _cmp_grads(c, g)


def _cmp_grads(c, g):
    """Against the bf16-emulating oracle: kernels to 3 %; per-channel vectors (bias / gamma / beta: mixed-sign
    sums over batch x pixels, which amplify the elementwise error of what they sum) to 6 %."""
    batch_bn = any(getattr(op, "mode", None) == "batch" for op in c.ops)
    for (k, gc), (_, gg) in zip(_grads(c).items(), _grads(g).items()):
        if batch_bn and k[1] == "bias" and float(gc.norm()) < 1e-5:
            continue        # a bias in front of batch-statistics BN has an exactly-zero gradient: only noise
        tol = 3e-2 if k[1] in ("kernel", "depthwise_kernel") else 6e-2
        assert _rel(gg.reshape(gc.shape), gc) < tol, k


def _grads(net):
    out = {}
    for p in net.param_table():
        if p.trainable:
            v = net.G[p.offset:p.offset + p.numel].view(p.shape)[p.corner()]
            out[(p.layer, p.wname)] = v.detach().float().cpu()
    return out


def test_generator_fwd_bwd():
    c, g = _pair(generator_spec(), 2)
    torch.manual_seed(0)
    z = (torch.rand(32, 128) * 2 - 1).bfloat16().float()
    yc = c.forward(c.prepare_input(z), training=True)
    yg = g.forward(g.prepare_input(z), training=True)
    assert _rel(g.to_logical(yg).cpu(), yc) < 1e-2
    dy = torch.randn_like(yc).abs().bfloat16().float()
    c.backward(dy)
    import distributeddeeplearning_amd.ops.hip as hip
    g.backward(hip.pad_channels(dy.cuda(), 8))
    _cmp_grads(c, g)


@pytest.mark.parametrize("policy", ["split", "inference", "mixed"])
def test_gan_trainer_steps_in_every_bn_policy(policy):
    """Plumbing only (finite losses, output range); learning is pinned in tests/test_gan_learning.py."""
    t = GANTrainer(GANConfig(batch=256, bn_policy=policy), "cuda")
    torch.manual_seed(0)
    real = (torch.rand(256, 28, 28, 1) > 0.7).float()
    losses = []
    for i in range(8):
        dl, gl = t.train_step(real)
        losses.append((float(dl), float(gl)))
    assert all(torch.isfinite(torch.tensor(l)).all() for l in losses)
    img = t.generate(4)
    assert img.shape == (4, 28, 28, 1) and float(img.min()) >= 0 and float(img.max()) <= 1


def test_captured_step_with_wgrad_branch_matches_eager_bitwise():
    """The hipGraph step forks every layer's weight-gradient chain onto a side stream
    (network.SideBranch); the eager step runs the same kernels on one stream.  All reductions have a
    fixed order, so after the same number of steps both trainers must hold identical bits."""
    from distributeddeeplearning_amd.models import network
    from distributeddeeplearning_amd.models.dcgan import GraphedGANStep
    assert network.BRANCH and network._BR is not None
    real = torch.rand(96, 28, 28, 1, generator=torch.Generator().manual_seed(3))
    a = GANTrainer(GANConfig(batch=96, seed=9), "cuda")
    b = GANTrainer(GANConfig(batch=96, seed=9), "cuda")
    a.D.branch_wgrad = a.G.branch_wgrad = True
    rp = a.D.prepare_input(real)
    step = GraphedGANStep(a, rp, warmup=2)
    assert network._BR.stream is not None
    for _ in range(4):
        step.step()
    for _ in range(6):
        b.train_step(rp, prepared=True)
    torch.cuda.synchronize()
    assert torch.equal(a.D.P, b.D.P) and torch.equal(a.G.P, b.G.P)
    assert torch.equal(a.D.S, b.D.S) and torch.equal(a.G.S, b.G.S)


def test_fused_activation_backward_is_bit_exact(monkeypatch):
    """act'(y) of the layer below applied in the input-gradient kernel's epilogue (pconv, col2im; other
    kernels fall back to an in-place pass) must give the same bits as the separate act_bwd pass."""
    from distributeddeeplearning_amd.models import network
    real = torch.rand(80, 28, 28, 1, generator=torch.Generator().manual_seed(4))
    a = GANTrainer(GANConfig(batch=80, seed=11), "cuda")
    rp = a.D.prepare_input(real)
    for _ in range(3):
        a.train_step(rp, prepared=True)
    monkeypatch.setattr(network, "FUSE_ACT_BWD", False)
    b = GANTrainer(GANConfig(batch=80, seed=11), "cuda")
    for _ in range(3):
        b.train_step(rp, prepared=True)
    torch.cuda.synchronize()
    assert torch.equal(a.D.P, b.D.P) and torch.equal(a.G.P, b.G.P)


def test_first_layer_bias_gradient_from_the_weight_gradient_launch(monkeypatch):
    """D's one-channel input conv takes its bias gradient out of the weight-gradient launch (ones column in the taps
    tensor, network.ConvOp._bias_tail): same kernel gradient bits as the separate column reduction, bias gradient
    equal to fp32 summation-order noise, nothing else in the flat gradient buffer touched."""
    from distributeddeeplearning_amd.models import network
    import distributeddeeplearning_amd.ops.hip as hip
    torch.manual_seed(0)
    x = torch.rand(96, 28, 28, 1).bfloat16().float()
    dy = hip.pad_channels((torch.randn(96, 1).abs() / 64).bfloat16().float().cuda(), 8)
    grads = []
    for fused in (True, False):
        monkeypatch.setattr(network, "BIAS_IN_WGRAD", fused)
        g = Network(discriminator_spec(), "cuda", 1, "batch")
        first = next(op for op in g.ops if isinstance(op, network.ConvOp))
        assert (first._bias_tail(25) is not None) == fused
        g.forward(g.prepare_input(x), training=True)
        g.backward(dy, need_input_grad=True)
        torch.cuda.synchronize()
        grads.append((g.G.clone(), first.gkernel.clone(), first.gbias.clone()))
    (Ga, ka, ba), (Gb, kb, bb) = grads
    assert torch.equal(ka, kb)
    assert float(bb.abs().max()) > 0
    assert float((ba - bb).abs().max()) <= 1e-4 * float(bb.abs().max()) + 1e-6
    same = Ga == Gb
    assert int((~same).sum()) <= ba.numel()          # only the bias slot may differ (in its last bits)


def test_image_layer_bias_gradient_from_the_weight_gradient_launch(monkeypatch):
    """G's one-channel transposed conv takes its scalar bias gradient out of the weight-gradient launch (row sum of
    the taps tensor's centre tap): same kernel gradient bits, bias gradient equal to summation-order noise."""
    from distributeddeeplearning_amd.models import network
    import distributeddeeplearning_amd.ops.hip as hip
    torch.manual_seed(0)
    z = (torch.rand(64, 128) * 2 - 1).bfloat16().float()
    dy = hip.pad_channels(torch.randn(64, 28, 28, 1).bfloat16().float().cuda(), 8)
    grads = []
    for fused in (True, False):
        monkeypatch.setattr(network, "BIAS_IN_WGRAD", fused)
        g = Network(generator_spec(), "cuda", 2, "batch")
        last = [op for op in g.ops if isinstance(op, network.ConvOp)][-1]
        assert last.onech == "out" and last._centre_tap() == 12
        assert (last._bias_tail(1, nbias=1) is not None) == fused
        g.forward(g.prepare_input(z), training=True)
        g.backward(dy)
        torch.cuda.synchronize()
        grads.append((g.G.clone(), last.gkernel.clone(), last.gbias.clone()))
    (Ga, ka, ba), (Gb, kb, bb) = grads
    assert torch.equal(ka, kb)
    assert float(bb.abs().max()) > 0
    assert float((ba - bb).abs().max()) <= 1e-4 * float(bb.abs().max()) + 1e-4
    # only bias slots may differ, in their last bits (the Dense layer's bias gradient takes the same route)
    nbias = sum(op.gbias.numel() for op in g.ops if isinstance(op, network.ConvOp) and op.kind == "dense") + 1
    assert int((Ga != Gb).sum()) <= nbias
    assert torch.allclose(Ga, Gb, rtol=1e-4, atol=1e-4 * float(Gb.abs().max()))


def test_generator_inference_forward_with_folded_batchnorm():
    """training=False: inference BN is folded into the producing conv's epilogue where the kernel allows
    (pconv AFFINE) -- same outputs as the CPU oracle with non-trivial moving statistics."""
    c, g = _pair(generator_spec(), 6)
    gen = torch.Generator().manual_seed(1)
    S = torch.rand(c.S.numel(), generator=gen) + 0.5
    c.S.copy_(S)
    g.S.copy_(S.cuda())
    z = (torch.rand(200, 128, generator=gen) * 2 - 1).bfloat16().float()
    yc = c.forward(c.prepare_input(z), training=False)
    yg = g.forward(g.prepare_input(z), training=False)
    assert _rel(g.to_logical(yg).cpu(), yc) < 1e-2
    assert not any(getattr(op, "applied_by_producer", False) for op in g.ops)      # flags consumed


def test_bn_on_load_matches_materialised_generator():
    """The generator's last BatchNorm is applied on load by the single-channel transposed conv (no BN output tensor):
    forward image, every gradient and the moving statistics must equal the materialised path bit for bit, for the
    plain forward and for the merged two-group forward narrowed to its second half.  With the BatchNorm's backward
    column sums coming out of the input-gradient kernel (one partial per image instead of colreduce's blocks) the
    gradients agree to summation order."""
    from distributeddeeplearning_amd.models import network as N
    from distributeddeeplearning_amd.models.dcgan import generator_spec
    torch.manual_seed(3)
    z = torch.rand(12, 128, device="cuda").bfloat16() * 2 - 1
    res = {}
    for on_load, fuse_bwd in ((True, False), (False, False), (True, True)):
        N.BN_ON_LOAD, N.BN_BWD_FUSE = on_load, fuse_bwd
        try:
            net = N.Network(generator_spec(28, 16), "cuda", seed=5)
            assert any(isinstance(op, N.BNOp) and op.defer_to is not None for op in net.ops)
            y = net.forward(z, training=True, bn_groups=2)
            net.narrow_saved(6, 12, group=1)
            net.G.zero_()
            net.backward(torch.ones_like(y[6:]) * 0.01, need_input_grad=False)
            torch.cuda.synchronize()
            res[(on_load, fuse_bwd)] = (y.clone(), net.G.clone(), net.S.clone())
        finally:
            N.BN_ON_LOAD, N.BN_BWD_FUSE = True, True
    for a, b in zip(res[(True, False)], res[(False, False)]):
        assert torch.equal(a, b)
    fy, fg, fs = res[(True, True)]
    ry, rg, rs = res[(True, False)]
    assert torch.equal(fy, ry) and torch.equal(fs, rs)
    assert float((fg - rg).norm() / rg.norm()) < 2e-3


@pytest.mark.parametrize("policy", ["split", "mixed"])
def test_fused_head_batchnorm_matches_unfused_trainer(policy, monkeypatch):
    """The discriminator head kernel applies the last BatchNorm on load and emits its backward reduction: losses
    equal the unfused path bit for bit (same bf16 activation), gradients to rounding (g is rounded once more)."""
    torch.manual_seed(5)
    real = (torch.rand(96, 28, 28, 1) > 0.6).float()
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DDL_GAN_HEAD_BN", flag)
        t = GANTrainer(GANConfig(batch=96, bn_policy=policy, seed=9), "cuda")
        assert (t._head_bn is not None) == (flag == "1")
        rp = t.D.prepare_input(real)
        t.D.G.zero_()
        t.G.G.zero_()
        dl = t.d_backward(rp)
        gD = t.D.G.clone()
        gl = t.g_backward(96)
        torch.cuda.synchronize()
        out[flag] = (float(dl), float(gl), gD, t.G.G.clone(), t.D.S.clone())
    a, b = out["1"], out["0"]
    assert a[0] == b[0] and a[1] == b[1]
    assert torch.equal(a[4], b[4])                                  # moving statistics
    for ga, gb in ((a[2], b[2]), (a[3], b[3])):
        assert float((ga - gb).norm() / gb.norm()) < 1e-2


@pytest.mark.parametrize("i", [0, 1, 2, 4])
def test_bn_on_load_matches_materialised_depthwise(i):
    """A BatchNorm in front of a depthwise layer is applied by the stencil kernels (forward and weight gradient) to
    their per-channel tap sums instead of writing its output.  The on-load path keeps the BatchNorm output in fp32
    where the materialised path rounds it to bf16: outputs, gradients and moving statistics agree to that rounding.
    Reference replicas model0/1/2/4 hold such pairs (/root/reference/DDL_Normal4.py random_layer: BatchNormalization
    after every layer)."""
    import os
    from distributeddeeplearning_amd.models import network as N
    from distributeddeeplearning_amd.utils import ckpt
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "reference_models", f"model{i}")
    if not os.path.isdir(ref):
        pytest.skip("reference checkpoints not present on this box")
    net = ckpt.load_model(ref, "cuda")
    assert any(isinstance(op, N.BNOp) and op.defer_to is not None and op.defer_to.g.depthwise for op in net.ops)
    torch.manual_seed(20 + i)
    x = net.prepare_input(torch.randint(0, 256, (24, 28, 28, 1), dtype=torch.uint8).cuda())
    s0 = net.S.clone()
    res = {}
    for on_load in (True, False):
        N.BN_ON_LOAD = on_load
        try:
            net.S.copy_(s0)
            y = net.forward(x, training=True)
            net.G.zero_()
            net.backward(torch.full_like(y, 0.01), need_input_grad=False)
            torch.cuda.synchronize()
            res[on_load] = (y.clone(), net.G.clone(), net.S.clone())
        finally:
            N.BN_ON_LOAD = True
    errs = [float((a.float() - b.float()).norm() / (b.float().norm() + 1e-6)) for a, b in zip(res[True], res[False])]
    print("ONLOAD_ERRS", i, errs)
    assert all(e < tol for e, tol in zip(errs, (3e-2, 6e-2, 1e-2))), errs


@pytest.mark.parametrize("i", [0, 2])          # the replicas with a stride-2 depthwise layer (few live taps per pixel)
def test_depthwise_bn_backward_fusion_matches_three_pass_form(i):
    """models/network.py `_dw_bn_bwd`: whole-network gradients of the reference replicas with a BatchNorm -> depthwise
    pair, fused (depthwise dgrad + BatchNorm backward in two recomputing passes) vs unfused (dgrad, column sums, apply).
    Same arithmetic per element; only the order of the column sums differs."""
    import os
    from distributeddeeplearning_amd.models import network as N
    from distributeddeeplearning_amd.utils import ckpt
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "reference_models", f"model{i}")
    if not os.path.isdir(ref):
        pytest.skip("reference checkpoints not present on this box")
    net = ckpt.load_model(ref, "cuda")
    torch.manual_seed(40 + i)
    x = net.prepare_input(torch.randint(0, 256, (24, 28, 28, 1), dtype=torch.uint8).cuda())
    s0 = net.S.clone()
    res, used = {}, {}
    real = N.ConvOp._dw_bn_bwd
    for fused in (True, False):
        N.DW_BN_FUSE = fused
        hits = []

        def spy(self, dy, need_dw, _real=real, _hits=hits):
            out = _real(self, dy, need_dw)
            _hits.append(out is not None)
            return out
        N.ConvOp._dw_bn_bwd = spy
        try:
            net.S.copy_(s0)
            y = net.forward(x, training=True)
            net.G.zero_()
            net.backward(torch.full_like(y, 0.01), need_input_grad=False)
            torch.cuda.synchronize()
            res[fused], used[fused] = net.G.clone(), list(hits)
        finally:
            N.DW_BN_FUSE = True
            N.ConvOp._dw_bn_bwd = real
    assert any(used[True]) and not any(used[False])
    err = float((res[True] - res[False]).norm() / (res[False].norm() + 1e-12))
    assert err < 2e-2, err


def test_streaming_one_channel_weight_gradients_match_the_taps_path(monkeypatch):
    """The streaming 5x5 weight gradient of the DCGAN pair's one-channel ends (network.S1_WGRAD_MIN_ELEMS forced to 0:
    by default it serves only operands of 2^27 elements and more) against the taps-tensor + split-K GEMM path: D's first
    conv with its bias gradient, G's last transposed conv with the BatchNorm in front of it on load and its scalar bias
    gradient; nothing else in the flat gradient buffers changes."""
    from distributeddeeplearning_amd.models import network
    import distributeddeeplearning_amd.ops.hip as hip
    torch.manual_seed(0)
    x = torch.rand(96, 28, 28, 1).bfloat16().float()
    dyD = hip.pad_channels((torch.randn(96, 1).abs() / 64).bfloat16().float().cuda(), 8)
    z = (torch.rand(64, 128) * 2 - 1).bfloat16().float()
    dyG = hip.pad_channels(torch.randn(64, 28, 28, 1).bfloat16().float().cuda(), 8)
    calls = []
    real = hip.stream1_wgrad
    monkeypatch.setattr(hip, "stream1_wgrad", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    res = {}
    for lim in (0, 1 << 40):
        monkeypatch.setattr(network, "S1_WGRAD_MIN_ELEMS", lim)
        n0 = len(calls)
        d = Network(discriminator_spec(), "cuda", 1, "batch")
        d.forward(d.prepare_input(x), training=True)
        d.backward(dyD, need_input_grad=True)
        g = Network(generator_spec(), "cuda", 2, "batch")
        g.forward(g.prepare_input(z), training=True)
        g.backward(dyG)
        torch.cuda.synchronize()
        assert (len(calls) - n0) == (2 if lim == 0 else 0)
        res[lim] = (d.G.clone(), g.G.clone())
    for a, b in zip(res[0], res[1 << 40]):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()))
        assert float((a - b).abs().max()) > 0 or torch.equal(a, b)
