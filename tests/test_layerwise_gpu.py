"""Layer-by-layer error attribution of the device executor against the CPU oracle.

Two kinds of check, for both GAN networks and the five imported reference replicas:

* LOCAL: every device op (a "unit": an op with the reshapes in front of it; a BatchNorm whose apply is deferred to
  the next conv is one unit with that conv) gets the ORACLE's bf16-rounded input and upstream gradient, so its error
  is its own and not what the layers before it accumulated.  Asserted per unit: forward rel-L2 <= 1 %, input
  gradient and every parameter gradient (kernels, biases, gamma, beta) <= 3 %.
* END TO END against a bf16-EMULATING oracle (`Network.emulate_bf16`: the fp32 executor rounds every op's output
  and input gradient to bf16 where the device stores bf16): what is left is fp32 summation order plus the odd
  1-ulp rounding flip, so the same 1 % / 3 % bounds hold for the whole chain; per-channel vectors (bias, gamma,
  beta) get 6 % there instead of round 1's 15 %.  Those are sums of mixed-sign terms over batch x pixels: an
  elementwise error eps of the incoming gradient shows up as eps x (typical / actual size of a random-walk sum),
  a factor of a few that no oracle removes.  The upstream gradient fed to the LAST layer is positive, so that the
  one-channel bias sums there (G's image layer, D's logit) do not cancel to a number whose relative error is
  meaningless."""
import os

import pytest
import torch

from distributeddeeplearning_amd.models import network as N
from distributeddeeplearning_amd.models.dcgan import discriminator_spec, generator_spec
from distributeddeeplearning_amd.utils import ckpt

pytestmark = pytest.mark.gpu
REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "reference_models")
FWD_TOL, GRAD_TOL, VEC_TOL_E2E = 1e-2, 3e-2, 6e-2


def _rbf(t):
    return t.bfloat16().float()


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def _share_weights(c, g):
    st = {k: _rbf(v) for k, v in c.logical_state().items()}
    c.load_logical_state(st)
    g.load_logical_state(st)


def _units(net):
    """Index lists: reshapes ride with the op behind them, a deferred BatchNorm with its consumer."""
    units, cur = [], []
    for i, op in enumerate(net.ops):
        cur.append(i)
        if isinstance(op, N.ViewOp):
            continue
        if isinstance(op, N.BNOp) and op.defer_to is not None:
            continue
        units.append(cur)
        cur = []
    if cur:
        units[-1].extend(cur)
    return units


def _to_dev(g, t):
    import distributeddeeplearning_amd.ops.hip as hip
    return hip.pad_channels(t.cuda().contiguous(), hip.pad8(t.shape[-1]))


def _grads(net):
    out = {}
    for p in net.param_table():
        if p.trainable:
            v = net.G[p.offset:p.offset + p.numel].view(p.shape)[p.corner()]
            out[(p.layer, p.wname)] = v.detach().float().cpu().clone()
    return out


def _run_local(c, g, x, dy_scale=1.0):
    """-> [(unit name, forward err, dx err)], {param: err}.  The oracle runs as a chain (rounding to bf16 after each
    op); each device unit is run on the oracle's tensors."""
    units = _units(g)
    assert [len(u) for u in units] == [len(u) for u in _units(g)]
    N._BR.enabled = False
    # units run on the ORACLE's tensors one at a time: a BatchNorm must not leave its backward apply step to the unit
    # below it (network.BNOp.defer_apply_to; covered end to end by tests/test_hip_ops.py and tests/test_network_gpu.py)
    for op in g.ops:
        if isinstance(op, N.BNOp):
            op.below_needs_dx = True
    # oracle forward, unit by unit
    xs, t = [], x
    for u in units:
        xs.append(t)
        for i in u:
            t = c.ops[i].fwd(t, True)
            if not isinstance(c.ops[i], N.ViewOp):
                t = _rbf(t)
    xs.append(t)
    torch.manual_seed(1)
    dy = _rbf(torch.randn_like(t).abs() * dy_scale)
    c.G.zero_()
    g.G.zero_()
    report = []
    first_param = next(k for k, u in enumerate(units) if any(g.ops[i].has_params for i in u))
    for k in range(len(units) - 1, -1, -1):
        u = units[k]
        # device unit on the oracle's input
        xg = _to_dev(g, xs[k])
        for i in u:
            if isinstance(g.ops[i], N.BNOp):
                g.ops[i].pre_stats = None
                g.ops[i].groups = 1
            xg = g.ops[i].fwd(xg, True)
        fe = _rel(g.to_logical(xg, xs[k + 1].shape[-1]).cpu(), xs[k + 1])
        # backward of the unit: oracle and device from the same upstream gradient
        need_dx = k > first_param
        dyc, dyg = dy, _to_dev(g, dy)
        for i in reversed(u):
            dyc = c.ops[i].bwd(dyc, need_dx=need_dx, need_dw=True)
            dyg = g.ops[i].bwd(dyg, need_dx=need_dx, need_dw=True)
            if dyc is not None and not isinstance(c.ops[i], N.ViewOp):
                dyc = _rbf(dyc)
        N._BR.join()
        de = None
        if need_dx:
            de = _rel(g.to_logical(dyg, dyc.shape[-1]).cpu(), dyc)
            dy = dyc
        report.append((g.ops[u[-1]].name, fe, de))
    torch.cuda.synchronize()
    gc, gg = _grads(c), _grads(g)
    perr = {}
    for key, ref in gc.items():
        if float(ref.norm()) < 1e-6 * max(1.0, float(ref.numel()) ** 0.5):
            assert float(gg[key].norm()) < 1e-2, key           # exactly-zero gradient (bias before batch-stat BN)
            continue
        perr[key] = _rel(gg[key].reshape(ref.shape), ref)
    return report[::-1], perr


def _check(report, perr, tag):
    lines = [f"{n}: fwd {fe:.4f}" + ("" if de is None else f" dx {de:.4f}") for n, fe, de in report]
    lines += [f"{k}: {v:.4f}" for k, v in perr.items()]
    msg = tag + "\n" + "\n".join(lines)
    for n, fe, de in report:
        assert fe <= FWD_TOL, msg
        assert de is None or de <= GRAD_TOL, msg
    for k, v in perr.items():
        assert v <= GRAD_TOL, msg


def _gan_pair(which):
    spec = discriminator_spec() if which == "D" else generator_spec()
    c, g = N.Network(spec, "cpu", 3), N.Network(spec, "cuda", 3)
    _share_weights(c, g)
    torch.manual_seed(0)
    if which == "D":
        x = _rbf(torch.rand(48, 28, 28, 1))
    else:
        x = _rbf(torch.rand(48, 128) * 2 - 1)
    return c, g, x


@pytest.mark.parametrize("which", ["D", "G"])
def test_gan_networks_layer_by_layer(which):
    c, g, x = _gan_pair(which)
    report, perr = _run_local(c, g, x, dy_scale=1.0 if which == "G" else 1 / 48)
    _check(report, perr, which)


@pytest.mark.parametrize("i", range(5))
def test_reference_replicas_layer_by_layer(i):
    if not os.path.isdir(f"{REF}/model{i}"):
        pytest.skip("reference checkpoints not present on this box")
    c = ckpt.load_model(f"{REF}/model{i}", "cpu")
    g = ckpt.load_model(f"{REF}/model{i}", "cuda")
    _share_weights(c, g)
    torch.manual_seed(i)
    x = c.prepare_input(torch.randint(0, 256, (16, 28, 28, 1), dtype=torch.uint8))
    report, perr = _run_local(c, g, x, dy_scale=1 / 16)
    _check(report, perr, f"model{i}")


def _end_to_end(c, g, x, dy_scale):
    c.emulate_bf16 = True
    yc = c.forward(x, training=True)
    yg = g.forward(_to_dev(g, x), training=True)
    fe = _rel(g.to_logical(yg, yc.shape[-1]).cpu(), yc)
    torch.manual_seed(2)
    dy = _rbf(torch.randn_like(yc).abs() * dy_scale)
    c.G.zero_()
    g.G.zero_()
    dxc = c.backward(dy, need_input_grad=True)
    dxg = g.backward(_to_dev(g, dy), need_input_grad=True)
    torch.cuda.synchronize()
    de = _rel(g.to_logical(dxg, dxc.shape[-1]).cpu(), dxc)
    gc, gg = _grads(c), _grads(g)
    perr = {}
    for key, ref in gc.items():
        if float(ref.norm()) < 1e-6 * max(1.0, float(ref.numel()) ** 0.5):
            continue
        err = float((gg[key].reshape(ref.shape) - ref).norm())
        scale = float(ref.norm())
        if key[1] == "beta":
            # d beta = sum g and d gamma = sum g * xhat run over the same terms; where the layer behind the BN
            # feeds another batch-statistics BN, sum g cancels to almost nothing (model1's batch_normalization_6:
            # |d beta| ~ 1e-2 |d gamma|) and only d gamma still shows the size of what was summed
            scale = max(scale, float(gc[(key[0], "gamma")].norm()))
        perr[key] = err / (scale + 1e-12)
    return fe, de, perr


@pytest.mark.parametrize("which", ["D", "G"])
def test_gan_networks_end_to_end_vs_bf16_emulating_oracle(which):
    c, g, x = _gan_pair(which)
    fe, de, perr = _end_to_end(c, g, x, 1.0 if which == "G" else 1 / 48)
    msg = f"{which}: fwd {fe:.4f} dx {de:.4f} " + " ".join(f"{k[0]}.{k[1]}={v:.4f}" for k, v in perr.items())
    assert fe <= FWD_TOL and de <= GRAD_TOL, msg
    assert all(v <= (GRAD_TOL if "kernel" in k[1] else VEC_TOL_E2E) for k, v in perr.items()), msg


@pytest.mark.parametrize("i", range(5))
def test_reference_replicas_end_to_end_vs_bf16_emulating_oracle(i):
    if not os.path.isdir(f"{REF}/model{i}"):
        pytest.skip("reference checkpoints not present on this box")
    c = ckpt.load_model(f"{REF}/model{i}", "cpu")
    g = ckpt.load_model(f"{REF}/model{i}", "cuda")
    _share_weights(c, g)
    torch.manual_seed(10 + i)
    x = c.prepare_input(torch.randint(0, 256, (16, 28, 28, 1), dtype=torch.uint8))
    fe, de, perr = _end_to_end(c, g, x, 1 / 16)
    msg = f"model{i}: fwd {fe:.4f} dx {de:.4f} " + " ".join(f"{k[0]}.{k[1]}={v:.4f}" for k, v in perr.items())
    assert fe <= FWD_TOL and de <= GRAD_TOL, msg
    assert all(v <= (GRAD_TOL if "kernel" in k[1] else VEC_TOL_E2E) for k, v in perr.items()), msg
