"""Does the DCGAN actually play the adversarial game, and does the GPU play the same game as the fp32 oracle?

Round 1 shipped a default mode whose two losses both collapsed to ~1e-4: the discriminator normalised
[real ; fake] as ONE batch in the D phase and a fake-only batch in the G phase, and the D-phase fakes
came from an inference-phase generator -- D told real from fake by the batch statistics, and a
fake-only batch read as "real".  These tests pin the fix (models/dcgan.py `bn_policy`):

  * the leak signature (mean logit of the fake half in the D phase vs of the fake batch in the G phase)
    is absent in "split" / "inference" and present in "mixed" -- CPU oracle, seconds;
  * on the GPU, 20 iterations of the reference-exact mode and of the default mode reproduce the CPU
    oracle's d_loss / g_loss sequence on identical weights and noise;
  * on the GPU, the default mode moves generated samples toward the data (nearest-neighbour distance to
    the training glyphs drops, samples stay diverse), where the round-1 mode collapses.

The reference's own evidence is ten recognisable digits (/root/reference/images/*.jpg written by
/root/reference/GAN_distributed.py:119-124,133-142); real-MNIST parity stays unpinned (no dataset on the
box): the data here are the synthetic glyphs of utils/data.py.
"""
import pytest
import torch

from distributeddeeplearning_amd.cli.gan import nn_distance
from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer, GraphedGANStep
from distributeddeeplearning_amd.utils.data import Data


def _glyphs(n, device="cpu"):
    d = Data(device, channels=1, n_train=n, n_test=16, synthetic=True, seed=3)
    return d.train_x[:n].float() / 255.0


# ------------------------------------------------------------------------------------ CPU oracle
@pytest.mark.parametrize("policy,leaks", [("split", False), ("inference", False), ("mixed", True)])
def test_bn_policy_leak_signature_cpu(policy, leaks):
    real = _glyphs(32)
    t = GANTrainer(GANConfig(batch=32, base=8, bn_policy=policy, lr=1e-3, seed=1), "cpu")
    rp = t.D.prepare_input(real)
    for _ in range(40):
        t.train_step(rp, prepared=True)
    d = t.diagnose(rp)
    gap = d["logit_gphase"] - d["logit_fake"]
    if leaks:
        # D rejects the fake half of a mixed batch and accepts the very same kind of images as a batch of their own
        assert gap > 4.0 and d["logit_fake"] < -2.0 and d["logit_gphase"] > 0.0, d
    else:
        assert abs(gap) < 1.5, d
        assert d["logit_real"] > d["logit_fake"], d


def test_diagnose_leaves_training_state_untouched_cpu():
    real = _glyphs(16)
    a = GANTrainer(GANConfig(batch=16, base=8, seed=2), "cpu")
    b = GANTrainer(GANConfig(batch=16, base=8, seed=2), "cpu")
    rp = a.D.prepare_input(real)
    for i in range(4):
        a.train_step(rp, prepared=True)
        b.train_step(rp, prepared=True)
        a.diagnose(rp)
    assert torch.equal(a.D.P, b.D.P) and torch.equal(a.G.P, b.G.P)
    assert torch.equal(a.D.S, b.D.S) and torch.equal(a.G.S, b.G.S)


def test_reference_exact_forces_inference_bn():
    cfg = GANConfig(reference_exact=True, bn_policy="split")
    assert cfg.bn_policy == "inference"
    with pytest.raises(ValueError):
        GANConfig(bn_policy="bogus")


# ------------------------------------------------------------------------------------ GPU vs oracle
class _FixedNoise:
    """Replays a list of pre-drawn noise batches (bf16-representable) on any device."""

    def __init__(self, batches, device, dtype):
        self.batches, self.i, self.device, self.dtype = batches, 0, device, dtype

    def uniform(self, shape, dtype, lo=-1.0, hi=1.0):
        z = self.batches[self.i]
        self.i += 1
        if shape[0] == 2 * z.shape[0]:      # the device trainer draws the D- and G-phase noise in one call
            z = torch.cat([z, self.batches[self.i]])
            self.i += 1
        assert tuple(z.shape) == tuple(shape)
        return z.to(self.device).to(self.dtype)

    def state_dict(self):
        return {"i": self.i}

    def load_state_dict(self, sd):
        self.i = sd["i"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["reference_exact", "split"])
def test_loss_sequence_matches_cpu_oracle_20_iterations(mode):
    """same weights, same noise, 20 iterations; the bf16 device trainer must trace the fp32
    oracle's loss curve.  Both optimizers start as sign descent (fresh Adam: m / sqrt(v) = sign(g)), so a
    gradient whose sign differs in bf16 moves its weight the other way: the curves drift apart slowly and
    the tolerance grows with the iteration."""
    B, n_it = 32, 20
    cfg = GANConfig(batch=B, seed=4, reference_exact=(mode == "reference_exact"))
    c, g = GANTrainer(cfg, "cpu"), GANTrainer(cfg, "cuda")
    for nc, ng in ((c.G, g.G), (c.D, g.D)):
        st = {k: v.bfloat16().float() for k, v in nc.logical_state().items()}
        nc.load_logical_state(st)
        ng.load_logical_state(st)
    gen = torch.Generator().manual_seed(0)
    noise = [(torch.rand(B, 128, generator=gen) * 2 - 1).bfloat16().float() for _ in range(2 * n_it)]
    c.rng = _FixedNoise(noise, "cpu", torch.float32)
    g.rng = _FixedNoise(noise, "cuda", torch.bfloat16)
    real = _glyphs(B).bfloat16().float()
    rc, rg = c.D.prepare_input(real), g.D.prepare_input(real)
    seq_c, seq_g = [], []
    for _ in range(n_it):
        seq_c.append([float(v) for v in c.train_step(rc, prepared=True)])
        seq_g.append([float(v) for v in g.train_step(rg, prepared=True)])
    for i, (lc, lg) in enumerate(zip(seq_c, seq_g)):
        tol = 0.02 + 0.004 * i
        for k in range(2):
            assert abs(lc[k] - lg[k]) <= tol + 0.05 * abs(lc[k]), (mode, i, seq_c, seq_g)
    # the curve did something: the discriminator's loss moved away from ln 2 * 2-ish start
    assert abs(seq_c[-1][0] - seq_c[0][0]) > 0.05


@pytest.mark.gpu
def test_default_mode_learns_glyphs_and_round1_mode_collapses(monkeypatch):
    """1500 graphed iterations on 500 synthetic glyphs.  Default ("split"): samples move toward the data and stay
    diverse, the discriminator sees a fake batch the same way in both phases.  Round-1 mode ("mixed"): the leak
    signature, both losses near zero, samples collapse onto one image.  Thresholds from two seeds measured on
    MI355X: split nn 0.42 -> 0.26 / diversity 0.21-0.25; mixed nn 0.42-0.47 / diversity 0.04-0.06."""
    # 1500 iterations of a chaotic game amplify the last-bit differences between kernel plans: run on the
    # heuristic default plans (no timing-dependent choices), so that every run of this test plays the same game
    from distributeddeeplearning_amd.ops import hip
    monkeypatch.setattr(hip, "AUTOTUNE", False)
    monkeypatch.setattr(hip, "_CONV_CHOICE", {})
    monkeypatch.setattr(hip, "_CHOICE_SPEC", {})
    real = _glyphs(500, "cuda")
    out = {}
    for policy in ("split", "mixed"):
        t = GANTrainer(GANConfig(batch=500, seed=5, bn_policy=policy), "cuda")
        rp = t.D.prepare_input(real)
        first = nn_distance(t.generate(128), real)
        step = GraphedGANStep(t, rp, warmup=3)
        for _ in range(1500):
            d_loss, g_loss = step.step()
        torch.cuda.synchronize()
        s = t.generate(128, training_bn=True)
        out[policy] = dict(first=first, last=nn_distance(s, real), div=float(s.std(0).mean()),
                           d_loss=float(d_loss), g_loss=float(g_loss), **t.diagnose(rp))
    sp, mx = out["split"], out["mixed"]
    # generated samples moved toward the data.  (Autotuned kernel plans differ from run to run, so 1500 iterations
    # of a chaotic game do not land on one value: 0.74-0.78 x first seen; an untrained / collapsed G stays >= 0.95.)
    assert sp["last"] < 0.85 * sp["first"], out
    assert sp["div"] > 0.12, out                                 # ... and did not collapse onto one image
    assert abs(sp["logit_gphase"] - sp["logit_fake"]) < 1.5, out
    assert sp["g_loss"] > 0.1, out                               # the game is still being played
    assert mx["logit_gphase"] - mx["logit_fake"] > 4.0, out      # round-1 failure, kept as a detector check
    assert mx["d_loss"] < 0.02 and mx["g_loss"] < 0.2, out
    # (collapse measured as diversity 0.04-0.10 against 0.21-0.26 in the default mode: compare the two, the absolute
    # value of a collapsed run depends on which kernel plans the autotuner picked)
    assert mx["div"] < 0.6 * sp["div"] and mx["div"] < 0.15 and mx["last"] > sp["last"] + 0.08, out


@pytest.mark.gpu
def test_graphed_resume_is_bitwise_identical_to_uninterrupted_run(tmp_path):
    """the checkpoint must carry the DEVICE step counter, the Philox offset and the BN statistics;
    graphed K steps + save + load + graphed K steps == 2K uninterrupted steps, bit for bit."""
    real = _glyphs(64, "cuda")
    K = 8

    def run(n, trainer):
        rp = trainer.D.prepare_input(real)
        step = GraphedGANStep(trainer, rp, warmup=3)          # three real steps
        for _ in range(n - 3):
            step.step()
        torch.cuda.synchronize()

    a = GANTrainer(GANConfig(batch=64, seed=6), "cuda")
    run(2 * K, a)
    b = GANTrainer(GANConfig(batch=64, seed=6), "cuda")
    run(K, b)
    assert b.optG.steps_taken() == K and b.optD.steps_taken() == K
    b.save_checkpoint(str(tmp_path / "gan.ckpt"), K)
    c = GANTrainer(GANConfig(batch=64, seed=99), "cuda")      # different init: everything must come from the file
    assert c.load_checkpoint(str(tmp_path / "gan.ckpt")) == K
    run(K, c)
    for x, y in ((a.G.P, c.G.P), (a.D.P, c.D.P), (a.G.S, c.G.S), (a.D.S, c.D.S), (a.optG.m, c.optG.m),
                 (a.optD.v, c.optD.v)):
        assert torch.equal(x, y)
    assert c.optG.steps_taken() == 2 * K


@pytest.mark.gpu
def test_discriminator_follows_a_new_real_batch_eager_and_graphed():
    """the merged generator pass reads the real images from its own [real | G(z1) | G(z2)] buffer.  A fresh
    minibatch must reach it on EVERY step -- a `prepare_input` result can reuse the address of the previous one, and a
    graph replay must see what `set_real` put into the static slot.  Oracle: a second trainer that only ever saw the
    second batch differs from one that saw batch A then batch B, but a trainer fed A, A, B must equal... itself re-run
    with the same feed; and the d_loss of a step on all-black images differs from one on glyphs."""
    glyph = _glyphs(64, "cuda")
    black = torch.zeros_like(glyph)

    def eager(feed):
        t = GANTrainer(GANConfig(batch=64, seed=11), "cuda")
        assert t.merged
        out = []
        for x in feed:
            d, _ = t.train_step(x)                        # unprepared input: a fresh prepared tensor per step
            out.append(float(d))
        return t, out

    ta, la = eager([glyph, glyph, black, black])
    tb, lb = eager([glyph, glyph, glyph, glyph])
    assert la[:2] == lb[:2]
    assert la[2] != lb[2] and not torch.equal(ta.D.P, tb.D.P)        # the third step saw different real images

    def graphed(feed):
        t = GANTrainer(GANConfig(batch=64, seed=11), "cuda")
        step = GraphedGANStep(t, t.D.prepare_input(feed[0]), warmup=1)
        out = []
        for x in feed[1:]:
            step.set_real(t.D.prepare_input(x))
            d, _ = step.step()
            out.append(float(d))
        torch.cuda.synchronize()
        return t, out

    tc, lc = graphed([glyph, glyph, black, black])
    assert lc == la[1:]                                   # replay == eager, bit for bit, on a changing feed
    assert torch.equal(tc.D.P, ta.D.P) and torch.equal(tc.G.P, ta.G.P)
