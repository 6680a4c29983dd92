"""The "explained" cases of the random-shape sweep (profiles/r03_fuzz.md: 195 / 200 agree, five cases explained by the builder and
re-checked by nobody -- VERDICT r3) as regression tests that assert the EXPLANATION: wherever a case sits outside the sweep's bar,
the fp64 oracle itself must move by a comparable amount when its weights move by ONE fp32 ulp (ill-conditioned net / chaotic
bf16 roundings / a one-point batch), and the kernel must stay within a small multiple of that movement.  If a kernel change breaks
one of these nets for real, the oracle's own sensitivity does not grow with it and the test fails."""
import os
import sys

import numpy as np
import pytest

from oracle import nif_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWEEP_SEED = 3


def _case(i):
    """configuration i of `python tools/fuzz_parity.py N 3` (the draws are sequential)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as F
    rng = np.random.default_rng(SWEEP_SEED)
    for _ in range(i + 1):
        cfg, B, desc = F.draw(rng, wide=False)
    return cfg, B, desc, SWEEP_SEED * 1000 + i


def _setup(i, policy="float32"):
    import nif_amd
    (kind, cs, cp), B, desc, seed = _case(i)
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(seed)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    names = [nm for nm, _ in spec.param_shapes()]
    if kind == "NIFMultiScale":
        ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 2.0).astype(np.float32)
    if kind == "NIFMultiScaleLastLayerParameterized":
        ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 30.0).astype(np.float32)
    m = getattr(nif_amd, kind)(cs, cp, mixed_policy=policy)
    model = m.build(); model.set_weights(ws)
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32)
    return m, model, spec, ws, x, y, sw, desc


def _one_ulp(ws, seed=7):
    rng = np.random.default_rng(seed)
    return [np.nextafter(w.astype(np.float32), (np.float32(np.inf) * rng.choice([-1.0, 1.0], size=w.shape)).astype(np.float32))
            .astype(np.float64) for w in ws]


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("case,what", [(52, "128 x 6 resblocks = 12 sine layers, hyper-weights x 2"), (105, "64 x 6, ParameterNet 128 x 4 tanh, latent 5")])
def test_ill_conditioned_nets_move_with_one_ulp_of_weight_noise(case, what):
    """sweep cases 52 (forward 9 % off) and 105 (7e-5 off): 'conditioning'.  Asserted: the fp64 oracle's own prediction moves by s
    under one-ulp weight noise, s is far above the 1e-5 bar, and the kernel sits within 3 s (+ the bar) of the oracle."""
    m, model, spec, ws, x, y, sw, desc = _setup(case)
    ws64 = [w.astype(np.float64) for w in ws]
    x64 = x.astype(np.float64)
    ref = O.forward(spec, ws64, x64)
    s = max(_rel(O.forward(spec, _one_ulp(ws, sd), x64), ref) for sd in (7, 8, 9))
    err = _rel(model.predict(x), ref)
    assert s > 2e-5, (desc, s)                       # the explanation holds: the net amplifies 6e-8 to more than the bar
    assert err < 3.0 * s + 1e-5, (desc, err, s)


def test_policy_rounding_flips_on_a_31_point_batch():
    """sweep case 98 (96 x 5 resblocks, B = 31, mixed_bfloat16: gradient 7.5e-3 from the emulation, bar 5e-3): 'rounding flips on 31
    points'.  Asserted: the emulating oracle's gradient moves by s under one-ulp weight noise (a different set of bf16 roundings
    flips) and the kernel is within max(5e-3, 3 s)."""
    m, model, spec, ws, x, y, sw, desc = _setup(98, "mixed_bfloat16")
    ws64 = [w.astype(np.float64) for w in ws]
    x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
    nb = (spec.n + 15) // 16
    ll = spec.kind == O.KIND_LL
    fn = O.ll_policy_loss_and_grad if ll else O.planes_loss_and_grad
    stash = (nb in (2, 4, 8)) if ll else (nb in (2, 4) or (nb == 8 and spec.r <= 1))
    rl, rg, _ = fn(spec, ws64, x64, y64, s64, rnd=O.bf16_round, stash_bf16=stash)
    s = max(_rel(O.flatten(fn(spec, _one_ulp(ws, sd), x64, y64, s64, rnd=O.bf16_round, stash_bf16=stash)[1]), O.flatten(rg)) for sd in (7, 8, 9))
    lb, gb = m._engine.loss_and_grad(x, y, sw)
    err = _rel(gb, O.flatten(rg))
    assert abs(lb - rl) < 1e-3 * abs(rl) + 3.0 * s * abs(rl), (desc, lb, rl)
    assert err < max(5e-3, 3.0 * s), (desc, err, s)
    assert err < 5e-2


def test_one_point_batches_are_judged_on_the_prediction():
    """sweep case 104 (B = 1: loss 2.7409e-5 vs 2.7406e-5 = 1.1e-4 relative) and case 189 (100 x 2 resblocks, B = 1: third loss of a
    three-step fit 5.04 % off after the loss jumped 260 x): one point.  Asserted: the PREDICTION agrees to the absolute fp32 level of
    an O(1) field; the loss e^2 of one point inherits 2 |du| / |e| (the residual e is small), which is what the sweep saw; and the
    fit trajectory of the one-point net moves by more than the sweep's 5 % bar when the oracle's weights move by one ulp."""
    import nif_amd
    for case in (104, 189):
        m, model, spec, ws, x, y, sw, desc = _setup(case)
        assert x.shape[0] == 1, desc
        ws64 = [w.astype(np.float64) for w in ws]
        x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
        u, ref = model.predict(x), O.forward(spec, ws64, x64)
        du = float(np.abs(u - ref).max())
        assert du < 2e-6 * max(1.0, float(np.abs(ref).max())), (desc, du)
        loss, _ = m._engine.loss_and_grad(x, y, sw)
        rl, _ = O.loss_and_grad(spec, ws64, x64, y64, s64)
        e = float(np.abs(ref - y64).max())
        assert abs(loss - rl) <= (2.0 * (du + 1e-7) / max(e, 1e-12) + 2e-5) * abs(rl), (desc, loss, rl, du, e)

    # case 189: the sweep's explanation ("one point") made precise.  One-ulp WEIGHT noise does not explain it (the oracle's trajectory
    # moves 4e-4 under it -- r3's explanation was never checked); what does: Adam's normalised step lr g / (|g| + eps) turns the fp32
    # error of the many near-zero gradient entries of a one-point batch into +-lr moves.  Asserted: (a) the oracle's own trajectory
    # moves by s when its gradients are perturbed INSIDE the gradient parity bar (2e-4 of each tensor's norm), and the free-running
    # fit is within 3 s; (b) step by step from the GPU's own state (teacher forced) every step is the oracle's to 1 % of lr wherever
    # the gradient entry is solid.
    m, model, spec, ws, x, y, sw, desc = _setup(189)
    x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
    f32 = lambda a: float(np.float32(a))
    LR = 1e-4

    def traj(w0, noise_seed=None):
        rng = np.random.default_rng(noise_seed)
        th = O.flatten(w0); mm = np.zeros_like(th); vv = np.zeros_like(th); ls = []
        for t in range(1, 4):
            l_, g_ = O.loss_and_grad(spec, O.unflatten(spec, th), x64, y64, s64)
            ls.append(l_)
            if noise_seed is not None:
                g_ = [gt + 2e-4 * np.linalg.norm(gt) / np.sqrt(gt.size) * rng.standard_normal(gt.shape) for gt in g_]
            th, mm, vv = O.adam_step(th, O.flatten(g_), mm, vv, t, lr=f32(LR), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
        return np.array(ls)
    ws64 = [w.astype(np.float64) for w in ws]
    base = traj(ws64)
    s = max(float(np.abs(traj(ws64, sd) / base - 1.0).max()) for sd in (7, 8, 9))
    model.compile(nif_amd.Adam(LR), "mse")
    h = model.fit(x, y, epochs=3, batch_size=1, shuffle=False, verbose=0, sample_weight=sw)
    err = float(np.abs(np.array(h.history["loss"]) / base - 1.0).max())
    assert abs(h.history["loss"][0] / base[0] - 1.0) < 1e-3, (h.history["loss"], base)       # before any step: the loss itself
    assert err < max(2e-3, 3.0 * s), (desc, err, s, h.history["loss"], base)
    # (b) teacher forced
    model.set_weights(ws)
    model.compile(nif_amd.Adam(LR), "mse")
    e = m._engine
    for t in range(1, 4):
        th0 = O.flatten(model.get_weights()).astype(np.float64)
        m0, v0, _ = e.get_opt_state() if t > 1 else (np.zeros_like(th0), np.zeros_like(th0), 0)
        hh = model.fit(x, y, epochs=1, batch_size=1, shuffle=False, verbose=0, sample_weight=sw)
        l_, g_ = O.loss_and_grad(spec, O.unflatten(spec, th0), x64, y64, s64)
        assert abs(hh.history["loss"][0] - l_) <= 1e-4 * abs(l_) + 1e-9, (t, hh.history["loss"][0], l_)
        th1, _, _ = O.adam_step(th0, O.flatten(g_), np.asarray(m0, np.float64), np.asarray(v0, np.float64), t, lr=f32(LR), b1=f32(0.9),
                                b2=f32(0.999), eps=f32(1e-7))
        solid = np.concatenate([(np.abs(gt.ravel()) > 0.02 * (np.sqrt(np.mean(gt ** 2)) + 1e-300)) for gt in g_])
        d = np.abs(O.flatten(model.get_weights()) - th1)
        assert d[solid].max() < 0.02 * LR, (t, d[solid].max() / LR)


def test_sobolev_step_takes_fewer_columns_per_pass_when_three_do_not_fit():
    """sweep r04 seed 7 case 5 (NIFMultiScale 128 x 6, latent 8, 3 coordinates + 4 parameters) was the round's only refusal: the Sobolev
    step over all seven input columns needs three parameter-column streams' per-wave LDS next to a 128-wide plane (165 KB).  The passes
    now take two columns (or one) where three do not fit (nif_api.hip sob_cols_per_pass): the step and the two-output predictions of
    that shape agree with the oracle like any other"""
    import nif_amd
    from tests.test_gpu_parity import _cfg, _per_tensor_rel
    kind, cs, cp = _cfg("NIFMultiScale", 128, 6, 16, 3, 8, 3, 3, 4, s_res=False, p_act="swish", p_res=True)
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(7005)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    B = 130
    m = getattr(nif_amd, kind)(cs, cp)
    model = m.build(); model.set_weights(ws)
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32)
    xi = [int(v) for v in rng.permutation(spec.pi + spec.si)]
    g = rng.uniform(-1, 1, size=(B, spec.so, len(xi))).astype(np.float32)
    ws64 = [w.astype(np.float64) for w in ws]
    sl, sg = m._engine.sobolev_loss_and_grad(x, y, g, xi, 0.05, sw)
    rl, rg, ru, rJ = O.sobolev_loss_and_grad(spec, ws64, x.astype(np.float64), y.astype(np.float64), g.astype(np.float64), xi, 0.05,
                                             sw.astype(np.float64))
    assert abs(sl - rl) <= 2e-5 * abs(rl), (sl, rl)
    rel = _per_tensor_rel(spec, sg, O.flatten(rg))
    assert max(rel.values()) < 4e-4, rel
    u, J = m._engine.sobolev_forward(x, xi)
    assert _rel(u, ru) < 1e-5 and _rel(J.reshape(rJ.shape), rJ) < 3e-5


@pytest.mark.parametrize("B", [1, 2, 17, 32])
def test_r6_sweep_case_16_snet6_ring_inside_its_allocation(B):
    """`tools/fuzz_parity.py 200 606`, case 16 (r6): NIFMultiScale 56 x 3 (the fused-gradient kernel k_snet6) with a ParameterNet of three
    hidden matrices (the HBM-stash adjoint) at a batch of <= 32 points: k_snet6's private `h` ring (every wave of a workgroup writes its
    slice) was larger than the ShapeNet stash of one 32-point tile and ran 32 KB into the ParameterNet's stash -- loss right, every
    ParameterNet gradient wrong by factors (1.4 ... 9.9 rel).  The workspace now covers the ring; all gradients within the plain bars."""
    from tests.test_gpu_parity import _cfg, _make, _per_tensor_rel
    m, model, spec, ws, x, y, sw = _make((_cfg("NIFMultiScale", 56, 3, 20, 3, 1, 1, 3, 3, p_act="swish"), B))
    loss, g = m._engine.loss_and_grad(x, y, sw)
    lref, gref = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))
    assert abs(loss - lref) <= 2e-6 * abs(lref)
    rel = _per_tensor_rel(spec, g, O.flatten(gref))
    assert max(rel.values()) < 1e-4, rel
    # ... and the same net through the k_snet4 + k_gw_* path agrees
    m._engine.set_option("fuse_gw", 0)
    _, g0 = m._engine.loss_and_grad(x, y, sw)
    assert float(np.linalg.norm(np.asarray(g) - np.asarray(g0)) / np.linalg.norm(g0)) < 5e-5
