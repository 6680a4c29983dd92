"""GPU parity tests proper: the HIP path (through the C-ABI / the drop-in Python surface) against
the fp64 NumPy oracle on identical inputs and weights.  Tolerances: predictions and loss within
1e-5 rel-L2 (BASELINE.json north_star, fp32); gradients within 5e-5 rel-L2 per tensor since r5 (fp32
accumulation over the batch vs fp64)."""
import numpy as np
import pytest

from oracle import nif_oracle as O

pytestmark = pytest.mark.gpu


def _cfg(kind, n, L, nst, lst, r, si, so, pi, s_res=False, p_act="sine", p_res=False, act="swish", omega=30.0):
    if kind == "LL":
        _, cs, cp = _cfg("NIFMultiScale", n, L, nst, lst, r, si, so, pi, s_res, p_act, p_res, act, omega)
        cs["connectivity"] = "last_layer"
        return "NIFMultiScaleLastLayerParameterized", cs, cp
    if kind == "NIF":
        cs = {"input_dim": si, "output_dim": so, "units": n, "nlayers": L, "activation": act}
        cp = {"input_dim": pi, "latent_dim": r, "units": nst, "nlayers": lst, "activation": act}
    else:
        cs = {"input_dim": si, "output_dim": so, "units": n, "nlayers": L, "use_resblock": s_res,
              "connectivity": "full", "omega_0": omega, "weight_init_factor": 0.01}
        cp = {"input_dim": pi, "latent_dim": r, "units": nst, "nlayers": lst, "activation": p_act,
              "use_resblock": p_res, "omega_0": omega}
    return kind, cs, cp


CONFIGS = {
    # name: (cfg, batch)
    "nif_cfg1_32x2": (_cfg("NIF", 32, 2, 32, 2, 1, 1, 1, 1), 300),
    "nif_pad_n30_tanh_r2_so2": (_cfg("NIF", 30, 2, 20, 1, 2, 2, 2, 2, act="tanh"), 257),
    "ms_cfg2_64x4": (_cfg("NIFMultiScale", 64, 4, 32, 2, 1, 1, 1, 1), 515),
    "ms_64x2_mlp_pnet_r3": (_cfg("NIFMultiScale", 64, 2, 32, 2, 3, 2, 1, 1, p_act="swish"), 129),
    "ms_res_48x2_pres": (_cfg("NIFMultiScale", 48, 2, 40, 2, 2, 2, 2, 1, s_res=True, p_res=True), 200),
    "ms_mlp_pres_so2": (_cfg("NIFMultiScale", 32, 1, 32, 1, 1, 1, 2, 2, p_act="tanh", p_res=True), 64),
    "ms_cfg3_128x3": (_cfg("NIFMultiScale", 128, 3, 64, 2, 1, 2, 1, 1), 160),
    "ms_res_128x1_nst128": (_cfg("NIFMultiScale", 128, 1, 128, 1, 1, 2, 1, 1, s_res=True), 96),
    "ms_tiny_b1": (_cfg("NIFMultiScale", 8, 1, 6, 1, 1, 1, 1, 1), 1),
    # boundaries of the LDS-DMA gradient kernels: (r+1) so = 8 (last of k_gw_out_lds), = 10 (register-load form),
    # (r+1)(si+1) = 32 rows of the first-layer MFMA operand, a batch smaller than one wave's share of tiles
    "ms_64x2_r1_so4": (_cfg("NIFMultiScale", 64, 2, 32, 2, 1, 2, 4, 1), 1031),
    "ms_64x2_r1_so5": (_cfg("NIFMultiScale", 64, 2, 32, 2, 1, 2, 5, 1), 333),
    "ms_32x2_r7_si3": (_cfg("NIFMultiScale", 32, 2, 32, 1, 7, 3, 1, 1), 4097),
    "ms_64x3_r3_so2_b33": (_cfg("NIFMultiScale", 64, 3, 24, 2, 3, 1, 2, 2), 33),
    # last-layer-parameterised class (config 4 family)
    "ll_plain_32x2_r3": (_cfg("LL", 32, 2, 32, 1, 3, 2, 2, 1), 257),
    "ll_res_48x2_r4": (_cfg("LL", 48, 2, 40, 2, 4, 3, 1, 2, s_res=True, p_res=True, p_act="swish"), 130),
    "ll_cfg4_128x2_r10_so3": (_cfg("LL", 128, 2, 32, 2, 10, 3, 3, 1), 96),
    # BASELINE.json's own shapes / the kernel instantiations they select: plain SIREN nets whose (nh+1)*4*NBL sign bits
    # do not fit the 128-bit ring run k_snet4<NBL,true,SINE,0,SGN=false> (act'(a) ring instead of the sign-bit cosine):
    # configs[2] = 6x128 (NBL 8), a 128-wide net with 4 layers, a 64-wide one with 8 (NBL 4); configs[3] = last-layer class
    # at 128x6 (and x4); configs[4] = 64x4 with two coordinates
    "ms_cfg3_128x6": (_cfg("NIFMultiScale", 128, 6, 32, 2, 1, 2, 1, 1, p_act="swish"), 200),
    "ms_128x4_r2": (_cfg("NIFMultiScale", 128, 4, 32, 2, 2, 2, 1, 1), 97),
    "ms_64x8": (_cfg("NIFMultiScale", 64, 8, 32, 2, 1, 1, 1, 1, p_act="swish"), 150),
    "ms_cfg5_64x4_si2": (_cfg("NIFMultiScale", 64, 4, 32, 2, 1, 2, 1, 1, p_act="swish"), 300),
    "ll_cfg4_128x6_r10_so3": (_cfg("LL", 128, 6, 32, 2, 10, 3, 3, 1, p_act="swish"), 160),
    "ll_128x4_r4": (_cfg("LL", 128, 4, 32, 2, 4, 2, 1, 1), 70),
    "ll_res_64x1_r4_so2": (_cfg("LL", 64, 1, 40, 2, 4, 3, 2, 2, s_res=True, p_res=True, p_act="swish"), 130),
    "ll_96x2_r5": (_cfg("LL", 96, 2, 32, 1, 5, 2, 3, 1), 77),
    # r4: k_llg (PT tiles per wave and stream pass) takes the plain 64- and 128-wide nets of the class; 331 points = 21 tiles:
    # two full tile groups of a workgroup under the policy (16 tiles each), a partly active wave in the second
    "ll_64x3_r5_so2": (_cfg("LL", 64, 3, 32, 2, 5, 3, 2, 1, p_act="swish"), 331),
    # 65..96 units = six 16-blocks in the fused kernels but 128-row stash tiles for the gradient kernels (stash_fp): r1 / early r2
    # builds disagreed on the tile stride there and produced wrong ShapeNet weight gradients without any error
    "ms_96x2_r2": (_cfg("NIFMultiScale", 96, 2, 32, 1, 2, 2, 1, 1), 77),
    "ms_res_80x1_so2": (_cfg("NIFMultiScale", 80, 1, 32, 1, 1, 2, 2, 1, s_res=True), 140),
    "ms_res_64x2": (_cfg("NIFMultiScale", 64, 2, 32, 2, 2, 2, 1, 1, s_res=True), 150),
    "nif_80x2_swish_r2": (_cfg("NIF", 80, 2, 32, 2, 2, 1, 1, 1, act="swish"), 100),
    # whole fp32 planes + the (r+1) copies of the small vectors exceed the LDS: the bf16-split kernel (chunked planes) still takes it
    "ms_res_128x4_r4_so2": (_cfg("NIFMultiScale", 128, 4, 32, 2, 4, 1, 2, 2, s_res=True), 97),
    # r5: 113..127 units = eight 16-feature blocks with PADDED features (the half-pair forms, and under the policy the 16-bit phase stash)
    "ms_pad_120x3": (_cfg("NIFMultiScale", 120, 3, 32, 2, 1, 2, 1, 1), 131),
    "ll_pad_120x2_r6_so2": (_cfg("LL", 120, 2, 32, 2, 6, 2, 2, 1), 99),
}


def _make(name, seed=0, boost=2.0):
    import nif_amd
    (kind, cs, cp), B = CONFIGS[name] if isinstance(name, str) else name
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(seed)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    if kind == "NIFMultiScale":
        names = [nm for nm, _ in spec.param_shapes()]
        ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * boost).astype(np.float32)
    if kind == "NIFMultiScaleLastLayerParameterized":
        # weight_init_factor = 0.01 makes the r x r map ~0: give the output a usable scale
        names = [nm for nm, _ in spec.param_shapes()]
        ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 30.0).astype(np.float32)
    cls = getattr(nif_amd, kind)
    m = cls(cs, cp)
    model = m.build()
    model.set_weights(ws)
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32)
    ws64 = [w.astype(np.float64) for w in ws]
    return m, model, spec, ws64, x, y, sw


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_forward_matches_oracle(name):
    m, model, spec, ws, x, y, sw = _make(name)
    u = model.predict(x)
    ref = O.forward(spec, ws, x.astype(np.float64))
    assert u.shape == ref.shape and u.dtype == np.float32
    assert _rel(u, ref) < 1e-5, _rel(u, ref)


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_weights_round_trip_and_layout(name):
    m, model, spec, ws, x, y, sw = _make(name)
    got = model.get_weights()
    assert len(got) == len(ws)
    for g, w in zip(got, ws):
        assert np.array_equal(g, w.astype(np.float32))
    lay = m._engine.layout()
    assert [(nm, r * (c if c else 1)) for nm, _, r, c in lay] == [(nm, int(np.prod(s))) for nm, s in spec.param_shapes()]


@pytest.mark.parametrize("name", ["nif_cfg1_32x2", "ms_cfg2_64x4", "ms_64x2_mlp_pnet_r3", "ms_res_48x2_pres",
                                  "ms_cfg3_128x3", "nif_pad_n30_tanh_r2_so2"])
def test_three_stage_factorisation(name):
    """README.md:99-117: lr = model_p_to_lr(p); w = model_lr_to_w(lr); u = model_x_to_u_given_w([x, w])."""
    m, model, spec, ws, x, y, sw = _make(name)
    p, xs = x[:, :spec.pi], x[:, spec.pi:]
    lr = m.model_p_to_lr().predict(p)
    lr_ref = O.model_p_to_lr(spec, ws, p.astype(np.float64))
    assert _rel(lr, lr_ref) < 1e-5
    w = m.model_lr_to_w().predict(lr)
    assert w.shape == (x.shape[0], spec.po)
    assert _rel(w, O.model_lr_to_w(spec, ws, lr.astype(np.float64))) < 1e-6
    u3 = m.model_x_to_u_given_w().predict([xs, w])
    ref = O.shapenet_given_w(spec, xs.astype(np.float64), w.astype(np.float64))
    assert _rel(u3, ref) < 1e-5, _rel(u3, ref)
    assert _rel(u3, model.predict(x).astype(np.float64)) < 2e-5
    w2 = m.model_p_to_w().predict(p)
    assert _rel(w2, w.astype(np.float64)) < 1e-6


@pytest.mark.parametrize("name", ["ll_plain_32x2_r3", "ll_res_48x2_r4", "ll_cfg4_128x2_r10_so3", "ll_cfg4_128x6_r10_so3"])
def test_last_layer_class_submodels(name):
    """model.py:1070-1145: p -> lr (= pnet output), x -> phi, u = Dot(phi, lr) + bias; lr_to_w raises."""
    m, model, spec, ws, x, y, sw = _make(name)
    p, xs = x[:, :spec.pi], x[:, spec.pi:]
    lr = m.model_p_to_lr().predict(p)
    assert lr.shape == (x.shape[0], spec.r)
    assert _rel(lr, O.model_p_to_lr(spec, ws, p.astype(np.float64))) < 1e-5
    phi = m.model_x_to_phi().predict(xs)
    assert phi.shape == (x.shape[0], spec.so, spec.r)
    assert _rel(phi, O.model_x_to_phi(spec, ws, xs.astype(np.float64))) < 1e-5
    u = m.model_x_to_u_given_w().predict([xs, lr])
    ref = np.einsum("bsj,bj->bs", phi.astype(np.float64), lr.astype(np.float64)) + ws[-1]
    assert _rel(u, ref) < 1e-5
    assert _rel(u, model.predict(x).astype(np.float64)) < 1e-5
    with pytest.raises(ValueError):
        m.model_lr_to_w()


@pytest.mark.parametrize("name", ["nif_cfg1_32x2", "nif_pad_n30_tanh_r2_so2", "ms_cfg2_64x4", "ms_64x2_mlp_pnet_r3",
                                  "ms_res_48x2_pres", "ms_cfg3_128x3", "ms_cfg5_64x4_si2", "ms_cfg3_128x6"])
def test_jacobian_layer_matches_oracle(name):
    """gradient.py:36-49: (y, dy/dx) w.r.t. the coordinate columns; oracle = analytic tangent in fp64 (itself
    pinned by central differences and torch autograd in tests/test_oracle.py)."""
    import nif_amd
    m, model, spec, ws, x, y, sw = _make(name)
    yi = list(range(spec.so))
    xi = list(range(spec.pi, spec.pi + spec.si))
    yv, J = nif_amd.JacobianLayer(model, yi, xi)(x)
    yr, Jr = O.jacobian_analytic(spec, ws, x.astype(np.float64), yi, xi)
    assert yv.shape == (x.shape[0], spec.so) and J.shape == (x.shape[0], len(yi), len(xi))
    assert _rel(yv, yr) < 1e-5
    assert _rel(J, Jr) < 2e-5, _rel(J, Jr)
    # index selection / ordering like tf.gather (gradient.py:228-229)
    if spec.si > 1:
        y2, J2 = nif_amd.JacobianLayer(model, [spec.so - 1], [xi[-1], xi[0]])(x)
        assert np.allclose(J2[:, 0, 0], J[:, spec.so - 1, -1], rtol=1e-4, atol=1e-5)
        assert np.allclose(J2[:, 0, 1], J[:, spec.so - 1, 0], rtol=1e-4, atol=1e-5)
    # parameter columns (e.g. du/dt): the hypernetwork weights move too; oracle = fp64 central differences
    pcols = list(range(spec.pi))
    _, Jp = nif_amd.JacobianLayer(model, yi, pcols)(x)
    _, Jpr = O.jacobian(spec, ws, x.astype(np.float64), yi, pcols)
    assert _rel(Jp, Jpr) < 1e-4, _rel(Jp, Jpr)
    # mixed request, more than 3 seeds' worth of columns when available
    allc = list(range(spec.pi + spec.si))
    _, Ja = nif_amd.JacobianLayer(model, yi, allc)(x)
    assert _rel(Ja[:, :, :spec.pi], Jp.astype(np.float64)) < 2e-5 and _rel(Ja[:, :, spec.pi:], J.astype(np.float64)) < 2e-5


@pytest.mark.parametrize("name", ["ll_plain_32x2_r3", "ll_res_48x2_r4", "ll_cfg4_128x2_r10_so3", "ll_cfg4_128x6_r10_so3"])
def test_jacobian_layer_last_layer_class(name):
    import nif_amd
    m, model, spec, ws, x, y, sw = _make(name)
    yi = list(range(spec.so))
    allc = list(range(spec.pi + spec.si))
    yv, J = nif_amd.JacobianLayer(model, yi, allc)(x)
    yr, Jr = O.jacobian(spec, ws, x.astype(np.float64), yi, allc)   # fp64 central differences
    assert _rel(yv, yr) < 1e-5
    assert _rel(J, Jr) < 1e-4, _rel(J, Jr)


def test_given_w_arbitrary_weights():
    """model_x_to_u_given_w must take ANY per-sample w, not only ones produced by the hypernetwork."""
    m, model, spec, ws, x, y, sw = _make("ms_cfg2_64x4")
    rng = np.random.default_rng(5)
    B = 77
    xs = rng.uniform(-1, 1, size=(B, spec.si)).astype(np.float32)
    w = (rng.standard_normal((B, spec.po)) * 0.01).astype(np.float32)  # 0.05 is ill-conditioned even for NumPy fp32
    u = m.model_x_to_u_given_w().predict([xs, w])
    ref = O.shapenet_given_w(spec, xs.astype(np.float64), w.astype(np.float64))
    assert _rel(u, ref) < 1e-5, _rel(u, ref)


@pytest.mark.parametrize("name", sorted(CONFIGS))
@pytest.mark.parametrize("weighted", [False, True])
def test_loss_and_grad_match_oracle(name, weighted):
    m, model, spec, ws, x, y, sw = _make(name)
    s = sw if weighted else None
    loss, g = m._engine.loss_and_grad(x, y, s)
    lref, gref = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64),
                                 None if s is None else s.astype(np.float64))
    # r5 bars (tools/exp/grad_survey.py over these 60 configs x 2: loss <= 5.0e-7, flat gradient <= 9.2e-6, worst tensor 1.9e-5 of
    # (its norm + 5e-3 of the whole gradient's) -- since the data adjoint runs on half pairs; r1-r4: 1e-5 / 1e-4 / 2e-4)
    assert abs(loss - lref) <= 2e-6 * abs(lref), (loss, lref)
    off = 0
    gnorm = np.linalg.norm(O.flatten(gref))
    for (nm, shp), gr in zip(spec.param_shapes(), gref):
        k = int(np.prod(shp))
        gg = g[off:off + k].reshape(shp)
        off += k
        err = np.linalg.norm(gg - gr)
        assert err <= 5e-5 * np.linalg.norm(gr) + 2.5e-7 * gnorm, (nm, err, np.linalg.norm(gr))
    assert _rel(g, O.flatten(gref)) < 3e-5


def test_adam_steps_follow_oracle():
    """three full-batch Adam steps at lr = 2e-3 (Keras' order of magnitude), every step checked on its own: from the weights and
    Adam slots the GPU holds BEFORE step t the oracle computes loss, gradient and the Keras-2.11 update; the GPU's weights after the
    step must sit within 1 % of lr of that prediction wherever the gradient entry is solid, and its slots must be the oracle's.
    (r3 compared free-running fp32 / fp64 trajectories: with w0 = 30 the SECOND gradient of two trajectories 1e-7 apart already
    differs by per cents, so that test had to be loosened to lr = 2e-4 and a 5 % / 40 % bar; step-by-step there is nothing chaotic
    to absorb and the bar is the one-step error.)"""
    import nif_amd
    LR = 2e-3
    m, model, spec, ws, x, y, sw = _make("ms_cfg2_64x4")
    model.compile(nif_amd.Adam(learning_rate=LR), loss="mse")
    e = m._engine
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    f32 = lambda a: float(np.float32(a))
    worst = 0.0
    for t in range(1, 4):
        th0 = O.flatten(model.get_weights()).astype(np.float64)
        m0, v0, step0 = e.get_opt_state()
        assert step0 == t - 1
        hist = model.fit(x, y, epochs=1, batch_size=x.shape[0], shuffle=False, verbose=0)
        l, g = O.loss_and_grad(spec, O.unflatten(spec, th0), x64, y64)
        assert abs(hist.history["loss"][0] - l) <= 2e-5 * abs(l), (t, hist.history["loss"][0], l)
        gf = O.flatten(g)
        th1, m1, v1 = O.adam_step(th0, gf, m0.astype(np.float64), v0.astype(np.float64), t, lr=f32(LR), b1=f32(0.9), b2=f32(0.999),
                                  eps=f32(1e-7))
        got = O.flatten(model.get_weights())
        gm, gv, step1 = e.get_opt_state()
        assert step1 == t
        # entries whose gradient is not small against the tensor's own scale: there the fp32 gradient (2e-4 of the tensor's norm,
        # test_loss_and_grad_match_oracle) has a small RELATIVE error and Adam's normalised step lr m / sqrt(v) is insensitive to it
        solid = np.ones_like(th0, dtype=bool)
        off = 0
        for gt in g:
            rms = np.sqrt(np.mean(gt ** 2)) + 1e-300
            solid[off:off + gt.size] = np.abs(gt.ravel()) > 0.02 * rms
            off += gt.size
        assert solid.mean() > 0.8
        d = np.abs(got - th1)
        worst = max(worst, d[solid].max() / LR)
        assert d[solid].max() < 0.01 * LR, (t, d[solid].max() / LR)
        assert d.max() <= 2.0 * LR * 1.001                  # elsewhere: at most the sign flip of one normalised step
        assert _rel(gm, m1) < 1e-3
        assert _rel(gv, v1) < 1e-3
    assert worst < 0.01


def test_fit_batches_partial_last_batch_and_sample_weight():
    import nif_amd
    m, model, spec, ws, x, y, sw = _make("nif_cfg1_32x2")
    model.compile("adam", loss="mse")
    hist = model.fit(x, y, epochs=2, batch_size=128, shuffle=False, verbose=0, sample_weight=sw)
    th = O.flatten(ws); mm = np.zeros_like(th); vv = np.zeros_like(th)
    t = 0
    ep_losses = []
    for ep in range(2):
        tot = 0.0
        for b0 in range(0, x.shape[0], 128):
            xb, yb, sb = x[b0:b0 + 128], y[b0:b0 + 128], sw[b0:b0 + 128]
            l, g = O.loss_and_grad(spec, O.unflatten(spec, th), xb.astype(np.float64), yb.astype(np.float64),
                                   sb.astype(np.float64))
            t += 1
            th, mm, vv = O.adam_step(th, O.flatten(g), mm, vv, t)
            tot += l * xb.shape[0]
        ep_losses.append(tot / x.shape[0])
    assert np.allclose(hist.history["loss"], ep_losses, rtol=5e-4)


def test_training_reduces_loss_on_travelling_wave():
    """End-to-end README workflow on the reference's own dataset (nif/demo/dataset/traveling_wave.npz)."""
    import os
    import nif_amd
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "traveling_wave.npz"))["data"]
    data, mean, std = O.standard_normalize(d.astype(np.float64))
    x, y = data[:, :2].astype(np.float32), data[:, 2:3].astype(np.float32)
    kind, cs, cp = _cfg("NIFMultiScale", 32, 2, 32, 2, 1, 1, 1, 1, p_act="swish")
    nif_amd.set_seed(1)
    model = nif_amd.NIFMultiScale(cs, cp).build()
    model.compile(nif_amd.Adam(1e-3), loss="mse")
    sched = nif_amd.callbacks.LearningRateScheduler(lambda ep, lr: lr if ep < 70 else 5e-4)
    h = model.fit(x, y, epochs=80, batch_size=500, shuffle=True, verbose=0, callbacks=[sched])
    assert h.history["loss"][-1] < 0.5 * h.history["loss"][0]
    # evaluate() = the engine's fp32 loss (r3: the TOTAL loss, regularisers included; none here)
    ref = O.mse_loss(model.predict(x).astype(np.float64), y)
    assert abs(model.evaluate(x, y) - ref) < 1e-5 * ref


def test_bad_arguments_raise():
    import nif_amd
    m, model, spec, ws, x, y, sw = _make("nif_cfg1_32x2")
    with pytest.raises(ValueError):
        model.set_weights(ws[:-1])
    with pytest.raises(ValueError):
        m.model_x_to_u_given_w().predict([x[:, 1:], np.zeros((x.shape[0], 3), np.float32)])
    with pytest.raises(RuntimeError):
        nif_amd.NIF(*CONFIGS["nif_cfg1_32x2"][0][1:]).build().fit(x, y)


# ---- BASELINE.json full size (2^20 points, ShapeNet 4x64): size-independent properties ---------------------
def _full_size_setup():
    import nif_amd
    import bench
    nif_amd.set_seed(1)
    m = nif_amd.NIFMultiScale(bench.CFG_SHAPE, bench.CFG_PARAM)
    model = m.build()
    x, y = nif_amd.data.synthetic_wave_batch(1 << 20, seed=5)
    return m, model, x, y


def test_full_size_shard_sum_equals_full_batch_and_is_deterministic():
    """What the 8-GPU run relies on: the SUM over 8 contiguous shards of [grad | loss], each scaled
    by 1/B_global inside the kernels, equals the full-batch result; and a repeated launch is bit-identical
    (fixed reduction order, no atomics)."""
    m, model, x, y = _full_size_setup()
    e = m._engine
    from nif_amd.engine import DeviceArray
    from nif_amd import distributed as dist
    B = x.shape[0]
    d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size)
    d_x.upload(x); d_y.upload(y)

    def grad_of(lo, hi):
        e.loss_grad_dev(d_x.at(lo * 2), d_y.at(lo), None, hi - lo, B)
        buf = DeviceArray.__new__(DeviceArray)
        buf.engine, buf.n, buf.ptr = e, e.n_params + 1, e.grad_dev_ptr()
        out = buf.download()
        buf.ptr = None
        return out.astype(np.float64)

    full = grad_of(0, B)
    again = grad_of(0, B)
    assert np.array_equal(full, again)
    acc = np.zeros_like(full)
    for r in range(8):
        lo, hi = dist.shard_bounds(B, 8, r)
        acc += grad_of(lo, hi)
    assert abs(acc[-1] - full[-1]) < 1e-6 * abs(full[-1])
    assert np.linalg.norm(acc[:-1] - full[:-1]) < 1e-5 * np.linalg.norm(full[:-1])
    # the oracle on a sample of the same batch pins the magnitude (it cannot run 2^20 points)
    ws = [w.astype(np.float64) for w in model.get_weights()]
    spec = O.Spec("NIFMultiScale", m.cfg_shape_net, m.cfg_parameter_net)
    lref, _ = O.loss_and_grad(spec, ws, x[:4096].astype(np.float64), y[:4096].astype(np.float64))
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, 4096, 4096)
    assert abs(e.last_loss() - lref) < 1e-5 * abs(lref)


def test_full_size_predict_factorisation_and_permutation():
    """model([p,x]) == model_x_to_u_given_w(x, model_lr_to_w(model_p_to_lr(p))) and row-permutation
    equivariance at 2^20 / 2^16 points (the factorised path materialises [B, po], so it runs on a slice)."""
    m, model, x, y = _full_size_setup()
    u = model.predict(x)
    assert u.shape == (x.shape[0], 1) and np.isfinite(u).all()
    perm = np.random.default_rng(0).permutation(x.shape[0])
    up = model.predict(x[perm])
    assert np.array_equal(up, u[perm])          # points are independent rows: bitwise equivariant
    sl = slice(0, 1 << 16)
    lr = m.model_p_to_lr().predict(x[sl, :1])
    w = m.model_lr_to_w().predict(lr)
    u3 = m.model_x_to_u_given_w().predict([x[sl, 1:], w])
    assert _rel(u3, u[sl].astype(np.float64)) < 2e-5
    ws = [wt.astype(np.float64) for wt in model.get_weights()]
    spec = O.Spec("NIFMultiScale", m.cfg_shape_net, m.cfg_parameter_net)
    assert _rel(u[:2048], O.forward(spec, ws, x[:2048].astype(np.float64))) < 1e-5


def test_weight_regularisers_l2_and_l1():
    """cfg_parameter_net l2_reg / l1_reg (model.py:109-117): Keras adds l2*sum(w^2) (or l1*sum|w|) over every
    ParameterNet kernel and bias to the loss; L2 takes precedence over L1."""
    import nif_amd
    (kind, cs, cp), B = CONFIGS["ms_64x2_mlp_pnet_r3"]
    for reg, val in (("l2_reg", 1e-3), ("l1_reg", 2e-4)):
        cp2 = dict(cp); cp2[reg] = val
        spec = O.Spec(kind, cs, cp)
        rng = np.random.default_rng(0)
        ws = O.init_weights(spec, rng, dtype=np.float32)
        m = nif_amd.NIFMultiScale(cs, cp2)
        model = m.build(); model.set_weights(ws)
        x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
        y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
        loss, g = m._engine.loss_and_grad(x, y)
        ws64 = [w.astype(np.float64) for w in ws]
        lref, gref = O.loss_and_grad(spec, ws64, x.astype(np.float64), y.astype(np.float64))
        th = O.flatten(ws64); gref = O.flatten(gref)
        if reg == "l2_reg":
            lref += val * (th ** 2).sum(); gref = gref + 2 * val * th
        else:
            lref += val * np.abs(th).sum(); gref = gref + val * np.sign(th)
        assert abs(loss - lref) < 1e-5 * abs(lref)
        assert _rel(g, gref) < 1e-4


@pytest.mark.parametrize("name", ["ll_plain_32x2_r3", "ll_res_48x2_r4", "ll_cfg4_128x2_r10_so3"])
@pytest.mark.parametrize("case", ["s_l2_p_l2", "s_l1_p_l1", "s_l2_only", "s_l1_p_l2"])
def test_last_layer_class_shapenet_regularisers(name, case):
    """cfg_shape_net l2_reg / l1_reg of the last-layer class (model.py:1028-1039): kernel and bias regulariser of every layer of the
    shared ShapeNet, with the reference's coefficient quirk (it hands cfg_parameter_net's number, or Keras' default 0.01 for None,
    to regularizers.L2 / L1).  r3 ignored these keys silently."""
    import nif_amd
    (kind, cs, cp), B = CONFIGS[name]
    cs2, cp2 = dict(cs), dict(cp)
    if case == "s_l2_p_l2": cs2["l2_reg"] = 5e-4; cp2["l2_reg"] = 2e-3          # ShapeNet L2 with the ParameterNet's 2e-3
    elif case == "s_l1_p_l1": cs2["l1_reg"] = 7e-4; cp2["l1_reg"] = 1e-3        # ShapeNet L1 with 1e-3
    elif case == "s_l2_only": cs2["l2_reg"] = 5e-4                              # ShapeNet L2 with Keras' default 0.01, no ParameterNet term
    else: cs2["l1_reg"] = 3e-4; cp2["l2_reg"] = 1e-3                            # ParameterNet L2 1e-3; ShapeNet L1 with p_l1_reg = None -> 0.01
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(3)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    names = [nm for nm, _ in spec.param_shapes()]
    ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 30.0).astype(np.float32)
    m = nif_amd.NIFMultiScaleLastLayerParameterized(cs2, cp2)
    model = m.build(); model.set_weights(ws)
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    loss, g = m._engine.loss_and_grad(x, y)
    ws64 = [w.astype(np.float64) for w in ws]
    lref, gref = O.loss_and_grad(spec, ws64, x.astype(np.float64), y.astype(np.float64))
    preg, sreg = O.weight_regularizer_coefficients(cs2, cp2, spec.kind)
    assert sreg != (0.0, 0.0)
    lreg, greg = O.weight_regularizer_term(spec, ws64, preg, sreg)
    assert lreg > 0.05 * abs(lref)          # the term is visible in the loss
    lref += lreg
    gref = O.flatten([a + b for a, b in zip(gref, greg)])
    assert abs(loss - lref) < 1e-5 * abs(lref)
    assert _rel(g, gref) < 2e-4
    # per tensor: the ShapeNet tensors carry the term, last_layer_bias does not
    off = 0
    for (nm, shp), gr in zip(spec.param_shapes(), greg):
        sz = int(np.prod(shp))
        if nm == "last_layer_bias":
            assert np.abs(gr).max() == 0.0
        if nm.startswith("snet_") and nm.endswith("_w"):
            assert _rel(g[off:off + sz], gref[off:off + sz]) < 5e-4, nm
        off += sz
    # the other classes refuse the call instead of ignoring it
    (k2, cs3, cp3), _ = CONFIGS["ms_mlp_pres_so2"]
    m2 = nif_amd.NIFMultiScale(cs3, cp3)
    with pytest.raises(nif_amd.NifError):
        m2._engine.set_shapenet_regularizer(0.0, 1e-3)


def test_lbfgs_fine_tuning_reduces_loss():
    """README.md:51-69: Adam first, then TFPLBFGS(model, loss, X, Y).minimize(rounds, max_iter)."""
    import os
    import nif_amd
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "traveling_wave.npz"))["data"]
    data, _, _ = O.standard_normalize(d.astype(np.float64))
    x, y = data[:, :2].astype(np.float32), data[:, 2:3].astype(np.float32)
    kind, cs, cp = _cfg("NIFMultiScale", 32, 2, 32, 2, 1, 1, 1, 1, p_act="swish")
    nif_amd.set_seed(1)
    model = nif_amd.NIFMultiScale(cs, cp).build()
    model.compile(nif_amd.Adam(1e-3), loss="mse")
    model.fit(x, y, epochs=20, batch_size=500, verbose=0)
    l0 = model.evaluate(x, y)
    tuner = nif_amd.optimizers.TFPLBFGS(model, "mse", x, y, display_epoch=10)
    hist = tuner.minimize(rounds=3, max_iter=20)
    l1 = model.evaluate(x, y)
    # lbfgs.py:123-126: history = {"iteration", "loss"} with one entry per closure evaluation
    assert l1 < l0 and len(hist["loss"]) > 5 and len(hist["iteration"]) == len(hist["loss"])
    assert min(hist["loss"]) <= hist["loss"][0] and abs(min(hist["loss"]) - l1) < 1e-4 * max(l1, 1e-8) + 1e-7


# ---- Sobolev training (BASELINE config 5): JacobianLayer as a trained output -----------------------------
SOB = ["ms_cfg2_64x4", "ms_64x2_mlp_pnet_r3", "ms_res_48x2_pres", "ms_mlp_pres_so2", "ms_cfg3_128x3", "ms_tiny_b1",
       "ms_cfg5_64x4_si2", "ms_cfg3_128x6", "ms_64x8", "nif_cfg1_32x2", "nif_pad_n30_tanh_r2_so2", "ms_96x2_r2", "ms_res_80x1_so2",
       "nif_80x2_swish_r2"]


@pytest.mark.parametrize("name", SOB)
@pytest.mark.parametrize("weighted", [False, True])
def test_sobolev_loss_and_grad_match_oracle(name, weighted):
    """loss = mse(u, y) + w * mse(du/dx, g): forward tangents + their adjoint in one kernel (k_sob) against the
    oracle's reference-formulation adjoint (itself pinned to torch double-backward in tests/test_oracle.py)."""
    m, model, spec, ws, x, y, sw = _make(name)
    B = x.shape[0]
    xi = list(range(spec.pi, spec.pi + spec.si))
    rng = np.random.default_rng(11)
    g = rng.uniform(-1, 1, size=(B, spec.so, len(xi))).astype(np.float32)
    sample_weight = sw if weighted else None
    wj = 0.05
    loss, grad = m._engine.sobolev_loss_and_grad(x, y, g, xi, wj, sample_weight)
    x64 = x.astype(np.float64)
    rl, rg, ru, rJ = O.sobolev_loss_and_grad(spec, ws, x64, y.astype(np.float64), g.astype(np.float64), xi, wj,
                                             None if sample_weight is None else sample_weight.astype(np.float64))
    assert abs(loss - rl) <= 2e-5 * abs(rl), (loss, rl)
    off = 0
    for (nm, shp), r_ in zip(spec.param_shapes(), rg):
        k = int(np.prod(shp))
        got = grad[off:off + k].reshape(shp)
        off += k
        err = _rel(got, r_) if np.linalg.norm(r_) > 1e-12 else float(np.abs(got).max())
        assert err < 3e-4, (nm, err)
    # predict() of the two-output model
    from nif_amd import JacobianLayer, SobolevModel
    sm = SobolevModel(JacobianLayer(model, list(range(spec.so)), xi))
    u, J = sm.predict(x)
    assert u.shape == (B, spec.so) and J.shape == (B, spec.so, len(xi))
    assert _rel(u, ru) < 1e-5 and _rel(J, rJ) < 2e-5, (_rel(u, ru), _rel(J, rJ))


# shapes of k_sobw (plain SIREN NIFMultiScale, <= 64 units, coordinate seeds: one wave per stream): two- and four-block widths,
# 1..3 seeds in any order, several outputs, latent dims up to 4, one hidden matrix, ragged batches
SOBW = {
    "n20_L1_r1_si2": (_cfg("NIFMultiScale", 20, 1, 16, 1, 1, 2, 1, 1), 97),
    "n32_L3_r4_si3_so2": (_cfg("NIFMultiScale", 32, 3, 24, 2, 4, 3, 2, 2), 1031),
    "n50_L2_r2_si2_so3": (_cfg("NIFMultiScale", 50, 2, 32, 2, 2, 2, 3, 1), 515),
    "n64_L5_r3_si3": (_cfg("NIFMultiScale", 64, 5, 32, 2, 3, 3, 1, 2), 4099),
}


@pytest.mark.parametrize("name", sorted(SOBW))
@pytest.mark.parametrize("ns", [1, 2, 3])
def test_sobolev_streams_on_waves_shapes(name, ns):
    m, model, spec, ws, x, y, sw = _make(SOBW[name])
    if ns > spec.si:
        pytest.skip("fewer coordinates than seeds")
    B = x.shape[0]
    xi = list(range(spec.pi, spec.pi + spec.si))[::-1][:ns]          # descending columns: the streams are not in x_index order
    g = np.random.default_rng(13).uniform(-1, 1, size=(B, spec.so, ns)).astype(np.float32)
    loss, grad = m._engine.sobolev_loss_and_grad(x, y, g, xi, 0.3, sw)
    rl, rg, ru, rJ = O.sobolev_loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), g.astype(np.float64), xi, 0.3,
                                             sw.astype(np.float64))
    assert abs(loss - rl) <= 2e-5 * abs(rl), (loss, rl)
    rel = _per_tensor_rel(spec, grad, O.flatten(rg))
    assert max(rel.values()) < 3e-4, rel
    # without sample weights, five points fewer (another ragged tail)
    l2, g2 = m._engine.sobolev_loss_and_grad(x[:B - 5], y[:B - 5], g[:B - 5], xi, 0.3, None)
    rl2, rg2 = O.sobolev_loss_and_grad(spec, ws, x[:B - 5].astype(np.float64), y[:B - 5].astype(np.float64),
                                       g[:B - 5].astype(np.float64), xi, 0.3, None)[:2]
    assert abs(l2 - rl2) <= 2e-5 * abs(rl2) and _rel(g2, O.flatten(rg2)) < 3e-4


def test_sobolev_streams_on_waves_many_tile_groups():
    """70 001 points = 1094 tile groups on 256 workgroups (the chunk stream wraps, the input rows of the next group are
    prefetched, the last group is ragged) against the oracle's plane formulation, float32 and under the policy"""
    cfg = (_cfg("NIFMultiScale", 64, 2, 32, 2, 2, 2, 1, 1), 70001)
    for policy, rnd, bl, bg in (("float32", None, 2e-5, 3e-4), ("mixed_bfloat16", O.bf16_round, 5e-4, 3e-3)):
        m, model, spec, ws, x, y, sw = _make_policy(cfg, policy)
        xi = [spec.pi, spec.pi + 1]
        g = np.random.default_rng(17).uniform(-1, 1, size=(x.shape[0], spec.so, 2)).astype(np.float32)
        loss, grad = m._engine.sobolev_loss_and_grad(x, y, g, xi, 0.1, sw)
        rl, rg = O.sobolev_planes_loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), g.astype(np.float64), xi, 0.1,
                                                sw.astype(np.float64), rnd=rnd, stash_bf16=rnd is not None and _stash_bf16(spec, xi))[:2]
        assert abs(loss - rl) <= bl * abs(rl), (policy, loss, rl)
        rel = _per_tensor_rel(spec, grad, O.flatten(rg))
        assert max(rel.values()) < bg, (policy, rel)



def test_sobolev_single_seed_and_zero_weight_degenerates():
    m, model, spec, ws, x, y, sw = _make("ms_64x2_mlp_pnet_r3")      # si = 2: differentiate w.r.t. the 2nd coordinate only
    B = x.shape[0]
    xi = [spec.pi + 1]
    g = np.random.default_rng(2).uniform(-1, 1, size=(B, spec.so, 1)).astype(np.float32)
    loss, grad = m._engine.sobolev_loss_and_grad(x, y, g, xi, 0.2, sw)
    rl, rg, _, _ = O.sobolev_loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), g.astype(np.float64), xi,
                                           0.2, sw.astype(np.float64))
    assert abs(loss - rl) <= 2e-5 * abs(rl)
    assert _rel(grad, O.flatten(rg)) < 3e-4
    # w = 0: the plain training step
    l0, g0 = m._engine.sobolev_loss_and_grad(x, y, g, xi, 0.0, sw)
    l1, g1 = m._engine.loss_and_grad(x, y, sw)
    assert abs(l0 - l1) <= 1e-6 * abs(l1) and _rel(g0, g1.astype(np.float64)) < 1e-5
    # plain step afterwards still right (stash geometry is shared)
    l2, g2 = m._engine.loss_and_grad(x, y, sw)
    assert l2 == l1 and np.array_equal(g1, g2)
    # out-of-range / repeated columns and the last-layer class are refused loudly
    import nif_amd
    with pytest.raises(nif_amd._lib.NifError):
        m._engine.sobolev_loss_and_grad(x, y, g, [spec.pi + spec.si], 0.2, sw)
    with pytest.raises(nif_amd._lib.NifError):
        m._engine.sobolev_loss_and_grad(x, y, np.concatenate([g, g], axis=2), [1, 1], 0.2, sw)


SOB_LL = ["ll_plain_32x2_r3", "ll_cfg4_128x2_r10_so3", "ll_cfg4_128x6_r10_so3", "ll_128x4_r4", "ll_res_64x1_r4_so2", "ll_96x2_r5",
          "ll_res_48x2_r4"]       # (48 units = three 16-blocks: no bf16-split planes, the f32-input MFMA form)


@pytest.mark.parametrize("name", SOB_LL)
@pytest.mark.parametrize("weighted", [False, True])
def test_sobolev_last_layer_class_matches_oracle(name, weighted):
    """JacobianLayer as a trained output of NIFMultiScaleLastLayerParameterized (model.py:1044-1068, :1219-1269): the shared
    SIREN ShapeNet carries the coordinate tangents, u = Dot(phi, a) + bias, du/dx = Dot(phi', a) (k_sob<.., LL>)"""
    m, model, spec, ws, x, y, sw = _make(name)
    B = x.shape[0]
    xi = list(range(spec.pi, spec.pi + spec.si))[::-1][:3]          # reversed order: stream <-> dydx column mapping
    rng = np.random.default_rng(13)
    g = rng.uniform(-1, 1, size=(B, spec.so, len(xi))).astype(np.float32)
    sample_weight = sw if weighted else None
    wj = 0.05
    loss, grad = m._engine.sobolev_loss_and_grad(x, y, g, xi, wj, sample_weight)
    rl, rg, ru, rJ = O.sobolev_loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), g.astype(np.float64), xi, wj,
                                             None if sample_weight is None else sample_weight.astype(np.float64))
    assert abs(loss - rl) <= 2e-5 * abs(rl), (loss, rl)
    off = 0
    for (nm, shp), r_ in zip(spec.param_shapes(), rg):
        k = int(np.prod(shp))
        got = grad[off:off + k].reshape(shp)
        off += k
        err = _rel(got, r_) if np.linalg.norm(r_) > 1e-12 else float(np.abs(got).max())
        assert err < 3e-4, (nm, err)
    from nif_amd import JacobianLayer, SobolevModel
    sm = SobolevModel(JacobianLayer(model, list(range(spec.so)), xi))
    u, J = sm.predict(x)
    assert _rel(u, ru) < 1e-5 and _rel(J, rJ) < 2e-5, (_rel(u, ru), _rel(J, rJ))
    # the plain step of the class afterwards (k_snet4<.., LL>) is untouched
    l1, g1 = m._engine.loss_and_grad(x, y, sw)
    rl1, rg1 = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))
    assert abs(l1 - rl1) <= 2e-5 * abs(rl1) and _rel(g1, O.flatten(rg1)) < 3e-4


@pytest.mark.parametrize("name", ["ll_plain_32x2_r3", "ll_cfg4_128x2_r10_so3", "ll_res_64x1_r4_so2", "ll_96x2_r5", "ll_res_48x2_r4"])
@pytest.mark.parametrize("cols", ["param_only", "mixed"])
def test_sobolev_last_layer_class_parameter_columns(name, cols):
    """x_index addressing ParameterNet inputs on the last-layer class: the ShapeNet does not see p, du/dp_c = Dot(phi, a'_c) with
    a'_c = (dz/dp_c) last_w -- one more contraction of the same phi in the kernel's epilogue, its adjoint through the r x r layer
    and the (primal, tangent) ParameterNet"""
    m, model, spec, ws, x, y, sw = _make(name)
    B = x.shape[0]
    if cols == "param_only":
        xi = list(range(min(spec.pi, 3)))
    else:
        xi = [spec.pi + spec.si - 1, spec.pi - 1, spec.pi]               # coordinate, parameter, coordinate
    rng = np.random.default_rng(14)
    g = rng.uniform(-1, 1, size=(B, spec.so, len(xi))).astype(np.float32)
    wj = 0.05
    loss, grad = m._engine.sobolev_loss_and_grad(x, y, g, xi, wj, sw)
    rl, rg, ru, rJ = O.sobolev_loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), g.astype(np.float64), xi, wj,
                                             sw.astype(np.float64))
    assert abs(loss - rl) <= 2e-5 * abs(rl), (loss, rl)
    off = 0
    for (nm, shp), r_ in zip(spec.param_shapes(), rg):
        k = int(np.prod(shp))
        got = grad[off:off + k].reshape(shp)
        off += k
        err = _rel(got, r_) if np.linalg.norm(r_) > 1e-12 else float(np.abs(got).max())
        assert err < 3e-4, (nm, err)
    from nif_amd import JacobianLayer, SobolevModel
    u, J = SobolevModel(JacobianLayer(model, list(range(spec.so)), xi)).predict(x)
    assert _rel(u, ru) < 1e-5 and _rel(J, rJ) < 2e-5, (_rel(u, ru), _rel(J, rJ))
    _, J2 = JacobianLayer(model, list(range(spec.so)), xi)(x)             # the forward-only kernels of the class
    assert _rel(J, J2.astype(np.float64)) < 2e-5
    l1, g1 = m._engine.loss_and_grad(x, y, sw)
    rl1, rg1 = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))
    assert abs(l1 - rl1) <= 2e-5 * abs(rl1) and _rel(g1, O.flatten(rg1)) < 3e-4


SOB_PAR = ["ms_cfg2_64x4", "ms_64x2_mlp_pnet_r3", "ms_res_48x2_pres", "ms_mlp_pres_so2", "ms_cfg3_128x3", "ms_tiny_b1",
           "ms_cfg5_64x4_si2", "nif_cfg1_32x2", "nif_pad_n30_tanh_r2_so2"]


@pytest.mark.parametrize("name", SOB_PAR)
@pytest.mark.parametrize("cols", ["param_only", "mixed"])
def test_sobolev_parameter_columns_match_oracle(name, cols):
    """x_index of JacobianLayer addressing ParameterNet inputs (gradient.py:207-231 takes any column): the tangent runs
    through the ParameterNet (k_pjac), the hyper layer and the product rule of every h W(p) (k_sob, PAR), alone and mixed
    with coordinate columns in an order that differs from the kernel's stream order"""
    m, model, spec, ws, x, y, sw = _make(name)
    B = x.shape[0]
    if cols == "param_only":
        xi = list(range(min(spec.pi, 3)))
    else:
        xi = [spec.pi + spec.si - 1, 0] + ([spec.pi] if spec.si > 1 else [])       # coordinate, parameter, coordinate
    rng = np.random.default_rng(12)
    g = rng.uniform(-1, 1, size=(B, spec.so, len(xi))).astype(np.float32)
    wj = 0.05
    loss, grad = m._engine.sobolev_loss_and_grad(x, y, g, xi, wj, sw)
    rl, rg, ru, rJ = O.sobolev_loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), g.astype(np.float64), xi, wj,
                                             sw.astype(np.float64))
    assert abs(loss - rl) <= 2e-5 * abs(rl), (loss, rl)
    off = 0
    for (nm, shp), r_ in zip(spec.param_shapes(), rg):
        k = int(np.prod(shp))
        got = grad[off:off + k].reshape(shp)
        off += k
        err = _rel(got, r_) if np.linalg.norm(r_) > 1e-12 else float(np.abs(got).max())
        assert err < 3e-4, (nm, err)
    from nif_amd import JacobianLayer, SobolevModel
    sm = SobolevModel(JacobianLayer(model, list(range(spec.so)), xi))
    u, J = sm.predict(x)
    assert _rel(u, ru) < 1e-5 and _rel(J, rJ) < 2e-5, (_rel(u, ru), _rel(J, rJ))
    # the same columns through the forward-only JacobianLayer kernels (k_jac)
    _, J2 = JacobianLayer(model, list(range(spec.so)), xi)(x)
    assert _rel(J, J2.astype(np.float64)) < 2e-5
    # a plain step afterwards is untouched by the side passes
    l1, g1 = m._engine.loss_and_grad(x, y, sw)
    rl1, rg1 = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))
    assert abs(l1 - rl1) <= 2e-5 * abs(rl1) and _rel(g1, O.flatten(rg1)) < 3e-4


WIDE_SOB = {
    # name: (cfg, batch, x_index, y_index)
    # a 3-D flow with time: all four input columns (r3 refused more than three), hypernetwork and last-layer class
    "ms_3d_time_all4": (_cfg("NIFMultiScale", 32, 2, 32, 1, 3, 3, 1, 1), 515, [0, 1, 2, 3], None),
    "ll_3d_time_all4_so3": (_cfg("LL", 64, 2, 32, 2, 5, 3, 3, 1), 130, [3, 0, 2, 1], None),
    "nif_4cols_pi2": (_cfg("NIF", 32, 2, 32, 2, 2, 2, 2, 2), 257, [2, 3, 0, 1], None),
    # a 5-parameter study: every parameter column + the coordinates = 7 columns, three passes
    "ms_pi5_all7": (_cfg("NIFMultiScale", 32, 2, 24, 2, 2, 2, 1, 5, p_act="swish"), 130, [5, 0, 6, 1, 2, 3, 4], None),
    "ll_pi4_params_only": (_cfg("LL", 32, 1, 32, 2, 3, 2, 2, 4, p_act="tanh"), 97, [3, 2, 1, 0], None),
    # any subset / order of outputs in y_index (r3: all outputs only)
    "ms_so3_y_2_0": (_cfg("NIFMultiScale", 50, 2, 32, 2, 2, 2, 3, 1), 257, [1, 2], [2, 0]),
    "ms_res_so2_y1_wave": (_cfg("NIFMultiScale", 48, 2, 40, 2, 2, 2, 2, 1, s_res=True, p_res=True), 200, [2], [1]),
    "ll_so3_y1_4cols": (_cfg("LL", 64, 2, 32, 2, 5, 3, 3, 1), 130, [0, 1, 2, 3], [1]),
    "nif_so2_y0": (_cfg("NIF", 30, 2, 20, 1, 2, 2, 2, 2, act="tanh"), 129, [3, 0], [0]),
    "ms_sobw_so3_y1": (_cfg("NIFMultiScale", 64, 3, 32, 2, 1, 2, 3, 1), 515, [1, 2], [1]),       # the streams-on-waves kernel
}


@pytest.mark.parametrize("name", sorted(WIDE_SOB))
@pytest.mark.parametrize("policy", ["float32"])
def test_sobolev_any_y_index_and_more_than_three_columns(name, policy):
    """JacobianLayer as a trained output with any y_index / any number of x_index columns (gradient.py:207-231).  The oracle's
    Sobolev step is the all-outputs one; a y_index subset is stated through it: residuals of the unlisted outputs are made zero
    (their targets = the oracle's own derivatives, constants) and the weight carries so / ny, which is the mean over the listed
    ny x nx entries.  fit() of the two-output model follows the oracle for two Adam steps; predict() returns [B, ny, nx]."""
    import nif_amd
    from nif_amd import JacobianLayer, SobolevModel
    (kind, cs, cp), B, xi, yi = WIDE_SOB[name]
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(17)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    names = [nm for nm, _ in spec.param_shapes()]
    ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * (30.0 if spec.kind == O.KIND_LL else 2.0)).astype(np.float32)
    m = getattr(nif_amd, kind)(cs, cp, mixed_policy=policy)
    model = m.build(); model.set_weights(ws)
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32)
    ys = list(range(spec.so)) if yi is None else yi
    g = rng.uniform(-1, 1, size=(B, len(ys), len(xi))).astype(np.float32)
    wj = 0.07
    ws64 = [w.astype(np.float64) for w in ws]
    x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)

    def oracle(w):
        gf = np.zeros((B, spec.so, len(xi)))
        if yi is not None:
            gf[:] = O.sobolev_loss_and_grad(spec, w, x64, y64, gf, xi, 0.0, s64)[3]      # the oracle's own du/dx: zero residuals
        gf[:, ys, :] = g
        return O.sobolev_loss_and_grad(spec, w, x64, y64, gf, xi, wj * spec.so / len(ys), s64)
    rl, rg, ru, rJ = oracle(ws64)
    sm = SobolevModel(JacobianLayer(model, ys, xi))
    u, J = sm.predict(x)
    assert J.shape == (B, len(ys), len(xi))
    assert _rel(u, ru) < 1e-5 and _rel(J, rJ[:, ys, :]) < 3e-5, (_rel(u, ru), _rel(J, rJ[:, ys, :]))
    _, J2 = JacobianLayer(model, ys, xi)(x)                       # the forward-only kernels give the same entries
    assert _rel(J, J2.astype(np.float64)) < 3e-5
    gfull = np.zeros((B, spec.so, len(xi)), dtype=np.float32); gfull[:, ys, :] = g
    loss, grad = m._engine.sobolev_loss_and_grad(x, y, gfull, xi, wj, sw, y_index=None if yi is None else yi)
    assert abs(loss - rl) <= 3e-5 * abs(rl), (loss, rl)
    rel = _per_tensor_rel(spec, grad, O.flatten(rg))
    assert max(rel.values()) < 4e-4, rel
    # the Keras surface: evaluate = that loss, fit = the oracle's Adam trajectory
    sm.compile(nif_amd.Adam(1e-4), "mse", loss_weights=[1.0, wj])
    assert abs(sm.evaluate(x, [y, g], sample_weight=sw) - rl) <= 3e-5 * abs(rl)
    h = sm.fit(x, [y, g], batch_size=B, epochs=2, shuffle=False, verbose=0, sample_weight=sw)
    th = O.flatten(ws64); mm = np.zeros_like(th); vv = np.zeros_like(th); ls = []
    f32 = lambda a: float(np.float32(a))
    for t in range(1, 3):
        l_, g_, _, _ = oracle(O.unflatten(spec, th))
        ls.append(l_)
        th, mm, vv = O.adam_step(th, O.flatten(g_), mm, vv, t, lr=f32(1e-4), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
    assert np.allclose(h.history["loss"], ls, rtol=2e-3), (h.history["loss"], ls)
    # a plain step afterwards is untouched by the passes
    l1, g1 = m._engine.loss_and_grad(x, y, sw)
    rl1, rg1 = O.loss_and_grad(spec, ws64, x64, y64, s64)
    # (the model has taken two Adam steps: compare at its current weights)
    wnow = [w.astype(np.float64) for w in model.get_weights()]
    rl1, rg1 = O.loss_and_grad(spec, wnow, x64, y64, s64)
    assert abs(l1 - rl1) <= 2e-5 * abs(rl1) and _rel(g1, O.flatten(rg1)) < 3e-4


def test_sobolev_passes_keep_the_regularisers_of_the_first_pass_only():
    """four columns = two passes; the weight / activity / latent-Jacobian regularisers of cfg_parameter_net must enter the total ONCE"""
    import nif_amd
    kind, cs, cp = _cfg("NIFMultiScale", 32, 2, 24, 2, 2, 2, 1, 2, p_act="swish")
    l2w, lact, ljac = 3e-3, 2e-3, 0.04
    cpr = dict(cp, l2_reg=l2w, act_l2_reg=lact, jac_reg=ljac)
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(23)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    mr = getattr(nif_amd, kind)(cs, cpr)
    model = mr.build(); model.set_weights(ws)
    B = 200
    x = rng.uniform(-1, 1, size=(B, 4)).astype(np.float32); y = rng.uniform(-1, 1, size=(B, 1)).astype(np.float32)
    xi = [0, 1, 2, 3]
    g = rng.uniform(-1, 1, size=(B, 1, 4)).astype(np.float32)
    ws64 = [w.astype(np.float64) for w in ws]; x64, y64 = x.astype(np.float64), y.astype(np.float64)
    mr._engine.set_jac_regularizer(model._jac_reg)
    loss, grad = mr._engine.sobolev_loss_and_grad(x, y, g, xi, 0.05)
    l0, g0, _, _ = O.sobolev_loss_and_grad(spec, ws64, x64, y64, g.astype(np.float64), xi, 0.05)
    la, ga = O.loss_and_grad(spec, ws64, x64, y64, act_reg=(0.0, lact))
    lp, gp = O.loss_and_grad(spec, ws64, x64, y64)
    lj, gj = O.jac_reg_loss_and_grad(spec, ws64, x64[:, :spec.pi], ljac)
    th = O.flatten(ws64)
    npn = sum(int(np.prod(s_)) for nm, s_ in spec.param_shapes() if nm.startswith("pnet_"))
    ref_l = l0 + (la - lp) + lj + l2w * float((th[:npn] ** 2).sum())
    ref_g = O.flatten(g0) + (O.flatten(ga) - O.flatten(gp)) + O.flatten(gj)
    ref_g[:npn] += 2.0 * l2w * th[:npn]
    assert abs(loss - ref_l) < 3e-5 * abs(ref_l), (loss, ref_l)
    assert _rel(grad, ref_g) < 3e-4
    mr._engine.set_jac_regularizer(0.0)


@pytest.mark.parametrize("act", ["selu", "softsign", "exponential", "hard_sigmoid"])
def test_remaining_keras_activations(act):
    """the rest of keras.activations (Keras 2.11) on every kernel family that evaluates an activation: class NIF (ShapeNet and
    ParameterNet, skip connections) and an MLP_ResNet ParameterNet of NIFMultiScale -- forward, loss / gradient, JacobianLayer on
    every column, HessianLayer, the Sobolev step, the latent-Jacobian regulariser"""
    import nif_amd
    from nif_amd import JacobianLayer, HessianLayer
    for kind, cs, cp, B in ((*_cfg("NIF", 40, 2, 24, 2, 2, 2, 2, 2, act=act), 130),
                            (*_cfg("NIFMultiScale", 32, 1, 40, 1, 2, 1, 1, 2, p_act=act, p_res=True), 97)):
        spec = O.Spec(kind, cs, cp)
        rng = np.random.default_rng(31)
        ws = O.init_weights(spec, rng, dtype=np.float32)
        if kind == "NIFMultiScale":
            names = [nm for nm, _ in spec.param_shapes()]
            ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 2.0).astype(np.float32)
        if act == "exponential":          # exp(exp(.)) with skip connections overflows at the default initial scale: a tamer draw
            ws = [(w * 0.3).astype(np.float32) for w in ws]
        m = getattr(nif_amd, kind)(cs, cp); model = m.build(); model.set_weights(ws)
        x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
        y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
        sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32)
        ws64 = [w.astype(np.float64) for w in ws]; x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
        assert _rel(model.predict(x), O.forward(spec, ws64, x64)) < 1e-5
        loss, g = m._engine.loss_and_grad(x, y, sw)
        rl, rg = O.loss_and_grad(spec, ws64, x64, y64, s64)
        assert abs(loss - rl) < 2e-5 * abs(rl)
        assert max(_per_tensor_rel(spec, g, O.flatten(rg)).values()) < 3e-4
        yi, xi = list(range(spec.so)), list(range(spec.pi + spec.si))
        _, J = JacobianLayer(model, yi, xi)(x)
        _, Jr = O.jacobian(spec, ws64, x64, yi, xi)
        assert _rel(J, Jr) < 2e-4
        _, Jh, H = HessianLayer(model, yi, xi)(x[:64])
        _, Jr2, Hr = O.hessian_analytic(spec, ws64, x64[:64], yi, xi)
        assert _rel(Jh, Jr2) < 3e-5 and _rel(H, Hr) < 3e-4, (_rel(Jh, Jr2), _rel(H, Hr))
        xs = [spec.pi + spec.si - 1, 0]
        gt = rng.uniform(-1, 1, size=(B, spec.so, 2)).astype(np.float32)
        sl, sg = m._engine.sobolev_loss_and_grad(x, y, gt, xs, 0.05, sw)
        rsl, rsg, _, _ = O.sobolev_loss_and_grad(spec, ws64, x64, y64, gt.astype(np.float64), xs, 0.05, s64)
        assert abs(sl - rsl) < 3e-5 * abs(rsl)
        assert max(_per_tensor_rel(spec, sg, O.flatten(rsg)).values()) < 4e-4
        m._engine.set_jac_regularizer(0.04)
        lj, gj = m._engine.loss_and_grad(x, y, sw)
        m._engine.set_jac_regularizer(0.0)
        rj, rgj = O.jac_reg_loss_and_grad(spec, ws64, x64[:, :spec.pi], 0.04)
        assert abs((lj - loss) - rj) < 3e-4 * rj + 1e-6 * abs(loss)
    with pytest.raises(ValueError):
        nif_amd.NIF(dict(cs, activation="softmax") if kind == "NIF" else {"input_dim": 1, "output_dim": 1, "units": 8, "nlayers": 1, "activation": "softmax"},
                    {"input_dim": 1, "latent_dim": 1, "units": 8, "nlayers": 1, "activation": "softmax"})


@pytest.mark.parametrize("loss", ["mae", "huber", "log_cosh"])
@pytest.mark.parametrize("name", ["ms_cfg2_64x4", "ms_cfg3_128x3", "ms_res_48x2_pres", "nif_cfg1_32x2", "ll_plain_32x2_r3", "ms_32x2_r7_si3"])
def test_compile_with_the_other_keras_losses(name, loss):
    """compile(loss='mae' | 'huber' | 'log_cosh') (keras.losses.get; README.md:33 passes 'mse'): loss and every gradient tensor of
    the plain step (the fused-gradient kernel, the 128-wide / resblock / class-NIF / last-layer forms, the f32-input MFMA path of an
    odd block count) and of the Sobolev step against the oracle; evaluate(); one model's loss does not leak into another model of
    the same engine"""
    import nif_amd
    from nif_amd import JacobianLayer, SobolevModel
    m, model, spec, ws, x, y, sw = _make(name)
    y = (3.0 * y).astype(np.float32)                       # |e| on both sides of Huber's delta
    model.compile(nif_amd.Adam(1e-3), loss)
    x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
    rl, rg = O.loss_and_grad(spec, ws, x64, y64, s64, loss=loss)
    assert abs(model.evaluate(x, y, sample_weight=sw) - rl) < 2e-5 * abs(rl)
    m._engine.set_loss(loss)
    l_, g_ = m._engine.loss_and_grad(x, y, sw)
    assert abs(l_ - rl) < 2e-5 * abs(rl), (l_, rl)
    def close(g, ref, bar):      # the metric of test_loss_and_grad_match_oracle: per tensor, relative + 2e-6 of the whole gradient's norm
        off, gn = 0, np.linalg.norm(ref)
        for nm, shp in spec.param_shapes():
            k = int(np.prod(shp))
            err = np.linalg.norm(g[off:off + k].astype(np.float64) - ref[off:off + k])
            assert err <= bar * np.linalg.norm(ref[off:off + k]) + 2e-6 * gn, (nm, err, np.linalg.norm(ref[off:off + k]), gn)
            off += k
    close(g_, O.flatten(rg), 3e-4 if loss != "mae" else 6e-4)        # (mae: sign(e) flips where |e| ~ 1e-7)
    xi = list(range(spec.pi, spec.pi + spec.si))[:2]
    gt = np.random.default_rng(5).uniform(-2, 2, size=(x.shape[0], spec.so, len(xi))).astype(np.float32)
    sl, sg = m._engine.sobolev_loss_and_grad(x, y, gt, xi, 0.1, sw)
    rsl, rsg, _, _ = O.sobolev_loss_and_grad(spec, ws, x64, y64, gt.astype(np.float64), xi, 0.1, s64, loss=loss)
    assert abs(sl - rsl) < 3e-5 * abs(rsl), (sl, rsl)
    close(sg, O.flatten(rsg), 4e-4 if loss != "mae" else 8e-4)
    m._engine.set_loss("mse")
    # the two-output model compiled with the same loss: fit()'s first logged loss is the oracle's
    sm = SobolevModel(JacobianLayer(model, list(range(spec.so)), xi))
    sm.compile(nif_amd.Adam(1e-4), loss, loss_weights=[1.0, 0.1])
    h = sm.fit(x, [y, gt], batch_size=x.shape[0], epochs=1, shuffle=False, verbose=0, sample_weight=sw)
    assert abs(h.history["loss"][0] - rsl) < 3e-5 * abs(rsl)
    # another model of the same engine still evaluates ITS loss (mse)
    other = m.model(); other.compile("adam", "mse")
    wnow = [w.astype(np.float64) for w in other.get_weights()]
    assert abs(other.evaluate(x, y) - O.loss_and_grad(spec, wnow, x64, y64)[0]) < 2e-5 * abs(rl)


def test_sobolev_fit_learns_value_and_slope_of_travelling_wave():
    """Train u(t,x) on values AND du/dx of the closed-form travelling wave; both errors must drop, and the
    derivative error must end lower than with value-only training on the same few points."""
    import nif_amd
    from nif_amd import JacobianLayer, SobolevModel
    kind, cs, cp = _cfg("NIFMultiScale", 32, 2, 16, 1, 1, 1, 1, 1, omega=30.0)
    rng = np.random.default_rng(0)
    N = 512
    t = rng.uniform(-1, 1, size=N); xx = rng.uniform(-1, 1, size=N)
    c, x0, om = 0.6, 0.2, 4.0   # u = exp(-50 s^2) sin(om s), s = x - x0 - c t   (normalised units)
    s_ = xx - x0 - c * t
    u = np.exp(-50 * s_ ** 2) * np.sin(om * s_)
    dudx = np.exp(-50 * s_ ** 2) * (om * np.cos(om * s_) - 100 * s_ * np.sin(om * s_))
    X = np.stack([t, xx], 1).astype(np.float32)
    Y = u[:, None].astype(np.float32)
    G = dudx[:, None, None].astype(np.float32)

    def run(sobolev):
        nif_amd.set_seed(3)
        m = nif_amd.NIFMultiScale(cs, cp)
        base = m.build()
        sm = SobolevModel(JacobianLayer(base, [0], [1]))
        sm.compile(nif_amd.Adam(2e-3), "mse", loss_weights=[1.0, 0.02 if sobolev else 0.0])
        sm._shuffle_seed = 0
        e0 = sm.evaluate(X, [Y, G])
        h = sm.fit(X, [Y, G], batch_size=128, epochs=150, verbose=0)
        uu, jj = sm.predict(X)
        return e0, h.history["loss"], float(np.mean((uu - Y) ** 2)), float(np.mean((jj - G) ** 2))

    e0, hist, mse_u, mse_j = run(True)
    assert hist[-1] < 0.2 * hist[0], (hist[0], hist[-1])
    _, _, mse_u_plain, mse_j_plain = run(False)
    assert mse_j < mse_j_plain, (mse_j, mse_j_plain)


@pytest.mark.parametrize("name", ["ms_64x2_mlp_pnet_r3", "ll_plain_32x2_r3"])
def test_sobolev_fit_with_a_time_derivative_follows_the_oracle(name):
    """SobolevModel.fit with x_index = [t, x]: d/dt is a PARAMETER column (PDE-constrained training differentiates w.r.t. time),
    d/dx a coordinate.  Three Adam steps through fit() against the oracle's Sobolev loss / gradient + Keras-Adam trajectory,
    hypernetwork and last-layer class."""
    import nif_amd
    from nif_amd import JacobianLayer, SobolevModel
    m, model, spec, ws, x, y, sw = _make(name)
    xi = [0, spec.pi]
    rng = np.random.default_rng(21)
    g = rng.uniform(-1, 1, size=(x.shape[0], spec.so, 2)).astype(np.float32)
    sm = SobolevModel(JacobianLayer(model, list(range(spec.so)), xi))
    sm.compile(nif_amd.Adam(1e-3), "mse", loss_weights=[1.0, 0.05])
    h = sm.fit(x, [y, g], batch_size=x.shape[0], epochs=3, shuffle=False, verbose=0)
    th = O.flatten(ws); mm = np.zeros_like(th); vv = np.zeros_like(th)
    f32 = lambda a: float(np.float32(a))
    ls = []
    for t in range(1, 4):
        l_, g_, _, _ = O.sobolev_loss_and_grad(spec, O.unflatten(spec, th), x.astype(np.float64), y.astype(np.float64),
                                               g.astype(np.float64), xi, 0.05)
        ls.append(l_)
        th, mm, vv = O.adam_step(th, O.flatten(g_), mm, vv, t, lr=f32(1e-3), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
    assert np.allclose(h.history["loss"], ls, rtol=2e-3), (h.history["loss"], ls)
    u, J = sm.predict(x)
    assert J.shape == (x.shape[0], spec.so, 2)


@pytest.mark.parametrize("name", ["ms_cfg5_64x4_si2", "ll_plain_32x2_r3", "nif_cfg1_32x2"])
@pytest.mark.parametrize("weighted", [False, True])
def test_sobolev_model_loss_weights_and_total_loss(name, weighted):
    """Keras total loss of the two-output model with loss_weights = [w0, w1], w0 != 1 (r3 refused w0 != 1):
    w0 mse(u) + w1 mse(du/dx) + the UNSCALED regularisation losses.  evaluate() and fit()'s logged loss are that total (r3's
    SobolevModel.evaluate returned the data term alone); two Adam steps follow the oracle (linearity: the data gradient of
    [w0, w1] is w0 x the gradient of [1, w1 / w0])."""
    import nif_amd
    from nif_amd import JacobianLayer, SobolevModel
    (kind, cs, cp), B = CONFIGS[name]
    cp2 = dict(cp); cp2["l2_reg"] = 2e-3
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(5)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    names = [nm for nm, _ in spec.param_shapes()]
    ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * (30.0 if spec.kind == O.KIND_LL else 2.0)).astype(np.float32)
    model = getattr(nif_amd, kind)(cs, cp2).build(); model.set_weights(ws)
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    xi = list(range(spec.pi, spec.pi + spec.si))
    g = rng.uniform(-1, 1, size=(B, spec.so, len(xi))).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32) if weighted else None
    w0, w1 = 0.3, 0.02
    preg, sreg = O.weight_regularizer_coefficients(cs, cp2, spec.kind)

    def total(th):
        w = O.unflatten(spec, th)
        l_, g_, _, _ = O.sobolev_loss_and_grad(spec, w, x.astype(np.float64), y.astype(np.float64), g.astype(np.float64), xi, w1 / w0,
                                               None if sw is None else sw.astype(np.float64))
        lreg, greg = O.weight_regularizer_term(spec, w, preg, sreg)
        return w0 * l_ + lreg, w0 * O.flatten(g_) + O.flatten(greg), lreg

    sm = SobolevModel(JacobianLayer(model, list(range(spec.so)), xi))
    sm.compile(nif_amd.Adam(1e-3), "mse", loss_weights=[w0, w1])
    th = O.flatten([w.astype(np.float64) for w in ws])
    l_ref, _, lreg = total(th)
    assert lreg > 1e-3 * l_ref                                  # the regulariser is visible: it must NOT be scaled by w0
    l_ev = sm.evaluate(x, [y, g], sample_weight=sw)
    assert abs(l_ev - l_ref) < 3e-5 * abs(l_ref), (l_ev, l_ref)
    h = sm.fit(x, [y, g], batch_size=B, epochs=2, shuffle=False, verbose=0, sample_weight=sw, validation_data=(x, [y, g]) if sw is None else (x, [y, g], sw))
    mm = np.zeros_like(th); vv = np.zeros_like(th); ls = []
    f32 = lambda a: float(np.float32(a))
    for t in range(1, 3):
        l_, g_, _ = total(th)
        ls.append(l_)
        th, mm, vv = O.adam_step(th, g_, mm, vv, t, lr=f32(1e-3), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
    assert np.allclose(h.history["loss"], ls, rtol=1e-3), (h.history["loss"], ls)
    assert abs(h.history["val_loss"][-1] - total(th)[0]) < 2e-3 * abs(ls[-1])      # val_loss = the same total, after the step
    # the sub-models are never compiled: Keras' evaluate raises, so does this one
    owner = model._owner
    with pytest.raises(RuntimeError):
        owner.model_p_to_lr().evaluate(x[:, :spec.pi], y)


def test_failed_allocation_leaves_no_sticky_error(monkeypatch):
    """HIP 7 keeps the last failure until it is read: a hipMalloc that fails inside nif_dev_alloc must not surface at the next launch
    check.  fit()'s host-shuffle fallback -- taken when HBM has no room for the gathered copy of the table -- is forced
    by failing the second copy's allocation with a REAL out-of-memory request."""
    import nif_amd
    from nif_amd.engine import Engine
    m, model, spec, ws, x, y, sw = _make("ms_cfg2_64x4")
    e = m._engine
    with pytest.raises(nif_amd.NifError):
        e.alloc(1 << 42)                                   # 16 TiB of floats
    loss, g = e.loss_and_grad(x, y)                        # the next kernel launches are checked with hipGetLastError
    lref, _ = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64))
    assert abs(loss - lref) < 1e-5 * abs(lref)
    real, calls = Engine.alloc, [0]

    def limited(self, n):
        calls[0] += 1
        if calls[0] > 2:                                   # x and y tables fit, the gathered copies do not
            return real(self, 1 << 42)
        return real(self, n)
    monkeypatch.setattr(Engine, "alloc", limited)
    model.compile(nif_amd.Adam(1e-3), "mse")
    model._shuffle_seed = 3
    h = model.fit(x, y, batch_size=128, epochs=2, shuffle=True, verbose=0)
    assert calls[0] >= 3 and np.isfinite(h.history["loss"]).all() and h.history["loss"][1] < h.history["loss"][0] * 1.5


@pytest.mark.parametrize("name", ["ll_plain_32x2_r3", "ll_cfg4_128x2_r10_so3", "ms_res_48x2_pres"])
def test_fit_follows_oracle_adam_and_checkpoint_round_trip(name, tmp_path):
    """Two epochs of fit() (partial last batch) against the oracle's loss/grad + Keras-Adam trajectory for the
    last-layer class (fused LL kernel) and a resblock model, then save_weights / load_weights into a fresh model:
    identical predictions and identical continuation (optimizer slots travel with the checkpoint)."""
    import nif_amd
    m, model, spec, ws, x, y, sw = _make(name)
    model.compile(nif_amd.Adam(1e-3), loss="mse")
    bs = 100
    hist = model.fit(x, y, epochs=2, batch_size=bs, shuffle=False, verbose=0)
    th = O.flatten(ws); mm = np.zeros_like(th); vv = np.zeros_like(th)
    t = 0
    ep_losses = []
    for ep in range(2):
        tot = 0.0
        for b0 in range(0, x.shape[0], bs):
            xb, yb = x[b0:b0 + bs], y[b0:b0 + bs]
            l, g = O.loss_and_grad(spec, O.unflatten(spec, th), xb.astype(np.float64), yb.astype(np.float64))
            t += 1
            th, mm, vv = O.adam_step(th, O.flatten(g), mm, vv, t)
            tot += l * xb.shape[0]
        ep_losses.append(tot / x.shape[0])
    assert np.allclose(hist.history["loss"], ep_losses, rtol=1e-3), (hist.history["loss"], ep_losses)
    assert _rel(O.flatten(model.get_weights()), th) < 2e-4
    ck = str(tmp_path / "ckpt")
    model.save_weights(ck)
    kind = type(m)
    m2 = kind(m.cfg_shape_net, m.cfg_parameter_net) if hasattr(m, "cfg_shape_net") else None
    if m2 is None:
        (k_, cs, cp), _ = CONFIGS[name]
        m2 = getattr(nif_amd, k_)(cs, cp)
    model2 = m2.build()
    model2.compile(nif_amd.Adam(1e-3), loss="mse")
    model2.load_weights(ck)
    assert np.array_equal(model2.predict(x), model.predict(x))
    h1 = model.fit(x, y, epochs=1, batch_size=bs, shuffle=False, verbose=0)
    h2 = model2.fit(x, y, epochs=1, batch_size=bs, shuffle=False, verbose=0)
    assert np.allclose(h1.history["loss"], h2.history["loss"], rtol=1e-6)
    assert np.array_equal(model2.predict(x), model.predict(x))


@pytest.mark.parametrize("which", ["cfg3_ms_128", "cfg4_last_layer", "cfg5_sobolev", "cfg5_sobolev_bf16", "cfg3_ms_128_bf16",
                                   "cfg4_last_layer_bf16", "wide_sobolev_6x128", "wide_sobolev_res_3x128"])
def test_full_size_shard_sum_other_configs(which):
    """The same size-independent properties at the TRUE per-GPU shard sizes and shapes of BASELINE configs[2..4]: the sum over 8
    contiguous shards of [grad | loss] equals the full-batch result, a repeated launch is bit-identical, and the
    oracle on a sample pins the loss (it cannot run 10^5..10^6 points)."""
    import nif_amd
    from nif_amd.engine import DeviceArray
    from nif_amd import distributed as dist
    rng = np.random.default_rng(4)
    xi = None
    if which.startswith("cfg3_ms_128"):          # configs[2]: NIFMultiScale 6x128, 4M points / 8 GPUs
        kind, cs, cp = _cfg("NIFMultiScale", 128, 6, 32, 2, 1, 2, 1, 1, p_act="swish")
        B = 1 << 19
    elif which.startswith("wide_sobolev"):       # r4: the Sobolev step of 128-wide nets on k_sobw<8, .., MODE> at a 2^18-point shard
        kind, cs, cp = _cfg("NIFMultiScale", 128, 3 if "res" in which else 6, 32, 2, 1, 2, 1, 1, p_act="swish", s_res="res" in which)
        B = 1 << 18
        xi = [1, 2]
    elif which.startswith("cfg4_last_layer"):    # configs[3]: last-layer class 128x6, 16M points / 8 GPUs
        kind, cs, cp = _cfg("LL", 128, 6, 32, 2, 10, 3, 3, 1, p_act="swish")
        B = 1 << 21
    else:                               # configs[4]: Sobolev, 64x4 with two coordinates, 8M points / 8 GPUs
        kind, cs, cp = _cfg("NIFMultiScale", 64, 4, 32, 2, 1, 2, 1, 1, p_act="swish")
        B = 1 << 20
        xi = [1, 2]
    spec = O.Spec(kind, cs, cp)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    names = [nm for nm, _ in spec.param_shapes()]
    ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * (30.0 if kind.endswith("Parameterized") else 2.0)).astype(np.float32)
    m = getattr(nif_amd, kind)(cs, cp, mixed_policy="mixed_bfloat16" if which.endswith("bf16") else "float32")
    model = m.build(); model.set_weights(ws)
    e = m._engine
    ncol, so = spec.pi + spec.si, spec.so
    x = rng.uniform(-1, 1, size=(B, ncol)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, so)).astype(np.float32)
    gt = rng.uniform(-1, 1, size=(B, so * 2)).astype(np.float32) if xi else None
    d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size)
    d_x.upload(x); d_y.upload(y)
    d_g = None
    if xi:
        d_g = DeviceArray(e, gt.size); d_g.upload(gt)

    def grad_of(lo, hi, bg):
        if xi:
            e.sobolev_loss_grad_dev(d_x.at(lo * ncol), d_y.at(lo * so), d_g.at(lo * so * 2), None, hi - lo, bg, xi, 0.1)
        else:
            e.loss_grad_dev(d_x.at(lo * ncol), d_y.at(lo * so), None, hi - lo, bg)
        buf = DeviceArray.__new__(DeviceArray)
        buf.engine, buf.n, buf.ptr = e, e.n_params + 1, e.grad_dev_ptr()
        out = buf.download()
        buf.ptr = None
        return out.astype(np.float64)

    full = grad_of(0, B, B)
    assert np.array_equal(full, grad_of(0, B, B))
    acc = np.zeros_like(full)
    for r_ in range(8):
        lo, hi = dist.shard_bounds(B, 8, r_)
        acc += grad_of(lo, hi, B)
    assert abs(acc[-1] - full[-1]) < 2e-6 * abs(full[-1])
    assert np.linalg.norm(acc[:-1] - full[:-1]) < 2e-5 * np.linalg.norm(full[:-1])
    ws64 = [w.astype(np.float64) for w in ws]
    n_s = 2048
    bf = which.endswith("bf16")
    if xi:
        # under the policy: the oracle that rounds where k_sob<..., BF = 2> rounds (sobolev_planes_loss_and_grad)
        lref, gref = O.sobolev_planes_loss_and_grad(spec, ws64, x[:n_s].astype(np.float64), y[:n_s].astype(np.float64),
                                                    gt[:n_s].astype(np.float64), xi, 0.1, rnd=O.bf16_round if bf else None,
                                                    stash_bf16=bf and _stash_bf16(spec, xi))[:2]
    elif bf:      # the policy on the 128-wide nets: bf16 dL/da stash rows through k_gw8<R, DAB> (r3)
        fn = O.ll_policy_loss_and_grad if spec.kind == O.KIND_LL else O.planes_loss_and_grad
        lref, gref = fn(spec, ws64, x[:n_s].astype(np.float64), y[:n_s].astype(np.float64), rnd=O.bf16_round,
                        stash_bf16=_stash_bf16(spec), stash_ph16=_stash_ph16(spec))[:2]
    else:
        lref, gref = O.loss_and_grad(spec, ws64, x[:n_s].astype(np.float64), y[:n_s].astype(np.float64))
    sub = grad_of(0, n_s, n_s)
    assert abs(sub[-1] - lref) < (5e-4 if bf else 2e-5) * abs(lref), (sub[-1], lref)
    if xi or bf:
        rel = _per_tensor_rel(spec, sub[:-1], O.flatten(gref))
        # (6 x 128 under the policy, this draw: the policy itself sits 9 % from exact arithmetic -- a flipped bf16 rounding of one
        # activation then moves the gradient by 4e-3 of its norm; another draw of the same net: 3.5e-4)
        bar = 3e-3 if bf else 3e-4
        if bf and not xi and max(rel.values()) >= bar:
            # r3 held the 128-wide nets to a blanket 6e-3 because "one draw sits there: a flipped bf16 rounding of one activation".
            # The explanation is now ASSERTED: the emulating oracle itself is evaluated on weights one fp32 ulp away; where ITS
            # gradient tensor moves by s, the kernel may sit 3 s from it -- and nowhere else above the typical 3e-3
            rng2 = np.random.default_rng(99)
            ws_n = [np.nextafter(w.astype(np.float32), np.float32(np.inf) * rng2.choice([-1.0, 1.0], size=w.shape).astype(np.float32))
                    .astype(np.float64) for w in ws64]
            gref_n = fn(spec, ws_n, x[:n_s].astype(np.float64), y[:n_s].astype(np.float64), rnd=O.bf16_round,
                        stash_bf16=_stash_bf16(spec), stash_ph16=_stash_ph16(spec))[1]
            sens = _per_tensor_rel(spec, O.flatten(gref_n), O.flatten(gref))
            for nm, v in rel.items():
                assert v < max(bar, 3.0 * sens[nm]), (nm, v, sens[nm])
            assert max(rel.values()) < 1e-2, rel
        else:
            assert max(rel.values()) < bar, rel
    d_x.free(); d_y.free()
    if d_g is not None:
        d_g.free()


@pytest.mark.parametrize("B", [1, 15, 17, 33, 63])
@pytest.mark.parametrize("name", ["ms_cfg2_64x4", "ll_plain_32x2_r3", "nif_cfg1_32x2"])
def test_ragged_tiny_batches_on_the_16_point_tile_kernels(name, B):
    """Batches smaller than / not a multiple of the 16- and 32-point tiles (clamped prefetch, masked tails)."""
    m, model, spec, ws, x, y, sw = _make(name)
    x, y, sw = x[:B], y[:B], sw[:B]
    u = model.predict(x)
    ref = O.forward(spec, ws, x.astype(np.float64))
    # a handful of O(0.1..1) values: absolute bar at the fp32 level of the O(1) field (a rel-L2 over one small value
    # only measures its conditioning); the same rows inside a big batch give bit-identical outputs
    assert np.abs(u - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    assert np.array_equal(u, model.predict(np.concatenate([x, x, x, x]))[:B])
    loss, grad = m._engine.loss_and_grad(x, y, sw)
    rl, rg = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))
    assert abs(loss - rl) <= 1e-5 * abs(rl)
    assert _rel(grad, O.flatten(rg)) < 2e-4


# ---- the gradient-precision story: bf16-split products vs the f32-input MFMA path vs the oracle ----------
def _per_tensor_rel(spec, g, gref):
    out, off = {}, 0
    gn = np.linalg.norm(gref)
    for nm, shp in spec.param_shapes():
        k = int(np.prod(shp))
        a, b = g[off:off + k].astype(np.float64), gref[off:off + k].astype(np.float64)
        out[nm] = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-6 * gn))
        off += k
    return out


def _oracle_grad_chunked(spec, ws, x, y, chunk=2048):
    """full-batch oracle loss/gradient as a sum over chunks (the oracle materialises [chunk, po] tensors in fp64)"""
    B = x.shape[0]
    loss, g = 0.0, None
    for lo in range(0, B, chunk):
        l, gr = O.loss_and_grad(spec, ws, x[lo:lo + chunk].astype(np.float64), y[lo:lo + chunk].astype(np.float64), batch_global=B)
        loss += l
        g = O.flatten(gr) if g is None else g + O.flatten(gr)
    return loss, g


def test_full_size_gradient_split_path_vs_fp32_mfma_vs_oracle():
    """The training step's products run as 16-bit pairs (r5: forward and data adjoint as HALF (hi, lo) pairs, three products, fp32-grade;
    weight gradients as bf16 hi/lo pairs).  At the benchmark size (2^20 points, 4x64): every tensor of the flat gradient against the
    same step on the f32-input MFMAs (`fp32_mfma` option: fmaf-exact products), and both against the fp64 oracle on a 65 536-point
    sub-batch.  Bars re-tightened in r5 to ~3x what tools/exp/graderr.py measures (default path vs oracle: flat 1.3e-6, worst tensor
    1.7e-5 -- the ParameterNet's first layer, k_pnet_bwg's bf16 pairs; fp32-MFMA path: 2.4e-7 / 2.9e-6; r4's bars: 1e-4 / 2e-4)."""
    m, model, x, y = _full_size_setup()
    e = m._engine
    spec = O.Spec("NIFMultiScale", m.cfg_shape_net, m.cfg_parameter_net)
    ws = [w.astype(np.float64) for w in model.get_weights()]
    ls, gs = e.loss_and_grad(x, y)
    e.set_option("fp32_mfma", 1)
    lf, gf = e.loss_and_grad(x, y)
    e.set_option("fp32_mfma", 0)
    assert abs(ls - lf) <= 5e-7 * abs(lf), (ls, lf)
    rel = _per_tensor_rel(spec, gs, gf)
    assert max(rel.values()) < 5e-5, rel
    assert _rel(gs, gf.astype(np.float64)) < 5e-6
    n_s = 1 << 16
    lo_, go_ = _oracle_grad_chunked(spec, ws, x[:n_s], y[:n_s])
    for fp32 in (0, 1):
        e.set_option("fp32_mfma", fp32)
        l_, g_ = e.loss_and_grad(x[:n_s], y[:n_s])
        assert abs(l_ - lo_) <= 1e-6 * abs(lo_), (fp32, l_, lo_)
        rel = _per_tensor_rel(spec, g_, go_)
        assert max(rel.values()) < (5e-5 if fp32 == 0 else 1e-5), (fp32, rel)
        assert _rel(g_, go_) < (5e-6 if fp32 == 0 else 1e-6), (fp32, _rel(g_, go_))
    e.set_option("fp32_mfma", 0)


def test_adam_trajectory_200_steps_split_vs_fp32_mfma_vs_oracle():
    """200 full-batch Adam steps on the reference's bundled travelling-wave dataset with the benchmark model: loss curves
    of the bf16-split path, the f32-input MFMA path (fmaf-exact products) and the fp64 oracle.

    Measured (tools/exp/traj_probe.py): all three agree to <= 2e-5 until the optimisation enters its steep phase
    (the loss falls 100x within ~40 steps); from there ANY fp32 evaluation drifts from the fp64 trajectory -- the exact-fp32
    path as much as the split path (2.8e-3 vs 2.5e-3 at lr = 3e-4; at lr = 1e-3 both reach O(1) around step 75).  The
    dynamics, not the product splitting, set that bound.  So: 1e-4 agreement over the first 100 steps, and over all 200
    steps the split path must stay within the drift the exact-fp32 path shows itself."""
    import os
    import nif_amd
    import bench
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "traveling_wave.npz"))["data"]
    data, _, _ = O.standard_normalize(d.astype(np.float64))
    x, y = data[:, :2].astype(np.float32), data[:, 2:3].astype(np.float32)
    steps, lr = 200, 3e-4
    curves = {}
    ws0 = None
    for mode in ("split", "fp32"):
        nif_amd.set_seed(21)
        m = nif_amd.NIFMultiScale(bench.CFG_SHAPE, bench.CFG_PARAM)
        model = m.build()
        ws0 = [w.astype(np.float64) for w in model.get_weights()]
        e = m._engine
        e.set_option("fp32_mfma", 1 if mode == "fp32" else 0)
        adam = nif_amd.Adam(lr).as_struct()
        d_x, d_y = e.alloc(x.size), e.alloc(y.size)
        d_x.upload(x); d_y.upload(y)
        losses = []
        for _ in range(steps):
            e.loss_grad_dev(d_x.at(0), d_y.at(0), None, x.shape[0], x.shape[0])
            losses.append(e.last_loss())
            e.adam_step_dev(adam)
        curves[mode] = np.array(losses)
    spec = O.Spec("NIFMultiScale", bench.CFG_SHAPE, bench.CFG_PARAM)
    th = O.flatten(ws0); mm = np.zeros_like(th); vv = np.zeros_like(th)
    f32 = lambda a: float(np.float32(a))
    ref = []
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    for t in range(1, steps + 1):
        l, g = O.loss_and_grad(spec, O.unflatten(spec, th), x64, y64)
        ref.append(l)
        th, mm, vv = O.adam_step(th, O.flatten(g), mm, vv, t, lr=f32(lr), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
    ref = np.array(ref)
    assert ref[-1] < 1e-2 * ref[0]                 # it does train (1.3 -> 5.6e-4)
    dev = {mode: np.abs(curves[mode] - ref) / ref for mode in curves}
    for mode in ("split", "fp32"):
        assert dev[mode][:100].max() < 1e-4, (mode, float(dev[mode][:100].max()))
        assert dev[mode].max() < 2e-2, (mode, float(dev[mode].max()), int(dev[mode].argmax()))
    assert dev["split"].max() <= 3.0 * dev["fp32"].max() + 1e-3, (float(dev["split"].max()), float(dev["fp32"].max()))


# ---- mixed_bfloat16 policy (BASELINE configs[4] names bf16; reference model.py:73,101-105) ----------------------------------
BF16 = ["ms_cfg2_64x4", "ms_cfg5_64x4_si2", "ms_64x2_mlp_pnet_r3", "ms_64x3_r3_so2_b33", "nif_cfg1_32x2", "ms_64x8",
        "ms_cfg3_128x3", "ms_pad_120x3"]      # (r5: the 128-wide policy step = k_snet4<8, .., PR = 1> with the 16-bit phase stash + k_gw8<1, true, true>)


def _snet6_shape(spec):
    """the plain training step of this shape runs on k_snet6 (csrc/k_snet6.hip snet6_supported): plain-SIREN NIFMultiScale, 49..64
    units, latent_dim 1, at most four hidden matrices, si / so <= 3"""
    return (spec.kind == O.KIND_MS and not spec.s_res and (spec.n + 15) // 16 == 4 and spec.r == 1 and 1 <= spec.n_hidden_mats <= 4
            and spec.si <= 3 and spec.so <= 3)


def _stash_bf16(spec, xi=None):
    """does the step of this shape keep its hidden-layer dL/da stash rows in bf16 under mixed_bfloat16?  The bf16 kernels' widths:
    two or four 16-feature blocks (k_gw_lds<DAB>), eight with at most two planes per layer (k_gw8<R, DAB>: the last-layer class,
    latent_dim 1); plain step: k_snet4<PR>; Sobolev step: only k_sobw<PR> (plain SIREN, coordinate seeds, <= 64 units)"""
    nbl = (spec.n + 15) // 16
    if xi is None and _snet6_shape(spec):
        return False        # late r4: the fused-gradient kernel's policy forms -- no stash at all, weight-gradient sums from exact (hi, lo) rows
    if spec.kind == O.KIND_LL:
        return xi is None and nbl in (2, 4, 8)
    if not (nbl in (2, 4) or (nbl == 8 and spec.r <= 1)):
        return False
    if xi is None:
        return True
    # (r4: k_sobw takes resblock nets and the 65..128-unit nets too)
    # (... and class NIF)
    return (spec.kind in (O.KIND_MS, O.KIND_NIF) and (nbl in (2, 4) or (nbl == 8 and spec.r <= 1)) and 1 <= len(xi) <= 3
            and all(j >= spec.pi for j in xi) and spec.r >= 1)


def _stash_ph16(spec):
    """... and the hidden matrices' INPUT rows as 16-bit phases (r5: plain step of the 128-wide plain-SIREN nets, k_snet4<8, .., PR = 1>
    -> k_gw8<R, true, true>; NIF_H_PH16=0 switches it off)"""
    import os
    return (os.environ.get("NIF_H_PH16", "1") != "0" and _stash_bf16(spec) and (spec.n + 15) // 16 == 8 and not spec.s_res
            and spec.kind in (O.KIND_LL, O.KIND_MS))


def _make_policy(name, policy, boost=1.0):
    import nif_amd
    (kind, cs, cp), B = CONFIGS[name] if isinstance(name, str) else name
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    if kind == "NIFMultiScale":
        names = [nm for nm, _ in spec.param_shapes()]
        ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * boost).astype(np.float32)
    m = getattr(nif_amd, kind)(cs, cp, mixed_policy=policy)
    model = m.build(); model.set_weights(ws)
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32)
    return m, model, spec, [w.astype(np.float64) for w in ws], x, y, sw


@pytest.mark.parametrize("name", BF16)
def test_mixed_bfloat16_policy_matches_the_oracle_with_the_same_casts(name):
    """Policy of the build (include/nif_hip.h nif_policy): operands of the hidden n x n products rounded to bf16, fp32
    accumulation, everything else fp32.  The oracle's plane formulation with bf16 rounding at the same points
    (oracle/nif_oracle.py planes_loss_and_grad(rnd=bf16_round)) pins it: predictions and loss to 5e-4 (a 1e-7
    difference of an fp32 activation flips a bf16 rounding now and then), gradients to 2e-3 per tensor; against the
    exact fp64 oracle the policy itself costs ~1e-3..1e-2, reported by the looser second bar."""
    m, model, spec, ws, x, y, sw = _make_policy(name, "mixed_bfloat16")
    assert m.compute_Dtype == "bfloat16" and m.variable_Dtype == "float32" and m.mixed_policy_name == "mixed_bfloat16"
    x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
    rl, rg, ru = O.planes_loss_and_grad(spec, ws, x64, y64, s64, rnd=O.bf16_round, stash_bf16=_stash_bf16(spec), stash_ph16=_stash_ph16(spec))
    u = model.predict(x)
    assert _rel(u, ru) < 5e-4, _rel(u, ru)
    loss, g = m._engine.loss_and_grad(x, y, sw)
    assert abs(loss - rl) <= 5e-4 * abs(rl), (loss, rl)
    rel = _per_tensor_rel(spec, g, O.flatten(rg))
    assert max(rel.values()) < 2e-3, rel
    if _stash_bf16(spec):
        # the bf16 dL/da stash is really what the weight-gradient sums took: the oracle without it is the farther one
        # (measured: 1.3e-5 .. 3.3e-4 from the matching form, 5e-4 .. 1.4e-3 from the other)
        rg0 = O.planes_loss_and_grad(spec, ws, x64, y64, s64, rnd=O.bf16_round, stash_bf16=False)[1]
        assert _rel(g, O.flatten(rg)) < 0.6 * _rel(g, O.flatten(rg0)), (_rel(g, O.flatten(rg)), _rel(g, O.flatten(rg0)))
    # distance of the policy from exact arithmetic: present (it IS a different computation) but bounded
    el, eg = O.loss_and_grad(spec, ws, x64, y64, s64)
    d_u = _rel(u, O.forward(spec, ws, x64))
    assert 1e-6 < d_u < 5e-2, d_u
    assert _rel(g, O.flatten(eg)) < 0.2
    # the fp32 model on the same weights is unaffected by the other one's policy
    m32, model32, *_ = _make_policy(name, "float32")
    assert _rel(model32.predict(x), O.forward(spec, ws, x64)) < 1e-5


@pytest.mark.parametrize("name", ["ll_plain_32x2_r3", "ll_cfg4_128x2_r10_so3", "ll_res_64x1_r4_so2", "ll_96x2_r5", "ll_64x3_r5_so2",
                                  "ll_cfg4_128x6_r10_so3", "ll_pad_120x2_r6_so2"])
def test_mixed_bfloat16_policy_on_the_last_layer_class(name):
    """the policy on NIFMultiScaleLastLayerParameterized: operands of the SHARED hidden n x n products rounded to bf16 (one product,
    fp32 accumulation), the phi layer / Dot / loss / weight-gradient sums fp32 -- against the oracle with the same casts
    (ll_policy_loss_and_grad(rnd=bf16_round)); the float32 model of the same class is untouched"""
    m, model, spec, ws, x, y, sw = _make_policy(name, "mixed_bfloat16")
    names = [nm for nm, _ in spec.param_shapes()]
    ws[names.index("pnet_last_w")] = ws[names.index("pnet_last_w")] * 30.0       # weight_init_factor 0.01 makes the r x r map ~0
    model.set_weights([w.astype(np.float32) for w in ws])
    ws = [w.astype(np.float32).astype(np.float64) for w in ws]
    assert m.compute_Dtype == "bfloat16"
    x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
    rl, rg, ru = O.ll_policy_loss_and_grad(spec, ws, x64, y64, s64, rnd=O.bf16_round, stash_bf16=_stash_bf16(spec), stash_ph16=_stash_ph16(spec))
    u = model.predict(x)
    assert _rel(u, ru) < 5e-4, _rel(u, ru)
    loss, g = m._engine.loss_and_grad(x, y, sw)
    assert abs(loss - rl) <= 5e-4 * abs(rl), (loss, rl)
    rel = _per_tensor_rel(spec, g, O.flatten(rg))
    assert max(rel.values()) < 3e-3, rel
    if _stash_bf16(spec):      # the bf16 dL/da stash (k_gw_lds<DAB> / k_gw8<0, DAB>) is what the hidden sums took
        rg0 = O.ll_policy_loss_and_grad(spec, ws, x64, y64, s64, rnd=O.bf16_round, stash_bf16=False)[1]
        assert _rel(g, O.flatten(rg)) < 0.7 * _rel(g, O.flatten(rg0)), (_rel(g, O.flatten(rg)), _rel(g, O.flatten(rg0)))
    d_u = _rel(u, O.forward(spec, ws, x64))
    assert 1e-6 < d_u < 5e-2, d_u                     # the policy IS a different computation
    m32, model32, *_ = _make_policy(name, "float32")
    model32.set_weights([w.astype(np.float32) for w in ws])
    assert _rel(model32.predict(x), O.forward(spec, ws, x64)) < 1e-5


def test_mixed_bfloat16_training_and_sobolev_step():
    """fit() under the policy follows the emulating oracle's Adam trajectory; the Sobolev step (configs[4]) runs on the
    single-product planes and stays within the policy's distance of the exact oracle"""
    import nif_amd
    m, model, spec, ws, x, y, sw = _make_policy("ms_cfg5_64x4_si2", "mixed_bfloat16")
    model.compile(nif_amd.Adam(2e-4), "mse")
    h = model.fit(x, y, epochs=3, batch_size=x.shape[0], shuffle=False, verbose=0)
    th = O.flatten(ws); mm = np.zeros_like(th); vv = np.zeros_like(th)
    f32 = lambda a: float(np.float32(a))
    losses = []
    for t in range(1, 4):
        l, g, _ = O.planes_loss_and_grad(spec, O.unflatten(spec, th), x.astype(np.float64), y.astype(np.float64), rnd=O.bf16_round,
                                         stash_bf16=_stash_bf16(spec))
        losses.append(l)
        th, mm, vv = O.adam_step(th, O.flatten(g), mm, vv, t, lr=f32(2e-4), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
    assert np.allclose(h.history["loss"], losses, rtol=2e-3), (h.history["loss"], losses)
    # Sobolev under the policy
    m, model, spec, ws, x, y, sw = _make_policy("ms_cfg5_64x4_si2", "mixed_bfloat16")
    xi = [1, 2]
    gt = np.random.default_rng(3).uniform(-1, 1, size=(x.shape[0], 1, 2)).astype(np.float32)
    loss, grad = m._engine.sobolev_loss_and_grad(x, y, gt, xi, 0.05, sw)
    x64, y64, g64, s64 = x.astype(np.float64), y.astype(np.float64), gt.astype(np.float64), sw.astype(np.float64)
    # cast for cast: the oracle's plane formulation rounds h_q, (w0 M^(k)), dL/da and nu^d where the kernels
    # (training: k_sobw<PR>, one wave per stream; predictions: k_sob<TRAIN = false, BF = 2>) round them; same bars as the plain step
    pl, pg, pu, pJ = O.sobolev_planes_loss_and_grad(spec, ws, x64, y64, g64, xi, 0.05, s64, rnd=O.bf16_round,
                                                    stash_bf16=_stash_bf16(spec, xi))
    assert abs(loss - pl) <= 5e-4 * abs(pl), (loss, pl)
    rel = _per_tensor_rel(spec, grad, O.flatten(pg))
    assert max(rel.values()) < 3e-3, rel
    u_p, J_p = m._engine.sobolev_forward(x, xi)
    # (a 1e-7 difference of an fp32 operand flips a bf16 rounding now and then: 2^-9 of that element)
    assert _rel(u_p, pu) < 1e-3 and _rel(J_p.reshape(pJ.shape), pJ) < 2e-3, (_rel(u_p, pu), _rel(J_p.reshape(pJ.shape), pJ))
    rl, rg, ru, rJ = O.sobolev_loss_and_grad(spec, ws, x64, y64, g64, xi, 0.05, s64)
    assert 1e-5 * abs(rl) < abs(pl - rl) < 5e-2 * abs(rl)          # the policy's own distance from exact arithmetic
    m32, model32, *_ = _make_policy("ms_cfg5_64x4_si2", "float32")
    l32, g32 = m32._engine.sobolev_loss_and_grad(x, y, gt, xi, 0.05, sw)
    assert abs(l32 - rl) < 2e-5 * abs(rl) and loss != l32          # the policy really changes the arithmetic
    with pytest.raises(NotImplementedError):
        nif_amd.NIFMultiScale(*CONFIGS["ms_cfg2_64x4"][0][1:], mixed_policy="float16")       # (r4: 'mixed_float16' is built; the pure 16-bit policies change the variable dtype)


@pytest.mark.parametrize("name", ["nif_cfg1_32x2", "ms_res_64x2", "ms_64x2_r1_so4", "ll_plain_32x2_r3", "ms_cfg5_64x4_si2",
                                  "ms_cfg3_128x3", "ms_96x2_r2", "ms_res_128x1_nst128"])      # r4: the wide k_sobw<6 | 8, PR> forms
def test_sobolev_step_under_the_policy_cast_for_cast(name):
    """configs[4] beyond its own shape: class NIF (skip connections, swish), a resblock net, several outputs -- the Sobolev
    step under mixed_bfloat16 against the oracle that rounds where k_sob<..., BF = 2> and k_sobw<PR> round (5e-4 loss / predictions, 3e-3 per
    gradient tensor); the last-layer class keeps exact products in k_sob<LL> under the policy: held to the exact
    oracle at the float32 bars"""
    (kind, cs, cp), _ = CONFIGS[name]
    m, model, spec, ws, x, y, sw = _make_policy(name, "mixed_bfloat16")
    x, y, sw = x[:200], y[:200], sw[:200]
    xi = list(range(spec.pi, spec.pi + spec.si))[:2]
    gt = np.random.default_rng(5).uniform(-1, 1, size=(x.shape[0], spec.so, len(xi))).astype(np.float32)
    loss, grad = m._engine.sobolev_loss_and_grad(x, y, gt, xi, 0.1, sw)
    x64, y64, g64, s64 = x.astype(np.float64), y.astype(np.float64), gt.astype(np.float64), sw.astype(np.float64)
    if spec.kind == O.KIND_LL:
        rl, rg, ru, rJ = O.sobolev_loss_and_grad(spec, ws, x64, y64, g64, xi, 0.1, s64)
        bar_l, bar_g = 2e-5, 3e-4
    else:
        rl, rg, ru, rJ = O.sobolev_planes_loss_and_grad(spec, ws, x64, y64, g64, xi, 0.1, s64, rnd=O.bf16_round,
                                                        stash_bf16=_stash_bf16(spec, xi))
        bar_l, bar_g = 5e-4, 3e-3
    assert abs(loss - rl) <= bar_l * abs(rl), (loss, rl)
    rel = _per_tensor_rel(spec, grad, O.flatten(rg))
    if max(rel.values()) >= bar_g and spec.kind != O.KIND_LL:
        # a tensor above the typical bar is accepted only where the EMULATING ORACLE itself moves that much when its weights move by
        # one fp32 ulp (a different set of bf16 roundings flips): the kernel may sit 3 s from it, and nowhere above 1e-2
        ws_n = _one_ulp_weights(ws)
        rg_n = O.sobolev_planes_loss_and_grad(spec, ws_n, x64, y64, g64, xi, 0.1, s64, rnd=O.bf16_round, stash_bf16=_stash_bf16(spec, xi))[1]
        sens = _per_tensor_rel(spec, O.flatten(rg_n), O.flatten(rg))
        for nm, v in rel.items():
            assert v < max(bar_g, 3.0 * sens[nm]) and v < 1e-2, (nm, v, sens[nm])
    else:
        assert max(rel.values()) < bar_g, rel


def _one_ulp_weights(ws64, seed=99):
    rng2 = np.random.default_rng(seed)
    return [np.nextafter(w.astype(np.float32), np.float32(np.inf) * rng2.choice([-1.0, 1.0], size=w.shape).astype(np.float32)).astype(np.float64)
            for w in ws64]


@pytest.mark.parametrize("name", ["ms_cfg5_64x4_si2", "ll_cfg4_128x2_r10_so3"])
def test_hessian_dev_entry_point_at_2_to_the_17_points(name):
    """nif_hessian_dev (r3): inputs and outputs resident in HBM, gather / last-layer contraction in kernels -- 131 072 points, every
    input column (parameter column included), equal to the host entry point on a sample and to the oracle on its first rows"""
    import nif_amd
    from nif_amd.engine import DeviceArray
    m, model, spec, ws, x, y, sw = _make(name, boost=1.0)
    e = m._engine
    B = 1 << 17
    rng = np.random.default_rng(2)
    xb = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    yi = list(range(spec.so))[::-1]
    xi = list(range(spec.pi + spec.si))
    ny, nx = len(yi), len(xi)
    d_x = DeviceArray(e, xb.size); d_x.upload(xb)
    d_y, d_d, d_h = DeviceArray(e, B * spec.so), DeviceArray(e, B * ny * nx), DeviceArray(e, B * ny * nx * nx)
    e.hessian_dev(d_x.at(0), B, yi, xi, d_y.at(0), d_d.at(0), d_h.at(0))
    e.sync()
    yv, J, H = d_y.download().reshape(B, spec.so), d_d.download().reshape(B, ny, nx), d_h.download().reshape(B, ny, nx, nx)
    assert np.isfinite(H).all() and np.array_equal(H, np.swapaxes(H, 2, 3))
    sel = np.r_[0:300, B - 300:B]
    y2, J2, H2 = e.hessian(xb[sel], yi, xi)
    assert np.array_equal(yv[sel], y2) and np.array_equal(J[sel], J2) and np.array_equal(H[sel], H2)
    ur, Jr, Hr = O.hessian_analytic(spec, ws, xb[:200].astype(np.float64), yi, xi)
    assert _rel(yv[:200], ur) < 1e-5 and _rel(J[:200], Jr) < 2e-5 and _rel(H[:200], Hr) < 1e-4
    for a_ in (d_x, d_y, d_d, d_h):
        a_.free()


# ---- round 3: the derivative layers and the optional regularisers take every shape the plain step trains ----------------------
WIDE = {
    # 128 units, 6 matrices, latent_dim 5: two fp32 planes + the small hyper-vectors exceed 160 KB of LDS -> single plane buffer
    "ms_128x6_r5_si2": _cfg("NIFMultiScale", 128, 6, 32, 2, 5, 2, 1, 1),
    # ParameterNet 96 units / 5 hidden matrices, two parameter inputs: k_pjac<128>, nm > 4
    "ms_64x2_pnet96x5_pi2": _cfg("NIFMultiScale", 64, 2, 96, 5, 2, 1, 1, 2, p_act="tanh"),
}


def test_derivative_layers_take_wide_nets_with_larger_latents():
    """JacobianLayer / HessianLayer / the Sobolev step on a shape whose two weight-plane buffers do not fit
    the LDS next to the small hyper-vectors (r2: refused) -- one plane buffer instead; against the oracle at the usual bars"""
    import nif_amd
    kind, cs, cp = WIDE["ms_128x6_r5_si2"]
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    m = getattr(nif_amd, kind)(cs, cp)
    model = m.build(); model.set_weights(ws)
    ws64 = [w.astype(np.float64) for w in ws]
    B = 150
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    x64 = x.astype(np.float64)
    xi = [1, 2]
    yv, J = nif_amd.JacobianLayer(model, [0], xi)(x)
    ur, Jr = O.jacobian_analytic(spec, ws64, x64, [0], xi)
    assert _rel(yv, ur) < 1e-5 and _rel(J, Jr) < 2e-5, (_rel(yv, ur), _rel(J, Jr))
    yv, J2, H = nif_amd.HessianLayer(model, [0], xi)(x)
    _, _, Hr = O.hessian_analytic(spec, ws64, x64, [0], xi)
    assert _rel(H, Hr) < 1e-4, _rel(H, Hr)
    gt = rng.uniform(-1, 1, size=(B, 1, 2)).astype(np.float32)
    loss, grad = m._engine.sobolev_loss_and_grad(x, y, gt, xi, 0.1, None)
    rl, rg, _, _ = O.sobolev_loss_and_grad(spec, ws64, x64, y.astype(np.float64), gt.astype(np.float64), xi, 0.1)
    assert abs(loss - rl) <= 2e-5 * abs(rl), (loss, rl)
    rel = _per_tensor_rel(spec, grad, O.flatten(rg))
    assert max(rel.values()) < 3e-4, rel


def test_parameter_column_terms_take_wide_deep_parameter_nets():
    """k_pjac / k_pjac2 beyond 64 units and 4 hidden matrices (r2: refused): latent Jacobian regulariser, Hessian and Sobolev
    step with parameter columns on a 96 x 5 ParameterNet with two inputs; activity regulariser at latent_dim 12"""
    import nif_amd
    kind, cs, cp = WIDE["ms_64x2_pnet96x5_pi2"]
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(1)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    ws64 = [w.astype(np.float64) for w in ws]
    l1 = 0.05
    m = getattr(nif_amd, kind)(cs, dict(cp, jac_reg=l1))
    model = m.build(); model.set_weights(ws)
    B = 130
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    m._engine.set_jac_regularizer(l1)
    loss, g = m._engine.loss_and_grad(x, y)
    l0, g0 = O.loss_and_grad(spec, ws64, x64, y64)
    lj, gj = O.jac_reg_loss_and_grad(spec, ws64, x64[:, :spec.pi], l1)
    assert lj > 1e-6 * l0 and abs(loss - (l0 + lj)) <= 1e-5 * (l0 + lj), (loss, l0, lj)
    rel = _per_tensor_rel(spec, g, O.flatten(g0) + O.flatten(gj))
    assert max(rel.values()) < 3e-4, rel
    m._engine.set_jac_regularizer(0.0)
    xa = [0, 1, 2]                                     # both parameter columns and the coordinate
    yv, J, H = nif_amd.HessianLayer(model, [0], xa)(x)
    ur, Jr, Hr = O.hessian_analytic(spec, ws64, x64, [0], xa)
    assert _rel(J, Jr) < 2e-5 and _rel(H, Hr) < 1e-4, (_rel(J, Jr), _rel(H, Hr))
    gt = rng.uniform(-1, 1, size=(B, 1, 2)).astype(np.float32)
    ls, gs = m._engine.sobolev_loss_and_grad(x, y, gt, [0, 2], 0.1, None)
    rl, rg, _, _ = O.sobolev_loss_and_grad(spec, ws64, x64, y64, gt.astype(np.float64), [0, 2], 0.1)
    assert abs(ls - rl) <= 2e-5 * abs(rl)
    assert max(_per_tensor_rel(spec, gs, O.flatten(rg)).values()) < 3e-4
    # activity regulariser at latent_dim 12 (r2: <= 8)
    kind2, cs2, cp2 = _cfg("NIFMultiScale", 32, 2, 32, 1, 12, 1, 1, 1)
    spec2 = O.Spec(kind2, cs2, cp2)
    ws2 = O.init_weights(spec2, rng, dtype=np.float32)
    m2 = getattr(nif_amd, kind2)(cs2, dict(cp2, act_l2_reg=2e-3))
    model2 = m2.build(); model2.set_weights(ws2)
    x2 = rng.uniform(-1, 1, size=(B, spec2.pi + spec2.si)).astype(np.float32)
    l2, g2 = m2._engine.loss_and_grad(x2, y)
    r2l, r2g = O.loss_and_grad(spec2, [w.astype(np.float64) for w in ws2], x2.astype(np.float64), y64, act_reg=(0.0, 2e-3))
    assert abs(l2 - r2l) <= 1e-5 * abs(r2l) and max(_per_tensor_rel(spec2, g2, O.flatten(r2g)).values()) < 3e-4


# ---- HessianLayer (N3; reference gradient.py:130-180) ------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["nif_cfg1_32x2", "nif_pad_n30_tanh_r2_so2", "ms_cfg2_64x4", "ms_64x2_mlp_pnet_r3", "ms_res_48x2_pres",
                                  "ms_cfg5_64x4_si2", "ms_cfg3_128x3", "ms_32x2_r7_si3", "ms_96x2_r2", "ll_plain_32x2_r3",
                                  "ll_cfg4_128x2_r10_so3", "ll_res_64x1_r4_so2", "ll_96x2_r5", "ll_res_48x2_r4"])
def test_hessian_layer_matches_oracle(name):
    """(y, dy/dx, d2y/dx2) for the coordinate columns: second-order forward-mode tangents in one kernel per coordinate pair
    against the fp64 oracle (pinned by torch double-backward in tests/test_oracle.py).  With w0 = 30 the second derivatives
    are O(w0^2) larger than the field: tolerance 1e-4 of the Hessian's own scale."""
    import nif_amd
    m, model, spec, ws, x, y, sw = _make(name, boost=1.0)
    x = x[:200]
    yi = list(range(spec.so))
    xi = list(range(spec.pi, spec.pi + spec.si))
    yv, J, H = nif_amd.HessianLayer(model, yi, xi)(x)
    ur, Jr, Hr = O.hessian_analytic(spec, ws, x.astype(np.float64), yi, xi)
    assert yv.shape == (x.shape[0], spec.so) and J.shape == Jr.shape and H.shape == (x.shape[0], spec.so, spec.si, spec.si)
    assert _rel(yv, ur) < 1e-5 and _rel(J, Jr) < 2e-5
    assert _rel(H, Hr) < 1e-4, _rel(H, Hr)
    assert np.array_equal(H, np.swapaxes(H, 2, 3))
    # index selection like tf.gather: a single output / a single (repeated) column
    y1, J1, H1 = nif_amd.HessianLayer(model, spec.so - 1, [xi[-1]])(x)
    assert H1.shape == (x.shape[0], 1, 1, 1)
    assert np.allclose(H1[:, 0, 0, 0], H[:, spec.so - 1, -1, -1], rtol=1e-5, atol=1e-6 * np.abs(H).max())
    # every input column, parameters included (the reference's tutorial 4 differentiates w.r.t. all of them), shuffled order:
    # second-order tangents through the ParameterNet (k_pjac2) and the second-order product rule of every layer
    xa = list(range(spec.pi + spec.si))[::-1]
    ya, Ja, Ha = nif_amd.HessianLayer(model, yi, xa)(x)
    ura, Jra, Hra = O.hessian_analytic(spec, ws, x.astype(np.float64), yi, xa)
    assert Ha.shape == (x.shape[0], spec.so, len(xa), len(xa))
    assert _rel(ya, ura) < 1e-5 and _rel(Ja, Jra) < 2e-5, (_rel(ya, ura), _rel(Ja, Jra))
    assert _rel(Ha, Hra) < 2e-4, _rel(Ha, Hra)
    for a_, b_ in ((0, 0), (0, len(xa) - 1), (len(xa) - 1, len(xa) - 1)):       # blocks: (x,x), (x,p), (p,p)
        blk, ref = Ha[:, :, a_, b_], Hra[:, :, a_, b_]
        assert np.linalg.norm(blk - ref) <= 3e-4 * max(np.linalg.norm(ref), 1e-3 * np.linalg.norm(Hra)), (a_, b_)
    assert np.array_equal(Ha, np.swapaxes(Ha, 2, 3))
    with pytest.raises(nif_amd._lib.NifError):
        nif_amd.HessianLayer(model, yi, [spec.pi + spec.si])(x)


def test_tutorial4_shape_contract_of_the_reference():
    """tests/golden/tutorial_logs.json: the reference's tutorial 4 wraps a model with 4 inputs and 5 outputs, x_index = all inputs,
    y_index = all outputs, 10 points, and prints (10, 5), (10, 5, 4), (10, 5, 4, 4)"""
    import json
    import os
    import nif_amd
    log = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tutorial_logs.json")))
    kind, cs, cp = _cfg("NIFMultiScale", 32, 2, 32, 1, 1, 3, 5, 1)           # (t; x, y, z) -> 5 outputs
    nif_amd.set_seed(0)
    model = nif_amd.NIFMultiScale(cs, cp).build()
    x = np.random.default_rng(0).uniform(0, 1, size=(log["y_shape"][0], len(log["x_index"]))).astype(np.float32)
    y, dydx = nif_amd.JacobianLayer(model, log["y_index"], log["x_index"])(x)
    assert list(y.shape) == log["y_shape"] and list(dydx.shape) == log["dydx_shape"]
    y2, dydx2, d2 = nif_amd.HessianLayer(model, log["y_index"], log["x_index"])(x)
    assert list(d2.shape) == log["d2ydx2_shape"] and np.allclose(dydx, dydx2, rtol=1e-4, atol=1e-6)


# ---- activity regularisers of the ParameterNet output (N3; reference model.py:118-125, :226, :659, :731) -----------------------
@pytest.mark.parametrize("name", ["nif_cfg1_32x2", "ms_cfg2_64x4", "ms_64x2_mlp_pnet_r3", "ms_32x2_r7_si3", "ll_plain_32x2_r3",
                                  "ll_cfg4_128x2_r10_so3", "ll_res_48x2_r4"])
@pytest.mark.parametrize("which", ["act_l2_reg", "act_l1_reg"])
def test_activity_regularisers_match_oracle(name, which):
    """loss += c/B sum phi(pnet_output) on the never materialised [B, po] tensor (two recompute passes, k_actreg_*) against
    the oracle's materialised formulation; L2 wins over L1 when both are given; fit() trains with it"""
    import nif_amd
    (kind, cs, cp), B = CONFIGS[name]
    B = min(B, 600)
    val = 3e-3 if which == "act_l2_reg" else 2e-4
    cp2 = dict(cp); cp2[which] = val
    if which == "act_l2_reg":
        cp2["act_l1_reg"] = 7.0      # ignored: L2 takes precedence (model.py:120-123)
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    m = getattr(nif_amd, kind)(cs, cp2)
    model = m.build(); model.set_weights(ws)
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32)
    ws64 = [w.astype(np.float64) for w in ws]
    act = (0.0, val) if which == "act_l2_reg" else (val, 0.0)
    loss, g = m._engine.loss_and_grad(x, y, sw)
    lr, gr = O.loss_and_grad(spec, ws64, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64), act_reg=act)
    l0, g0 = O.loss_and_grad(spec, ws64, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))
    assert abs(lr - l0) > 1e-4 * abs(l0)                      # the term is visible
    assert abs(loss - lr) <= 1e-5 * abs(lr), (loss, lr, l0)
    rel = _per_tensor_rel(spec, g, O.flatten(gr))
    assert max(rel.values()) < 3e-4, rel
    # a shard of a larger batch (what a rank computes): scaled by 1/B_global
    d_x, d_y = m._engine.alloc(x.size), m._engine.alloc(y.size)
    d_x.upload(x); d_y.upload(y)
    m._engine.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, 3 * B)
    l3, g3 = m._engine.grad_read()
    lr3, gr3 = O.loss_and_grad(spec, ws64, x.astype(np.float64), y.astype(np.float64), None, batch_global=3 * B, act_reg=act)
    assert abs(l3 - lr3) <= 1e-5 * abs(lr3) and _rel(g3, O.flatten(gr3)) < 2e-4
    model.compile(nif_amd.Adam(1e-3), "mse")
    h = model.fit(x, y, epochs=2, batch_size=256, shuffle=False, verbose=0)
    assert np.isfinite(h.history["loss"]).all()


def test_parameter_output_l1_act_reg_layer_matches_oracle():
    """nif.layers.ParameterOutputL1ActReg (regularization.py:4-32): loss += l1 * ||pnet_output||_1 over the WHOLE batch tensor --
    no division by the batch size -- through model.evaluate and one fit step (partial last batch: the term follows the batch)"""
    import nif_amd
    from nif_amd.layers import ParameterOutputL1ActReg
    (kind, cs, cp), B = CONFIGS["ms_cfg2_64x4"]
    B = 300
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    m = getattr(nif_amd, kind)(cs, cp)
    base = m.build(); base.set_weights(ws)
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    ws64 = [w.astype(np.float64) for w in ws]
    l1 = 3e-6
    reg = ParameterOutputL1ActReg(base, l1=l1)
    po = O.model_lr_to_w(spec, ws64, O.model_p_to_lr(spec, ws64, x[:, :spec.pi].astype(np.float64)))
    l0 = O.loss_and_grad(spec, ws64, x.astype(np.float64), y.astype(np.float64))[0]
    want = l0 + l1 * np.abs(po).sum()
    assert l1 * np.abs(po).sum() > 1e-3 * l0
    assert abs(reg.evaluate(x, y) - want) <= 2e-5 * want and abs(base.evaluate(x, y) - l0) <= 1e-5 * l0
    # one epoch of two steps (200 + 100 rows) with a vanishing learning rate: the logged loss is the row-weighted mean of the batches' totals
    reg.compile(nif_amd.Adam(1e-12), "mse")
    h = reg.fit(x, y, epochs=1, batch_size=200, shuffle=False, verbose=0)
    parts = []
    for lo, hi in ((0, 200), (200, 300)):
        lb = O.loss_and_grad(spec, ws64, x[lo:hi].astype(np.float64), y[lo:hi].astype(np.float64))[0]
        parts.append((hi - lo) * (lb + l1 * np.abs(po[lo:hi]).sum()))
    assert abs(h.history["loss"][0] - sum(parts) / B) <= 2e-5 * sum(parts) / B
    assert abs(base.evaluate(x, y) - l0) <= 1e-5 * l0           # the owner's configuration is restored


# ---- latent Jacobian regulariser (N3; reference model.py:353-375, gradient.py:52-127) ------------------------------------------
JAC = {
    "nif_swish_pi2": _cfg("NIF", 32, 2, 32, 2, 2, 1, 1, 2),
    "ms_siren_pnet": _cfg("NIFMultiScale", 64, 2, 32, 2, 1, 1, 1, 1),
    "ms_siren_res_pnet_r3": _cfg("NIFMultiScale", 32, 1, 40, 2, 3, 2, 1, 2, p_res=True),
    "ms_mlp_res_pnet": _cfg("NIFMultiScale", 32, 1, 24, 1, 2, 1, 2, 3, p_act="tanh", p_res=True),
    "ms_mlp_short_pnet_64": _cfg("NIFMultiScale", 32, 2, 64, 3, 2, 2, 1, 1, p_act="swish"),
    "ll_siren_pnet_r3": _cfg("LL", 32, 2, 32, 2, 3, 2, 2, 2),
    "ll_mlp_res_pnet_48": _cfg("LL", 48, 1, 24, 1, 4, 1, 1, 1, s_res=True, p_act="tanh", p_res=True),
    # r4: more than three parameter inputs (k_pjac in passes over groups of three columns; r3 refused pi_dim > 3)
    "ms_pi5_swish": _cfg("NIFMultiScale", 32, 2, 24, 2, 2, 2, 1, 5, p_act="swish"),
    "ms_pi4_siren_res": _cfg("NIFMultiScale", 32, 1, 40, 1, 3, 1, 2, 4, p_res=True),
    "ll_pi7_tanh": _cfg("LL", 32, 1, 32, 2, 3, 2, 2, 7, p_act="tanh"),
}


@pytest.mark.parametrize("name", sorted(JAC))
def test_jac_reg_matches_oracle(name):
    """cfg_parameter_net['jac_reg']: loss and every gradient tensor = main step + the oracle's regulariser term (itself
    pinned by torch double-backward, tests/test_oracle.py); all four ParameterNet layer types; ragged batch"""
    import nif_amd
    kind, cs, cp = JAC[name]
    l1 = 0.05
    cp2 = dict(cp, jac_reg=l1)
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    m = getattr(nif_amd, kind)(cs, cp2)
    model = m.build(); model.set_weights(ws)
    assert model._jac_reg == l1 and m.model()._jac_reg == 0.0      # only build() wraps the model (model.py:353-375)
    m._engine.set_jac_regularizer(l1)                              # what model.fit / model.evaluate do before they compute a loss
    ws64 = [w.astype(np.float64) for w in ws]
    for B, bg in ((333, 333), (64, 200)):
        x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
        y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
        d_x, d_y = m._engine.alloc(x.size), m._engine.alloc(y.size)
        d_x.upload(x); d_y.upload(y)
        m._engine.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, bg)
        loss, g = m._engine.grad_read()
        x64 = x.astype(np.float64)
        l0, g0 = O.loss_and_grad(spec, ws64, x64, y.astype(np.float64), batch_global=bg)
        lj, gj = O.jac_reg_loss_and_grad(spec, ws64, x64[:, :spec.pi], l1, batch_global=bg)
        assert lj > 1e-6 * l0
        assert abs(loss - (l0 + lj)) <= 1e-5 * (l0 + lj), (loss, l0, lj)
        ref = O.flatten(g0) + O.flatten(gj)
        rel = _per_tensor_rel(spec, g, ref)
        assert max(rel.values()) < 3e-4, rel
        # the regulariser's own part, isolated: gradient(with) - gradient(without) against the oracle's term
        m._engine.set_jac_regularizer(0.0)
        m._engine.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, bg)
        lw, gw = m._engine.grad_read()
        m._engine.set_jac_regularizer(l1)
        dj = (g.astype(np.float64) - gw)[:spec.n_params()]
        npn = sum(int(np.prod(s_)) for nm, s_ in spec.param_shapes() if nm.startswith("pnet_") and not nm.startswith("pnet_last"))
        assert _rel(dj[:npn], O.flatten(gj)[:npn]) < 2e-3 and abs((loss - lw) - lj) < 2e-4 * lj + 1e-7 * l0
    # evaluate() = the total loss incl. the regulariser for build()'s model, without it for .model() (Keras: model.losses)
    ev_b, ev_m = model.evaluate(x, y), m.model().evaluate(x, y)
    lj_e, _ = O.jac_reg_loss_and_grad(spec, ws64, x.astype(np.float64)[:, :spec.pi], l1)
    assert abs((ev_b - ev_m) - lj_e) < 2e-4 * lj_e + 1e-6 * ev_m
    model.compile(nif_amd.Adam(1e-3), "mse")
    assert np.isfinite(model.fit(x, y, epochs=2, batch_size=32, verbose=0).history["loss"]).all()


# ---- the HIP path against the FROZEN oracle vectors (tests/golden/oracle_v1.npz): forward, loss, per-tensor gradient, Jacobian and one
# Adam step of 13 configurations at B in {7, 64, 257} against numbers on disk -- an edit that moved a kernel and the live oracle
# together would still fail here -------------------------------------------------------------------------------------
def _frozen_cases():
    import importlib.util
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sp = importlib.util.spec_from_file_location("make_oracle_goldens", os.path.join(gold, "make_oracle_goldens.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    return mod, os.path.join(gold, "oracle_v1.npz")


@pytest.mark.parametrize("name", sorted(_frozen_cases()[0].CASES))
def test_hip_path_matches_frozen_oracle_vectors(name):
    import nif_amd
    mod, path = _frozen_cases()
    z = np.load(path)
    kind, cs, cp = mod.CASES[name]
    spec = O.Spec(kind, cs, cp)
    ws = [w.astype(np.float32) for w in O.unflatten(spec, z["%s/theta" % name])]
    for B in mod.BATCHES:
        k = "%s/%d/" % (name, B)
        x, y, sw = z[k + "x"], z[k + "y"], z[k + "sw"]
        m = getattr(nif_amd, kind)(cs, cp)
        model = m.build()
        model.set_weights(ws)
        assert _rel(model.predict(x), z[k + "u"]) < 1e-5, (name, B)
        loss, g = m._engine.loss_and_grad(x, y, sw)
        lref, gref = float(z[k + "loss"]), z[k + "grad"]
        assert abs(loss - lref) <= 2e-6 * abs(lref) + 1e-12, (name, B, loss, lref)
        gnorm, off = np.linalg.norm(gref), 0
        for nm, shp in spec.param_shapes():
            n_ = int(np.prod(shp))
            gr = gref[off:off + n_]
            err = np.linalg.norm(g[off:off + n_] - gr)
            assert err <= 5e-5 * np.linalg.norm(gr) + 2.5e-7 * gnorm, (name, B, nm, err, np.linalg.norm(gr))
            off += n_
        yi, xi = list(range(spec.so)), list(range(spec.pi, spec.pi + spec.si))
        _, J = nif_amd.JacobianLayer(model, yi, xi)(x)
        assert _rel(J, z[k + "jac"]) < 2e-5, (name, B, _rel(J, z[k + "jac"]))
        # one Adam step from zero moments (Keras 2.11, lr 1e-3): the step is +-lr per entry wherever the gradient is solid
        model.compile(nif_amd.Adam(learning_rate=1e-3), loss="mse")
        model.fit(x, y, epochs=1, batch_size=B, shuffle=False, verbose=0, sample_weight=sw)
        got = O.flatten(model.get_weights()).astype(np.float64)
        # Keras' first step is lr g / (|g| + eps / sqrt(1 - beta2)) = lr g / (|g| + 3.2e-6): insensitive to the fp32 gradient's relative
        # error wherever |g| is well above that floor and not small against its tensor's scale
        solid, off = np.abs(gref) > 1e-4, 0
        for nm, shp in spec.param_shapes():
            n_ = int(np.prod(shp))
            rms = np.sqrt(np.mean(gref[off:off + n_] ** 2)) + 1e-300
            solid[off:off + n_] &= np.abs(gref[off:off + n_]) > 0.02 * rms
            off += n_
        if not solid.any():
            solid = np.abs(gref) >= np.abs(gref).max()
        d = np.abs(got - z[k + "theta1"])
        assert d[solid].max() < 0.01 * 1e-3, (name, B, d[solid].max())
        assert d.max() <= 2.0 * 1e-3 * 1.001, (name, B)
        m._engine.close()


# ---- the 16-bit Keras policies against the EXACT fp64 oracle: the cast-for-cast tests above pin WHERE the kernels
# round; this one bounds HOW FAR the rounded computation is from exact arithmetic.  Bars = 3x the worst value measured over these shapes
# on MI355X (tools/exp/policy_distance.py, r6: bf16 predictions <= 1.1e-2, loss <= 5.7e-3, flat gradient <= 9.7e-3, worst tensor <= 2.3e-2;
# half: 1.3e-3 / 5.5e-4 / 1.4e-3 / 5.6e-3); bf16 has 8 significand bits, half 11: the half policy must sit >= 3x closer on the flat gradient
POLICY_BARS = {"mixed_bfloat16": (3e-2, 2e-2, 3e-2, 7e-2), "mixed_float16": (4e-3, 2e-3, 4.5e-3, 1.7e-2)}


@pytest.mark.parametrize("name", ["ms_cfg2_64x4", "ms_cfg5_64x4_si2", "ms_cfg3_128x3", "nif_cfg1_32x2", "ll_plain_32x2_r3",
                                  "ll_cfg4_128x6_r10_so3", "ms_64x8"])
def test_policy_distance_from_exact_arithmetic(name):
    got = {}
    for pol, (bu, bl, bg, bt) in POLICY_BARS.items():
        m, model, spec, ws, x, y, sw = _make_policy(name, pol)
        x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
        u = model.predict(x)
        loss, g = m._engine.loss_and_grad(x, y, sw)
        el, eg = O.loss_and_grad(spec, ws, x64, y64, s64)        # EXACT arithmetic: no rounding emulation
        d_u, d_l, d_g = _rel(u, O.forward(spec, ws, x64)), abs(loss - el) / abs(el), _rel(g, O.flatten(eg))
        d_t = max(_per_tensor_rel(spec, g, O.flatten(eg)).values())
        assert d_u < bu and d_l < bl and d_g < bg and d_t < bt, (pol, name, d_u, d_l, d_g, d_t)
        assert d_u > 1e-6, (pol, name, d_u)                       # (the policy really ran: fp32 sits at 1e-7)
        got[pol] = d_g
    assert got["mixed_float16"] * 3.0 < got["mixed_bfloat16"], got


# ---- r6: the small-batch step (k_small: loss + every gradient of <= 2048 points in one launch) and the tile kernels it replaces for those
# batches: both against the oracle, and against each other ---------------------------------------------------------------------------------
SMALL_CASES = {
    "nif_cfg1_32x2": CONFIGS["nif_cfg1_32x2"],
    "nif_pad_n30_tanh_r2_so2": CONFIGS["nif_pad_n30_tanh_r2_so2"],
    "ms_tiny_b1": CONFIGS["ms_tiny_b1"],
    "cfg0_nif_2x32_b512": (_cfg("NIF", 32, 2, 32, 2, 1, 1, 1, 1), 512),            # configs[0]'s own step
    "nif_32x2_r2_si3_so3_b2047": (_cfg("NIF", 32, 2, 32, 3, 2, 3, 3, 2, act="tanh"), 2047),
    "ms_plain_32x3_r2_b17": (_cfg("NIFMultiScale", 32, 3, 24, 2, 2, 2, 2, 1), 17),
    "ms_plain_24x2_mlp_pnet_b333": (_cfg("NIFMultiScale", 24, 2, 32, 1, 3, 1, 1, 3, p_act="swish"), 333),
}


@pytest.mark.parametrize("name", sorted(SMALL_CASES))
@pytest.mark.parametrize("weighted", [False, True])
def test_small_batch_step_matches_oracle_and_tile_kernels(name, weighted):
    m, model, spec, ws, x, y, sw = _make(SMALL_CASES[name])
    s = sw if weighted else None
    e = m._engine
    # the FIRST training call of a fresh context already takes k_small (r6 sweep: it fell back to the tile kernels once, silently)
    e.profile_enable(True); e.profile_read(reset=True)
    e.loss_and_grad(x, y, s)
    prof0 = e.profile_read(reset=True); e.profile_enable(False)
    assert prof0["snet"][1] == 1 and prof0["pnet_fwd"][1] == 0 and prof0["pnet_bwd"][1] == 0 and prof0["gw"][1] == 0, prof0
    lref, gref = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), None if s is None else s.astype(np.float64))
    gref = O.flatten(gref)
    res = {}
    for small in (1, 0):
        e.set_option("small_step", small)
        loss, g = e.loss_and_grad(x, y, s)
        res[small] = (loss, np.asarray(g, dtype=np.float64))
        assert abs(loss - lref) <= 2e-6 * abs(lref), (small, loss, lref)
        rel = _per_tensor_rel(spec, g, gref)
        # k_small runs plain fp32 FMAs (no split products): 1e-5 per tensor; the tile kernels keep their r5 bar
        assert max(rel.values()) < (1e-5 if small else 5e-5), (small, rel)
    e.set_option("small_step", 1)
    assert _rel(res[1][1], res[0][1]) < 3e-5
    # the step really went through k_small: its launch is booked on the ShapeNet group and NOTHING on the ParameterNet / weight-gradient groups
    e.profile_enable(True); e.profile_read(reset=True)
    e.loss_and_grad(x, y, s)
    prof = e.profile_read(reset=True); e.profile_enable(False)
    assert prof["snet"][1] == 1 and prof["pnet_fwd"][1] == 0 and prof["pnet_bwd"][1] == 0 and prof["gw"][1] == 0, prof


def test_small_batch_fit_follows_the_oracle_adam_steps():
    """Model.fit at configs[0]'s shape (batch 512 of a 2 000-point table, no shuffle): four Adam steps through k_small + k_reduce + k_adam
    against the oracle's Keras-2.11 updates, teacher-forced per step as in test_adam_steps_follow_oracle"""
    import nif_amd
    m, model, spec, ws, x, y, sw = _make((_cfg("NIF", 32, 2, 32, 2, 1, 1, 1, 1), 2000))
    model.compile(nif_amd.Adam(learning_rate=1e-3), loss="mse")
    e = m._engine
    f32 = lambda a: float(np.float32(a))      # noqa: E731
    th = O.flatten(model.get_weights()).astype(np.float64)
    h = model.fit(x, y, epochs=1, batch_size=512, shuffle=False, verbose=0)
    mm = np.zeros_like(th); vv = np.zeros_like(th)
    tot, cnt = 0.0, 0
    for t, b0 in enumerate(range(0, 2000, 512), start=1):
        xb, yb = x[b0:b0 + 512].astype(np.float64), y[b0:b0 + 512].astype(np.float64)
        l_, g_ = O.loss_and_grad(spec, O.unflatten(spec, th), xb, yb)
        tot += l_ * len(xb); cnt += len(xb)
        th, mm, vv = O.adam_step(th, O.flatten(g_), mm, vv, t, lr=f32(1e-3), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
    assert abs(h.history["loss"][0] - tot / cnt) <= 1e-4 * abs(tot / cnt), (h.history["loss"][0], tot / cnt)
    got = O.flatten(model.get_weights()).astype(np.float64)
    assert np.abs(got - th).max() <= 4 * 2.0 * 1e-3 * 1.001        # four steps of at most 2 lr each
    assert np.median(np.abs(got - th)) < 0.05 * 1e-3               # ... and almost everywhere on the oracle's trajectory
    assert e.get_opt_state()[2] == 4
