"""Pins the NumPy oracle: independent torch-autograd restatement, internal identities,
finite differences, and the reference's bundled dataset / normalisers (via golden fixtures)."""
import os

import numpy as np
import pytest

from oracle import nif_oracle as O
from tests.cfgs import ALL_SMALL, cfg_ll

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _setup(name, B=9, seed=0, wscale=None):
    kind, cs, cp = ALL_SMALL[name]
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(seed)
    ws = O.init_weights(spec, rng)
    if kind != "NIF":
        # weight_init_factor=0.01 makes z-dependence tiny; enlarge so the tests see it
        names = [nm for nm, _ in spec.param_shapes()]
        ws[names.index("pnet_last_w")] *= 20.0
    inputs = rng.uniform(-1, 1, size=(B, spec.pi + spec.si))
    y = rng.uniform(-1, 1, size=(B, spec.so))
    sw = rng.uniform(0.5, 1.5, size=(B,))
    return kind, cs, cp, spec, ws, inputs, y, sw


@pytest.mark.parametrize("name", sorted(ALL_SMALL))
def test_forward_and_grad_match_torch_autograd(name):
    torch = pytest.importorskip("torch")
    from tests import torch_ref as T
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name)
    for sample_weight in (None, sw):
        loss, grads = O.loss_and_grad(spec, ws, inputs, y, sample_weight)
        tl, tg, tu = T.loss_and_grad(kind, cs, cp, ws, inputs, y, sample_weight)
        u = O.forward(spec, ws, inputs)
        assert np.allclose(u, tu, rtol=1e-12, atol=1e-12)
        assert abs(loss - tl) <= 1e-12 * max(1.0, abs(tl))
        assert len(grads) == len(tg)
        for (nm, _), g, t in zip(spec.param_shapes(), grads, tg):
            assert t is not None, nm
            denom = max(np.abs(t).max(), 1e-30)
            assert np.abs(g - t).max() / denom < 1e-9, nm


@pytest.mark.parametrize("name", ["nif_swish", "ms_plain_r3_si2", "ms_res"])
def test_three_stage_factorisation_identity(name):
    # README.md:99-117: model_x_to_u_given_w(x, model_lr_to_w(model_p_to_lr(p))) == model([p,x])
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name)
    p, x = inputs[:, :spec.pi], inputs[:, spec.pi:]
    lr = O.model_p_to_lr(spec, ws, p)
    w = O.model_lr_to_w(spec, ws, lr)
    assert w.shape == (inputs.shape[0], spec.po)
    u = O.shapenet_given_w(spec, x, w)
    assert np.allclose(u, O.forward(spec, ws, inputs), rtol=1e-13, atol=1e-13)


def test_last_layer_class_raises_for_lr_to_w():
    kind, cs, cp, spec, ws, inputs, y, sw = _setup("ll_plain")
    with pytest.raises(ValueError):
        O.model_lr_to_w(spec, ws, inputs[:, :1])


@pytest.mark.parametrize("name", ["nif_swish", "ms_plain_r3_si2", "ms_res"])
def test_grad_matches_finite_differences(name):
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name, B=5)
    loss, grads = O.loss_and_grad(spec, ws, inputs, y, sw)
    flat = O.flatten(ws)
    g = O.flatten(grads)
    rng = np.random.default_rng(3)
    for idx in rng.choice(flat.size, size=25, replace=False):
        d = np.zeros_like(flat); d[idx] = 1e-6
        lp = O.mse_loss(O.forward(spec, O.unflatten(spec, flat + d), inputs), y, sw)
        lm = O.mse_loss(O.forward(spec, O.unflatten(spec, flat - d), inputs), y, sw)
        fd = (lp - lm) / 2e-6
        assert abs(fd - g[idx]) <= 2e-5 * max(1.0, abs(g[idx])) + 1e-8  # FD truncation (omega_0=30)


@pytest.mark.parametrize("name", ["nif_tanh_r2_so2", "ms_plain_r3_si2", "ms_res"])
def test_jacobian_fd_vs_analytic_vs_autograd(name):
    torch = pytest.importorskip("torch")
    from tests import torch_ref as T
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name, B=6)
    yi = list(range(spec.so))
    xi = list(range(spec.pi, spec.pi + spec.si))
    y1, J1 = O.jacobian(spec, ws, inputs, yi, xi)
    y2, J2 = O.jacobian_analytic(spec, ws, inputs, yi, xi)
    tu, tJ = T.jacobian(kind, cs, cp, ws, inputs)
    assert np.allclose(y1, y2) and np.allclose(y1, tu)
    assert np.allclose(J2, tJ[:, :, xi], rtol=1e-10, atol=1e-12)
    assert np.allclose(J1, tJ[:, :, xi], rtol=1e-5, atol=1e-7)
    # parameter columns too (finite differences only)
    _, Jp = O.jacobian(spec, ws, inputs, yi, list(range(spec.pi)))
    assert np.allclose(Jp, tJ[:, :, :spec.pi], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", ["ms_plain", "ms_plain_r3_si2", "ms_res", "ms_mlp_pres", "nif_swish", "nif_tanh_r2_so2", "ll_plain", "ll_res"])
def test_sobolev_loss_and_grad_match_torch_double_backward(name):
    """Sobolev step (JacobianLayer as a trained output): the oracle's hand-derived adjoint of the tangent
    program against torch autograd through the input gradient, with and without sample weights."""
    torch = pytest.importorskip("torch")
    from tests import torch_ref as T
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name, B=7)
    xi = list(range(spec.pi, spec.pi + spec.si))
    rng = np.random.default_rng(5)
    dydx = rng.uniform(-1, 1, size=(7, spec.so, len(xi)))
    for sample_weight in (None, sw):
        loss, grads, u, J = O.sobolev_loss_and_grad(spec, ws, inputs, y, dydx, xi, 0.3, sample_weight)
        tl, tg, tu, tJ = T.sobolev_loss_and_grad(kind, cs, cp, ws, inputs, y, dydx, xi, 0.3, sample_weight)
        assert np.allclose(u, tu, rtol=1e-12, atol=1e-12) and np.allclose(J, tJ, rtol=1e-10, atol=1e-12)
        assert abs(loss - tl) <= 1e-12 * max(1.0, abs(tl))
        for (nm, _), g, t in zip(spec.param_shapes(), grads, tg):
            assert t is not None, nm
            assert np.abs(g - t).max() / max(np.abs(t).max(), 1e-30) < 1e-9, nm
    # w_jac = 0 degenerates to the plain step
    l0, g0, _, _ = O.sobolev_loss_and_grad(spec, ws, inputs, y, dydx, xi, 0.0, sw)
    l1, g1 = O.loss_and_grad(spec, ws, inputs, y, sw)
    assert abs(l0 - l1) < 1e-14 and all(np.allclose(a, b, rtol=1e-12, atol=1e-14) for a, b in zip(g0, g1))


@pytest.mark.parametrize("name", ["ms_plain", "ms_plain_r3_si2", "ms_res", "ms_mlp_pres", "nif_swish", "nif_tanh_r2_so2", "ll_plain", "ll_res"])
@pytest.mark.parametrize("cols", ["param_only", "mixed"])
def test_sobolev_parameter_columns_match_torch_double_backward(name, cols):
    """JacobianLayer works for ANY input column (gradient.py:207-231): x_index addressing ParameterNet inputs -- the tangent
    then runs through the ParameterNet, the hyper layer and every product h W(p) -- alone and mixed with coordinates"""
    torch = pytest.importorskip("torch")
    from tests import torch_ref as T
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name, B=7)
    if cols == "param_only":
        xi = list(range(spec.pi))
    else:
        xi = [spec.pi + spec.si - 1, 0] + ([spec.pi] if spec.si > 1 else [])       # coordinate, parameter, coordinate
    rng = np.random.default_rng(6)
    dydx = rng.uniform(-1, 1, size=(7, spec.so, len(xi)))
    for sample_weight in (None, sw):
        loss, grads, u, J = O.sobolev_loss_and_grad(spec, ws, inputs, y, dydx, xi, 0.3, sample_weight)
        tl, tg, tu, tJ = T.sobolev_loss_and_grad(kind, cs, cp, ws, inputs, y, dydx, xi, 0.3, sample_weight)
        assert np.allclose(u, tu, rtol=1e-12, atol=1e-12) and np.allclose(J, tJ, rtol=1e-9, atol=1e-11)
        assert abs(loss - tl) <= 1e-12 * max(1.0, abs(tl))
        for (nm, _), g, t in zip(spec.param_shapes(), grads, tg):
            assert t is not None, nm
            assert np.abs(g - t).max() / max(np.abs(t).max(), 1e-30) < 1e-9, nm


def test_jacobian_shape_contract_notebook4():
    # tutorial/4 cells 12-18: JacobianLayer output shapes (B, ny), (B, len(y_index), len(x_index))
    kind, cs, cp, spec, ws, inputs, y, sw = _setup("nif_tanh_r2_so2", B=10)
    yv, J = O.jacobian(spec, ws, inputs, [0, 1], [1, 2])
    assert yv.shape == (10, 2) and J.shape == (10, 2, 2)


def test_po_dim_and_param_counts_match_survey_table():
    # SURVEY section 8 size table (po_dim, trainable P)
    def spec_of(kind, n, L, si=1, so=1, res=False):
        if kind == "NIF":
            cs = {"input_dim": si, "output_dim": so, "units": n, "nlayers": L, "activation": "swish"}
            cp = {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
        else:
            cs = {"input_dim": si, "output_dim": so, "units": n, "nlayers": L, "use_resblock": res,
                  "connectivity": "full", "omega_0": 30.0, "weight_init_factor": 0.01}
            cp = {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "sine",
                  "use_resblock": False, "omega_0": 30.0}
        return O.Spec(kind, cs, cp)
    s = spec_of("NIF", 32, 2); assert (s.po, s.n_params()) == (2209, 6627)
    s = spec_of("NIF", 64, 4); assert (s.po, s.n_params()) == (16833, 35875)
    s = spec_of("NIFMultiScale", 64, 4); assert (s.po, s.n_params()) == (16833, 35875)
    s = spec_of("NIFMultiScale", 128, 6, si=2); assert (s.po, s.n_params()) == (99585, 201379)
    s = spec_of("NIFMultiScale", 128, 6, si=2, res=True); assert (s.po, s.n_params()) == (198657, 399523)
    s = spec_of("NIFMultiScale", 64, 4, si=2); assert s.po == 16897


def test_adam_matches_closed_form_first_step():
    th = np.array([1.0, -2.0]); g = np.array([0.5, -0.25])
    th1, m, v = O.adam_step(th, g, np.zeros(2), np.zeros(2), 1, lr=1e-3)
    # first Adam step moves by ~lr*sign(g)
    assert np.allclose(th1, th - 1e-3 * np.sign(g), atol=1e-8)


def test_closed_form_matches_bundled_dataset():
    # the reference's only real fixture: nif/demo/dataset/traveling_wave*.npz (copied as data)
    for fn, om, tol in (("traveling_wave.npz", 4.0, 1e-6), ("traveling_wave_high_freq.npz", 400.0, 1e-4)):
        d = np.load(os.path.join(GOLD, fn))["data"]
        assert d.shape == (2000, 3)
        u = O.traveling_wave(d[:, 0].astype(np.float64), d[:, 1].astype(np.float64), om)
        assert np.abs(u - d[:, 2]).max() < tol


def test_normalisers_match_reference_golden():
    # golden produced by importing /root/reference/nif/data/point_wise_data.py (NumPy only)
    g = np.load(os.path.join(GOLD, "normalisers.npz"))
    raw = g["raw"]
    d, m, s = O.standard_normalize(raw.copy())
    assert np.array_equal(d, g["std_data"]) and np.array_equal(m, g["std_mean"]) and np.array_equal(s, g["std_std"])
    d, m, s = O.minmax_normalize(raw.copy(), 1, 1, 1)
    assert np.array_equal(d, g["mm_data"]) and np.array_equal(m, g["mm_mean"]) and np.array_equal(s, g["mm_std"])


@pytest.mark.parametrize("name", ["nif_swish", "nif_tanh_r2_so2", "ms_plain", "ms_plain_r3_si2", "ms_res", "ms_res_pres", "ms_mlp_pres"])
def test_plane_formulation_equals_the_materialised_reference_formulation(name):
    """h.W(a) = sum_k zt_k (h.M^(k)): the oracle's second restatement, which never forms [B, po], against the
    reference formulation (materialised pnet_output + per-sample einsum chain); and its bf16 emulation is a rounding"""
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name)
    for sample_weight, bg in ((None, None), (sw, 13)):
        l, g = O.loss_and_grad(spec, ws, inputs, y, sample_weight, batch_global=bg)
        l2, g2, u2 = O.planes_loss_and_grad(spec, ws, inputs, y, sample_weight, batch_global=bg)
        assert abs(l - l2) <= 1e-12 * max(1.0, abs(l))
        assert np.allclose(u2, O.forward(spec, ws, inputs), rtol=1e-12, atol=1e-12)
        for (nm, _), a, b in zip(spec.param_shapes(), g, g2):
            assert np.abs(a - b).max() <= 1e-10 * max(np.abs(a).max(), 1e-30), nm
    lb, gb, ub = O.planes_loss_and_grad(spec, ws, inputs, y, sw, rnd=O.bf16_round)
    assert np.isfinite(lb) and lb != l
    r = O.bf16_round(np.array([1.0, 1.00390625, 1.005859375, 1.01171875, -3.14159265, 0.0]))
    assert np.array_equal(r, np.array([1.0, 1.0, 1.0078125, 1.015625, -3.140625, 0.0]))      # ties to even, 8-bit significand


@pytest.mark.parametrize("name", ["nif_swish", "ms_plain", "ms_res", "ll_plain", "ll_res"])
def test_mixed_float16_emulation_of_the_oracle(name):
    """f16_round is IEEE half with RNE and saturation (what v_med3_f32 + v_cvt_pk_f16_f32 do); its .grad form rounds under the 2^15
    loss scale, so adjoints below half's normal range survive; the emulated step is a rounding of the exact one -- finer than the
    bf16 emulation (11 significand bits against 8) -- and torch's own float16 cast agrees bit for bit"""
    r = O.f16_round(np.array([1.0, 1.00048828125, 1.000732421875, 1.00146484375, 65519.0, 65520.0, 1e9, -1e9, 5.96e-8, 2.9e-8]))
    assert np.array_equal(r, np.array([1.0, 1.0, 1.0009765625, 1.001953125, 65504.0, 65504.0, 65504.0, -65504.0, 2.0 ** -24, 0.0]))
    torch = pytest.importorskip("torch")
    v = np.random.default_rng(0).standard_normal(4096).astype(np.float32) * np.float32(10.0) ** np.random.default_rng(1).integers(-9, 5, 4096)
    assert np.array_equal(O.f16_round(v), torch.from_numpy(np.clip(v, -65504, 65504)).to(torch.float16).to(torch.float64).numpy())
    g = np.array([[1e-7, 3e-9, 1e-20, 0.0], [3.0, 1.0, 1e-3, 1e-13], [0.0, 0.0, 0.0, 0.0]])
    assert np.all(O.f16_round(g[0, :2]) != g[0, :2]) and O.f16_round(3e-9) == 0.0    # unscaled: subnormal or flushed
    rg = O.f16_round.grad(g)        # per-point power of two: the row's largest entry lands in [2^14, 2^15)
    assert np.allclose(rg[0, :2], g[0, :2], rtol=1e-3) and np.allclose(rg[1, :3], g[1, :3], rtol=1e-3) and np.all(rg[2] == 0.0)
    assert rg[0, 2] == 0.0 and rg[1, 3] == 0.0            # more than 38 binades below the row's maximum: below half's subnormals
    assert np.all(np.isfinite(O.f16_round.grad(np.array([[1e30, -3e38, 1.0]]))))
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name)
    if spec.kind == O.KIND_LL:
        fn = O.ll_policy_loss_and_grad
        le, ge = O.loss_and_grad(spec, ws, inputs, y, sw)
    else:
        fn = O.planes_loss_and_grad
        le, ge = O.loss_and_grad(spec, ws, inputs, y, sw)
    lh, gh, uh = fn(spec, ws, inputs, y, sw, rnd=O.f16_round)
    lb, gb, ub = fn(spec, ws, inputs, y, sw, rnd=O.bf16_round)
    fl = O.flatten
    dh = np.linalg.norm(fl(gh) - fl(ge)) / np.linalg.norm(fl(ge))
    db = np.linalg.norm(fl(gb) - fl(ge)) / np.linalg.norm(fl(ge))
    assert lh != le and 0 < dh < 5e-2 and dh < 0.5 * db, (dh, db)      # (ms_res: large weights, bf16 sits 0.26 away, half 0.03)
    # without rounding the .grad hook changes nothing: rnd=None is the exact formulation
    l0, g0, _ = fn(spec, ws, inputs, y, sw, rnd=None)
    assert abs(l0 - le) <= 1e-12 * max(1.0, abs(le))


@pytest.mark.parametrize("name", ["nif_tanh_r2_so2", "nif_swish", "ms_plain_r3_si2", "ms_res", "ms_mlp_pres", "ll_plain", "ll_res"])
def test_hessian_analytic_vs_fd_of_the_jacobian_and_torch(name):
    """HessianLayer oracle: second-order forward mode against central differences of the analytic Jacobian and against
    torch autograd (double backward) of the independent torch restatement"""
    torch = pytest.importorskip("torch")
    from tests import torch_ref as T
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name, B=5)
    yi = list(range(spec.so))
    xi = list(range(spec.pi, spec.pi + spec.si))
    u, J, H = O.hessian_analytic(spec, ws, inputs, yi, xi)
    u2, J2 = O.jacobian_analytic(spec, ws, inputs, yi, xi)
    assert np.allclose(u, u2) and np.allclose(J, J2, rtol=1e-12, atol=1e-14)
    assert H.shape == (5, spec.so, spec.si, spec.si) and np.allclose(H, np.swapaxes(H, 2, 3))
    eps = 1e-6
    for kk, col in enumerate(xi):
        d = np.zeros_like(inputs); d[:, col] = eps
        _, Jp = O.jacobian_analytic(spec, ws, inputs + d, yi, xi)
        _, Jm = O.jacobian_analytic(spec, ws, inputs - d, yi, xi)
        fd = (Jp - Jm) / (2 * eps)
        assert np.allclose(H[:, :, :, kk], fd, rtol=2e-3, atol=1e-5 * max(1.0, np.abs(H).max())), (name, kk)   # FD truncation (w0 = 30)
    tH = T.hessian(kind, cs, cp, ws, inputs)            # [B, so, ncol, ncol] over all input columns
    assert np.allclose(H, tH[:, :, xi][:, :, :, xi], rtol=1e-9, atol=1e-11 * max(1.0, np.abs(H).max()))
    # every input column, parameters included (tutorial 4 differentiates w.r.t. all of them), in a shuffled order
    xa = list(range(spec.pi + spec.si))[::-1]
    ua, Ja, Ha = O.hessian_analytic(spec, ws, inputs, yi, xa)
    tu, tJ = T.jacobian(kind, cs, cp, ws, inputs)
    assert np.allclose(ua, tu) and np.allclose(Ja, tJ[:, :, xa], rtol=1e-9, atol=1e-11 * max(1.0, np.abs(Ja).max()))
    assert np.allclose(Ha, tH[:, :, xa][:, :, :, xa], rtol=1e-8, atol=1e-10 * max(1.0, np.abs(Ha).max()))


@pytest.mark.parametrize("name", ["nif_swish", "nif_tanh_r2_so2", "ms_plain", "ms_plain_r3_si2", "ms_res_pres", "ms_mlp_pnet", "ms_mlp_pres"])
def test_jac_reg_matches_torch_double_backward(name):
    """latent Jacobian regulariser: the oracle's tangent + adjoint program against torch autograd through the Jacobian,
    for every ParameterNet layer type (Dense+shortcut, MLP_ResNet, SIREN, SIREN_ResNet)"""
    torch = pytest.importorskip("torch")
    from tests import torch_ref as T
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name, B=6)
    p = inputs[:, :spec.pi]
    loss, grads = O.jac_reg_loss_and_grad(spec, ws, p, 0.7)
    tl, tg = T.jac_reg_loss_and_grad(kind, cs, cp, ws, p, 0.7)
    assert abs(loss - tl) <= 1e-12 * max(1.0, abs(tl)) and loss > 0
    for (nm, _), g, t in zip(spec.param_shapes(), grads, tg):
        if t is None:
            assert not np.any(g), nm          # downstream of the latent
        else:
            assert np.abs(g - t).max() <= 1e-9 * max(np.abs(t).max(), 1e-30), nm
    l2, g2 = O.jac_reg_loss_and_grad(spec, ws, p, 0.7, batch_global=18)
    assert abs(l2 - loss / 3) < 1e-15 * max(1.0, loss)


@pytest.mark.parametrize("name", ["nif_swish", "ms_plain_r3_si2", "ms_res", "ll_plain", "ll_res"])
@pytest.mark.parametrize("which", ["l1", "l2"])
def test_activity_regulariser_matches_torch_autograd(name, which):
    """Keras activity_regularizer on the last ParameterNet layer (model.py:118-125, :226): loss += c * sum(phi(pnet_output)) /
    batch, phi = |.| or (.)^2 -- the oracle's term and its gradient against torch autograd of the independent restatement"""
    torch = pytest.importorskip("torch")
    from tests import torch_ref as T
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name, B=6)
    act = (0.03, 0.0) if which == "l1" else (0.0, 0.02)
    loss, grads = O.loss_and_grad(spec, ws, inputs, y, sw, act_reg=act)
    wt = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in ws]
    xin = torch.tensor(inputs, dtype=torch.float64)
    u = T.forward(kind, cs, cp, wt, xin)
    per = ((u - torch.tensor(y)) ** 2).mean(dim=1) * torch.tensor(sw)
    z = T.latent(kind, cs, cp, wt, xin[:, :spec.pi])
    npn = len([nm for nm, _ in spec.param_shapes() if nm.startswith("pnet_")])
    pout = z @ wt[npn - 2] + wt[npn - 1]                       # the last ParameterNet layer (Dense / HyperLinearForSIREN)
    reg = (act[0] * pout.abs().sum() + act[1] * (pout ** 2).sum()) / u.shape[0]
    tl = per.sum() / u.shape[0] + reg
    tg = torch.autograd.grad(tl, wt, allow_unused=True)
    assert abs(loss - tl.item()) <= 1e-12 * max(1.0, abs(tl.item()))
    for (nm, _), g, t in zip(spec.param_shapes(), grads, tg):
        assert np.abs(g - t.numpy()).max() / max(np.abs(t.numpy()).max(), 1e-30) < 1e-9, nm


@pytest.mark.parametrize("name", [k for k, v in ALL_SMALL.items() if v[0] in ("NIF", "NIFMultiScale")])
def test_sobolev_plane_formulation_equals_the_materialised_one(name):
    """sobolev_planes_loss_and_grad (what k_sob executes; carries the mixed_bfloat16 emulation of configs[4]) == the reference
    formulation sobolev_loss_and_grad to 1e-10 in exact mode, for one and two coordinate columns; under rnd=bf16_round it is a
    DIFFERENT computation (1e-4 .. 1e-2 away) with a finite, non-degenerate gradient"""
    kind, cs, cp = ALL_SMALL[name]
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = O.init_weights(spec, rng)
    B = 19
    x = rng.uniform(-1, 1, (B, spec.pi + spec.si)); y = rng.uniform(-1, 1, (B, spec.so)); sw = rng.uniform(0.5, 1.5, (B,))
    for xi in ([spec.pi], list(range(spec.pi, spec.pi + spec.si))[:2]):
        dy = rng.uniform(-1, 1, (B, spec.so, len(xi)))
        l0, g0, u0, j0 = O.sobolev_loss_and_grad(spec, ws, x, y, dy, xi, 0.3, sw, batch_global=31)
        l1, g1, u1, j1 = O.sobolev_planes_loss_and_grad(spec, ws, x, y, dy, xi, 0.3, sw, batch_global=31)
        assert abs(l0 - l1) <= 1e-12 * abs(l0) and np.abs(u0 - u1).max() < 1e-12 and np.abs(j0 - j1).max() < 1e-10
        f0, f1 = O.flatten(g0), O.flatten(g1)
        assert np.abs(f0 - f1).max() <= 1e-10 * np.abs(f0).max()
        l2, g2, _, _ = O.sobolev_planes_loss_and_grad(spec, ws, x, y, dy, xi, 0.3, sw, batch_global=31, rnd=O.bf16_round)
        d = np.linalg.norm(O.flatten(g2) - f0) / np.linalg.norm(f0)
        assert 1e-5 < d < 5e-2 and np.isfinite(l2)
        # the bf16 dL/da stash form (k_sobw<PR> / k_gw_lds<DAB>): the same sums without rounding, other weight gradients with it
        l3, g3, _, _ = O.sobolev_planes_loss_and_grad(spec, ws, x, y, dy, xi, 0.3, sw, batch_global=31, stash_bf16=True)
        assert l3 == l1 and np.abs(O.flatten(g3) - f1).max() <= 1e-12 * np.abs(f1).max()
        l4, g4, _, _ = O.sobolev_planes_loss_and_grad(spec, ws, x, y, dy, xi, 0.3, sw, batch_global=31, rnd=O.bf16_round, stash_bf16=True)
        d4 = np.linalg.norm(O.flatten(g4) - O.flatten(g2)) / np.linalg.norm(f0)
        assert l4 == l2 and 1e-6 < d4 < 2e-2
    # and of the plain step
    l5, g5, _ = O.planes_loss_and_grad(spec, ws, x, y, sw, batch_global=31)
    l6, g6, _ = O.planes_loss_and_grad(spec, ws, x, y, sw, batch_global=31, stash_bf16=True)
    assert l5 == l6 and np.abs(O.flatten(g5) - O.flatten(g6)).max() <= 1e-12 * np.abs(O.flatten(g5)).max()
    l7, g7, _ = O.planes_loss_and_grad(spec, ws, x, y, sw, batch_global=31, rnd=O.bf16_round)
    l8, g8, _ = O.planes_loss_and_grad(spec, ws, x, y, sw, batch_global=31, rnd=O.bf16_round, stash_bf16=True)
    d8 = np.linalg.norm(O.flatten(g8) - O.flatten(g7)) / np.linalg.norm(O.flatten(g7))
    assert l7 == l8 and 1e-6 < d8 < 2e-2
    with pytest.raises(AssertionError):
        O.sobolev_planes_loss_and_grad(spec, ws, x, y, dy, [0], 0.3)        # parameter columns: the materialised form has them


def test_weight_regulariser_term_is_the_gradient_of_its_loss():
    """oracle.weight_regularizer_term (model.py:109-117, :1028-1039): central differences of the loss term, last-layer class with
    a ParameterNet L1 and a ShapeNet L2 term; last_layer_bias untouched"""
    kind, cs, cp = cfg_ll()
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = [w.astype(np.float64) for w in O.init_weights(spec, rng, dtype=np.float32)]
    preg, sreg = (2e-3, 0.0), (0.0, 0.01)
    loss, grads = O.weight_regularizer_term(spec, ws, preg, sreg)
    names = [nm for nm, _ in spec.param_shapes()]
    assert np.abs(grads[names.index("last_layer_bias")]).max() == 0.0
    for nm in ("pnet_first_w", "snet_first_b", "snet_bottleneck_w"):
        i = names.index(nm)
        w = ws[i]
        idx = tuple(rng.integers(0, s) for s in w.shape)
        if abs(w[idx]) < 1e-4:
            w[idx] = 0.01
            loss, grads = O.weight_regularizer_term(spec, ws, preg, sreg)
        h = 1e-6
        w0 = w[idx]
        w[idx] = w0 + h; lp, _ = O.weight_regularizer_term(spec, ws, preg, sreg)
        w[idx] = w0 - h; lm, _ = O.weight_regularizer_term(spec, ws, preg, sreg)
        w[idx] = w0
        assert abs((lp - lm) / (2 * h) - grads[i][idx]) < 1e-6 * max(1.0, abs(grads[i][idx])), nm


NEW_ACTS = ["selu", "softsign", "exponential", "hard_sigmoid"]


@pytest.mark.parametrize("act", NEW_ACTS)
def test_remaining_keras_activations_match_torch_and_their_own_derivatives(act):
    """keras.activations of Keras 2.11 beyond the nine r3 had (model.py:303 takes any name keras.activations.get knows): selu,
    softsign, exponential, hard_sigmoid (= clip(0.2 x + 0.5, 0, 1) in 2.11) -- forward / gradient of class NIF and of an MLP
    ParameterNet against torch autograd, f' and f'' against central differences of f and f'."""
    torch = pytest.importorskip("torch")
    from tests import torch_ref as T
    from tests.cfgs import cfg_nif, cfg_ms
    f, df = O.act_fn(act)
    d2 = O.act_d2(act)
    a = np.linspace(-3.0, 3.0, 601) + 1e-3          # (off the kinks of hard_sigmoid at +-2.5 and of softsign / selu at 0)
    hstep = 1e-6
    assert np.allclose(df(a), (f(a + hstep) - f(a - hstep)) / (2 * hstep), atol=2e-6)
    assert np.allclose(d2(a), (df(a + hstep) - df(a - hstep)) / (2 * hstep), atol=2e-5)
    for kind, cs, cp in (cfg_nif(r=2, so=2, si=2, act=act), cfg_ms(p_act=act, p_res=True, so=2)):
        spec = O.Spec(kind, cs, cp)
        rng = np.random.default_rng(1)
        ws = O.init_weights(spec, rng)
        x = rng.uniform(-1, 1, size=(9, spec.pi + spec.si)); y = rng.uniform(-1, 1, size=(9, spec.so))
        loss, grads = O.loss_and_grad(spec, ws, x, y)
        tl, tg, tu = T.loss_and_grad(kind, cs, cp, ws, x, y, None)
        assert np.allclose(O.forward(spec, ws, x), tu, rtol=1e-12, atol=1e-12) and abs(loss - tl) <= 1e-12 * max(1.0, abs(tl))
        for g, t in zip(grads, tg):
            assert np.abs(g - t).max() / max(np.abs(t).max(), 1e-30) < 1e-9


@pytest.mark.parametrize("loss", ["mae", "huber", "log_cosh"])
@pytest.mark.parametrize("name", ["nif_swish", "ms_plain_r3_si2", "ll_plain"])
def test_other_keras_losses_match_torch_autograd(name, loss):
    """compile(loss=...) beyond 'mse' (keras.losses.get: 'mae', 'huber' with delta 1, 'log_cosh'): the oracle's loss / gradient against
    torch autograd; targets scaled so that |e| straddles Huber's delta"""
    pytest.importorskip("torch")
    from tests import torch_ref as T
    kind, cs, cp, spec, ws, inputs, y, sw = _setup(name, B=17)
    y = 3.0 * y
    l_, g_ = O.loss_and_grad(spec, ws, inputs, y, sw, loss=loss)
    tl, tg, _ = T.loss_and_grad(kind, cs, cp, ws, inputs, y, sw, loss=loss)
    assert abs(l_ - tl) <= 1e-12 * max(1.0, abs(tl))
    for g, t in zip(g_, tg):
        assert np.abs(g - t).max() / max(np.abs(t).max(), 1e-30) < 1e-9
    # the Sobolev step takes the same loss on both outputs: against central differences of the loss in two random directions
    xi = list(range(spec.pi, spec.pi + spec.si))
    rng = np.random.default_rng(3)
    gt = rng.uniform(-2, 2, size=(17, spec.so, len(xi)))
    l0, g0, _, _ = O.sobolev_loss_and_grad(spec, ws, inputs, y, gt, xi, 0.3, sw, loss=loss)
    for _ in range(2):
        d = [rng.standard_normal(w.shape) for w in ws]
        h = 1e-6
        lp = O.sobolev_loss_and_grad(spec, [w + h * q for w, q in zip(ws, d)], inputs, y, gt, xi, 0.3, sw, loss=loss)[0]
        lm = O.sobolev_loss_and_grad(spec, [w - h * q for w, q in zip(ws, d)], inputs, y, gt, xi, 0.3, sw, loss=loss)[0]
        dd = sum((a * b).sum() for a, b in zip(g0, d))
        assert abs((lp - lm) / (2 * h) - dd) <= 2e-5 * max(1.0, abs(dd)), ((lp - lm) / (2 * h), dd)


def test_phase16_stash_emulation():
    """the 16-bit phase stash of the 128-wide policy step (oracle.phase16, r5): the rebuilt argument is within 2 pi 2^-17 of the exact
    one modulo 2 pi, and the gradient that takes every stashed sine / cosine from it is a small, non-zero perturbation of the
    bf16-stash form (both classes that have it)"""
    rng = np.random.default_rng(5)
    a = rng.normal(size=4000) * 40.0
    a2 = O.phase16(a)
    d = np.abs(np.exp(1j * a) - np.exp(1j * a2))
    assert d.max() <= 2.0 * np.pi * 2.0 ** -17 * 1.0001 and d.max() > 1e-6
    assert np.abs(a2).max() <= np.pi * 1.0001
    from tests.test_gpu_parity import CONFIGS
    for name in ("ms_cfg3_128x3", "ll_cfg4_128x2_r10_so3"):
        (kind, cs, cp), B = CONFIGS[name]
        spec = O.Spec(kind, cs, cp)
        ws = O.init_weights(spec, np.random.default_rng(11))
        x = rng.uniform(-1, 1, size=(37, spec.pi + spec.si)); y = rng.uniform(-1, 1, size=(37, spec.so))
        fn = O.ll_policy_loss_and_grad if spec.kind == O.KIND_LL else O.planes_loss_and_grad
        l0, g0, _ = fn(spec, ws, x, y, rnd=O.bf16_round, stash_bf16=True)
        l1, g1, _ = fn(spec, ws, x, y, rnd=O.bf16_round, stash_bf16=True, stash_ph16=True)
        assert l0 == l1                                       # the forward pass does not see the stash
        e = np.linalg.norm(O.flatten(g1) - O.flatten(g0)) / np.linalg.norm(O.flatten(g0))
        assert 1e-7 < e < 1e-3, (name, e)


# ---- the frozen oracle (tests/golden/oracle_v1.npz, written by tests/golden/make_oracle_goldens.py): any later edit of the oracle's
# arithmetic fails here (or shows up as a diff of the fixture) -- SURVEY 7 step 1, VERDICT r5 item 7 -----------------------------------
def _frozen():
    import importlib.util
    sp = importlib.util.spec_from_file_location("make_oracle_goldens", os.path.join(GOLD, "make_oracle_goldens.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    return mod, np.load(os.path.join(GOLD, "oracle_v1.npz"))


def test_frozen_fixture_covers_its_case_list():
    mod, z = _frozen()
    assert sorted(z["names"].tolist()) == sorted(mod.CASES) and tuple(z["batches"].tolist()) == mod.BATCHES
    assert len(mod.CASES) >= 12 and mod.BATCHES == (7, 64, 257)
    kinds = {mod.CASES[n][0] for n in mod.CASES}
    assert kinds == {"NIF", "NIFMultiScale", "NIFMultiScaleLastLayerParameterized"}


def test_live_oracle_reproduces_frozen_vectors():
    """forward, loss, flat gradient, Jacobian and one Adam step of 13 configurations x B in {7, 64, 257}: today's oracle against the
    numbers it wrote when the fixture was made, from the STORED weights and inputs (nothing is redrawn), to 1e-13."""
    mod, z = _frozen()
    for name in sorted(mod.CASES):
        kind, cs, cp = mod.CASES[name]
        spec = O.Spec(kind, cs, cp)
        ws = O.unflatten(spec, z["%s/theta" % name])
        for B in mod.BATCHES:
            k = "%s/%d/" % (name, B)
            got = mod.evaluate(spec, ws, z[k + "x"], z[k + "y"], z[k + "sw"])
            for q in ("u", "loss", "grad", "jac", "theta1"):
                ref = z[k + q]
                scale = max(float(np.abs(ref).max()), 1e-300)
                assert np.asarray(got[q]).shape == ref.shape, (name, B, q)
                assert float(np.abs(np.asarray(got[q]) - ref).max()) <= 1e-13 * scale, (name, B, q)
            # ... and the stored inputs are what the generator would draw today (the fixture and its script belong together)
            _, ws2, x2, y2, sw2 = mod.draw(name, B)
            assert np.array_equal(O.flatten(ws2).astype(np.float32), z["%s/theta" % name]) and np.array_equal(x2, z[k + "x"])
            assert np.array_equal(y2, z[k + "y"]) and np.array_equal(sw2, z[k + "sw"])
