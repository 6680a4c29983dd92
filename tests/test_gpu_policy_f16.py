"""Keras' `mixed_float16` policy (reference nif/model.py:101-105: the name goes to tf.keras.mixed_precision.Policy) on the HIP path:
k_snet4<.., PR = 2> (csrc/k_snet4_f16.hip) against the oracle with the same casts -- half-precision operands of the hidden n x n
products (RNE, saturated), dL/da rounded under the 2^15 loss scale, everything else fp32 (oracle/nif_oracle.py f16_round)."""
import numpy as np
import pytest

from oracle import nif_oracle as O
from tests.test_gpu_parity import CONFIGS, _make_policy, _per_tensor_rel, _rel

pytestmark = pytest.mark.gpu

F16 = ["ms_cfg2_64x4", "ms_cfg5_64x4_si2", "ms_64x2_mlp_pnet_r3", "ms_64x3_r3_so2_b33", "nif_cfg1_32x2", "ms_64x8", "ms_cfg3_128x6",
       "ms_res_128x4_r4_so2"]


@pytest.mark.parametrize("name", F16)
def test_mixed_float16_policy_matches_the_oracle_with_the_same_casts(name):
    """predictions and loss to 5e-4 of the emulating oracle (half has 11 significand bits: a flipped rounding is 2^-11 of one
    operand), gradients to 2e-3 per tensor; the policy itself sits 1e-4 .. 1e-2 from exact arithmetic -- closer than
    mixed_bfloat16 (8 bits) on the same shapes, which the last assertion checks on the predictions"""
    m, model, spec, ws, x, y, sw = _make_policy(name, "mixed_float16")
    assert m.compute_Dtype == "float16" and m.variable_Dtype == "float32" and m.mixed_policy_name == "mixed_float16"
    x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
    rl, rg, ru = O.planes_loss_and_grad(spec, ws, x64, y64, s64, rnd=O.f16_round)
    u = model.predict(x)
    assert _rel(u, ru) < 5e-4, _rel(u, ru)
    loss, g = m._engine.loss_and_grad(x, y, sw)
    assert abs(loss - rl) <= 5e-4 * abs(rl), (loss, rl)
    rel = _per_tensor_rel(spec, g, O.flatten(rg))
    # (128-wide nets: 3e-3, the bar the mixed_bfloat16 cases of that width hold -- six 128 x 128 layers carry a flipped rounding
    # of one operand further; measured 1.0e-3 .. 2.2e-3 on ms_cfg3_128x6, <= 1e-3 on the 64-wide nets)
    assert max(rel.values()) < (3e-3 if spec.n > 64 else 2e-3), rel
    # the policy IS a different computation, bounded, and finer than bf16's
    exact_u = O.forward(spec, ws, x64)
    d_u = _rel(u, exact_u)
    assert 1e-7 < d_u < 2e-2, d_u
    assert _rel(g, O.flatten(O.loss_and_grad(spec, ws, x64, y64, s64)[1])) < 0.1
    mb, modelb, *_ = _make_policy(name, "mixed_bfloat16")
    assert d_u < _rel(modelb.predict(x), exact_u), (d_u, _rel(modelb.predict(x), exact_u))
    # the fp32 model on the same weights is untouched
    m32, model32, *_ = _make_policy(name, "float32")
    assert _rel(model32.predict(x), exact_u) < 1e-5


@pytest.mark.parametrize("name", ["ll_plain_32x2_r3", "ll_cfg4_128x2_r10_so3", "ll_res_64x1_r4_so2", "ll_96x2_r5", "ll_64x3_r5_so2"])
def test_mixed_float16_policy_on_the_last_layer_class(name):
    """the SHARED hidden products of NIFMultiScaleLastLayerParameterized in half precision (k_snet4<.., LL, 2>); phi layer, Dot, loss
    and weight-gradient sums fp32 -- oracle ll_policy_loss_and_grad(rnd=f16_round)"""
    m, model, spec, ws, x, y, sw = _make_policy(name, "mixed_float16")
    names = [nm for nm, _ in spec.param_shapes()]
    ws[names.index("pnet_last_w")] = ws[names.index("pnet_last_w")] * 30.0       # weight_init_factor 0.01 makes the r x r map ~0
    model.set_weights([w.astype(np.float32) for w in ws])
    ws = [w.astype(np.float32).astype(np.float64) for w in ws]
    x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
    rl, rg, ru = O.ll_policy_loss_and_grad(spec, ws, x64, y64, s64, rnd=O.f16_round)
    u = model.predict(x)
    assert _rel(u, ru) < 5e-4, _rel(u, ru)
    loss, g = m._engine.loss_and_grad(x, y, sw)
    assert abs(loss - rl) <= 5e-4 * abs(rl), (loss, rl)
    rel = _per_tensor_rel(spec, g, O.flatten(rg))
    assert max(rel.values()) < 3e-3, rel
    d_u = _rel(u, O.forward(spec, ws, x64))
    assert 1e-7 < d_u < 2e-2, d_u


def test_loss_scale_keeps_small_and_large_adjoints():
    """what the loss scale is for: at a global batch of 2^20 points dL/da ~ 1e-7 sits below half's normal range (6.1e-5) and
    partly below its subnormals (6e-8).  The kernel's gradient stays within 3e-3 per tensor of the oracle that rounds
    half(s dL/da) under the per-point power of two, while the oracle WITHOUT a scale is far off (most of dL/da flushes to zero);
    and targets 1e6 times larger (|dL/da| ~ 1e3: a fixed 2^15 scale would overflow half) change nothing either"""
    m, model, spec, ws, x, y, sw = _make_policy("ms_cfg2_64x4", "mixed_float16")
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    Bg = 1 << 20
    e = m._engine
    from nif_amd.engine import DeviceArray
    d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size)
    d_x.upload(x); d_y.upload(y)
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, x.shape[0], Bg)          # this rank's shard of a 2^20-point global batch
    buf = DeviceArray.__new__(DeviceArray)
    buf.engine, buf.n, buf.ptr = e, e.n_params + 1, e.grad_dev_ptr()
    g = buf.download()[:-1]
    buf.ptr = None
    d_x.free(); d_y.free()
    rl, rg, _ = O.planes_loss_and_grad(spec, ws, x64, y64, batch_global=Bg, rnd=O.f16_round)
    rel = _per_tensor_rel(spec, g, O.flatten(rg))
    assert max(rel.values()) < 3e-3, rel
    unscaled = lambda a: O.f16_round(a)
    unscaled.grad = O.f16_round        # no loss scale: dL/da rounded as it is
    rg0 = O.planes_loss_and_grad(spec, ws, x64, y64, batch_global=Bg, rnd=unscaled)[1]
    assert _rel(g, O.flatten(rg0)) > 20 * _rel(g, O.flatten(rg)), (_rel(g, O.flatten(rg0)), _rel(g, O.flatten(rg)))
    ybig = (y * np.float32(1e6)).astype(np.float32)
    loss, g = e.loss_and_grad(x, ybig)
    rl, rg, _ = O.planes_loss_and_grad(spec, ws, x64, ybig.astype(np.float64), rnd=O.f16_round)
    assert np.all(np.isfinite(g)) and abs(loss - rl) <= 5e-4 * abs(rl)
    rel = _per_tensor_rel(spec, g, O.flatten(rg))
    assert max(rel.values()) < 3e-3, rel


def test_mixed_float16_fit_and_derivative_layers():
    """fit() under the policy follows the emulating oracle's Adam trajectory; the Sobolev step and the Jacobian layer run the exact
    products under this policy (include/nif_hip.h): equal to the float32 model's"""
    import nif_amd
    m, model, spec, ws, x, y, sw = _make_policy("ms_cfg5_64x4_si2", "mixed_float16")
    model.compile(nif_amd.Adam(2e-4), "mse")
    h = model.fit(x, y, epochs=3, batch_size=x.shape[0], shuffle=False, verbose=0)
    th = O.flatten(ws); mm = np.zeros_like(th); vv = np.zeros_like(th)
    f32 = lambda a: float(np.float32(a))
    losses = []
    for t in range(1, 4):
        l, g, _ = O.planes_loss_and_grad(spec, O.unflatten(spec, th), x.astype(np.float64), y.astype(np.float64), rnd=O.f16_round)
        losses.append(l)
        th, mm, vv = O.adam_step(th, O.flatten(g), mm, vv, t, lr=f32(2e-4), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
    assert np.allclose(h.history["loss"], losses, rtol=2e-3), (h.history["loss"], losses)
    m, model, spec, ws, x, y, sw = _make_policy("ms_cfg5_64x4_si2", "mixed_float16")
    m32, model32, *_ = _make_policy("ms_cfg5_64x4_si2", "float32")
    xi = [1, 2]
    gt = np.random.default_rng(3).uniform(-1, 1, size=(x.shape[0], 1, 2)).astype(np.float32)
    l16, g16 = m._engine.sobolev_loss_and_grad(x, y, gt, xi, 0.05, sw)
    l32, g32 = m32._engine.sobolev_loss_and_grad(x, y, gt, xi, 0.05, sw)
    assert abs(l16 - l32) <= 1e-6 * abs(l32) and _rel(g16, g32) < 1e-6
    pl, pg, *_ = O.sobolev_loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), gt.astype(np.float64), xi, 0.05,
                                         sw.astype(np.float64))
    assert abs(l16 - pl) <= 2e-5 * abs(pl) and max(_per_tensor_rel(spec, g16, O.flatten(pg)).values()) < 3e-4
