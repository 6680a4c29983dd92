"""The deferred row reduction of a plain step (r6: nif_adam_step_dev runs it fused with the update, k_reduce_adam; every other entry point of
the library runs it first).  Same summation order and update expressions as k_reduce + k_adam: the two forms must agree bit for bit, and no
API order may see a stale [grad | loss] buffer."""
import numpy as np
import pytest

from oracle import nif_oracle as O
from tests.test_gpu_parity import _cfg, _make, _rel

pytestmark = pytest.mark.gpu

CASES = {
    "snet6_4x64": (_cfg("NIFMultiScale", 64, 4, 32, 2, 1, 1, 1, 1), 4099),      # the benchmark net's kernels (k_snet6 + k_pnet_bwg)
    "small_nif_32x2": (_cfg("NIF", 32, 2, 32, 2, 1, 1, 1, 1), 512),             # k_small
    "ms_6x128": (_cfg("NIFMultiScale", 128, 3, 64, 2, 1, 2, 1, 1), 1031),       # k_snet4 + k_gw_*
}


def _steps(name, fuse, nsteps=3, weighted=False):
    import nif_amd
    m, model, spec, ws, x, y, sw = _make(CASES[name])
    e = m._engine
    e.set_option("fuse_tail", fuse)
    adam = nif_amd.Adam(1e-3).as_struct()
    d_x, d_y, d_sw = e.alloc(x.size), e.alloc(y.size), e.alloc(sw.size)
    d_x.upload(x); d_y.upload(y); d_sw.upload(sw)
    losses = []
    for _ in range(nsteps):
        e.loss_grad_dev(d_x.at(0), d_y.at(0), d_sw.at(0) if weighted else None, x.shape[0], x.shape[0])
        e.adam_step_dev(adam)
        losses.append(e.last_loss())          # AFTER the update: [grad | loss] must hold this step's sums in both forms
    th = O.flatten(model.get_weights())
    mm, vv, step = e.get_opt_state()
    _, g = e.grad_read()
    return np.array(losses), th, mm, vv, step, g


@pytest.mark.parametrize("name", sorted(CASES))
def test_fused_tail_is_bit_identical_to_reduce_then_adam(name):
    a = _steps(name, 1, weighted=(name == "small_nif_32x2"))
    b = _steps(name, 0, weighted=(name == "small_nif_32x2"))
    assert a[4] == b[4] == 3
    assert np.array_equal(a[0], b[0]), (a[0], b[0])
    for i in (1, 2, 3, 5):
        assert np.array_equal(a[i], b[i]), (name, i, float(np.abs(a[i] - b[i]).max()))


@pytest.mark.parametrize("name", ["snet6_4x64", "small_nif_32x2"])
def test_no_entry_point_sees_a_stale_gradient(name):
    import nif_amd
    m, model, spec, ws, x, y, sw = _make(CASES[name])
    e = m._engine
    B = x.shape[0]
    lref, gref = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64))
    gref = O.flatten(gref)
    d_x, d_y = e.alloc(x.size), e.alloc(y.size)
    d_x.upload(x); d_y.upload(y)
    xb = (x * 0.5).astype(np.float32)
    d_x2 = e.alloc(x.size); d_x2.upload(xb)
    # (1) loss_grad -> grad_read: the reduction runs inside the read
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
    loss, g = e.grad_read()
    assert abs(loss - lref) <= 3e-6 * abs(lref) and _rel(g, gref) < 3e-5
    # (2) loss_grad -> last_loss
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
    assert abs(e.last_loss() - lref) <= 3e-6 * abs(lref)
    # (3) loss_grad on OTHER inputs -> loss_grad on x (the first one's rows are dropped, not mixed in) -> raw pointer + d2h
    e.loss_grad_dev(d_x2.at(0), d_y.at(0), None, B, B)
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
    buf = type(d_x).__new__(type(d_x)); buf.engine, buf.n, buf.ptr = e, e.n_params + 1, e.grad_dev_ptr()
    gl = buf.download()
    buf.ptr = None                                          # (a view of the library's buffer: not this object's to free)
    assert abs(gl[-1] - lref) <= 3e-6 * abs(lref) and _rel(gl[:-1], gref) < 3e-5
    # (4) loss_grad -> zero_grad -> adam: an update with a ZERO gradient (first step: m = v = 0 -> theta unchanged)
    th0 = O.flatten(model.get_weights())
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
    e.zero_grad()
    e.adam_step_dev(nif_amd.Adam(1e-3).as_struct())
    assert np.array_equal(O.flatten(model.get_weights()), th0)
    # (5) a regulariser set between the gradient and the update is part of the update (reduce, add, then Adam -- not the fused form)
    e.set_opt_state(np.zeros_like(th0), np.zeros_like(th0), 0)
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
    e.set_regularizer(0.0, 0.5, 0, e.n_params)
    e.adam_step_dev(nif_amd.Adam(1e-3).as_struct())
    th1 = O.flatten(model.get_weights()).astype(np.float64)
    g_tot = gref + 2 * 0.5 * th0.astype(np.float64)
    c_eps = 1e-7 / np.sqrt(1.0 - 0.999)                     # Adam's first step: lr g / (|g| + eps / sqrt(1 - beta_2))
    want = th0 - 1e-3 * g_tot / (np.abs(g_tot) + c_eps)
    solid = np.abs(g_tot) > 1e-4
    assert np.abs(th1 - want)[solid].max() < 2e-6
    e.set_regularizer(0.0, 0.0, 0, 0)
    # (6) predictions between the gradient and the update do not disturb either
    e.set_opt_state(np.zeros_like(th0), np.zeros_like(th0), 0)
    model.set_weights(O.unflatten(spec, th0.astype(np.float32)))
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
    u = model.predict(x, batch_size=B)
    assert _rel(u, O.forward(spec, ws, x.astype(np.float64))) < 1e-5
    e.adam_step_dev(nif_amd.Adam(1e-3).as_struct())
    th2 = O.flatten(model.get_weights()).astype(np.float64)
    want2 = th0 - 1e-3 * gref / (np.abs(gref) + c_eps)
    solid2 = np.abs(gref) > 1e-4
    assert np.abs(th2 - want2)[solid2].max() < 2e-6
