"""A/B of the alternative kernel paths on identical inputs, each in its own process (the switches are read once):
bf16-split products (k_snet4, default) vs f32-input MFMAs (NIF_FP32_MFMA=1 -> k_snet3), and the stash-free
ParameterNet adjoint (k_pnet_bwg, default) vs the stash path (NIF_PNET_STASH=1), LDS-DMA vs register-load
weight-gradient kernels (NIF_GW_LDS=0).  All must agree with the fp64
oracle to the parity bar AND with each other far inside it -- the split products are an fp32-exact reformulation,
not a lower-precision mode."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import nif_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import bench, nif_amd
from oracle import nif_oracle as O
out = sys.argv[1]
spec = O.Spec("NIFMultiScale", bench.CFG_SHAPE, bench.CFG_PARAM)
rng = np.random.default_rng(5)
ws = O.init_weights(spec, rng, dtype=np.float32)
names = [nm for nm, _ in spec.param_shapes()]
ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 2.0).astype(np.float32)
m = nif_amd.NIFMultiScale(bench.CFG_SHAPE, bench.CFG_PARAM)
model = m.build(); model.set_weights(ws)
x, y = nif_amd.data.synthetic_wave_batch(20000, seed=3)
u = model.predict(x)
loss, grad = m._engine.loss_and_grad(x, y, None)
np.savez(out, u=u, loss=np.float64(loss), grad=grad, x=x, y=y, **{"w%%d" %% i: w for i, w in enumerate(ws)})
''' % ROOT


def _run(tmp_path, tag, env_extra):
    out = str(tmp_path / (tag + ".npz"))
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT, out], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    return np.load(out)


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))


def test_bf16_split_and_fp32_mfma_paths_agree(tmp_path):
    a = _run(tmp_path, "split", {"NIF_FP32_MFMA": "0", "NIF_PNET_STASH": "0", "NIF_FUSE_EDGE": "0"})
    b = _run(tmp_path, "fp32", {"NIF_FP32_MFMA": "1", "NIF_PNET_STASH": "1"})
    c = _run(tmp_path, "edge", {"NIF_FUSE_EDGE": "1"})       # opt-in: first/last-layer gradients fused into k_snet4
    assert _rel(c["u"], a["u"]) < 1e-7 and _rel(c["grad"], a["grad"]) < 2e-5, _rel(c["grad"], a["grad"])
    # weight-gradient kernels with register loads (k_gw_mfma / k_gw_first_mfma / k_gw_out) instead of the LDS-DMA forms
    d = _run(tmp_path, "gwreg", {"NIF_GW_LDS": "0"})
    assert _rel(d["u"], a["u"]) == 0.0 and _rel(d["grad"], a["grad"]) < 2e-5, _rel(d["grad"], a["grad"])
    # the two GPU formulations against each other
    assert _rel(a["u"], b["u"]) < 2e-6, _rel(a["u"], b["u"])
    assert abs(float(a["loss"]) - float(b["loss"])) <= 2e-6 * abs(float(b["loss"]))
    assert _rel(a["grad"], b["grad"]) < 5e-5, _rel(a["grad"], b["grad"])
    # and each against the fp64 oracle (same weights, same inputs)
    import bench
    spec = O.Spec("NIFMultiScale", bench.CFG_SHAPE, bench.CFG_PARAM)
    ws = [a["w%d" % i].astype(np.float64) for i in range(len(spec.param_shapes()))]
    ref = O.forward(spec, ws, a["x"].astype(np.float64))
    rl, rg = O.loss_and_grad(spec, ws, a["x"].astype(np.float64), a["y"].astype(np.float64))
    for d in (a, b):
        assert _rel(d["u"], ref) < 1e-5, _rel(d["u"], ref)
        assert abs(float(d["loss"]) - rl) <= 1e-5 * abs(rl)
        assert _rel(d["grad"], O.flatten(rg)) < 2e-4, _rel(d["grad"], O.flatten(rg))


@pytest.mark.parametrize("name", ["ms_cfg2_64x4", "ms_64x2_r1_so4", "ms_cfg5_64x4_si2", "ms_32x2_r3_si2"])
def test_fused_weight_gradient_kernel_matches_the_stash_path_and_the_oracle(name):
    """k_snet5 (opt-in `fused_gw`): the hidden layers' weight gradients accumulated inside the forward/adjoint kernel in
    per-wave MFMA accumulators (deposits through LDS, no h / dL/da stash round trip through HBM) against the default
    k_snet4 + k_gw_lds path and the fp64 oracle; ragged batch sizes; bit-identical when repeated."""
    import nif_amd
    from tests.test_gpu_parity import CONFIGS, _cfg, _per_tensor_rel
    if name == "ms_32x2_r3_si2":
        (kind, cs, cp), B = _cfg("NIFMultiScale", 32, 2, 32, 1, 3, 2, 1, 1), 777
    else:
        (kind, cs, cp), B = CONFIGS[name]
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    names = [nm for nm, _ in spec.param_shapes()]
    ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 2.0).astype(np.float32)
    m = nif_amd.NIFMultiScale(cs, cp)
    model = m.build(); model.set_weights(ws)
    e = m._engine
    for Bn in (B, 17, 4099):
        x = rng.uniform(-1, 1, size=(Bn, spec.pi + spec.si)).astype(np.float32)
        y = rng.uniform(-1, 1, size=(Bn, spec.so)).astype(np.float32)
        sw = rng.uniform(0.5, 1.5, size=(Bn,)).astype(np.float32)
        e.set_option("fused_gw", 0)
        l0, g0 = e.loss_and_grad(x, y, sw)
        e.set_option("fused_gw", 1)
        l1, g1 = e.loss_and_grad(x, y, sw)
        l2, g2 = e.loss_and_grad(x, y, sw)
        assert l1 == l2 and np.array_equal(g1, g2)
        assert abs(l1 - l0) <= 2e-6 * abs(l0)
        assert max(_per_tensor_rel(spec, g1, g0.astype(np.float64)).values()) < 5e-5
        if Bn <= 1100:
            ws64 = [w.astype(np.float64) for w in ws]
            lr, gr = O.loss_and_grad(spec, ws64, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))
            assert abs(l1 - lr) <= 1e-5 * abs(lr)
            assert max(_per_tensor_rel(spec, g1, O.flatten(gr)).values()) < 2e-4
    e.set_option("fused_gw", 0)
