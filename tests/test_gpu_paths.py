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
    a = _run(tmp_path, "split", {"NIF_FP32_MFMA": "0", "NIF_PNET_STASH": "0"})
    b = _run(tmp_path, "fp32", {"NIF_FP32_MFMA": "1", "NIF_PNET_STASH": "1"})
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
