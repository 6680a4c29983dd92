"""The C/OpenMP restatement of the reference formulation (oracle/nif_ref_cpu.c, the CPU baseline of
bench.py) against the NumPy oracle."""
import numpy as np
import pytest

from oracle import nif_oracle as O
from oracle import ref_cpu as R
from tests.cfgs import cfg_ms, cfg_nif


@pytest.mark.parametrize("cfg", [cfg_nif(n=16, L=2, nst=8, lst=2, r=2, si=2, so=2, pi=1),
                                 cfg_ms(n=16, L=3, nst=8, lst=2, r=1, si=1, so=1, pi=1),
                                 cfg_ms(n=8, L=1, nst=6, lst=1, r=3, si=2, so=1, pi=2, p_act="swish")])
def test_c_restatement_matches_numpy_oracle(cfg):
    kind, cs, cp = cfg
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    B = 300
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32)
    lib = R.load()
    c = R.make_cfg(spec)
    assert lib.nifref_nparams(R.C.byref(c)) == spec.n_params()
    for weights in (None, sw):
        loss, g = R.loss_and_grad(lib, c, O.flatten(ws), x, y, weights, micro=128)
        lref, gref = O.loss_and_grad(spec, [w.astype(np.float64) for w in ws], x.astype(np.float64),
                                     y.astype(np.float64), None if weights is None else weights.astype(np.float64))
        gref = O.flatten(gref)
        assert abs(loss - lref) < 1e-5 * abs(lref)
        assert np.linalg.norm(g - gref) < 1e-4 * np.linalg.norm(gref)


def test_c_adam_matches_oracle():
    lib = R.load()
    rng = np.random.default_rng(1)
    th = rng.standard_normal(1000).astype(np.float32)
    g = rng.standard_normal(1000).astype(np.float32)
    m = np.zeros_like(th); v = np.zeros_like(th)
    th2 = th.copy()
    lib.nifref_adam(th2.ctypes.data, g.ctypes.data, m.ctypes.data, v.ctypes.data, 1000, 1, 1e-3, 0.9, 0.999, 1e-7)
    ref, _, _ = O.adam_step(th.astype(np.float64), g.astype(np.float64), 0.0, 0.0, 1)
    assert np.abs(th2 - ref).max() < 1e-6
