"""k_snet5 (hidden-layer weight gradients fused into the forward/adjoint kernel) against the k_snet4 + k_gw_lds path"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nif_amd, bench
from oracle import nif_oracle as O
spec = O.Spec("NIFMultiScale", bench.CFG_SHAPE, bench.CFG_PARAM)
for B in (100, 4096, 70001, 1 << 20):
    nif_amd.set_seed(1)
    m = nif_amd.NIFMultiScale(bench.CFG_SHAPE, bench.CFG_PARAM); model = m.build(); e = m._engine
    x, y = nif_amd.data.synthetic_wave_batch(B, seed=5)
    sw = np.random.default_rng(0).uniform(0.5, 1.5, B).astype(np.float32)
    l0, g0 = e.loss_and_grad(x, y, sw)
    e.set_option("fused_gw", 1)
    l1, g1 = e.loss_and_grad(x, y, sw)
    l2, g2 = e.loss_and_grad(x, y, sw)
    off = 0; worst = ("", 0.0)
    for nm, shp in spec.param_shapes():
        k = int(np.prod(shp)); a, b = g1[off:off + k].astype(np.float64), g0[off:off + k].astype(np.float64); off += k
        rel = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        if rel > worst[1]: worst = (nm, rel)
    print("B=%d loss %.8e vs %.8e  grad rel %.3e  worst tensor %s %.3e  deterministic %s" % (
        B, l1, l0, np.linalg.norm(g1 - g0) / np.linalg.norm(g0), worst[0], worst[1], np.array_equal(g1, g2)), flush=True)
    if B <= 4096:
        ws = [w.astype(np.float64) for w in model.get_weights()]
        lr, gr = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))
        gr = O.flatten(gr)
        print("   vs oracle: fused %.3e  unfused %.3e" % (np.linalg.norm(g1 - gr) / np.linalg.norm(gr), np.linalg.norm(g0 - gr) / np.linalg.norm(gr)))
    e.close()
