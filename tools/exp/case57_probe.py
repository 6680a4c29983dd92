import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nif_amd
from oracle import nif_oracle as O
from tests.test_gpu_parity import _cfg, _per_tensor_rel
for seed in (0, 1, 2):
  for B in (16, 15, 17, 32, 48):
    rng = np.random.default_rng(seed)
    kind, cs, cp = _cfg("NIFMultiScale", 49, 4, 6, 1, 1, 2, 1, 2, p_act="tanh")
    spec = O.Spec(kind, cs, cp)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    names = [nm for nm, _ in spec.param_shapes()]
    ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 2.0).astype(np.float32)
    m = nif_amd.NIFMultiScale(cs, cp); model = m.build(); model.set_weights(ws); e = m._engine
    x = rng.uniform(-1, 1, size=(B, 4)).astype(np.float32); y = rng.uniform(-1, 1, size=(B, 1)).astype(np.float32)
    lref, gref = O.loss_and_grad(spec, [w.astype(np.float64) for w in ws], x.astype(np.float64), y.astype(np.float64))
    out = []
    for fuse in (1, 0):
        e.set_option("fuse_gw", fuse)
        loss, g = e.loss_and_grad(x, y)
        rel = _per_tensor_rel(spec, g, O.flatten(gref))
        out.append({k: "%.1e" % v for k, v in rel.items() if v > 5e-5})
    print("seed", seed, "B", B, "fused", out[0], "| stash path", out[1], flush=True)
    e.close()
