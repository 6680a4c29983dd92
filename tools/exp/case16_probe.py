import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import nif_oracle as O
from tests.test_gpu_parity import _cfg, _make, _per_tensor_rel
for B in (1, 2, 16, 17, 33):
    for fuse in (1, 0):
        m, model, spec, ws, x, y, sw = _make((_cfg("NIFMultiScale", 56, 3, 20, 3, 1, 1, 3, 3, p_act="swish"), B))
        m._engine.set_option("fuse_gw", fuse)
        loss, g = m._engine.loss_and_grad(x, y, sw)
        lref, gref = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))
        rel = _per_tensor_rel(spec, g, O.flatten(gref))
        print("B", B, "fuse_gw", fuse, "loss", abs(loss - lref) / abs(lref), {k: "%.1e" % v for k, v in rel.items() if v > 1e-4})
