"""per-config gradient error against the oracle (what tests/test_gpu_parity.py::test_loss_and_grad_match_oracle bounds): flat rel-L2 and the
worst tensor (err / (norm + 1e-6 |g|)) -- to set the test's bars from measurements"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_parity import CONFIGS, _make, _rel
from oracle import nif_oracle as O
rows = []
for name in sorted(CONFIGS):
    for weighted in (False, True):
        m, model, spec, ws, x, y, sw = _make(name)
        s = sw if weighted else None
        loss, g = m._engine.loss_and_grad(x, y, s)
        lref, gref = O.loss_and_grad(spec, ws, x.astype(np.float64), y.astype(np.float64), None if s is None else s.astype(np.float64))
        off, worst, wn = 0, 0.0, ""
        gnorm = np.linalg.norm(O.flatten(gref))
        for (nm, shp), gr in zip(spec.param_shapes(), gref):
            k = int(np.prod(shp)); gg = g[off:off + k].reshape(shp); off += k
            q = np.linalg.norm(gg - gr) / (np.linalg.norm(gr) + 5e-3 * gnorm)
            if q > worst: worst, wn = q, nm
        rows.append((name, weighted, abs(loss - lref) / abs(lref), _rel(g, O.flatten(gref)), worst, wn))
        m._engine.close()
for r in sorted(rows, key=lambda r: -r[4])[:25]:
    print("%-28s w=%d loss %.1e flat %.1e worst %.1e %s" % r)
print("max flat %.2e  max worst %.2e  max loss %.2e" % (max(r[3] for r in rows), max(r[4] for r in rows), max(r[2] for r in rows)))
