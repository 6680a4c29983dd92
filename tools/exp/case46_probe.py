import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np
import fuzz_parity as F
import nif_amd
from oracle import nif_oracle as O
rng = np.random.default_rng(8)
for i in range(47):
    cfg, B, desc = F.draw(rng)
print(desc)
kind, cs, cp = cfg
spec = O.Spec(kind, cs, cp)
r = np.random.default_rng(8 * 1000 + 46)
ws = O.init_weights(spec, r, dtype=np.float32)
m = getattr(nif_amd, kind)(cs, cp); model = m.build(); model.set_weights(ws)
x = r.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32); y = r.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
for xi in ([4], [5, 6], [4, 5, 6], [0], [0, 1], [0, 1, 2], [0, 4], [0, 1, 4], [0, 1, 2, 3, 4, 5, 6]):
    g = r.uniform(-1, 1, size=(B, spec.so, len(xi))).astype(np.float32)
    try:
        l, gr = m._engine.sobolev_loss_and_grad(x, y, g, xi, 0.05, None)
        print(xi, "ok", l)
    except Exception as ex:
        print(xi, "FAIL", str(ex)[:160])
