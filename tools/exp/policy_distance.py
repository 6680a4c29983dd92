"""distance of the 16-bit Keras policies from the EXACT fp64 oracle (not the emulating one): forward rel-L2, loss, flat gradient and
the worst tensor, per config -- the numbers behind tests/test_gpu_parity.py::test_policy_distance_from_exact_arithmetic"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nif_oracle as O
from tests.test_gpu_parity import _make_policy, _rel, _per_tensor_rel

for pol in ("mixed_bfloat16", "mixed_float16"):
    for name in ("ms_cfg2_64x4", "ms_cfg5_64x4_si2", "ms_cfg3_128x3", "nif_cfg1_32x2", "ll_plain_32x2_r3", "ll_cfg4_128x6_r10_so3", "ms_64x8"):
        try:
            m, model, spec, ws, x, y, sw = _make_policy(name, pol)
            x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
            u = model.predict(x)
            loss, g = m._engine.loss_and_grad(x, y, sw)
            el, eg = O.loss_and_grad(spec, ws, x64, y64, s64)
            rel = _per_tensor_rel(spec, g, O.flatten(eg))
            print(pol, name, "u %.2e loss %.2e grad %.2e worst %.2e" % (_rel(u, O.forward(spec, ws, x64)), abs(loss - el) / abs(el),
                                                                         _rel(g, O.flatten(eg)), max(rel.values())), flush=True)
        except Exception as ex:
            print(pol, name, "ERR", repr(ex)[:120], flush=True)
