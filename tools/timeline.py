"""Print the per-phase cycle timeline of one wavefront of k_snet3 (needs a -DNIF_TIMELINE build)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nif_amd  # noqa: E402
from nif_amd.engine import DeviceArray  # noqa: E402

nif_amd.set_seed(1)
m = nif_amd.NIFMultiScale(bench.CFG_SHAPE, bench.CFG_PARAM)
model = m.build()
e = m._engine
B = 1 << 20
x, y = nif_amd.data.synthetic_wave_batch(B, seed=100)
d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size)
d_x.upload(x); d_y.upload(y)
for _ in range(3):
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
e.sync()
e.lib.nif_debug_timeline(e.ctx, None, 0)  # arm
e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
e.sync()
buf = (C.c_int64 * 500)()
e.lib.nif_debug_timeline(e.ctx, buf, 250)
a = np.array(buf[:]).reshape(-1, 2)
a = a[a[:, 0] != 0]
t0 = a[0, 1]
prev = t0
names = {1: "tile start", 2: "first layer done", 3: "hidden fwd done", 4: "last layer + loss done", 5: "hidden bwd done"}
for i, (idv, t) in enumerate(a[:60]):
    nm = names.get(int(idv), ("fwd j=%d planes start" % (idv - 10)) if 10 <= idv < 30 else
                   ("fwd j=%d planes done" % (idv - 30)) if 30 <= idv < 50 else
                   ("bwd j=%d planes start" % (idv - 50)) if 50 <= idv < 70 else ("bwd j=%d planes done" % (idv - 70)))
    print("%3d %-28s +%7d  (t=%8d)" % (idv, nm, t - prev, t - t0))
    prev = t
