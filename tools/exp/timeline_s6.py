"""s_memtime timeline of producer wave 0 and consumer wave 8 of block 0 of k_snet6, third tile round (needs a -DNIF_TIMELINE build:
python tools/build_variant.py tl "-DNIF_TIMELINE" k_snet6.hip; NIF_LIB=nif_amd/libnif_hip_tl.so python tools/exp/timeline_s6.py)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
for _ in range(30):
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
e.sync()
e.lib.nif_debug_timeline(e.ctx, None, 0)  # arm
e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
e.sync()
buf = (C.c_int64 * 4096)()
e.lib.nif_debug_timeline(e.ctx, buf, 2048)
a = np.array(buf[:]).reshape(-1, 2)
for name, lo in (("producer wave 0", 0), ("consumer wave 8", 1024)):
    q = a[lo:lo + 1024]
    q = q[q[:, 0] != 0]
    if not len(q):
        print(name, "no stamps"); continue
    print("==", name, len(q), "stamps; round length", q[-1, 1] - q[0, 1], "ticks")
    prev = q[0, 1]
    step = 0
    line = []
    for idv, t in q:
        if idv == 100 or idv == 1:
            if line: print(" ".join(line))
            line = ["step %2d" % step]; step += idv == 100
        line.append("%s+%d" % ({1: "round", 100: "V", 200: "dma", 300: "mfma", 400: "wait", 500: "bar"}.get(int(idv), str(idv)), t - prev))
        prev = t
    print(" ".join(line))
