import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import nif_amd
from nif_amd.engine import DeviceArray
sys.path.insert(0, "/root/repo/tools")
import bench_configs as bc
cls, (cs, cp), B, xi = bc.WORK["cfg4_linear_nif_3d"]
B = 1 << 19
nif_amd.set_seed(1)
m = getattr(nif_amd, cls)(cs, cp); m.build(); e = m._engine
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, size=(B, 4)).astype(np.float32); y = rng.uniform(-1, 1, size=(B, 3)).astype(np.float32)
d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size); d_x.upload(x); d_y.upload(y)
for _ in range(3): e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
e.sync()
e.lib.nif_debug_timeline(e.ctx, None, 0)
e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B); e.sync()
buf = (C.c_int64 * 500)(); e.lib.nif_debug_timeline(e.ctx, buf, 250)
a = np.array(buf[:]).reshape(-1, 2); a = a[a[:, 0] != 0]
prev = a[0, 1]
for idv, t in a[:40]:
    print(int(idv), int(t - prev)); prev = t
