"""k_latent_to_w_flat alone: GB/s of the [B, po] write stream at the benchmark's shape (po = 16 833, r = 1), B = 2^17 rows"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, nif_amd
from nif_amd.engine import DeviceArray
from nif_amd._lib import check
nif_amd.set_seed(1)
WIDE = len(sys.argv) > 1 and sys.argv[1] == "wide"      # configs[2]'s 6 x 128 net: the column-window form (the hyper layer does not fit the LDS)
cs = dict(bench.CFG_SHAPE, units=128, nlayers=6, input_dim=2) if WIDE else bench.CFG_SHAPE
m = nif_amd.NIFMultiScale(cs, bench.CFG_PARAM); m.build(); e = m._engine; s = m._spec
Bw = 1 << (14 if WIDE else 17)
d_lr = DeviceArray(e, Bw * s.pi_hidden); d_lr.upload(np.random.default_rng(7).standard_normal(Bw * s.pi_hidden).astype(np.float32))
d_w = DeviceArray(e, Bw * s.po_dim)
for _ in range(3):
    check(e.lib.nif_latent_to_w_dev(e.ctx, d_lr.at(0), Bw, d_w.at(0)))
e.sync()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        check(e.lib.nif_latent_to_w_dev(e.ctx, d_lr.at(0), Bw, d_w.at(0)))
    e.sync()
    best = min(best, (time.perf_counter() - t0) / 10)
print("%s latent_to_w%s po=%d %.1f GB/s (%.3f ms)" % (os.environ.get("NIF_LIB", "product"), " [window form]" if WIDE else "", s.po_dim, 4.0 * s.po_dim * Bw / best / 1e9, best * 1e3))
