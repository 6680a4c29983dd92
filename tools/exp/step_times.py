"""per-step times of the benchmark step over a long run (host-synchronised single steps and back-to-back groups): is the gap between
bench.py's mean and median a clock ramp, a periodic throttle or noise?   python tools/exp/step_times.py [n_steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, nif_amd
from nif_amd.engine import DeviceArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = 1 << 20
nif_amd.set_seed(1)
m = nif_amd.NIFMultiScale(bench.CFG_SHAPE, bench.CFG_PARAM); m.build(); e = m._engine
x, y = nif_amd.data.synthetic_wave_batch(B, seed=100)
d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size); d_x.upload(x); d_y.upload(y)
adam = nif_amd.Adam(1e-3).as_struct(); e.reserve(B, 0)
def step():
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B); e.adam_step_dev(adam)
e.sync()
# back-to-back groups of 5 steps from a cold start (what the driver's warm-up + timed region see)
g = []
for i in range(n // 5):
    t = time.perf_counter()
    for _ in range(5): step()
    e.sync(); g.append((time.perf_counter() - t) / 5 * 1e3)
print("groups of 5 (ms/step):", " ".join("%.3f" % v for v in g))
per = []
for i in range(n):
    t = time.perf_counter(); step(); e.sync(); per.append((time.perf_counter() - t) * 1e3)
per = np.array(per)
print("single synced: min %.3f med %.3f mean %.3f p90 %.3f max %.3f" % (per.min(), np.median(per), per.mean(), np.percentile(per, 90), per.max()))
print("first 40:", " ".join("%.3f" % v for v in per[:40]))
