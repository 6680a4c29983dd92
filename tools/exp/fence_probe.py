import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as td
import bench, nif_amd
from nif_amd import distributed as dist
from nif_amd.engine import DeviceArray
dist.init("nccl")
nif_amd.set_seed(1)
m = nif_amd.NIFMultiScale(bench.CFG_SHAPE, bench.CFG_PARAM); m.build(); e = m._engine
B = 1 << 20
x, y = nif_amd.data.synthetic_wave_batch(B, seed=100)
d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size); d_x.upload(x); d_y.upload(y)
adam = nif_amd.Adam(1e-3).as_struct()
def step():
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B); dist.all_reduce_grad(e); e.adam_step_dev(adam)
def t(f, n=5):
    out = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); out.append((time.perf_counter() - t0) * 1e3)
    return ["%.3f" % v for v in out]
for _ in range(5): step()
e.sync()
print("e.sync (idle)      ", t(e.sync))
print("td.barrier         ", t(td.barrier))
print("cuda.synchronize   ", t(torch.cuda.synchronize))
def region(K):
    e.sync(); td.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    e.sync(); t2 = time.perf_counter()
    td.barrier(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print("K=%d: launch loop %.3f ms, e.sync %.3f, barrier %.3f, sync %.3f -> %.4f ms/step" % (K, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4-t0)*1e3/K))
for K in (10, 20, 10, 20):
    region(K)
dist.shutdown()
