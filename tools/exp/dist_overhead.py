"""Where does the per-step cost of the data-parallel path come from at world size 1?  Times the same device-resident
step (a) bare, (b) after torch.cuda is initialised in the process, (c) after the nccl(=RCCL) process group exists,
(d) with the per-step all-reduce.   RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29514 python tools/exp/dist_overhead.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import nif_amd
from nif_amd import distributed as dist
from nif_amd.engine import DeviceArray

nif_amd.set_seed(1)
m = nif_amd.NIFMultiScale(bench.CFG_SHAPE, bench.CFG_PARAM)
m.build()
e = m._engine
B = 1 << 20
x, y = nif_amd.data.synthetic_wave_batch(B, seed=100)
d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size)
d_x.upload(x); d_y.upload(y)
adam = nif_amd.Adam(1e-3).as_struct()


def run(tag, allreduce=False, n=30):
    for _ in range(5):
        e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
        if allreduce:
            dist.all_reduce_grad(e)
        e.adam_step_dev(adam)
    e.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
        if allreduce:
            dist.all_reduce_grad(e)
        e.adam_step_dev(adam)
    e.sync()
    print("%-34s %.4f ms/step" % (tag, (time.perf_counter() - t0) / n * 1e3), flush=True)


run("bare")
run("bare (again)")
import torch
torch.cuda.init()
z = torch.zeros(1, device="cuda")
torch.cuda.synchronize()
run("after torch.cuda init")
dist.init("nccl")
run("after init_process_group")
run("with all_reduce", allreduce=True)
run("bare again, pg alive")
