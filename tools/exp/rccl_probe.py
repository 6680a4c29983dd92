"""step-by-step probe of the direct RCCL path at world size 1 (prints flushed: shows where a hang sits)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
def say(*a):
    print("[%.2f]" % (time.time() - T0), *a, flush=True)
T0 = time.time()
import numpy as np
import nif_amd
from nif_amd import _lib
from nif_amd._lib import check
say("imported")
lib = _lib.load()
say("lib loaded, devices:", lib.nif_device_count())
from tests.test_gpu_parity import _cfg
kind, cs, cp = _cfg("NIFMultiScale", 64, 2, 32, 2, 1, 1, 1, 1)
m = nif_amd.NIFMultiScale(cs, cp); model = m.build(); e = m._engine
say("engine created")
buf = C.create_string_buffer(128)
check(lib.nif_comm_unique_id(buf)); say("unique id ok")
mode = sys.argv[1] if len(sys.argv) > 1 else "rank"
if mode == "rank":
    check(lib.nif_comm_init_rank(e.ctx, buf.raw, 0, 1)); say("comm_init_rank ok")
else:
    arr = (C.c_void_p * 1)(e.ctx)
    check(lib.nif_comm_init_all(arr, 1)); say("comm_init_all ok")
x, y = nif_amd.data.synthetic_wave_batch(4096, seed=1)
d_x, d_y = e.alloc(x.size), e.alloc(y.size); d_x.upload(x); d_y.upload(y)
e.loss_grad_dev(d_x.at(0), d_y.at(0), None, 4096, 4096); e.sync(); say("loss_grad ok")
check(lib.nif_allreduce_grad(e.ctx)); say("allreduce enqueued")
e.sync(); say("allreduce done")
check(lib.nif_comm_barrier(e.ctx)); say("barrier ok")
check(lib.nif_comm_destroy(e.ctx)); say("destroy ok")
