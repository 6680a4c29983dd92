"""GPU-side mechanics of the one collective of the step, exercised at world_size 1 over the RCCL backend:
torch tensor aliasing the library's [grad | loss] buffer (__cuda_array_interface__), all-reduce issued on the
library's own HIP stream (ExternalStream), bench.py's distributed code path."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_SCRIPT = r'''
import os, sys, numpy as np
sys.path.insert(0, %(root)r)
import nif_amd
from nif_amd import distributed as dist
from nif_amd.engine import DeviceArray
from oracle import nif_oracle as O
from tests.test_gpu_parity import _cfg
rank, world = dist.init("nccl")
assert world == 1
kind, cs, cp = _cfg("NIFMultiScale", 64, 2, 32, 2, 1, 1, 1, 1)
nif_amd.set_seed(3)
m = nif_amd.NIFMultiScale(cs, cp); model = m.build(); e = m._engine
x, y = nif_amd.data.synthetic_wave_batch(4096, seed=1)
loss, g = e.loss_and_grad(x, y)
d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size)
d_x.upload(x); d_y.upload(y)
e.loss_grad_dev(d_x.at(0), d_y.at(0), None, 4096, 4096)
t, stream = dist.grad_tensor(e)
dist.all_reduce_grad(e)          # SUM over 1 rank: must leave the buffer unchanged
e.sync()
import torch
torch.cuda.synchronize()
alias = t.cpu().numpy()
assert alias.shape == (e.n_params + 1,)
assert np.array_equal(alias[:-1], g) and abs(alias[-1] - loss) < 1e-12, (alias[-1], loss)
# the alias really is the library buffer: Adam consumes what the all-reduce left there
adam = nif_amd.Adam(1e-3).as_struct()
w0 = O.flatten(model.get_weights())
e.adam_step_dev(adam); e.sync()
w1 = O.flatten(model.get_weights())
th, _, _ = O.adam_step(w0.astype(np.float64), g.astype(np.float64), 0.0, 0.0, 1, lr=1e-3)
assert np.abs(w1 - th).max() < 1e-6
# fit() through its multi-rank code path (agreed batch sizes, all-reduce on the library stream every step): with one
# real rank the sums are identities, so the history must equal the single-process one exactly
nif_amd.set_seed(5)
ma = nif_amd.NIFMultiScale(cs, cp); a = ma.build(); a.compile(nif_amd.Adam(1e-3), "mse")
nif_amd.set_seed(5)
mb = nif_amd.NIFMultiScale(cs, cp); b = mb.build(); b.compile(nif_amd.Adam(1e-3), "mse")
ha = a.fit(x, y, epochs=2, batch_size=1000, shuffle=False, verbose=0)
real_ws = dist.world_size
dist.world_size = lambda: 2
hb = b.fit(x, y, epochs=2, batch_size=1000, shuffle=False, verbose=0)
dist.world_size = real_ws
assert ha.history["loss"] == hb.history["loss"], (ha.history, hb.history)
assert np.array_equal(O.flatten(a.get_weights()), O.flatten(b.get_weights()))
dist.zero_grad(mb._engine); mb._engine.sync(); torch.cuda.synchronize()
assert not dist.grad_tensor(mb._engine)[0].cpu().numpy().any()
assert dist.all_reduce_ints([3, 4, 5]) == [3, 4, 5] and dist.all_reduce_ints([7], op="max") == [7]
dist.shutdown()
print("OK")
'''


def test_allreduce_aliases_library_buffer_world1():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT}], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-3000:]


def test_bench_runs_under_torchrun_single_rank():
    """bench.py's N>1 code path (process group, barrier, max-over-ranks) launched the way the driver does."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps",
           "2", "--warmup", "1", "--points", "65536", "--no-cpu-baseline", "--given-w-points", "4096",
           "--force-dist"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    import json
    # the contract: ONE line on stdout (RCCL's version banner and everything else goes to stderr)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "roofline" in d
