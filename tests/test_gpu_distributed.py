"""GPU-side mechanics of the one collective of the step, exercised with one real rank over RCCL called through the
C-ABI (nif_comm.hip): the communicator built from a rendezvoused unique id (one process per GPU) and by
ncclCommInitAll (one process, n GPUs), the all-reduce on the library's own stream and buffer, Model.fit through its
multi-rank code path, the zero-gradient step, bench.py launched the way the driver does and launching itself."""
import ctypes as C
import json
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
assert "torch" not in sys.modules
rank, world = dist.init()
comm = dist.get()
assert (rank, world) == (0, 1) and dist.local_device() == 0
kind, cs, cp = _cfg("NIFMultiScale", 64, 2, 32, 2, 1, 1, 1, 1)
nif_amd.set_seed(3)
m = nif_amd.NIFMultiScale(cs, cp); model = m.build(); e = m._engine
x, y = nif_amd.data.synthetic_wave_batch(4096, seed=1)
loss, g = e.loss_and_grad(x, y)
d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size)
d_x.upload(x); d_y.upload(y)
e.loss_grad_dev(d_x.at(0), d_y.at(0), None, 4096, 4096)
comm.all_reduce_grad(e)          # builds the RCCL communicator (NIF_FORCE_RCCL=1); SUM over 1 rank: buffer unchanged
import ctypes as C
r_, w_ = C.c_int32(-1), C.c_int32(-1)
assert e.lib.nif_comm_info(e.ctx, C.byref(r_), C.byref(w_)) == 0 and (r_.value, w_.value) == (0, 1)
loss2, g2 = e.grad_read()
assert np.array_equal(g2, g) and loss2 == loss, (loss2, loss)
comm.barrier(e)
assert comm.all_reduce_ints(e, [3, 4, 5]) == [3, 4, 5] and comm.all_reduce_ints(e, [7], op="max") == [7]
assert comm.all_reduce_float(e, 1.25, op="max") == 1.25
# the all-reduced buffer is what Adam consumes
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
dist.install(None)
ha = a.fit(x, y, epochs=2, batch_size=1000, shuffle=False, verbose=0)
dist.install(comm)
comm.attach(mb._engine)           # second engine, second communicator (its own id file), still 1 rank wide
comm.world = 2                    # ... then take fit's N > 1 branches with one real rank
hb = b.fit(x, y, epochs=2, batch_size=1000, shuffle=False, verbose=0)
comm.world = 1
assert ha.history["loss"] == hb.history["loss"], (ha.history, hb.history)
assert np.array_equal(O.flatten(a.get_weights()), O.flatten(b.get_weights()))
assert comm._seq == 2
dist.shutdown()
print("OK")
'''


def test_rccl_allreduce_on_library_buffer_world1():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", NIF_FORCE_RCCL="1")
    r = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT}], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True, timeout=240)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-3000:]


def _model(seed=3, l2=None):
    import nif_amd
    from tests.test_gpu_parity import _cfg
    kind, cs, cp = _cfg("NIFMultiScale", 64, 2, 32, 2, 1, 1, 1, 1)
    if l2 is not None:
        cp = dict(cp, l2_reg=l2)
    nif_amd.set_seed(seed)
    m = nif_amd.NIFMultiScale(cs, cp)
    return m, m.build()


def test_zero_grad_step_still_applies_the_regulariser():
    """a rank that joins a step with a zero gradient (shard ran out of rows) must add the same
    weight-regulariser term as the ranks that computed one -- also right after an ordinary step."""
    import nif_amd
    from oracle import nif_oracle as O
    l2 = 1e-2
    m, model = _model(l2=l2)
    e = m._engine
    x, y = nif_amd.data.synthetic_wave_batch(2048, seed=1)
    adam = nif_amd.Adam(1e-3).as_struct()
    d_x, d_y = e.alloc(x.size), e.alloc(y.size)
    d_x.upload(x); d_y.upload(y)
    e.loss_grad_dev(d_x.at(0), d_y.at(0), None, 2048, 2048)
    e.adam_step_dev(adam)                                    # ordinary step: leaves no stale "already applied" state
    w0 = O.flatten(model.get_weights()).astype(np.float64)
    mom, var, step = e.get_opt_state()
    e.zero_grad()
    e.adam_step_dev(adam)
    w1 = O.flatten(model.get_weights()).astype(np.float64)
    n_pnet = sum(int(np.prod(s)) for nm, s in m._spec.param_shapes() if nm.startswith("pnet_"))
    g = np.zeros_like(w0); g[:n_pnet] = 2 * l2 * w0[:n_pnet]
    th, _, _ = O.adam_step(w0, g, mom.astype(np.float64), var.astype(np.float64), step + 1, lr=1e-3)
    assert np.abs(w1 - th).max() < 2e-6
    assert np.abs(w1[:n_pnet] - w0[:n_pnet]).max() > 1e-5     # the regulariser did move the ParameterNet


def test_single_process_group_and_sharding_train_step():
    """ncclCommInitAll over the contexts of ONE process (n = 1 here) and nif_train_step_multi = nif_train_step"""
    import nif_amd
    from nif_amd._lib import check, ptr
    from oracle import nif_oracle as O
    ma, a = _model(seed=7)
    mb, b = _model(seed=7)
    ea, eb = ma._engine, mb._engine
    x, y = nif_amd.data.synthetic_wave_batch(3000, seed=2)
    sw = np.random.default_rng(0).uniform(0.5, 1.5, 3000).astype(np.float32)
    adam = nif_amd.Adam(2e-3).as_struct()
    arr = (C.c_void_p * 1)(eb.ctx)
    check(eb.lib.nif_comm_init_all(arr, 1))
    r_, w_ = C.c_int32(-1), C.c_int32(-1)
    check(eb.lib.nif_comm_info(eb.ctx, C.byref(r_), C.byref(w_)))
    assert (r_.value, w_.value) == (0, 1)
    for _ in range(3):
        la = ea.train_step(x, y, sw, adam)
        lb = C.c_float()
        check(eb.lib.nif_train_step_multi(arr, 1, ptr(x), ptr(y), ptr(sw), 3000, C.byref(adam), C.byref(lb)))
        assert la == lb.value
    check(eb.lib.nif_allreduce_grad_multi(arr, 1))
    assert np.array_equal(O.flatten(a.get_weights()), O.flatten(b.get_weights()))
    check(eb.lib.nif_comm_destroy(eb.ctx))


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_the_drivers_launcher_single_rank():
    """bench.py's N>1 code path (RCCL communicator, barrier, max-over-ranks) launched the way the driver does"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps",
           "2", "--warmup", "1", "--points", "65536", "--no-cpu-baseline", "--given-w-points", "4096",
           "--force-dist"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _one_json_line(r.stdout)      # the contract: ONE line on stdout
    assert d["n_gpus"] == 1 and d["value"] > 0 and "roofline" in d and "RCCL" in d["config"]["collective"]
    # r4: the run checks its own collective (rank + 1 through nif_allreduce_grad's buffer and stream) and says so
    di = d["dist"]
    assert di["rccl_ranks_seen"] == 1 and di["world"] == 1 and di["comm_build_s"] > 0 and di["selftest"]
    assert 0 < di["ms_per_step_rank_min"] <= di["ms_per_step_rank_max"]
    ro = d["roofline"]
    for k in ("frac_hbm", "frac_bf16_pipe", "speedup_vs_f32_input_mfma_peak", "algorithmic_bytes_per_point", "design_bytes_per_point", "traffic_stale",
              "csrc_sha", "bound"):
        assert k in ro, k
    assert ro["algorithmic_bytes_per_point"] == 12.0


def test_comm_selftest_world1_and_pci_bus_id():
    """nif_comm_selftest with one real RCCL rank: the sum accounts for one rank, the buffer is left zeroed; the PCI bus id of the
    device (NUMA placement of a rank's process) has the sysfs form"""
    import ctypes as C
    import nif_amd
    from nif_amd import distributed as dist
    from nif_amd._lib import check
    m, _ = _model()
    e = m._engine
    comm = dist.RcclComm(0, 1, 0, key="selftest_%d" % os.getpid())
    os.environ["NIF_FORCE_RCCL"] = "1"
    try:
        assert comm.selftest(e) == 1
    finally:
        os.environ.pop("NIF_FORCE_RCCL", None)
    buf = C.create_string_buffer(64)
    check(e.lib.nif_device_pci_bus_id(0, buf, 64))
    bdf = buf.value.decode()
    assert len(bdf.split(":")) == 3 and "." in bdf, bdf
    check(e.lib.nif_comm_destroy(e.ctx))


def test_bench_plain_invocation():
    """`python bench.py` with no launcher around it"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--points",
           "65536", "--no-cpu-baseline", "--given-w-points", "4096"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["median_ms_per_step_host_synced"] > 0 and d["ms_per_step_fp32_mfma"] > 0


# ---- round 5 (VERDICT r4 Next #7): the 8-process START on the one GPU there is -------------------------------------------------
_REHEARSAL = r'''
import os, sys, time, json
t_start = float(os.environ["NIF_T0"])
sys.path.insert(0, %(root)r)
import bench                                   # (the benchmark's own model configuration)
import nif_amd
from nif_amd import distributed as dist
t_import = time.time()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ["LOCAL_RANK"] = "0"                 # eight ranks, ONE device: everything a rank does up to, but excluding, ncclCommInitRank
nif_amd.set_seed(1)
m = nif_amd.NIFMultiScale(bench.CFG_SHAPE, bench.CFG_PARAM)      # dlopen of libnif_hip.so (all kernels) + nif_create
model = m.build()
e = m._engine
e.reserve(1 << 20, 0)                          # the benchmark batch's workspaces
t_engine = time.time()
node = dist.pin_to_device_numa(0)
comm = dist.RcclComm(rank, world, 0)           # NIF_RDZV_KEY from the launcher, NIF_COMM_TIMEOUT as in bench.py's self_launch
raw, mine = comm._exchange_id(e.lib)           # the file rendezvous with a REAL nif_comm_unique_id from rank 0
t_rdzv = time.time()
# (the product removes rank 0's files once ncclCommInitRank has returned = every rank has read the id; here the ranks say so)
done = os.path.join(os.environ["NIF_RDZV_DIR"], "done_%%d")
open(done %% rank, "w").close()
if rank == 0:
    while not all(os.path.exists(done %% r) for r in range(world)):
        assert time.time() - t_rdzv < 120.0
        time.sleep(0.01)
    comm._remove(mine)
assert len(raw) == 128
print(json.dumps({"rank": rank, "import_s": t_import - t_start, "engine_s": t_engine - t_start, "rendezvous_s": t_rdzv - t_start,
                  "rdzv_wait_s": t_rdzv - t_engine, "numa": node, "id": raw.hex()}))
'''


def test_eight_process_start_rehearsal_on_one_gpu(tmp_path):
    """What a rank of `bench.py --gpus 8` does between process start and ncclCommInitRank -- import, dlopen of the whole library,
    nif_create, the benchmark's workspaces, NUMA pinning, the nonce handshake around rank 0's real RCCL unique id -- by EIGHT real
    processes at once on the one device a test box has (row (e) of SURVEY 8 is otherwise unmeasured on hardware).  Every rank must
    hold the same 128-byte id, and the slowest rank's time to rendezvous must sit well inside NIF_COMM_TIMEOUT (a third of it)."""
    import time
    timeout = 300.0
    script = tmp_path / "rehearsal.py"
    script.write_text(_REHEARSAL % {"root": ROOT})
    port = _free_port()
    procs = []
    t0 = time.time()
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NIF_T0=repr(t0),
                   NIF_RDZV_KEY="rehearsal_%d_%d" % (os.getpid(), port), NIF_RDZV_DIR=str(tmp_path), NIF_COMM_TIMEOUT=str(int(timeout)),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    recs = []
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, "rank %d: %s" % (r, err.decode()[-2000:])
        recs.append(json.loads(out.decode().strip().splitlines()[-1]))
    assert sorted(x["rank"] for x in recs) == list(range(8))
    assert len({x["id"] for x in recs}) == 1                       # one id, the one rank 0 drew, on every rank
    slowest = max(x["rendezvous_s"] for x in recs)
    assert slowest < timeout / 3.0, recs
    assert not list(tmp_path.glob("nif_rccl_*"))                    # rank 0 removed the id / open / hello files once every rank had said "done"
    sys.stderr.write("8-process start on one GPU: slowest import %.1f s, engine %.1f s, rendezvous %.1f s (timeout %.0f s)\n"
                     % (max(x["import_s"] for x in recs), max(x["engine_s"] for x in recs), slowest, timeout))
