"""N>1 path on CPU: two real processes (gloo) run the REAL `nif_amd.model.Model.fit` -- shard walking, the once-per-call
agreement of every step's global batch size (`distributed.plan_steps`), the zero-gradient step of the rank whose shard
is a batch shorter, one gradient all-reduce per step, replicated Adam -- with the oracle standing in for the per-shard
HIP compute (tests/doubles.py) and a gloo communicator with RcclComm's interface installed in nif_amd.distributed.
What the GPU path all-reduces over RCCL is exactly this buffer: [grad | loss], pre-scaled by 1/B_global."""
import os
import socket
import types

import numpy as np
import pytest

from oracle import nif_oracle as O
from tests.cfgs import ALL_SMALL


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_LOCAL = (40, 25)      # rows per rank: batches 16,16,8 on rank 0 and 16,9,- on rank 1 -> rank 1 joins step 3 with zeros
BS = 16
EPOCHS = 2
L2 = 1e-3


def _problem(name):
    kind, cs, cp = ALL_SMALL[name]
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = O.init_weights(spec, rng)
    n = sum(N_LOCAL)
    x = rng.uniform(-1, 1, size=(n, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(n, spec.so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)
    n_pnet = sum(int(np.prod(s)) for nm, s in spec.param_shapes() if nm.startswith("pnet_"))
    return kind, cs, cp, spec, ws, x, y, sw, (0.0, L2, 0, n_pnet)


def _worker(rank, world, port, name, outdir):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    import torch.distributed as td
    td.init_process_group("gloo")
    import nif_amd
    from nif_amd import distributed as dist
    from nif_amd.model import Model
    from nif_amd.spec import Spec
    from tests.doubles import GlooComm, OracleEngine
    comm = dist.install(GlooComm())
    assert dist.is_initialized() and dist.world_size() == world and dist.rank() == rank and dist.local_device() == rank
    kind, cs, cp, spec, ws, x, y, sw, reg = _problem(name)
    eng = OracleEngine(spec, ws, reg)
    owner = types.SimpleNamespace(_spec=Spec(kind, cs, cp), _engine=eng)
    model = Model(owner, "full")
    model.compile(nif_amd.Adam(1e-2), "mse")
    lo = sum(N_LOCAL[:rank]); hi = lo + N_LOCAL[rank]
    h = model.fit(x[lo:hi], y[lo:hi], sample_weight=sw[lo:hi], batch_size=BS, epochs=EPOCHS, shuffle=False, verbose=0)
    steps = [c for c in eng.calls if c[0] in ("loss_grad", "zero_grad")]
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), theta=eng.theta, loss=np.array(h.history["loss"]),
             steps=np.array([c[1] if c[0] == "loss_grad" else 0 for c in steps]),
             gsizes=np.array([c[2] if c[0] == "loss_grad" else -1 for c in steps]), nred=comm.n_grad_reduces)
    dist.shutdown()
    if td.is_initialized():
        td.destroy_process_group()


@pytest.mark.parametrize("name", ["ms_plain", "nif_swish"])
def test_two_rank_fit_equals_global_batch_training(name, tmp_path):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # both ranks took the same number of steps, one all-reduce each, and ended with IDENTICAL replicated weights
    assert int(r0["nred"]) == int(r1["nred"]) == 3 * EPOCHS
    assert np.array_equal(r0["theta"], r1["theta"])
    assert np.array_equal(r0["loss"], r1["loss"])
    # the agreed global sizes: 32, 25, 8; rank 1 contributed zeros to the third step
    assert list(r0["steps"]) == [16, 16, 8] * EPOCHS and list(r1["steps"]) == [16, 9, 0] * EPOCHS
    assert list(r0["gsizes"]) == [32, 25, 8] * EPOCHS and list(r1["gsizes"]) == [32, 25, -1] * EPOCHS
    # serial emulation: every step sees the union of the ranks' rows of that step as ONE batch (+ L2 on the pnet)
    kind, cs, cp, spec, ws, x, y, sw, reg = _problem(name)
    theta = O.flatten(ws); m = np.zeros_like(theta); v = np.zeros_like(theta); t = 0
    losses = []
    for _ in range(EPOCHS):
        tot, cnt = 0.0, 0.0
        for ib in range(3):
            rows = []
            for r in range(2):
                lo = sum(N_LOCAL[:r]) + ib * BS
                rows += list(range(lo, min(lo + BS, sum(N_LOCAL[:r + 1]))))
            rows = np.array(rows)
            loss, grads = O.loss_and_grad(spec, O.unflatten(spec, theta), x[rows].astype(np.float64), y[rows].astype(np.float64),
                                          sw[rows].astype(np.float64))
            g = O.flatten(grads)
            w = theta[reg[2]:reg[3]]
            g[reg[2]:reg[3]] += 2 * L2 * w
            loss += L2 * np.sum(w * w)
            t += 1
            f32 = lambda a: float(np.float32(a))     # nif_adam carries float32 hyper-parameters
            theta, m, v = O.adam_step(theta, g, m, v, t, lr=f32(1e-2), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
            tot += loss * len(rows); cnt += len(rows)
        losses.append(tot / cnt)
    assert np.allclose(r0["theta"], theta, rtol=1e-9, atol=1e-12)
    assert np.allclose(r0["loss"], losses, rtol=1e-9)


def _worker_empty(rank, world, port, outdir):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    import torch.distributed as td
    td.init_process_group("gloo")
    import nif_amd
    from nif_amd import distributed as dist
    from nif_amd.model import Model
    from nif_amd.spec import Spec
    from tests.doubles import GlooComm, OracleEngine
    comm = dist.install(GlooComm())
    kind, cs, cp, spec, ws, x, y, sw, reg = _problem("ms_plain")
    eng = OracleEngine(spec, ws, reg)
    model = Model(types.SimpleNamespace(_spec=Spec(kind, cs, cp), _engine=eng), "full")
    model.compile(nif_amd.Adam(1e-2), "mse")
    n = 40 if rank == 0 else 0                      # rank 1 has NO rows at all
    h = model.fit(x[:n], y[:n], sample_weight=sw[:n], batch_size=BS, epochs=1, shuffle=True, verbose=0)
    steps = [c[0] for c in eng.calls if c[0] in ("loss_grad", "zero_grad")]
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), theta=eng.theta, loss=np.array(h.history["loss"]), nred=comm.n_grad_reduces,
             steps=np.array(steps), reserve=np.array([c[1] for c in eng.calls if c[0] == "reserve"]))
    dist.shutdown()
    if td.is_initialized():
        td.destroy_process_group()


def test_two_rank_fit_with_an_empty_shard(tmp_path):
    """a rank without rows reserves a 1-row workspace, takes part in every step's all-reduce with a zero gradient
    (which still carries the weight-regulariser term) and ends with the same replicated weights"""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    mp.spawn(_worker_empty, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["nred"]) == int(r1["nred"]) == 3
    assert list(r1["steps"]) == ["zero_grad"] * 3 and list(r0["steps"]) == ["loss_grad"] * 3
    assert list(r1["reserve"]) == [1] and list(r0["reserve"]) == [16]
    assert np.array_equal(r0["theta"], r1["theta"]) and np.array_equal(r0["loss"], r1["loss"])


def test_plan_steps_single_process_and_shard_bounds():
    from nif_amd import distributed as dist
    assert dist.plan_steps(37, 16) == ([16, 16, 5], [16, 16, 5])
    assert dist.plan_steps(32, 16) == ([16, 16], [16, 16])
    for n in (1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            cuts = [dist.shard_bounds(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_defaults():
    from nif_amd import distributed as dist
    assert dist.get() is None and dist.world_size() == 1 and dist.rank() == 0 and dist.local_device() == 0


def test_rccl_id_rendezvous_through_files(tmp_path, monkeypatch):
    """the only host-side exchange of the RCCL path: the others publish a nonce, rank 0 answers with [id | nonces]; a stale
    answer of an earlier launch under the same key is never accepted; private directory, private files; bounded waits"""
    import stat
    import threading
    from nif_amd import _lib
    from nif_amd.distributed import RcclComm

    class FakeLib(object):
        def __init__(self, fill):
            self.fill = fill

        def nif_comm_unique_id(self, buf):
            buf.raw = bytes([self.fill]) * 128
            return 0

    d = str(tmp_path / "rdzv")
    a = RcclComm(0, 3, 0, key="k/ey 1", directory=d)
    b = RcclComm(1, 3, 1, key="k/ey 1", directory=d, timeout=20)
    c = RcclComm(2, 3, 2, key="k/ey 1", directory=d, timeout=20)
    assert stat.S_IMODE(os.stat(d).st_mode) == 0o700
    # a crashed earlier launch left an answer behind (other id, other nonces): nobody may take it
    with open(a._id_path(0), "wb") as f:
        f.write(bytes([9]) * (128 + 2 * RcclComm.NONCE))
    got = {}
    ths = [threading.Thread(target=lambda q=q, nm=nm: got.setdefault(nm, q._exchange_id(FakeLib(7)))) for q, nm in ((b, "b"), (c, "c"))]
    for th in ths:
        th.start()
    raw, mine = a._exchange_id(FakeLib(7))
    for th in ths:
        th.join()
    assert raw == bytes([7]) * 128 and got["b"][0] == raw and got["c"][0] == raw
    assert all(os.path.dirname(q) == d for q in mine) and len(mine) == 4      # the answer, the token, two hello files: rank 0 removes them
    assert stat.S_IMODE(os.stat(a._id_path(0)).st_mode) == 0o600
    assert a._id_path(1) != a._id_path(0)          # one file set per communicator
    # bounded waits on both sides
    lone = RcclComm(1, 2, 1, key="nobody", directory=d, timeout=0.05)
    with pytest.raises(_lib.NifError):
        lone._exchange_id(FakeLib(1))
    lone0 = RcclComm(0, 2, 0, key="nobody2", directory=d, timeout=0.05)
    with pytest.raises(_lib.NifError):
        lone0._exchange_id(FakeLib(1))
    assert not [f for f in os.listdir(d) if "nobody" in f]       # a rank that gives up removes what it wrote
    # a killed earlier launch left a HELLO behind under the same key (torchrun's default port = the same key for every launch) and
    # rank 1 of this launch starts late: rank 0 must not answer the stale nonce (r3: rank 1 then never saw its own and timed out)
    k0 = RcclComm(0, 2, 0, key="again", directory=d, timeout=20)
    k1 = RcclComm(1, 2, 1, key="again", directory=d, timeout=20)
    with open(k0._path("hello", 0, 1), "wb") as f:
        f.write(bytes([5]) * (2 * RcclComm.NONCE))
    with open(k0._path("open", 0), "wb") as f:
        f.write(bytes([6]) * RcclComm.NONCE)                     # ... and its token
    late = {}

    def late_rank():
        import time as _t
        _t.sleep(0.2)
        late["r"] = k1._exchange_id(FakeLib(3))
    th = threading.Thread(target=late_rank); th.start()
    raw0, mine0 = k0._exchange_id(FakeLib(3))
    th.join()
    assert raw0 == bytes([3]) * 128 and late["r"][0] == raw0
    # no key, no launcher environment, more than one rank: refuse instead of guessing
    monkeypatch.delenv("NIF_RDZV_KEY", raising=False); monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(_lib.NifError):
        RcclComm(0, 2, 0, directory=d)
    monkeypatch.setenv("NIF_COMM_TIMEOUT", "7")
    monkeypatch.setenv("MASTER_PORT", "1234")
    assert RcclComm(0, 2, 0, directory=d)._timeout == 7.0


def _run_bench(extra_env, args, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NIF_BENCH_ENGINE="tests.doubles:bench_double", PYTHONPATH=root)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, cwd=root, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, universal_newlines=True, timeout=timeout)


def test_bench_self_launch_two_ranks_prints_one_json_line():
    """`python bench.py --gpus 2` (how a user, and the driver's N > 1 legs via torchrun, start it): two real processes, the
    launcher's environment, ONE JSON line from rank 0 with the aggregate over both ranks -- on an engine double (CPU)."""
    import json
    pytest.importorskip("torch")
    r = _run_bench({}, ["--gpus", "2", "--steps", "2", "--warmup", "1", "--points", "48", "--no-cpu-baseline", "--no-extras"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 96 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 96 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]


def test_bench_self_launch_eight_ranks_self_check_fields():
    """world 8 -- the size of the driver's scaling run -- on the engine double: the all-reduce self-check (rank + 1 through the step's
    own collective: 36 on every rank) runs before the warm-up and the line carries what lets the driver see that N ranks took part"""
    import json
    pytest.importorskip("torch")
    r = _run_bench({}, ["--gpus", "8", "--steps", "2", "--warmup", "1", "--points", "32", "--no-cpu-baseline", "--no-extras"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 256
    di = d["dist"]
    assert di["rccl_ranks_seen"] == 8 and di["world"] == 8
    assert di["comm_build_s"] is not None and di["comm_build_s"] >= 0.0
    assert 0 < di["ms_per_step_rank_min"] <= di["ms_per_step_rank_max"] == pytest.approx(d["ms_per_step"])


def test_numa_cpulist_parsing(tmp_path):
    from nif_amd import distributed as dist
    (tmp_path / "node1").mkdir()
    (tmp_path / "node1" / "cpulist").write_text("32-35,160-161\n")
    assert dist.cpus_of_numa_node(1, sysfs=str(tmp_path)) == [32, 33, 34, 35, 160, 161]
    assert dist.cpus_of_numa_node(7, sysfs=str(tmp_path)) == []
    # no GPU / no library symbol / no sysfs entry: nothing is pinned, no exception

    class NoLib(object):
        def nif_device_pci_bus_id(self, dev, buf, n):
            return -1
    before = os.sched_getaffinity(0)
    assert dist.pin_to_device_numa(0, lib=NoLib()) is None
    assert os.sched_getaffinity(0) == before


def test_bench_self_launch_fails_fast_when_a_rank_dies_early():
    """rank 1 exits before the rendezvous: rank 0 (blocked in the process-group handshake) is taken down within seconds and the
    launcher returns rank 1's exit code, with the rank's stderr passed on under its prefix"""
    import time
    pytest.importorskip("torch")
    t0 = time.time()
    r = _run_bench({"NIF_BENCH_DOUBLE_FAIL_RANK": "1"},
                   ["--gpus", "2", "--steps", "2", "--warmup", "1", "--points", "48", "--no-cpu-baseline", "--no-extras"])
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 60
    assert "[rank 1] double: rank 1 fails on purpose" in r.stderr and "stopping the other ranks" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip()]


def test_bench_under_torchrun_two_ranks_on_the_double():
    """the DRIVER's command for its N > 1 legs -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- with two ranks on the engine double: bench.py takes RANK / WORLD_SIZE /
    MASTER_* from the launcher's environment, does not launch ranks of its own, and rank 0 alone prints the one JSON line"""
    import json
    import socket
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, NIF_BENCH_ENGINE="tests.doubles:bench_double", PYTHONPATH=root)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--points", "40",
           "--no-cpu-baseline", "--no-extras", "--ramp-steps", "2"]
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["config"]["global_batch"] == 80
    assert d["dist"]["rccl_ranks_seen"] == 2 and d["dist"]["world"] == 2
    assert abs(d["value"] - 80 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
