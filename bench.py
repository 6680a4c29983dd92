"""bench.py -- train-step throughput of the NIF hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launches its own N ranks, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (ranks from the launcher)

Workload (BASELINE.json configs[1]): 1D travelling wave, NIFMultiScale, ShapeNet 4x64 SIREN,
ParameterNet 2x32, latent_dim 1, fp32, batch = 2^20 (t;x)->u points PER GPU (weak scaling),
synthetic data from the closed form of the reference's bundled dataset, reference init.
A step = loss+gradient of the local shard (HIP), ONE sum all-reduce of [grad|loss] over ranks
(RCCL through the C-ABI: ncclAllReduce on the library's stream; no torch anywhere) when N > 1, Adam update.
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG_SHAPE = {"input_dim": 1, "output_dim": 1, "units": 64, "nlayers": 4, "use_resblock": False,
             "connectivity": "full", "omega_0": 30.0, "weight_init_factor": 0.01}
CFG_PARAM = {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "swish",
             "use_resblock": False, "omega_0": 30.0}
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
HBM_PEAK_GBS = 8000.0        # spec; 6290 measured-achievable


BF16_PEAK_TFLOPS = 2500.0    # dense v_mfma_*_bf16 peak
HBM_MEASURED_GBS = 6290.0    # measured-achievable (MI355X_MICROARCH.md)


def csrc_sha():
    """content hash of the HIP sources the library is built from: ties profiles/traffic.json to the code that produced the timing"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nif_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def roofline_block(dims, B, snet_ms, fused, traffic_json, sha):
    """The bench line's `roofline` object for the dominant kernel (pure function: tests/test_host_logic.py checks the stale path).
    dims = (pi, si, so, n, nh, r).  All three fractions are always present; `bound` names the larger of the two pipes the kernel
    really uses (HBM, bf16 matrix cores) and achieved / peak / unit / frac follow it.  SURVEY 8d-ii's figure -- algorithmic fp32
    flops against the 157.3 TF fp32-MFMA peak, a pipe the kernel does not run on -- is `speedup_vs_f32_input_mfma_peak` (a ratio
    that exceeds 1, not a fraction of a roof: VERDICT r5).
    `traffic` = HBM bytes per launch from the committed PMC passes, dropped (traffic_stale) when profiles/traffic.json was measured
    on other sources than the ones this run was built from."""
    pi, si, so, n, nh, r = dims
    n_w = si * n + nh * n * n + n * so
    sn_s = snet_ms * 1e-3
    alg_bytes = 4.0 * (pi + si + so)                            # what a training step must read per point
    nblk = (n + 31) // 32
    if fused:     # k_snet6: forward + data adjoint + weight gradients; h_1 .. h_{nh-1} through the private ring (write + read; h_0 is
                  # recomputed in the adjoint), latent in, dL/dz out
        alg_flop = 6.0 * (r + 1) * n_w
        # r5: three 16-bit products per fp32 product in all three sweeps (forward / data adjoint: half (hi, lo) pairs on
        # v_mfma_f32_16x16x32_f16; weight gradients: bf16 (hi, lo) pairs on v_mfma_f32_32x32x16_bf16) -- r4: 6 + 3 + 3
        exec_bf16 = (3.0 + 3.0 + 3.0) * 2.0 * (r + 1) * nh * n * n
        design_bytes = 4.0 * 32 * nblk * 2 * (nh - 1) + 4.0 * (si + so + 1) + 8.0 * r
        kernel = ("k_snet6<4> (ShapeNet forward + MSE + data adjoint + every ShapeNet weight gradient; fp32 products as 3-product half / bf16 "
                  "(hi, lo) pairs on v_mfma_f32_16x16x32_f16 / 32x32x16_bf16, 8 producer + 8 consumer waves per workgroup)")
    else:         # k_snet4: forward + data adjoint; h and dL/da stash rows written, h re-read
        alg_flop = 4.0 * (r + 1) * n_w
        # r5: a SIREN net on k_snet4 runs PR = 3 (three half products forward, three in the data adjoint); the six + three bf16 products
        # are class NIF's (this benchmark's net is a SIREN: ADVICE r5)
        exec_bf16 = (3.0 + 3.0) * 2.0 * (r + 1) * nh * n * n if (((n + 15) // 16) % 2) == 0 else 0.0
        design_bytes = 4.0 * 32 * nblk * (2 * (nh + 1) + nh)
        kernel = "k_snet4<4,true,SINE,0,tagged-sine> (ShapeNet forward + MSE + data adjoint; k_gw_* reduce the stash rows)"
    traffic, stale, tnote = None, None, None
    if traffic_json is not None:
        tj = traffic_json
        want = "snet6" if fused else "snet"
        if tj.get("csrc_sha") == sha and want in tj:
            traffic = (2.0 * tj[want]["FETCH_SIZE_KB"] + tj[want]["WRITE_SIZE_KB"]) * 1024.0 * B / tj["points"]
            tnote = tj.get("source", "") + "; " + tj.get("calibration", "")
            stale = False
        else:
            stale = True
            tnote = "profiles/traffic.json was measured on csrc %s (head %s), this run is built from csrc %s: traffic dropped" % (
                tj.get("csrc_sha"), tj.get("head"), sha)
    hbm_bytes = traffic if traffic is not None else design_bytes * B
    hbm_gbs = hbm_bytes / sn_s / 1e9 if sn_s > 0 else 0.0
    bf16_tf = exec_bf16 * B / sn_s / 1e12 if sn_s > 0 else 0.0
    f32_tf = alg_flop * B / sn_s / 1e12 if sn_s > 0 else 0.0
    frac_hbm, frac_bf16, frac_f32 = hbm_gbs / HBM_PEAK_GBS, bf16_tf / BF16_PEAK_TFLOPS, f32_tf / FP32_PEAK_TFLOPS
    hbm_bound = frac_hbm >= frac_bf16
    return {"kernel": kernel, "bound": "hbm" if hbm_bound else "mfma",
            "achieved": hbm_gbs if hbm_bound else bf16_tf, "peak": HBM_PEAK_GBS if hbm_bound else BF16_PEAK_TFLOPS,
            "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": frac_hbm if hbm_bound else frac_bf16,
            "frac_hbm": frac_hbm, "frac_bf16_pipe": frac_bf16, "speedup_vs_f32_input_mfma_peak": frac_f32,
            "frac_bf16_pipe_algorithmic": f32_tf / BF16_PEAK_TFLOPS,
            "speedup_vs_f32_input_mfma_peak_note": "NOT a roofline fraction (r5 called it frac_fp32_equiv and it read 1.25): algorithmic fp32 "
                                    "flops (SURVEY 8d-ii) over the 157.3 TF f32-input MFMA peak, i.e. how much faster than the best possible "
                                    "f32-input-MFMA kernel this one runs.  The kernel's own pipe is the 16-bit one: each fp32 product is THREE "
                                    "16-bit products (frac = frac_bf16_pipe counts those executed flops against 2.5 PF = algorithmic flops "
                                    "against the 833 TF a 3-product emulation can reach; frac_bf16_pipe_algorithmic the algorithmic ones)",
            "frac_hbm_of_measured_6290": hbm_gbs / HBM_MEASURED_GBS,
            "hbm_GBs": hbm_gbs, "hbm_bytes_source": "pmc" if traffic is not None else "design_bytes_per_point",
            "executed_bf16_TFLOPs": bf16_tf, "fp32_equiv_TFLOPs": f32_tf,
            "algorithmic_bytes_per_point": alg_bytes, "design_bytes_per_point": design_bytes,
            "algorithmic_flop_per_point": alg_flop, "executed_bf16_flop_per_point": exec_bf16,
            "traffic": traffic, "traffic_stale": stale,
            "traffic_ratio": (traffic / (alg_bytes * B)) if traffic is not None else None,
            "design_traffic_ratio": design_bytes / alg_bytes,
            "traffic_unit": "HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, scaled to this batch)",
            "traffic_note": tnote, "csrc_sha": sha, "avg_ms": snet_ms,
            "avg_ms_note": "HIP events on the library's stream in a separate instrumented leg after the timed region "
                           "(events between the kernels: the groups do not overlap there, so their sum exceeds ms_per_step)"}


def cpu_baseline(sample_points=1 << 20, micro=4096, probe_points=65536):
    """The reference FORMULATION restated in C/OpenMP (oracle/nif_ref_cpu.c: materialised pnet_output
    [b, po], per-sample einsum chain, reverse sweep with a materialised [b, po] gradient, Adam) timed on this
    box's host cores on a bounded sample of the same workload: whole train steps of the benchmark's `sample_points`
    points (the same batch the GPU steps over) executed as micro-batches of `micro` (fp32), ~10-25 s in total.  It stands in for the reference's tf.distribute CPU
    path, which cannot run here (TensorFlow 2.11.1 is not installed and the reference's Python cannot travel)."""
    from oracle import nif_oracle as O
    from oracle import ref_cpu as R
    import ctypes as C
    spec = O.Spec("NIFMultiScale", CFG_SHAPE, CFG_PARAM)
    rng = np.random.default_rng(1)
    th = O.flatten(O.init_weights(spec, rng, dtype=np.float32)).astype(np.float32)
    x_all, y_all = O.synthetic_wave_batch(sample_points, seed=0)
    lib = R.load()
    cfg = R.make_cfg(spec)
    ncpu = min(os.cpu_count() or 1, lib.nifref_max_threads())

    def run(cores, budget_s, max_rep, npts):
        th_ = th.copy(); m = np.zeros_like(th_); v = np.zeros_like(th_)
        x, y = x_all[:npts], y_all[:npts]

        def step(t):
            _, g = R.loss_and_grad(lib, cfg, th_, x, y, None, micro=micro, nthreads=cores)
            lib.nifref_adam(th_.ctypes.data, g.ctypes.data, m.ctypes.data, v.ctypes.data, th_.size, t, 1e-3, 0.9, 0.999, 1e-7)

        if npts <= probe_points:
            step(1)  # warm-up
        t0 = time.perf_counter()
        nrep, t = 0, 2
        while True:
            step(t); t += 1; nrep += 1
            if time.perf_counter() - t0 > budget_s or nrep >= max_rep:
                break
        return nrep, time.perf_counter() - t0

    # the formulation is memory-bound ([b,po] tensors): more threads than ~a quarter socket LOSE throughput on a
    # 2-socket host (measured: 16 thr 3.3e5, 32 thr 1.8e5, 64 thr 0.95e5 points/s on 2x EPYC 9575F), so probe a
    # few counts briefly and time the best one
    best, best_rate = 1, 0.0
    for c in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        nrep, dt = run(c, 1.0, 3, probe_points)
        if nrep * probe_points / dt > best_rate:
            best, best_rate = c, nrep * probe_points / dt
    cores = best
    nrep, dt = run(cores, 10.0, 6, sample_points)
    return {"value": sample_points * nrep / dt, "unit": "points/s", "cores": int(cores), "kind": "port",
            "sample": "%d whole train steps of %d points (the benchmark batch; micro-batches of %d) of the benchmark model; C/OpenMP fp32 "
                      "restatement of the reference formulation (materialised [b,po] + per-sample einsum), "
                      "OMP threads = %d; not TensorFlow" % (nrep, sample_points, micro, cores)}


def weak_scaling_reference(sha):
    """The committed 1-GPU bench line of THIS build (profiles/r*_bench.json whose roofline.csrc_sha is the running sources' hash): an
    N-GPU line then states its own weak-scaling efficiency value / (N x reference).  None when no committed line matches the build."""
    import glob
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json"))):
        try:
            d = json.loads(open(fn).read().strip().splitlines()[-1])
            if d.get("n_gpus") == 1 and (d.get("roofline") or {}).get("csrc_sha") == sha:
                best = (float(d["value"]), os.path.relpath(fn, ROOT))
        except (OSError, ValueError, KeyError, IndexError):
            continue
    return best


def gradient_error(m, model, x, y):
    """(flat rel-L2, worst per-tensor rel-L2, note) of the engine's gradient on (x, y) against the NumPy fp64 oracle at the engine's
    CURRENT weights.  The oracle is the checker here (extras leg, after the timed region), never the thing measured."""
    from oracle import nif_oracle as O
    ws = model.get_weights()
    spec = O.Spec("NIFMultiScale", CFG_SHAPE, CFG_PARAM)
    ws64 = [w.astype(np.float64) for w in ws]
    _, g_ref = O.loss_and_grad(spec, ws64, x.astype(np.float64), y.astype(np.float64))
    _, g = m._engine.loss_and_grad(x, y)
    g = np.asarray(g, dtype=np.float64)
    flat_ref = O.flatten(g_ref)
    flat = float(np.linalg.norm(g - flat_ref) / np.linalg.norm(flat_ref))
    worst, off = 0.0, 0
    for t in g_ref:
        nrm = float(np.linalg.norm(t))
        if nrm > 0:
            worst = max(worst, float(np.linalg.norm(g[off:off + t.size] - t.ravel()) / nrm))
        off += t.size
    return flat, worst, "%d points of the benchmark batch, the engine's weights after the timed steps, %d tensors; fp64 NumPy oracle" % (len(x), len(g_ref))


def self_launch(args):
    """`python bench.py --gpus N` outside any launcher: start N ranks of this script (one process per GPU), hand them
    the torchrun-style environment, watch them.  Rank 0 inherits stdout and prints the JSON line; every rank's stderr is
    passed on line by line with a `[rank r]` prefix.  ALL ranks are polled: the first one that exits non-zero (or the
    NIF_BENCH_TIMEOUT watchdog) takes the others down within seconds and its code becomes ours -- a rank that dies before
    ncclCommInitRank must not leave the rest blocked in the rendezvous until somebody's outer timeout."""
    import socket
    import subprocess
    import threading
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []

    def pump(r, pipe):
        for line in iter(pipe.readline, b""):
            sys.stderr.buffer.write(b"[rank %d] " % r + line)
            sys.stderr.buffer.flush()
        pipe.close()

    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NIF_RDZV_KEY="bench_%d_%d" % (os.getpid(), port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("NIF_COMM_TIMEOUT", "300")     # counted from the rank's first rendezvous call (after the library and the engine exist), not from process start
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                             stdout=None if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE)
        threading.Thread(target=pump, args=(r, p.stderr), daemon=True).start()
        procs.append(p)
    deadline = time.time() + float(os.environ.get("NIF_BENCH_TIMEOUT", "1500"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc = abs(bad[0][1]) or 1
            sys.stderr.write("[bench] rank %d exited with code %d: stopping the other ranks\n" % bad[0])
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            rc = 124
            sys.stderr.write("[bench] NIF_BENCH_TIMEOUT expired: stopping all ranks\n")
            break
        time.sleep(0.05)
    if rc:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    sys.exit(rc)


def load_double():
    """NIF_BENCH_ENGINE=module:function -- CPU tests of the launcher / JSON contract put an engine + communicator double here
    (tests/doubles.py); the product path below is the HIP engine and has no fallback."""
    spec = os.environ.get("NIF_BENCH_ENGINE")
    if not spec:
        return None
    import importlib
    mod, fn = spec.split(":")
    return getattr(importlib.import_module(mod), fn)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=1 << 20, help="points per GPU")
    ap.add_argument("--ramp-steps", type=int, default=40,
                    help="untimed steps BEFORE the warm-up: the SMU needs ~30 ms of load to raise the shader clock from the idle "
                         "state (first 25 steps after a cold start: 1.52 -> 1.35 ms, tools/exp/step_times.py); reported as clock_ramp_steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the A/B and per-kernel legs after the timed region")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block: the other BASELINE.json configs (configs[0], [2], [3], [4] at their per-GPU shard sizes, "
                         "3 warm-up + 8 timed steps each) run AFTER the headline region on one GPU")
    ap.add_argument("--config-steps", type=int, default=8)
    ap.add_argument("--given-w-points", type=int, default=1 << 17)
    ap.add_argument("--force-dist", action="store_true", help="build the RCCL communicator and all-reduce even at world size 1")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    # The contract is ONE line on stdout.  RCCL prints a version banner on the C-level stdout of rank 0 (buffered, so
    # it would even land AFTER the JSON line at exit): keep the real stdout aside for the JSON and point fd 1 at
    # stderr for everything else in the process (libraries, child threads).
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import nif_amd
    from nif_amd import distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or args.force_dist
    rank = 0
    comm = None
    double = load_double()
    B = args.points
    if double is not None:
        e, comm, DeviceArray = double(B)
        m = None
        rank = comm.rank if comm is not None else 0
        use_dist = comm is not None
    else:
        from nif_amd.engine import DeviceArray
        if use_dist:
            if args.force_dist:
                os.environ["NIF_FORCE_RCCL"] = "1"
            rank, world = dist.init()
            comm = dist.get()
    assert world == args.gpus, "launch with one rank per GPU: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    numa_node = dist.pin_to_device_numa() if (double is None and use_dist) else None     # this rank's cores = its GPU's socket

    if double is None:
        nif_amd.set_seed(1)  # identical initial weights on every rank (mirrored variables)
        m = nif_amd.NIFMultiScale(CFG_SHAPE, CFG_PARAM)
        model = m.build()
        e = m._engine
    x, y = nif_amd.data.synthetic_wave_batch(B, seed=100 + rank)
    d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size)
    d_x.upload(x); d_y.upload(y)
    adam = nif_amd.Adam(1e-3).as_struct()
    Bg = B * world
    e.reserve(B, 0)      # every workspace sized now: no hipMalloc inside a step

    def step():
        e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, Bg)
        if use_dist:
            comm.all_reduce_grad(e)       # ONE ncclAllReduce(sum, f32, P+1) on the library's stream
        e.adam_step_dev(adam)

    def fence():
        if use_dist:
            comm.barrier(e)               # all ranks here + this rank's stream drained
        else:
            e.sync()

    comm_build_s, ranks_seen = None, 1
    if use_dist:
        t_c = time.perf_counter()
        comm.attach(e)    # id rendezvous + ncclCommInitRank (RCCL loads its code object here: seconds per process)
        fence()   # the first collective builds RCCL's channels (~10 ms of idle GPU): pay that before the warm-up, not
                  # between the warm-up and the timed region, where the idle gap lets the clocks drop
        comm_build_s = time.perf_counter() - t_c
        # the first multi-GPU run is also the first test of the collective: rank + 1 through the step's own all-reduce (same buffer,
        # count, stream) must sum to N (N + 1) / 2 on every rank, or the run stops here
        ranks_seen = comm.selftest(e)
        assert ranks_seen == world, "all-reduce self-check accounts for %d ranks, WORLD_SIZE is %d" % (ranks_seen, world)
    # the contract's measurement from a COLD start first (W warm-up steps, K timed steps right after the set-up): reported as
    # cold_start_ms_per_step next to the steady-state figure, so that the effect of the clock ramp is in the line itself
    cold_ms = None
    if args.ramp_steps > 0:
        for _ in range(args.warmup):
            step()
        fence()
        tc = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        cold_ms = (time.perf_counter() - tc) / args.steps * 1e3
        if use_dist:
            cold_ms = comm.all_reduce_float(e, cold_ms, op="max")
    for _ in range(max(args.ramp_steps - (args.warmup + args.steps if cold_ms is not None else 0), 0)):      # clock ramp (same count on every rank)
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dt_min = dt_max = dt
    if use_dist:
        dt_min = comm.all_reduce_float(e, dt, op="min")
        dt_max = dt = comm.all_reduce_float(e, dt, op="max")
        comm_build_s = comm.all_reduce_float(e, comm_build_s, op="max")
    loss = e.last_loss()
    dist_info = {"rccl_ranks_seen": ranks_seen, "world": world, "comm_build_s": comm_build_s,
                 "ms_per_step_rank_min": dt_min / args.steps * 1e3, "ms_per_step_rank_max": dt_max / args.steps * 1e3,
                 "numa_node_rank0": numa_node,
                 "selftest": "rank + 1 through nif_allreduce_grad's buffer / stream before the warm-up: N (N + 1) / 2 on every rank" if use_dist else None}

    # SURVEY 8d's statistic next to the contract's: median of host-synchronised single steps (all ranks in lock step)
    per = []
    if not args.no_extras:
        for _ in range(max(5, min(args.steps, 20))):
            fence()
            t1 = time.perf_counter()
            step()
            e.sync()
            per.append(time.perf_counter() - t1)
    med_ms = float(np.median(per)) * 1e3 if per else None
    if use_dist and med_ms is not None:
        med_ms = comm.all_reduce_float(e, med_ms, op="max")

    out = None
    if rank == 0 and double is not None:
        out = {"metric": "train-step points/sec (1D-wave, batch=1M per GPU)", "value": Bg * args.steps / dt, "unit": "points/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "ENGINE DOUBLE (%s): launcher / contract test, not a measurement" % os.environ["NIF_BENCH_ENGINE"],
                          "global_batch": Bg, "parallelism": "dp%d" % world, "final_loss": loss},
               "roofline": None, "cpu_baseline": None, "dist": dist_info}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    elif rank == 0:
        # ---- per-kernel durations, live, with HIP events on the library's stream -------------
        e.profile_enable(True)
        nprof = max(3, min(args.steps, 10))
        for _ in range(3):      # (the leg's own warm-up: the event pool is created on first use)
            e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, Bg)
            e.adam_step_dev(adam)
        e.profile_read(reset=True)
        for _ in range(nprof):
            e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, Bg)
            e.adam_step_dev(adam)
        prof = e.profile_read(reset=True)
        e.profile_enable(False)
        kern_ms = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in prof.items()}
        s = m._spec
        tj, traffic_gw = None, None
        sha = csrc_sha()
        try:  # HBM traffic of the same kernels from the committed PMC run (bench.py cannot run rocprofv3 on itself)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("csrc_sha") == sha:
                traffic_gw = (2.0 * tj["k_given_w"]["FETCH_SIZE_KB"] + tj["k_given_w"]["WRITE_SIZE_KB"]) * 1024.0 \
                    * args.given_w_points / tj["k_given_w"]["points"]
        except Exception:
            tj = None
        # the weight-gradient group is empty when the fused-gradient kernel (k_snet6) took the step
        fused = os.environ.get("NIF_FUSE_GW", "1") != "0" and kern_ms.get("gw", 0.0) < 0.05 * kern_ms["snet"]
        roofline = roofline_block((s.pi_dim, s.si_dim, s.so_dim, s.n_sx, s.n_hidden_mats, s.pi_hidden), B, kern_ms["snet"], fused, tj, sha)
        # ---- A/B: the same step with every ShapeNet product on the f32-input MFMAs (no bf16 splits) -----------
        fp32_ms = None
        if not args.no_extras:
            e.set_option("fp32_mfma", 1)
            for _ in range(2):
                e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, Bg); e.adam_step_dev(adam)
            e.sync()
            t1 = time.perf_counter()
            for _ in range(5):
                e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, Bg); e.adam_step_dev(adam)
            e.sync()
            fp32_ms = (time.perf_counter() - t1) / 5 * 1e3
            e.set_option("fp32_mfma", 0)
        # ---- the measured gradient error of the DEFAULT path: a 4 096-point sub-batch of the benchmark batch against the fp64 oracle
        # (checker only, outside the timed region) -- flat rel-L2 and the worst tensor ----------------------------------------------
        grad_err = None
        if not args.no_extras:
            try:
                grad_err = gradient_error(m, model, x[:4096], y[:4096])
            except Exception as ex:      # (the oracle is test infrastructure: a box without it still benches)
                grad_err = (None, None, "not run: %r" % (ex,))
        # ---- the HBM-bound kernel north_star names: model_x_to_u_given_w ------------------------
        Bw = args.given_w_points
        rng = np.random.default_rng(7)
        d_lr = DeviceArray(e, Bw * s.pi_hidden)
        d_lr.upload(rng.standard_normal(Bw * s.pi_hidden).astype(np.float32))
        d_w = DeviceArray(e, Bw * s.po_dim)
        d_xs = DeviceArray(e, Bw * s.si_dim)
        d_xs.upload(rng.uniform(-1, 1, Bw * s.si_dim).astype(np.float32))
        d_u = DeviceArray(e, Bw * s.so_dim)
        from nif_amd._lib import check
        check(e.lib.nif_latent_to_w_dev(e.ctx, d_lr.at(0), Bw, d_w.at(0)))
        for _ in range(2):
            check(e.lib.nif_shapenet_given_w_dev(e.ctx, d_xs.at(0), d_w.at(0), Bw, d_u.at(0)))
        e.sync()
        e.profile_enable(True)
        for _ in range(5):
            check(e.lib.nif_shapenet_given_w_dev(e.ctx, d_xs.at(0), d_w.at(0), Bw, d_u.at(0)))
            check(e.lib.nif_latent_to_w_dev(e.ctx, d_lr.at(0), Bw, d_w.at(0)))
        prof2 = e.profile_read(reset=True)
        e.profile_enable(False)
        gw_ms = prof2["given_w"][0] / max(prof2["given_w"][1], 1)
        l2w_ms = prof2["latent_to_w"][0] / max(prof2["latent_to_w"][1], 1)
        bytes_gw = 4.0 * (s.si_dim + s.po_dim + s.so_dim) * Bw
        gbs = bytes_gw / (gw_ms * 1e-3) / 1e9 if gw_ms > 0 else 0.0
        roofline_given_w = {"kernel": "k_given_w<64> (model_x_to_u_given_w, per-sample batched matvec)", "bound": "hbm",
                            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "frac_of_measured_peak_6290": gbs / 6290.0, "traffic": traffic_gw, "algorithmic_bytes": bytes_gw,
                            "avg_ms": gw_ms,
                            "points": Bw, "bytes_per_point": 4.0 * (s.si_dim + s.po_dim + s.so_dim),
                            "latent_to_w_GBs": 4.0 * s.po_dim * Bw / (l2w_ms * 1e-3) / 1e9 if l2w_ms > 0 else 0.0}
        out = {
            "metric": "train-step points/sec (1D-wave, batch=1M per GPU)",
            "value": Bg * args.steps / dt, "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "clock_ramp_steps": max(args.ramp_steps, args.warmup + args.steps) if args.ramp_steps > 0 else 0,
            "cold_start_ms_per_step": cold_ms,
            "value_cold_start": (Bg / (cold_ms * 1e-3)) if cold_ms else None,
            "clock_ramp_note": "untimed steps in front of the W warm-up steps: MI355X raises the shader clock over ~30 ms of load; a "
                               "timed region that starts 7 ms after idle measures the ramp instead of the step.  The first W + K of them "
                               "ARE the contract's measurement from a cold start: cold_start_ms_per_step",
            "config": {"workload": "configs[1]: 1D travelling wave, NIFMultiScale ShapeNet 4x64 SIREN (omega_0=30), "
                                   "ParameterNet 2x32 swish, latent_dim 1, P=%d, %d points/GPU" % (e.n_params, B),
                       "arithmetic": "fp32 results from 16-bit split products: every fp32 product is three f16 / bf16 MFMA products of (hi, lo) "
                                     "pairs with fp32 accumulation (dtype f32 names the variables, inputs, accumulators and results)",
                       "global_batch": Bg, "parallelism": "dp%d" % world, "final_loss": loss,
                       "collective": "RCCL ncclAllReduce(sum, f32, P+1) on the library stream, one per step" if use_dist else "none (1 GPU)"},
            "median_ms_per_step_host_synced": med_ms,
            "value_from_median": (Bg / (med_ms * 1e-3)) if med_ms else None,
            "grad_products": "forward and data adjoint: fp32 products as 3-product HALF (hi, lo) pairs (11 + 11 significand bits per operand, "
                             "power-of-two scales; as accurate as the f32-input MFMA: tools/exp/f16_split_mfma.hip); weight gradients: "
                             "2-way hi/lo bf16 split (3 products, 16 significand bits per operand); fp32 accumulation everywhere",
            "grad_rel_l2_vs_oracle": grad_err[0] if grad_err else None,
            "grad_max_tensor_rel_vs_oracle": grad_err[1] if grad_err else None,
            "grad_check": grad_err[2] if grad_err else None,
            "fused_weight_gradients": fused,
            "ms_per_step_fp32_mfma": fp32_ms,
            "roofline": roofline,
            "roofline_given_w": roofline_given_w,
            "dist": dist_info,
            "kernel_ms": kern_ms,
        }
        if world > 1:      # the N-GPU line states its own weak-scaling efficiency against the committed 1-GPU line of the same sources
            ref = weak_scaling_reference(sha)
            out["weak_scaling_ref_pps_1gpu"] = ref[0] if ref else None
            out["weak_scaling_ref_source"] = ref[1] if ref else "no profiles/r*_bench.json of csrc %s" % sha
            out["weak_scaling_efficiency_vs_ref"] = (out["value"] / (world * ref[0])) if ref else None
        if world == 1 and not args.no_extras and not args.no_configs:
            # every other BASELINE config, driver-run: after the headline's timed region and its legs, on fresh
            # engines of their own (tools/bench_configs.py -- the function profiles/rNN_configs.json comes from)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_configs as BC
                out["configs"] = BC.baseline_configs_block(steps=args.config_steps, warmup=3)
            except Exception as ex:      # (a config that fails must not take the headline line with it: say so in the line)
                out["configs"] = {"error": repr(ex)[:300]}
        if not args.no_cpu_baseline and world == 1:     # rank 0 at N = 1 only (the other ranks would idle in the fence)
            out["cpu_baseline"] = cpu_baseline(sample_points=B)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        fence()
        if double is None:
            dist.shutdown()
        elif comm is not None:
            comm.shutdown()


if __name__ == "__main__":
    main()
