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


def STASH_BYTES_PER_POINT(s):
    """h and dL/da stash rows the fused kernel writes, h re-read in the adjoint"""
    return 4.0 * 32 * ((s.n_sx + 31) // 32) * (2 * (s.n_hidden_mats + 1) + s.n_hidden_mats)


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
        env.setdefault("NIF_COMM_TIMEOUT", "120")
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the A/B and per-kernel legs after the timed region")
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

    if use_dist:
        fence()   # the first collective builds RCCL's channels (~10 ms of idle GPU): pay that before the warm-up, not
                  # between the warm-up and the timed region, where the idle gap lets the clocks drop
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        dt = comm.all_reduce_float(e, dt, op="max")
    loss = e.last_loss()

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
               "roofline": None, "cpu_baseline": None}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    elif rank == 0:
        # ---- per-kernel durations, live, with HIP events on the library's stream -------------
        e.profile_enable(True)
        nprof = max(3, min(args.steps, 10))
        for _ in range(nprof):
            e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, Bg)
            e.adam_step_dev(adam)
        prof = e.profile_read(reset=True)
        e.profile_enable(False)
        kern_ms = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in prof.items()}
        s = m._spec
        n_w = s.si_dim * s.n_sx + s.n_hidden_mats * s.n_sx ** 2 + s.n_sx * s.so_dim
        flops_snet = 4.0 * (s.pi_hidden + 1) * n_w * B          # fwd + data-adjoint GEMMs of the fused kernel
        ach = flops_snet / (kern_ms["snet"] * 1e-3) / 1e12 if kern_ms["snet"] > 0 else 0.0
        traffic, traffic_gw, tnote = None, None, None
        try:  # HBM traffic of the same kernels from the committed PMC run (bench.py cannot run rocprofv3 on itself)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            traffic = (2.0 * tj["snet"]["FETCH_SIZE_KB"] + tj["snet"]["WRITE_SIZE_KB"]) * 1024.0 * B / tj["points"]
            traffic_gw = (2.0 * tj["k_given_w"]["FETCH_SIZE_KB"] + tj["k_given_w"]["WRITE_SIZE_KB"]) * 1024.0 \
                * args.given_w_points / tj["k_given_w"]["points"]
            tnote = tj["source"] + "; " + tj["calibration"]
        except Exception:
            pass
        # the fused kernel's n x n products run on the bf16 matrix cores as exact 3-way splits of the fp32 operands
        # (k_snet4.hip: 6 bf16 products per fp32 product forward, 3 in the adjoint): `achieved` is the ALGORITHMIC
        # fp32 work per second, priced against the fp32 MFMA peak as the survey prescribes; the bf16
        # flops actually executed and the stash traffic are given next to it
        nbl_even = (((s.n_sx + 15) // 16) % 2) == 0
        exec_bf16 = (6.0 + 3.0) * 2.0 * (s.pi_hidden + 1) * (s.n_hidden_mats * s.n_sx ** 2) * B if nbl_even else 0.0
        stash_bytes = STASH_BYTES_PER_POINT(s) * B
        sn_s = kern_ms["snet"] * 1e-3
        # What binds the fused kernel: neither matrix pipe nor VALU issue -- the HBM write path of the h / dL/da stash
        # rows (no-stash build: 0.71 ms, with: 1.03; the same store pattern alone streams at 6.6 TB/s).  Primary figure = the
        # larger of its HBM fraction (this design's own algorithmic bytes: the stash rows it must write and re-read)
        # and its bf16-matrix-pipe fraction; SURVEY 8d-ii's fp32-equivalent figure is kept as a secondary key.
        hbm_gbs = stash_bytes / sn_s / 1e9 if sn_s > 0 else 0.0
        bf16_frac = exec_bf16 / sn_s / 1e12 / 2500.0 if sn_s > 0 else 0.0
        hbm_bound = hbm_gbs / HBM_PEAK_GBS >= bf16_frac
        roofline = {"kernel": "k_snet4<4,true,SINE,0,tagged-sine> (ShapeNet fwd + MSE + data adjoint; fp32 products as bf16 splits on "
                              "v_mfma_f32_16x16x32_bf16)" if nbl_even else "k_snet3 (16x16x4 fp32 MFMA)",
                    "bound": "hbm" if hbm_bound else "mfma",
                    "achieved": hbm_gbs if hbm_bound else exec_bf16 / sn_s / 1e12,
                    "peak": HBM_PEAK_GBS if hbm_bound else 2500.0, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": hbm_gbs / HBM_PEAK_GBS if hbm_bound else bf16_frac,
                    "algorithmic_bytes_per_point": STASH_BYTES_PER_POINT(s),
                    "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC, scaled to this batch)",
                    "traffic_note": tnote, "avg_ms": kern_ms["snet"],
                    "avg_ms_note": "HIP events on the library's stream in a separate instrumented leg after the timed region "
                                   "(events between the kernels: the groups do not overlap there, so their sum exceeds ms_per_step)",
                    "frac_of_measured_hbm_6290": hbm_gbs / 6290.0,
                    "fp32_equiv_TFLOPs": ach, "flop_per_point": 4.0 * (s.pi_hidden + 1) * n_w,
                    "frac_of_fp32_mfma_peak_157": ach / FP32_PEAK_TFLOPS,
                    "executed_bf16_TFLOPs": exec_bf16 / sn_s / 1e12 if sn_s > 0 else 0.0,
                    "frac_of_bf16_mfma_peak_2500": bf16_frac}
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
            "config": {"workload": "configs[1]: 1D travelling wave, NIFMultiScale ShapeNet 4x64 SIREN (omega_0=30), "
                                   "ParameterNet 2x32 swish, latent_dim 1, P=%d, %d points/GPU" % (e.n_params, B),
                       "global_batch": Bg, "parallelism": "dp%d" % world, "final_loss": loss,
                       "collective": "RCCL ncclAllReduce(sum, f32, P+1) on the library stream, one per step" if use_dist else "none (1 GPU)"},
            "median_ms_per_step_host_synced": med_ms,
            "value_from_median": (Bg / (med_ms * 1e-3)) if med_ms else None,
            "grad_products": "forward: fp32-exact 6-product bf16 split; data adjoint: 3-product bf16 split; weight gradients: "
                             "2-way hi/lo bf16 split; fp32 accumulation everywhere",
            "ms_per_step_fp32_mfma": fp32_ms,
            "roofline": roofline,
            "roofline_given_w": roofline_given_w,
            "kernel_ms": kern_ms,
        }
        if not args.no_cpu_baseline and world == 1:     # rank 0 at N = 1 only (the other ranks would idle in the fence)
            out["cpu_baseline"] = cpu_baseline(sample_points=B)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        fence()
        if double is None:
            dist.shutdown()


if __name__ == "__main__":
    main()
