"""Per-GPU train-step timings of the BASELINE.json configs other than the bench line (configs[2..4]) plus
the bench config itself, with the per-kernel-group HIP-event breakdown.  1 GPU, per-GPU shard sizes of the
8-GPU configs.  Not part of the driver contract -- numbers quoted in DESIGN.md come from here.

    python tools/bench_configs.py [--steps 10] [--only cfg5_sobolev]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ms(n, L, nst, lst, r, si, so, pi, res=False, p_act="swish", conn="full"):
    cs = {"input_dim": si, "output_dim": so, "units": n, "nlayers": L, "use_resblock": res, "connectivity": conn,
          "omega_0": 30.0, "weight_init_factor": 0.01}
    cp = {"input_dim": pi, "latent_dim": r, "units": nst, "nlayers": lst, "activation": p_act, "use_resblock": False,
          "omega_0": 30.0}
    return cs, cp


WORK = {
    # name: (class, cfgs, points per GPU, sobolev x_index or None)
    "cfg2_wave_4x64": ("NIFMultiScale", ms(64, 4, 32, 2, 1, 1, 1, 1), 1 << 20, None),
    "cfg3_ms_6x128_2d": ("NIFMultiScale", ms(128, 6, 64, 2, 1, 2, 1, 1), 1 << 19, None),
    "cfg3_ms_res_3x128_2d": ("NIFMultiScale", ms(128, 3, 64, 2, 1, 2, 1, 1, res=True), 1 << 19, None),
    "cfg4_linear_nif_3d": ("NIFMultiScaleLastLayerParameterized", ms(128, 2, 32, 2, 10, 3, 3, 1, conn="last_layer"), 1 << 21, None),
    "cfg5_sobolev_2d_4x64": ("NIFMultiScale", ms(64, 4, 32, 2, 1, 2, 1, 1), 1 << 20, [1, 2]),
    "sobolev_6x128_2d": ("NIFMultiScale", ms(128, 6, 64, 2, 1, 2, 1, 1), 1 << 18, [1, 2]),      # (not a BASELINE config: the wide Sobolev kernels)
    "sobolev_res_3x128_2d": ("NIFMultiScale", ms(128, 3, 64, 2, 1, 2, 1, 1, res=True), 1 << 18, [1, 2]),
    "sobolev_res_2x64_2d": ("NIFMultiScale", ms(64, 2, 32, 2, 1, 2, 1, 1, res=True), 1 << 20, [1, 2]),
    "sobolev_nif_2x64_swish": ("NIF", ({"input_dim": 2, "output_dim": 1, "units": 64, "nlayers": 2, "activation": "swish"},
                                      {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}), 1 << 20, [1, 2]),
    "sobolev_3x96_2d": ("NIFMultiScale", ms(96, 3, 32, 2, 2, 2, 1, 1), 1 << 18, [1, 2]),
    "cfg5_sobolev_2d_dx_only": ("NIFMultiScale", ms(64, 4, 32, 2, 1, 2, 1, 1), 1 << 20, [1]),
    # configs[4] names bf16: the mixed_bfloat16 policy of the build (single bf16 product per n x n operand pair)
    "cfg5_sobolev_2d_4x64_bf16": ("NIFMultiScale", ms(64, 4, 32, 2, 1, 2, 1, 1), 1 << 20, [1, 2], "mixed_bfloat16"),
    "cfg2_wave_4x64_bf16": ("NIFMultiScale", ms(64, 4, 32, 2, 1, 1, 1, 1), 1 << 20, None, "mixed_bfloat16"),
    "cfg3_ms_6x128_2d_bf16": ("NIFMultiScale", ms(128, 6, 64, 2, 1, 2, 1, 1), 1 << 19, None, "mixed_bfloat16"),
    "cfg4_linear_nif_3d_128x6": ("NIFMultiScaleLastLayerParameterized", ms(128, 6, 32, 2, 10, 3, 3, 1, conn="last_layer"), 1 << 21, None),
    "cfg4_linear_nif_3d_128x6_bf16": ("NIFMultiScaleLastLayerParameterized", ms(128, 6, 32, 2, 10, 3, 3, 1, conn="last_layer"), 1 << 21, None,
                                      "mixed_bfloat16"),
    # Keras' mixed_float16 (r4: k_snet4<.., PR = 2>; the dL/da stash rows and the weight-gradient sums stay fp32)
    "cfg2_wave_4x64_f16": ("NIFMultiScale", ms(64, 4, 32, 2, 1, 1, 1, 1), 1 << 20, None, "mixed_float16"),
    "cfg3_ms_6x128_2d_f16": ("NIFMultiScale", ms(128, 6, 64, 2, 1, 2, 1, 1), 1 << 19, None, "mixed_float16"),
    "cfg4_linear_nif_3d_128x6_f16": ("NIFMultiScaleLastLayerParameterized", ms(128, 6, 32, 2, 10, 3, 3, 1, conn="last_layer"), 1 << 21, None,
                                     "mixed_float16"),
    # (not BASELINE configs: the !SMALL forms of k_pnet_bwg -- latent_dim 3, two parameter inputs -- with a fixed and a generic activation)
    "aux_pnet_r3_swish_4x64": ("NIFMultiScale", ms(64, 4, 32, 2, 3, 1, 1, 2), 1 << 20, None),
    "aux_pnet_r3_tanh_4x64": ("NIFMultiScale", ms(64, 4, 32, 2, 3, 1, 1, 2, p_act="tanh"), 1 << 20, None),
    "cfg1_nif_swish_2x32": ("NIF", ({"input_dim": 1, "output_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"},
                                    {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}), 1 << 20, None),
}


def csrc_sha():
    import bench
    return bench.csrc_sha()


def load_config_traffic():
    """profiles/configs_traffic.json (tools/pmc_configs.sh + tools/pmc_configs_md.py: FETCH_SIZE / WRITE_SIZE of every kernel of every
    named config, stamped with the content hash of nif_amd/csrc/ the passes ran on) -- None when absent"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "configs_traffic.json")))
    except (OSError, ValueError):
        return None


def roofline(name, s, B, nx, policy, kernel_ms, tj=None, sha=None):
    """The roofline block of one config's DOMINANT kernel (the fused ShapeNet kernel: k_snet6 when the weight-gradient group is empty,
    else k_snet4 / k_sobw / k_sob) -- r5: from MEASURED bytes.  `traffic` = 2 x FETCH_SIZE + WRITE_SIZE of that kernel per launch from
    the committed PMC passes of the same config, dropped (`traffic_stale`) when they ran on other sources than this build;
    `traffic_ratio` against the step's algorithmic bytes 4 (pi + si + so [+ so nx]) per point; achieved HBM GB/s = traffic / the
    kernel's time in THIS run; the 16-bit matrix work the kernel executes against 2.5 PF; SURVEY 8d's fp32-equivalent figure as a
    secondary key.  Without usable counters the HBM side is null (no design formula stands in for a measurement any more)."""
    snet_ms = kernel_ms.get("snet", 0.0)
    if snet_ms <= 0:
        return None
    ns = len(nx) if nx else 0
    n, nh = s.n_sx, s.n_hidden_mats
    ll = s.connectivity == "last_layer"
    planes = 1 if ll else s.pi_hidden + 1
    fused = kernel_ms.get("gw", 0.0) < 0.05 * snet_ms            # (k_snet6: no weight-gradient launches)
    so_eff = s.so_dim * (s.pi_hidden if ll else 1)
    n_w = s.si_dim * n + nh * n * n + n * so_eff
    sweeps = 3.0 if fused else 2.0                               # forward + data adjoint (+ the weight gradients inside the kernel)
    flop32 = sweeps * 2.0 * planes * n_w * (1 + ns) * B          # fp32-equivalent
    pol = policy in ("mixed_bfloat16", "mixed_float16")
    if fused:
        prod = (1.0 + 1.0 + 3.0) if pol else 9.0                 # k_snet6: three half products forward and adjoint, three bf16 products per weight-gradient pair
    else:
        # k_snet4 on a SIREN class (NIFMultiScale, last-layer class) since r5: PR = 3, three half products forward + three in the data
        # adjoint; class NIF and the Sobolev kernels (k_sobw / k_sob) still stream the bf16 split groups: 6 + 3; policies 1 + 1
        siren_s4 = (not ns) and s.kind != "NIF"
        prod = 2.0 if pol else (6.0 if siren_s4 else 9.0)
    nbl_even = (((n + 15) // 16) % 2) == 0 and n > 16
    exec16 = prod * 2.0 * planes * nh * n * n * (1 + ns) * B if nbl_even else 0.0
    alg_bytes = 4.0 * (s.pi_dim + s.si_dim + s.so_dim + s.so_dim * ns) * B
    t = snet_ms * 1e-3
    traffic, stale, kname = None, None, ("k_snet6" if fused else ("k_sobw / k_sob" if ns else "k_snet4"))
    if tj is not None:
        if tj.get("csrc_sha") == sha and name in tj.get("configs", {}):
            ks = tj["configs"][name]
            cand = [k for k in ks if k.replace("void ", "").startswith(("k_snet6<", "k_snet4<", "k_sobw<", "k_sob<", "k_snet3<", "k_snet<"))]
            if cand:
                k = max(cand, key=lambda k: 2.0 * ks[k]["FETCH_SIZE_KB"] + ks[k]["WRITE_SIZE_KB"])
                traffic = (2.0 * ks[k]["FETCH_SIZE_KB"] + ks[k]["WRITE_SIZE_KB"]) * 1024.0 * B / tj["configs_points"].get(name, B)
                kname = k
            stale = False
        else:
            stale = True
    hbm = traffic / t / 1e9 if traffic is not None else None
    mf = exec16 / t / 1e12
    frac_hbm = hbm / 8000.0 if hbm is not None else None
    bound = "hbm" if (frac_hbm is not None and frac_hbm >= mf / 2500.0) else "mfma"
    return {"kernel": kname, "bound": bound, "achieved": round(hbm if bound == "hbm" else mf, 2),
            "peak": 8000.0 if bound == "hbm" else 2500.0, "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
            "frac": round(frac_hbm if bound == "hbm" else mf / 2500.0, 4), "avg_ms": snet_ms,
            "frac_hbm": None if frac_hbm is None else round(frac_hbm, 4), "frac_bf16_pipe": round(mf / 2500.0, 4),
            "hbm_GBs": None if hbm is None else round(hbm, 1), "executed_bf16_TFLOPs": round(mf, 1),
            "fp32_equiv_TFLOPs": round(flop32 / t / 1e12, 1), "speedup_vs_f32_input_mfma_peak": round(flop32 / t / 1e12 / 157.3, 4),
            "frac_bf16_pipe_algorithmic": round(flop32 / t / 1e12 / 2500.0, 4),
            "algorithmic_bytes_per_point": alg_bytes / B, "traffic": traffic, "traffic_stale": stale,
            "traffic_ratio": None if traffic is None else round(traffic / alg_bytes, 1),
            "traffic_unit": "HBM bytes per launch of the dominant kernel (PMC: 2 x FETCH_SIZE + WRITE_SIZE, profiles/configs_traffic.json)",
            "fused_weight_gradients": bool(fused), "csrc_sha": sha}


def run_config(name, steps=10, warmup=3, tj=None, sha=None, host_buffers=False):
    """One named config: build the engine, put synthetic inputs of the config's shape in HBM, ramp the clock, time `steps` train steps
    (loss + gradient + Adam; HIP events on the library's stream), then an instrumented leg for the per-kernel-group breakdown.
    -> the config's record (bench.py's `configs` block and this tool's JSON document use the same function)."""
    import time as _t
    import nif_amd
    from nif_amd.engine import DeviceArray
    work = WORK[name]
    cls, (cs, cp), B, xi = work[:4]
    policy = work[4] if len(work) > 4 else "float32"
    nif_amd.set_seed(0)
    m = getattr(nif_amd, cls)(cs, cp, mixed_policy=policy)
    m.build()
    e = m._engine
    ncol = cp["input_dim"] + cs["input_dim"]
    so = cs["output_dim"]
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(B, ncol)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, so)).astype(np.float32)
    d_x, d_y = DeviceArray(e, x.size), DeviceArray(e, y.size)
    d_x.upload(x); d_y.upload(y)
    d_g = None
    if xi:
        g = rng.uniform(-1, 1, size=(B, so * len(xi))).astype(np.float32)
        d_g = DeviceArray(e, g.size); d_g.upload(g)
    adam = nif_amd.Adam(1e-3).as_struct()

    def step():
        if xi:
            e.sobolev_loss_grad_dev(d_x.at(0), d_y.at(0), d_g.at(0), None, B, B, xi, 0.1)
        else:
            e.loss_grad_dev(d_x.at(0), d_y.at(0), None, B, B)
        e.adam_step_dev(adam)

    # clock ramp (the shader clock needs ~30 ms of load): steps for >= 80 ms, then the clean timing, then the
    # instrumented leg (HIP events between the kernels: its per-kernel times add up to more than a clean step)
    t0 = _t.perf_counter(); nr = 0
    while nr < warmup or _t.perf_counter() - t0 < 0.08:
        step(); nr += 1
        if nr % 4 == 0:
            e.sync()
    e.sync()
    e.timer_start()
    for _ in range(steps):
        step()
    total = e.timer_stop()
    e.profile_enable(True)
    e.profile_read(reset=True)
    for _ in range(steps):
        step()
    e.sync()
    prof = e.profile_read(reset=True)
    e.profile_enable(False)
    msstep = total / steps
    rec = {"points": B, "policy": policy, "ms_per_step": round(msstep, 4), "Mpts_per_s": round(B / msstep / 1e3, 2),
           "steps": steps, "params": int(e.n_params),
           "kernel_ms": {k: round(v[0] / steps, 4) for k, v in prof.items() if v[1] > 0}}
    rec["roofline"] = roofline(name, m._spec, B, xi, policy, rec["kernel_ms"], tj, sha)
    if host_buffers:
        # the same step when the boundary hands over HOST buffers (nif_train_step: H2D of x, y + step + loss readback)
        for _ in range(2):
            e.train_step(x, y, None, adam)
        t0 = _t.perf_counter()
        for _ in range(steps):
            e.train_step(x, y, None, adam)
        dt = (_t.perf_counter() - t0) / steps
        rec["host_buffers_ms_per_step"] = round(dt * 1e3, 4)
        rec["host_buffers_Mpts_per_s"] = round(B / dt / 1e6, 2)
    d_x.free(); d_y.free()
    if d_g is not None:
        d_g.free()
    e.close()
    return rec


def run_cfg0(epochs=10):
    """configs[0]: tutorial 1 -- NIF 2x32 + 2x32, the 10k-point lattice of SURVEY 8d, Model.fit with batch 512, 20 steps per epoch"""
    import time
    import nif_amd
    nif_amd.set_seed(0)
    cs = {"input_dim": 1, "output_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
    cp = {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
    m = nif_amd.NIF(cs, cp)
    model = m.build()
    model.compile(nif_amd.Adam(1e-3), "mse")
    t = np.repeat(np.linspace(0, 90, 50), 200); xx = np.tile(np.arange(200) * 0.005, 50)
    raw = np.stack([t, xx, nif_amd.data.traveling_wave(t, xx, 4.0)], axis=1)
    data, _, _ = nif_amd.data.PointWiseData.standard_normalize(raw)
    x0, y0 = data[:, :2].astype(np.float32), data[:, 2:3].astype(np.float32)
    model.fit(x0, y0, epochs=2, batch_size=512, verbose=0)
    t0 = time.perf_counter()
    h = model.fit(x0, y0, epochs=epochs, batch_size=512, verbose=0)
    dt = time.perf_counter() - t0
    nsteps = epochs * ((x0.shape[0] + 511) // 512)
    rec = {"points": int(x0.shape[0]), "batch": 512, "us_per_step": round(dt / nsteps * 1e6, 1), "ms_per_step": round(dt / nsteps * 1e3, 5),
           "Mpts_per_s": round(epochs * x0.shape[0] / dt / 1e6, 2), "epochs": epochs, "final_loss": float(h.history["loss"][-1]),
           "params": int(m._engine.n_params), "roofline": None,
           "note": "Model.fit on the resident table (shuffle, gather, loss + gradient + Adam per 512-point step); launch-bound: no roofline applies at 512 points per step"}
    m._engine.close()
    return rec


# the BASELINE.json configs other than the bench line's own (configs[1]) at their per-GPU shard sizes: what bench.py's `configs` block runs
BASELINE_CONFIGS = [("configs[2] NIFMultiScale 6x128, 2-D, 2^19 points/GPU", "cfg3_ms_6x128_2d"),
                    ("configs[3] last-layer class 128x6, r = 10, so = 3, 3-D, 2^21 points/GPU", "cfg4_linear_nif_3d_128x6"),
                    ("configs[4] Sobolev 4x64, 2-D, du/dx + du/dy targets, fp32, 2^20 points/GPU", "cfg5_sobolev_2d_4x64"),
                    ("configs[4] Sobolev 4x64 under mixed_bfloat16 (the config names bf16), 2^20 points/GPU", "cfg5_sobolev_2d_4x64_bf16")]


def baseline_configs_block(steps=8, warmup=3):
    """bench.py's `configs` object: every BASELINE config besides the headline one, driver-run -- ms_per_step,
    the dominant kernel with its roofline fraction and traffic ratio, the per-kernel-group breakdown."""
    tj, sha = load_config_traffic(), csrc_sha()
    out = {"cfg0_tutorial1_fit_10k_b512": dict(run_cfg0(epochs=6), workload="configs[0] tutorial/1: NIF 2x32 + 2x32, 10k points, Model.fit batch 512")}
    for label, name in BASELINE_CONFIGS:
        rec = run_config(name, steps=steps, warmup=warmup, tj=tj, sha=sha)
        rf = rec.get("roofline") or {}
        out[name] = {"workload": label, "points": rec["points"], "policy": rec["policy"], "ms_per_step": rec["ms_per_step"],
                     "Mpts_per_s": rec["Mpts_per_s"], "steps": steps, "warmup": warmup,
                     "dominant_kernel": rf.get("kernel"), "dominant_kernel_ms": rf.get("avg_ms"), "bound": rf.get("bound"),
                     "frac": rf.get("frac"), "frac_hbm": rf.get("frac_hbm"), "frac_bf16_pipe": rf.get("frac_bf16_pipe"),
                     "traffic_ratio": rf.get("traffic_ratio"), "traffic_stale": rf.get("traffic_stale"),
                     "kernel_ms": rec["kernel_ms"]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--only", default=None)
    ap.add_argument("--exact", action="store_true", help="--only names ONE config (default: every config whose name contains it)")
    ap.add_argument("--out", default=None, help="where the JSON document goes (default: gpurun_out/bench_configs.json for a full run, "
                    "nothing for an --only run: tools/pmc_configs.sh calls this per config under the profiler)")
    a = ap.parse_args()
    out = {}
    tj, sha = load_config_traffic(), csrc_sha()
    for name in WORK:
        if a.only and (a.only != name if a.exact else a.only not in name):
            continue
        rec = run_config(name, a.steps, a.warmup, tj, sha, host_buffers=name.startswith("cfg2"))
        out[name] = rec
        print(name, json.dumps(rec), flush=True)
    if not a.only or ("cfg0" in a.only and not a.exact):
        rec = run_cfg0()
        out["cfg0_tutorial1_fit_10k_b512"] = rec
        print("cfg0_tutorial1_fit_10k_b512", json.dumps(rec), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    # ONE JSON document (json.load works on it; r4's profiles/r04_configs.json was the per-line log): copy to profiles/rNN_configs.json
    doc = {"csrc_sha": sha, "traffic_source": None if tj is None else tj.get("source"), "configs": out}
    path = a.out or (None if a.only else os.path.join(ROOT, "gpurun_out", "bench_configs.json"))
    if path:
        with open(path, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
