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
    "cfg5_sobolev_2d_dx_only": ("NIFMultiScale", ms(64, 4, 32, 2, 1, 2, 1, 1), 1 << 20, [1]),
    # configs[4] names bf16: the mixed_bfloat16 policy of the build (single bf16 product per n x n operand pair)
    "cfg5_sobolev_2d_4x64_bf16": ("NIFMultiScale", ms(64, 4, 32, 2, 1, 2, 1, 1), 1 << 20, [1, 2], "mixed_bfloat16"),
    "cfg2_wave_4x64_bf16": ("NIFMultiScale", ms(64, 4, 32, 2, 1, 1, 1, 1), 1 << 20, None, "mixed_bfloat16"),
    "cfg4_linear_nif_3d_128x6": ("NIFMultiScaleLastLayerParameterized", ms(128, 6, 32, 2, 10, 3, 3, 1, conn="last_layer"), 1 << 21, None),
    "cfg4_linear_nif_3d_128x6_bf16": ("NIFMultiScaleLastLayerParameterized", ms(128, 6, 32, 2, 10, 3, 3, 1, conn="last_layer"), 1 << 21, None,
                                      "mixed_bfloat16"),
    "cfg1_nif_swish_2x32": ("NIF", ({"input_dim": 1, "output_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"},
                                    {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}), 1 << 20, None),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    import nif_amd
    from nif_amd.engine import DeviceArray
    out = {}
    for name, work in WORK.items():
        cls, (cs, cp), B, xi = work[:4]
        policy = work[4] if len(work) > 4 else "float32"
        if a.only and a.only not in name:
            continue
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

        for _ in range(a.warmup):
            step()
        e.sync()
        e.profile_enable(True)
        e.profile_read(reset=True)
        e.timer_start()
        for _ in range(a.steps):
            step()
        total = e.timer_stop()
        prof = e.profile_read(reset=True)
        e.profile_enable(False)
        msstep = total / a.steps
        rec = {"points": B, "ms_per_step": round(msstep, 4), "Mpts_per_s": round(B / msstep / 1e3, 2),
               "params": int(e.n_params),
               "kernel_ms": {k: round(v[0] / a.steps, 4) for k, v in prof.items() if v[1] > 0}}
        if name.startswith("cfg2"):
            # the same step when the boundary hands over HOST buffers (nif_train_step: H2D of x, y + step + loss readback)
            import time
            for _ in range(2):
                e.train_step(x, y, None, adam)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                e.train_step(x, y, None, adam)
            dt = (time.perf_counter() - t0) / a.steps
            rec["host_buffers_ms_per_step"] = round(dt * 1e3, 4)
            rec["host_buffers_Mpts_per_s"] = round(B / dt / 1e6, 2)
        out[name] = rec
        print(name, json.dumps(rec), flush=True)
        d_x.free(); d_y.free()
        if d_g is not None:
            d_g.free()
        e.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
