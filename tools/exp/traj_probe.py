"""How far do 200-step Adam loss curves of the bf16-split path, the f32-MFMA path and the fp64 oracle drift apart?"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nif_amd, bench
from oracle import nif_oracle as O
d = np.load(os.path.join(ROOT, "tests", "golden", "traveling_wave.npz"))["data"]
data, _, _ = O.standard_normalize(d.astype(np.float64))
x, y = data[:, :2].astype(np.float32), data[:, 2:3].astype(np.float32)
steps = 200
out = {}
for lr in (1e-3, 3e-4, 1e-4):
    curves = {}
    for mode in ("split", "fp32"):
        nif_amd.set_seed(21)
        m = nif_amd.NIFMultiScale(bench.CFG_SHAPE, bench.CFG_PARAM); model = m.build()
        ws0 = [w.astype(np.float64) for w in model.get_weights()]
        e = m._engine
        e.set_option("fp32_mfma", 1 if mode == "fp32" else 0)
        adam = nif_amd.Adam(lr).as_struct()
        d_x, d_y = e.alloc(x.size), e.alloc(y.size); d_x.upload(x); d_y.upload(y)
        losses = []
        for _ in range(steps):
            e.loss_grad_dev(d_x.at(0), d_y.at(0), None, x.shape[0], x.shape[0]); losses.append(e.last_loss()); e.adam_step_dev(adam)
        curves[mode] = np.array(losses)
    spec = O.Spec("NIFMultiScale", bench.CFG_SHAPE, bench.CFG_PARAM)
    th = O.flatten(ws0); mm = np.zeros_like(th); vv = np.zeros_like(th)
    f32 = lambda a: float(np.float32(a))
    ref = []
    for t in range(1, steps + 1):
        l, g = O.loss_and_grad(spec, O.unflatten(spec, th), x.astype(np.float64), y.astype(np.float64))
        ref.append(l)
        th, mm, vv = O.adam_step(th, O.flatten(g), mm, vv, t, lr=f32(lr), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
    ref = np.array(ref)
    ds, df = np.abs(curves["split"] - ref) / ref, np.abs(curves["fp32"] - ref) / ref
    print("lr", lr, "ref[0,50,100,150,199]", ref[[0, 50, 100, 150, 199]])
    for k in (10, 25, 50, 75, 100, 125, 150, 199):
        print("  step %3d  split %.2e  fp32 %.2e   max-so-far split %.2e fp32 %.2e" % (k, ds[k], df[k], ds[:k + 1].max(), df[:k + 1].max()))
    out[str(lr)] = {"ref": ref.tolist(), "split": curves["split"].tolist(), "fp32": curves["fp32"].tolist()}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "traj_probe.json"), "w"))
