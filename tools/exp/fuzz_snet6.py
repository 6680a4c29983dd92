"""random shapes INSIDE k_snet6's domain (plain-SIREN NIFMultiScale, 49-64 units, 1-4 hidden matrices, latent_dim 1, 1-3 coordinates /
outputs; any ParameterNet): loss and every gradient tensor against the fp64 oracle at a small batch, and -- the barrier schedule over
several tile rounds per workgroup -- the fused kernel against the k_snet4 + k_gw_* path of the same step at 10^4 .. 10^5 points; float32
and the two 16-bit policies.   usage: python tools/exp/fuzz_snet6.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nif_amd
from oracle import nif_oracle as O
from tests.test_gpu_parity import _per_tensor_rel

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
nbad = 0
for i in range(ncase):
    n, L = int(rng.integers(49, 65)), int(rng.integers(1, 5))
    si, so, pi = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
    nst, lst = int(rng.choice([6, 20, 32, 40, 64])), int(rng.integers(1, 4))
    p_act = str(rng.choice(["sine", "swish", "tanh"])); p_res = bool(rng.integers(0, 2))
    policy = str(rng.choice(["float32", "float32", "mixed_bfloat16", "mixed_float16"]))
    cs = {"input_dim": si, "output_dim": so, "units": n, "nlayers": L, "use_resblock": False, "connectivity": "full", "omega_0": 30.0, "weight_init_factor": 0.01}
    cp = {"input_dim": pi, "latent_dim": 1, "units": nst, "nlayers": lst, "activation": p_act, "use_resblock": p_res, "omega_0": 30.0}
    spec = O.Spec("NIFMultiScale", cs, cp)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    names = [nm for nm, _ in spec.param_shapes()]
    ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 2.0).astype(np.float32)
    m = nif_amd.NIFMultiScale(cs, cp, mixed_policy=policy); model = m.build(); model.set_weights(ws)
    e = m._engine
    msg = []
    # (a) small batch against the oracle (float32 only: the policies have their own emulating oracles in the test suite)
    B = int(rng.choice([1, 16, 17, 33, 100, 257, 1000]))
    x = rng.uniform(-1, 1, size=(B, pi + si)).astype(np.float32); y = rng.uniform(-1, 1, size=(B, so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32)
    if policy == "float32":
        lref, gref = O.loss_and_grad(spec, [w.astype(np.float64) for w in ws], x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))
        loss, g = e.loss_and_grad(x, y, sw)
        rel = _per_tensor_rel(spec, g, O.flatten(gref))
        if not (abs(loss - lref) <= 3e-6 * abs(lref) and max(rel.values()) < 1e-4):
            msg.append(("oracle", B, abs(loss - lref) / abs(lref), max(rel.values()), {k: "%.1e" % v for k, v in rel.items() if v > 3e-5},
                        {nm: "%.1e" % float(np.linalg.norm(t)) for (nm, _), t in zip(spec.param_shapes(), gref)}))
            e.set_option("fuse_gw", 0)
            _, g0_ = e.loss_and_grad(x, y, sw)
            e.set_option("fuse_gw", 1)
            msg.append(("same batch on the stash path", {k: "%.1e" % v for k, v in _per_tensor_rel(spec, g0_, O.flatten(gref)).items() if v > 3e-5}))
    # (b) several tile rounds per workgroup: fused kernel vs the k_snet4 + k_gw_* path
    Bb = int(rng.choice([10000, 40000, 70001, 131072]))
    xb = rng.uniform(-1, 1, size=(Bb, pi + si)).astype(np.float32); yb = rng.uniform(-1, 1, size=(Bb, so)).astype(np.float32)
    l1, g1 = e.loss_and_grad(xb, yb)
    e.set_option("fuse_gw", 0)
    l0, g0 = e.loss_and_grad(xb, yb)
    e.set_option("fuse_gw", 1)
    d = float(np.linalg.norm(np.asarray(g1, dtype=np.float64) - g0) / np.linalg.norm(g0))
    bar = 2e-5 if policy == "float32" else 5e-3      # (policies: the two paths round the weight-gradient operands differently)
    if not (abs(l1 - l0) <= 1e-5 * abs(l0) + (0 if policy == "float32" else 2e-3 * abs(l0)) and d < bar):
        msg.append(("fused vs stash path", Bb, abs(l1 - l0) / abs(l0), d))
    nbad += bool(msg)
    print("FAIL" if msg else "ok  ", i, "n", n, "L", L, "si", si, "so", so, "pi", pi, "pnet", nst, lst, p_act, "res" if p_res else "-", policy, "B", B, Bb, msg if msg else "", flush=True)
    e.close()
print("cases %d, failing %d" % (ncase, nbad))
