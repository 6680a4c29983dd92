"""Freezes the NumPy fp64 oracle (oracle/nif_oracle.py) into tests/golden/oracle_v1.npz.

For every case -- the ten small configurations of tests/cfgs.py plus three shapes that select the production kernels (the benchmark's
4 x 64 SIREN net, class NIF 2 x 32, the last-layer class) -- at B in {7, 64, 257}: the weights (float32 values), inputs, targets and
sample weights that went in, and what the oracle returned for them in fp64:

    u       forward(spec, ws, x)                                          [B, so]
    loss    loss_and_grad(spec, ws, x, y, sw)[0]                           scalar (Keras 'mse', SUM_OVER_BATCH_SIZE, sample-weighted)
    grad    flatten(loss_and_grad(...)[1])                                 [P]   (Keras variable order)
    jac     jacobian_analytic(spec, ws, x, all outputs, coordinate cols)   [B, so, si]
    theta1  one Keras-2.11 Adam step from zero moments (lr 1e-3)           [P]

The fixture is DATA (inputs and expected outputs), written by the oracle as it stood when this script last ran:

    python tests/golden/make_oracle_goldens.py          (in the build container; NumPy only)

tests/test_oracle.py::test_live_oracle_reproduces_frozen_vectors asserts that today's oracle still returns these numbers to 1e-13
(so any later edit of the oracle's arithmetic shows up as a failing CPU test or as a diff of this file), and
tests/test_gpu_parity.py::test_hip_path_matches_frozen_oracle_vectors compares the HIP path with the FROZEN numbers -- a kernel and
the oracle can no longer move together unnoticed.  The oracle itself is a restatement (TensorFlow is absent here: parity unpinned,
DESIGN 0 row c); this file pins the restatement in time, not against the reference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import nif_oracle as O      # noqa: E402
from tests.cfgs import ALL_SMALL, cfg_ms, cfg_nif, cfg_ll     # noqa: E402

BATCHES = (7, 64, 257)
CASES = dict(ALL_SMALL)
CASES.update({
    # shapes that select the production kernels: k_snet6 (the benchmark net), k_snet4<2, NIF>, k_snet4<2, LL>
    "prod_ms_cfg2_64x4": cfg_ms(n=64, L=4, nst=32, lst=2, r=1, si=1, so=1, pi=1, p_act="swish"),
    "prod_nif_cfg1_32x2": cfg_nif(n=32, L=2, nst=32, lst=2, r=1, si=1, so=1, pi=1),
    "prod_ll_32x2_r3": cfg_ll(n=32, L=2, nst=32, lst=1, r=3, si=2, so=2, pi=1),
})


def draw(name, B):
    """(spec, ws [float32-valued], x, y, sw) of a case: deterministic in (name, B); the weights depend on the name only"""
    kind, cs, cp = CASES[name]
    spec = O.Spec(kind, cs, cp)
    seed = sum(ord(c) * (i + 1) for i, c in enumerate(name)) % 100003
    ws = O.init_weights(spec, np.random.default_rng(seed), dtype=np.float32)
    names = [nm for nm, _ in spec.param_shapes()]
    if kind == "NIFMultiScale":            # weight_init_factor = 0.01 makes the z-dependence tiny: enlarge so the vectors see it
        ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 2.0).astype(np.float32)
    if kind == "NIFMultiScaleLastLayerParameterized":
        ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 30.0).astype(np.float32)
    rng = np.random.default_rng(seed + 7919 * B)
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, spec.so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32)
    return spec, ws, x, y, sw


def evaluate(spec, ws, x, y, sw):
    """what is frozen: the oracle's answers in fp64 for float32-valued weights and inputs"""
    ws64 = [w.astype(np.float64) for w in ws]
    x64, y64, sw64 = x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64)
    u = O.forward(spec, ws64, x64)
    loss, g = O.loss_and_grad(spec, ws64, x64, y64, sw64)
    grad = O.flatten(g)
    yi, xi = list(range(spec.so)), list(range(spec.pi, spec.pi + spec.si))
    _, jac = O.jacobian_analytic(spec, ws64, x64, yi, xi)
    th = O.flatten(ws64)
    f32 = lambda a: float(np.float32(a))    # noqa: E731  (the engine holds the hyper-parameters as floats)
    th1, _, _ = O.adam_step(th, grad, np.zeros_like(th), np.zeros_like(th), 1, lr=f32(1e-3), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
    return {"u": u, "loss": np.float64(loss), "grad": grad, "jac": np.asarray(jac), "theta1": th1}


def main():
    out = {"names": np.array(sorted(CASES)), "batches": np.array(BATCHES)}
    for name in sorted(CASES):
        for B in BATCHES:
            spec, ws, x, y, sw = draw(name, B)
            if B == BATCHES[0]:
                out["%s/theta" % name] = O.flatten(ws).astype(np.float32)
            k = "%s/%d/" % (name, B)
            out[k + "x"], out[k + "y"], out[k + "sw"] = x, y, sw
            for q, v in evaluate(spec, ws, x, y, sw).items():
                out[k + q] = v
    path = os.path.join(HERE, "oracle_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d cases x %d batch sizes, %.2f MB" % (path, len(CASES), len(BATCHES), os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
