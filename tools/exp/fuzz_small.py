"""random shapes INSIDE k_small's domain (class NIF with any activation / plain-SIREN NIFMultiScale, units <= 32, <= 4 hidden matrices per net,
latent_dim / inputs / outputs <= 4, batches 1 .. 2048, the four losses, sample weights): loss and every gradient tensor against the fp64
oracle, and the step must really have run on k_small (one launch on the ShapeNet group, nothing on the others).
usage: python tools/exp/fuzz_small.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nif_amd
from oracle import nif_oracle as O
from tests.test_gpu_parity import _per_tensor_rel

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ACTS = ["swish", "tanh", "relu", "sigmoid", "elu", "softplus", "gelu", "selu", "softsign", "hard_sigmoid", "linear"]
nbad = nsmall = 0
for i in range(ncase):
    kind = "NIF" if rng.random() < 0.5 else "NIFMultiScale"
    n, L, nst, lst = int(rng.integers(2, 33)), int(rng.integers(1, 5)), int(rng.integers(2, 33)), int(rng.integers(1, 5))
    r, si, so, pi = (int(rng.integers(1, 5)) for _ in range(4))
    act = str(rng.choice(ACTS))
    if kind == "NIF":
        cs = {"input_dim": si, "output_dim": so, "units": n, "nlayers": L, "activation": act}
        cp = {"input_dim": pi, "latent_dim": r, "units": nst, "nlayers": lst, "activation": act}
    else:
        p_act = "sine" if rng.random() < 0.5 else act
        cs = {"input_dim": si, "output_dim": so, "units": n, "nlayers": L, "use_resblock": False, "connectivity": "full", "omega_0": 30.0,
              "weight_init_factor": 0.01}
        cp = {"input_dim": pi, "latent_dim": r, "units": nst, "nlayers": lst, "activation": p_act, "use_resblock": False, "omega_0": 30.0}
    B = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 64, 257, 512, 1000, 2047, 2048]))
    loss_kind = str(rng.choice(["mse", "mse", "mae", "huber", "log_cosh"]))
    spec = O.Spec(kind, cs, cp)
    ws = O.init_weights(spec, rng, dtype=np.float32)
    names = [nm for nm, _ in spec.param_shapes()]
    if kind == "NIFMultiScale":
        ws[names.index("pnet_last_w")] = (ws[names.index("pnet_last_w")] * 2.0).astype(np.float32)
    m = getattr(nif_amd, kind)(cs, cp); model = m.build(); model.set_weights(ws)
    model.compile(nif_amd.Adam(1e-3), loss_kind)
    e = m._engine
    e.set_loss(loss_kind)
    x = rng.uniform(-1, 1, size=(B, pi + si)).astype(np.float32)
    y = rng.uniform(-1, 1, size=(B, so)).astype(np.float32)
    sw = rng.uniform(0.5, 1.5, size=(B,)).astype(np.float32) if rng.random() < 0.5 else None
    ws64 = [w.astype(np.float64) for w in ws]
    lref, gref = O.loss_and_grad(spec, ws64, x.astype(np.float64), y.astype(np.float64), None if sw is None else sw.astype(np.float64), loss=loss_kind)
    e.profile_enable(True); e.profile_read(reset=True)
    loss, g = e.loss_and_grad(x, y, sw)
    prof = e.profile_read(reset=True); e.profile_enable(False)
    small = prof["snet"][1] == 1 and prof["pnet_fwd"][1] == 0 and prof["gw"][1] == 0
    nsmall += small
    rel = _per_tensor_rel(spec, g, O.flatten(gref))
    ok = abs(loss - lref) <= 3e-6 * abs(lref) + 1e-12 and max(rel.values()) < 2e-5
    nbad += not ok
    print("ok  " if ok else "FAIL", i, kind, "n", n, "L", L, "nst", nst, "lst", lst, "r", r, "si", si, "so", so, "pi", pi, act, cp["activation"], "B", B, loss_kind,
          "sw" if sw is not None else "-", "k_small" if small else "TILE", "loss %.1e worst %.1e" % (abs(loss - lref) / abs(lref), max(rel.values())), flush=True)
    e.close()
print("cases %d, failing %d, on k_small %d" % (ncase, nbad, nsmall))
