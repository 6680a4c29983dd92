"""gradient error of the default (16-bit split) path per tensor: 2^20 points against the f32-input MFMA path, a 65 536-point sub-batch
against the fp64 oracle (tests/test_gpu_parity.py::test_full_size_gradient_split_path_vs_fp32_mfma_vs_oracle takes its bars from here)"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_parity import _full_size_setup, _per_tensor_rel, _rel, _oracle_grad_chunked
from oracle import nif_oracle as O
m, model, x, y = _full_size_setup()
e = m._engine
spec = O.Spec("NIFMultiScale", m.cfg_shape_net, m.cfg_parameter_net)
ws = [w.astype(np.float64) for w in model.get_weights()]
ls, gs = e.loss_and_grad(x, y)
e.set_option("fp32_mfma", 1); lf, gf = e.loss_and_grad(x, y); e.set_option("fp32_mfma", 0)
rel = _per_tensor_rel(spec, gs, gf)
print("2^20: default vs fp32-mfma: loss rel %.2e  max tensor %.2e  flat %.2e" % (abs(ls-lf)/abs(lf), max(rel.values()), _rel(gs, gf.astype(np.float64))))
n_s = 1 << 16
lo_, go_ = _oracle_grad_chunked(spec, ws, x[:n_s], y[:n_s])
for fp32 in (0, 1):
    e.set_option("fp32_mfma", fp32)
    l_, g_ = e.loss_and_grad(x[:n_s], y[:n_s])
    rel = _per_tensor_rel(spec, g_, go_)
    print("65536 vs oracle fp32_mfma=%d: loss rel %.2e max tensor %.2e (%s) flat %.2e" % (fp32, abs(l_-lo_)/abs(lo_), max(rel.values()), max(rel, key=rel.get), _rel(g_, go_)))
    print("   per tensor: " + "  ".join("%s %.1e" % (k, v) for k, v in rel.items()))
