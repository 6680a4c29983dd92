"""ctypes wrapper of oracle/libnif_ref_cpu.so (C/OpenMP restatement of the reference formulation).
TEST / MEASUREMENT INFRASTRUCTURE ONLY -- see the header of nif_ref_cpu.c."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ACT = {"sine": 1, "swish": 2, "silu": 2, "tanh": 3}


class ref_cfg(C.Structure):
    _fields_ = [("kind", C.c_int), ("pi", C.c_int), ("si", C.c_int), ("so", C.c_int), ("n", C.c_int), ("L", C.c_int),
                ("nst", C.c_int), ("lst", C.c_int), ("r", C.c_int), ("s_act", C.c_int), ("p_act", C.c_int),
                ("omega_s", C.c_float), ("omega_p", C.c_float)]


def load():
    so = os.path.join(_HERE, "libnif_ref_cpu.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE])
    lib = C.CDLL(so)
    lib.nifref_po.restype = C.c_long
    lib.nifref_nparams.restype = C.c_long
    lib.nifref_loss_grad.argtypes = [C.POINTER(ref_cfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                                     C.c_long, C.c_void_p, C.POINTER(C.c_double), C.c_int]
    lib.nifref_adam.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_float,
                                C.c_float, C.c_float, C.c_float]
    return lib


def make_cfg(spec):
    """spec: oracle.nif_oracle.Spec.  Only NIF and plain NIFMultiScale with shortcut/SIREN ParameterNet."""
    if spec.kind not in ("NIF", "NIFMultiScale") or spec.s_res or spec.p_res:
        raise ValueError("nif_ref_cpu.c covers class NIF and NIFMultiScale without resblocks")
    c = ref_cfg()
    c.kind = 0 if spec.kind == "NIF" else 1
    c.pi, c.si, c.so, c.n, c.L, c.nst, c.lst, c.r = spec.pi, spec.si, spec.so, spec.n, spec.L, spec.nst, spec.lst, spec.r
    c.s_act = _ACT[spec.s_act]
    c.p_act = _ACT[spec.p_act]
    c.omega_s, c.omega_p = float(spec.omega_s), float(spec.omega_p)
    return c


def loss_and_grad(lib, cfg, theta, x, y, sw=None, micro=4096, nthreads=0):
    """One global batch as micro-batches of `micro` points (the [b, po] tensors would not fit otherwise)."""
    theta = np.ascontiguousarray(theta, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    g = np.zeros_like(theta)
    loss = C.c_double(0.0)
    B = x.shape[0]
    for b0 in range(0, B, micro):
        b = min(micro, B - b0)
        xs, ys = x[b0:b0 + b], y[b0:b0 + b]
        sws = None if sw is None else np.ascontiguousarray(sw[b0:b0 + b], dtype=np.float32)
        rc = lib.nifref_loss_grad(C.byref(cfg), theta.ctypes.data, xs.ctypes.data, ys.ctypes.data,
                                  None if sws is None else sws.ctypes.data, b, B, g.ctypes.data, C.byref(loss), nthreads)
        assert rc == 0
    return float(loss.value), g
