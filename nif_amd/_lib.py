"""ctypes binding of libnif_hip.so (include/nif_hip.h).  No torch, no numpy-side compute: every
numeric result of the package comes out of the HIP library.  There is deliberately no fallback:
if the shared object is missing or no gfx950 device is visible, we raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnif_hip.so")

NIF_ABI_VERSION = 1
KIND_NIF, KIND_MULTISCALE, KIND_LASTLAYER = 0, 1, 2

PROF_NAMES = ["pack", "pnet_fwd", "snet", "pnet_bwd", "gw", "reduce", "adam", "given_w", "latent_to_w", "snet_fwd"]

ACT_IDS = {
    None: 0, "linear": 0, "sine": 1, "swish": 2, "silu": 2, "tanh": 3, "relu": 4, "sigmoid": 5,
    "elu": 6, "softplus": 7, "gelu": 8,
}


class NifError(RuntimeError):
    pass


class nif_cfg(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("kind", C.c_int32), ("pi_dim", C.c_int32), ("si_dim", C.c_int32),
        ("so_dim", C.c_int32), ("n_sx", C.c_int32), ("l_sx", C.c_int32), ("n_st", C.c_int32),
        ("l_st", C.c_int32), ("latent_dim", C.c_int32), ("s_act", C.c_int32), ("s_resblock", C.c_int32),
        ("s_omega0", C.c_float), ("p_act", C.c_int32), ("p_resblock", C.c_int32), ("p_omega0", C.c_float),
        ("reserved", C.c_int32 * 8),
    ]


class nif_tensor_desc(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("offset", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32)]


class nif_adam(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float)]


_FP = C.POINTER(C.c_float)
_VP = C.c_void_p
_CTX = C.c_void_p

# name -> (restype, argtypes); mirrors include/nif_hip.h one to one
SIGNATURES = {
    "nif_last_error": (C.c_char_p, []),
    "nif_abi_version": (C.c_int, []),
    "nif_device_count": (C.c_int, []),
    "nif_create": (C.c_int, [C.POINTER(nif_cfg), C.c_int, C.POINTER(_CTX)]),
    "nif_destroy": (C.c_int, [_CTX]),
    "nif_param_count": (C.c_int, [_CTX, C.POINTER(C.c_int64)]),
    "nif_po_dim": (C.c_int, [_CTX, C.POINTER(C.c_int64)]),
    "nif_param_layout": (C.c_int, [_CTX, C.POINTER(nif_tensor_desc), C.POINTER(C.c_int32)]),
    "nif_set_params": (C.c_int, [_CTX, _VP, C.c_int64]),
    "nif_get_params": (C.c_int, [_CTX, _VP, C.c_int64]),
    "nif_get_opt_state": (C.c_int, [_CTX, _VP, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    "nif_set_opt_state": (C.c_int, [_CTX, _VP, _VP, C.c_int64, C.c_int64]),
    "nif_dev_alloc": (C.c_int, [_CTX, C.c_int64, C.POINTER(_VP)]),
    "nif_dev_free": (C.c_int, [_CTX, _VP]),
    "nif_h2d": (C.c_int, [_CTX, _VP, _VP, C.c_int64]),
    "nif_d2h": (C.c_int, [_CTX, _VP, _VP, C.c_int64]),
    "nif_sync": (C.c_int, [_CTX]),
    "nif_stream": (_VP, [_CTX]),
    "nif_grad_dev": (_VP, [_CTX]),
    "nif_params_dev": (_VP, [_CTX]),
    "nif_forward": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_forward_dev": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_pnet_latent": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_x_to_phi": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_jacobian": (C.c_int, [_CTX, _VP, C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.c_int32, _VP, _VP]),
    "nif_latent_to_w": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_latent_to_w_dev": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_shapenet_given_w": (C.c_int, [_CTX, _VP, _VP, C.c_int64, _VP]),
    "nif_shapenet_given_w_dev": (C.c_int, [_CTX, _VP, _VP, C.c_int64, _VP]),
    "nif_loss_grad_dev": (C.c_int, [_CTX, _VP, _VP, _VP, C.c_int64, C.c_int64]),
    "nif_sobolev_loss_grad_dev": (C.c_int, [_CTX, _VP, _VP, _VP, _VP, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.c_int32,
                                            C.c_float]),
    "nif_sobolev_forward_dev": (C.c_int, [_CTX, _VP, C.c_int64, C.POINTER(C.c_int32), C.c_int32, _VP, _VP]),
    "nif_adam_step_dev": (C.c_int, [_CTX, C.POINTER(nif_adam)]),
    "nif_loss_and_grad": (C.c_int, [_CTX, _VP, _VP, _VP, C.c_int64, _FP, _VP]),
    "nif_train_step": (C.c_int, [_CTX, _VP, _VP, _VP, C.c_int64, C.POINTER(nif_adam), _FP]),
    "nif_set_regularizer": (C.c_int, [_CTX, C.c_float, C.c_float, C.c_int64, C.c_int64]),
    "nif_metric_accumulate": (C.c_int, [_CTX, C.c_float]),
    "nif_metric_read": (C.c_int, [_CTX, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
    "nif_last_loss": (C.c_int, [_CTX, _FP]),
    "nif_profile_enable": (C.c_int, [_CTX, C.c_int]),
    "nif_profile_read": (C.c_int, [_CTX, _FP, C.POINTER(C.c_int64), C.c_int, C.c_int]),
    "nif_debug_timeline": (C.c_int, [_CTX, C.POINTER(C.c_int64), C.c_int32]),
    "nif_timer_start": (C.c_int, [_CTX]),
    "nif_timer_stop": (C.c_int, [_CTX, _FP]),
}

_lib = None


def _one_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's) and asks for
    it by the unversioned file name, so the loader does NOT reuse a copy that libnif_hip.so pulled in from /opt/rocm
    earlier: a process that builds a model first and joins the RCCL process group later would hold two runtimes and
    the second one sees no GPU.  The other order is fine (libnif_hip.so needs `libamdhip64.so.7`, which matches
    whatever is loaded).  So: when torch is installed but not imported yet, load ITS runtime first.  Locating the
    package does not import it; NIF_NO_TORCH_HIP=1 skips this."""
    import sys
    if "torch" in sys.modules or os.environ.get("NIF_NO_TORCH_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=getattr(os, "RTLD_GLOBAL", 0x100) | getattr(os, "RTLD_NOW", 2))
    except (ImportError, OSError, ValueError):
        pass


def load():
    """dlopen libnif_hip.so and bind every symbol of include/nif_hip.h.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NifError(
            "libnif_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  nif_amd has no CPU fallback." % LIB_PATH)
    _one_hip_runtime()
    lib = C.CDLL(LIB_PATH, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.nif_abi_version() != NIF_ABI_VERSION:
        raise NifError("libnif_hip.so ABI version %d != %d" % (lib.nif_abi_version(), NIF_ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().nif_last_error()
        raise NifError("libnif_hip error %d: %s" % (rc, msg.decode("utf-8", "replace") if msg else "?"))


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)
