"""ctypes binding of libnif_hip.so (include/nif_hip.h).  No tensor framework, no numpy-side compute: every
numeric result of the package comes out of the HIP library.  There is deliberately no fallback:
if the shared object is missing or no gfx950 device is visible, we raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NIF_LIB") or os.path.join(_HERE, "libnif_hip.so")   # NIF_LIB: measurement builds

NIF_ABI_VERSION = 2
KIND_NIF, KIND_MULTISCALE, KIND_LASTLAYER = 0, 1, 2
COMM_ID_BYTES = 128
DT_F32, DT_F64, DT_I64 = 0, 1, 2
POLICY_IDS = {"float32": 0, "mixed_bfloat16": 1, "mixed_float16": 2}
OP_SUM, OP_MAX, OP_MIN = 0, 1, 2

PROF_NAMES = ["pack", "pnet_fwd", "snet", "pnet_bwd", "gw", "reduce", "adam", "given_w", "latent_to_w", "snet_fwd"]

ACT_IDS = {
    None: 0, "linear": 0, "sine": 1, "swish": 2, "silu": 2, "tanh": 3, "relu": 4, "sigmoid": 5,
    "elu": 6, "softplus": 7, "gelu": 8, "selu": 9, "softsign": 10, "exponential": 11, "hard_sigmoid": 12,
}


# keras.losses.get names -> nif_loss
LOSS_IDS = {
    "mse": 0, "MSE": 0, "mean_squared_error": 0, "MeanSquaredError": 0,
    "mae": 1, "MAE": 1, "mean_absolute_error": 1, "MeanAbsoluteError": 1,
    "huber": 2, "huber_loss": 2, "Huber": 2,
    "log_cosh": 3, "logcosh": 3, "LogCosh": 3,
}


class NifError(RuntimeError):
    pass


class nif_cfg(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("kind", C.c_int32), ("pi_dim", C.c_int32), ("si_dim", C.c_int32),
        ("so_dim", C.c_int32), ("n_sx", C.c_int32), ("l_sx", C.c_int32), ("n_st", C.c_int32),
        ("l_st", C.c_int32), ("latent_dim", C.c_int32), ("s_act", C.c_int32), ("s_resblock", C.c_int32),
        ("s_omega0", C.c_float), ("p_act", C.c_int32), ("p_resblock", C.c_int32), ("p_omega0", C.c_float),
        ("mixed_policy", C.c_int32), ("reserved", C.c_int32 * 7),
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
    "nif_host_alloc": (C.c_int, [_CTX, C.c_int64, C.POINTER(_VP)]),
    "nif_host_free": (C.c_int, [_CTX, _VP]),
    "nif_h2d_async": (C.c_int, [_CTX, _VP, _VP, C.c_int64, C.c_int32]),
    "nif_copy_acquire": (C.c_int, [_CTX, C.c_int32]),
    "nif_copy_release": (C.c_int, [_CTX, C.c_int32]),
    "nif_copy_wait_host": (C.c_int, [_CTX, C.c_int32]),
    "nif_gather_rows_dev": (C.c_int, [_CTX, _VP, _VP, C.c_int64, C.c_int32, _VP]),
    "nif_stream": (_VP, [_CTX]),
    "nif_grad_dev": (_VP, [_CTX]),
    "nif_params_dev": (_VP, [_CTX]),
    "nif_forward": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_forward_dev": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_pnet_latent": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_x_to_phi": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_jacobian": (C.c_int, [_CTX, _VP, C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.c_int32, _VP, _VP]),
    "nif_hessian": (C.c_int, [_CTX, _VP, C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.c_int32, _VP, _VP, _VP]),
    "nif_hessian_dev": (C.c_int, [_CTX, _VP, C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.c_int32, _VP, _VP, _VP]),
    "nif_latent_to_w": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_latent_to_w_dev": (C.c_int, [_CTX, _VP, C.c_int64, _VP]),
    "nif_shapenet_given_w": (C.c_int, [_CTX, _VP, _VP, C.c_int64, _VP]),
    "nif_shapenet_given_w_dev": (C.c_int, [_CTX, _VP, _VP, C.c_int64, _VP]),
    "nif_loss_grad_dev": (C.c_int, [_CTX, _VP, _VP, _VP, C.c_int64, C.c_int64]),
    "nif_sobolev_loss_grad_dev": (C.c_int, [_CTX, _VP, _VP, _VP, _VP, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.c_int32,
                                            C.c_float]),
    "nif_sobolev_loss_grad_dev_y": (C.c_int, [_CTX, _VP, _VP, _VP, _VP, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.c_int32,
                                              C.POINTER(C.c_int32), C.c_int32, C.c_float]),
    "nif_graph_begin": (C.c_int, [_CTX]),
    "nif_graph_end": (C.c_int, [_CTX, C.POINTER(C.c_int32)]),
    "nif_graph_launch": (C.c_int, [_CTX, C.c_int32, C.POINTER(nif_adam)]),
    "nif_graph_destroy": (C.c_int, [_CTX, C.c_int32]),
    "nif_set_loss": (C.c_int, [_CTX, C.c_int32]),
    "nif_sobolev_forward_dev": (C.c_int, [_CTX, _VP, C.c_int64, C.POINTER(C.c_int32), C.c_int32, _VP, _VP]),
    "nif_adam_step_dev": (C.c_int, [_CTX, C.POINTER(nif_adam)]),
    "nif_zero_grad": (C.c_int, [_CTX]),
    "nif_reserve": (C.c_int, [_CTX, C.c_int64, C.c_int32]),
    "nif_device_pci_bus_id": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
    "nif_comm_unique_id": (C.c_int, [_VP]),
    "nif_comm_init_rank": (C.c_int, [_CTX, _VP, C.c_int32, C.c_int32]),
    "nif_comm_init_all": (C.c_int, [C.POINTER(_CTX), C.c_int32]),
    "nif_comm_destroy": (C.c_int, [_CTX]),
    "nif_comm_info": (C.c_int, [_CTX, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "nif_allreduce_grad": (C.c_int, [_CTX]),
    "nif_comm_selftest": (C.c_int, [_CTX, C.POINTER(C.c_int32)]),
    "nif_allreduce_grad_multi": (C.c_int, [C.POINTER(_CTX), C.c_int32]),
    "nif_comm_allreduce": (C.c_int, [_CTX, _VP, C.c_int64, C.c_int32, C.c_int32]),
    "nif_comm_barrier": (C.c_int, [_CTX]),
    "nif_train_step_multi": (C.c_int, [C.POINTER(_CTX), C.c_int32, _VP, _VP, _VP, C.c_int64, C.POINTER(nif_adam), _FP]),
    "nif_loss_and_grad": (C.c_int, [_CTX, _VP, _VP, _VP, C.c_int64, _FP, _VP]),
    "nif_train_step": (C.c_int, [_CTX, _VP, _VP, _VP, C.c_int64, C.POINTER(nif_adam), _FP]),
    "nif_set_regularizer": (C.c_int, [_CTX, C.c_float, C.c_float, C.c_int64, C.c_int64]),
    "nif_set_shapenet_regularizer": (C.c_int, [_CTX, C.c_float, C.c_float]),
    "nif_set_jac_regularizer": (C.c_int, [_CTX, C.c_float]),
    "nif_set_activity_regularizer": (C.c_int, [_CTX, C.c_float, C.c_float]),
    "nif_metric_accumulate": (C.c_int, [_CTX, C.c_float]),
    "nif_metric_read": (C.c_int, [_CTX, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
    "nif_last_loss": (C.c_int, [_CTX, _FP]),
    "nif_grad_read": (C.c_int, [_CTX, _FP, _VP]),
    "nif_set_option": (C.c_int, [_CTX, C.c_char_p, C.c_int32]),
    "nif_profile_enable": (C.c_int, [_CTX, C.c_int]),
    "nif_profile_read": (C.c_int, [_CTX, _FP, C.POINTER(C.c_int64), C.c_int, C.c_int]),
    "nif_debug_timeline": (C.c_int, [_CTX, C.POINTER(C.c_int64), C.c_int32]),
    "nif_timer_start": (C.c_int, [_CTX]),
    "nif_timer_stop": (C.c_int, [_CTX, _FP]),
}

_lib = None


def load():
    """dlopen libnif_hip.so and bind every symbol of include/nif_hip.h.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NifError(
            "libnif_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  nif_amd has no CPU fallback." % LIB_PATH)
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
