"""Data parallelism over the point batch: one process per GPU, `torch.distributed` as plumbing
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  Replaces the reference's
`tf.distribute.MirroredStrategy().scope()` recipe (README.md:39-49): rows (points) are independent, so
every rank computes loss and gradient of its shard, pre-scaled by 1/B_global inside the HIP kernels, and
ONE sum all-reduce of the flat float32 buffer [grad(P) | loss] per step makes every rank hold the
global-batch gradient; each rank then applies the identical Adam update (replicated state).

torch is imported lazily and only here; the compute path (libnif_hip.so) never sees it.  The
all-reduce runs on the context's own HIP stream (torch.cuda.ExternalStream) on a tensor aliasing the
library's gradient buffer (__cuda_array_interface__), so there is no host synchronisation."""
import os

_state = {"pg": False, "grad_alias": {}}


def _td():
    import torch.distributed as td
    return td


def is_initialized():
    if not _state["pg"]:
        return False
    return _td().is_initialized()


def world_size():
    return _td().get_world_size() if is_initialized() else 1


def rank():
    return _td().get_rank() if is_initialized() else 0


def local_device():
    """HIP device of this process: LOCAL_RANK under torchrun, else 0."""
    return int(os.environ.get("LOCAL_RANK", "0")) if is_initialized() else 0


def init(backend=None):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT (torchrun env)."""
    import torch
    td = _td()
    if not td.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        td.init_process_group(backend=backend)
    _state["pg"] = True
    return td.get_rank(), td.get_world_size()


def shutdown():
    if _state["pg"] and _td().is_initialized():
        _td().destroy_process_group()
    _state["pg"] = False
    _state["grad_alias"].clear()


def shard_bounds(n, world, r):
    """Contiguous, balanced split of n rows over `world` ranks: rank r gets [lo, hi)."""
    base, rem = divmod(n, world)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def all_reduce_scalar_sum(v):
    import torch
    td = _td()
    dev = "cuda" if td.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(v)], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return int(round(t.item()))


def all_reduce_ints(values, op="sum"):
    """Element-wise SUM (or MAX) over ranks of a short list of integers: ONE collective and one host read-back.
    fit() uses it once per call to learn every step's global batch size instead of syncing the host every step."""
    import torch
    td = _td()
    dev = "cuda" if td.get_backend() == "nccl" else "cpu"
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.MAX if op == "max" else td.ReduceOp.SUM)
    return [int(v) for v in t.cpu().tolist()]


def zero_grad(engine):
    """A rank whose shard has no rows left for a step still takes part in the collective: with a zero buffer."""
    import torch
    t, stream = grad_tensor(engine)
    with torch.cuda.stream(stream):
        t.zero_()


class _DevPtr(object):
    """Zero-copy view of a raw device buffer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def grad_tensor(engine):
    """torch float32 tensor [P+1] aliasing engine's flat gradient||loss buffer (device memory)."""
    import torch
    key = id(engine)
    if key not in _state["grad_alias"]:
        dev = torch.device("cuda", torch.cuda.current_device())
        t = torch.as_tensor(_DevPtr(engine.grad_dev_ptr(), engine.n_params + 1), device=dev)
        stream = torch.cuda.ExternalStream(engine.stream_ptr(), device=dev)
        _state["grad_alias"][key] = (t, stream)
    return _state["grad_alias"][key]


def all_reduce_grad(engine):
    """The one collective of the training step: SUM over ranks of [grad | loss], in place, enqueued on
    the engine's HIP stream."""
    import torch
    td = _td()
    t, stream = grad_tensor(engine)
    with torch.cuda.stream(stream):
        td.all_reduce(t, op=td.ReduceOp.SUM)


def all_reduce_host(arr):
    """SUM all-reduce of a host float32 array (gloo path used by the CPU tests)."""
    import torch
    td = _td()
    t = torch.from_numpy(arr)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return arr
