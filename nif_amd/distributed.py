"""Data parallelism over the point batch: one process per GPU, RCCL over xGMI called directly through the C-ABI
(include/nif_hip.h, "multi-GPU" section; nif_amd/csrc/nif_comm.hip).  Replaces the reference's
`tf.distribute.MirroredStrategy().scope()` recipe (reference README.md:39-49): rows (points) are independent, so
every rank computes loss and gradient of its shard, pre-scaled by 1/B_global inside the HIP kernels, and ONE
ncclAllReduce(sum) of the flat float32 buffer [grad(P) | loss] per step makes every rank hold the global-batch
gradient; each rank then applies the identical Adam update (replicated state).  The collective is enqueued on the
context's own HIP stream on the library's own buffer: no host synchronisation, no copy, no tensor framework.

Process model: RANK / WORLD_SIZE / LOCAL_RANK from the environment (what the usual one-process-per-GPU launchers export; `bench.py
--gpus N` launches its own ranks the same way).  The only thing that has to travel between the processes on
the host is RCCL's 128-byte unique id (single node, like the reference's MirroredStrategy): a handshake through files in a
PRIVATE directory (mode 0700, files 0600) named after NIF_RDZV_KEY, else MASTER_ADDR / MASTER_PORT -- what all ranks of one
launch share, whatever wrapper shells sit between the launcher and the ranks.  Rank 0 publishes a per-launch token, every other
rank answers with [token | a random nonce], rank 0 answers with [id | the nonces of THIS launch's hellos]: a reader only accepts an
answer that carries ITS nonce and rank 0 only a hello that carries ITS token, so no file a crashed earlier launch left behind under
the same key can be taken for this launch's; every rank removes what it wrote when it fails, rank 0 everything in a `finally`.  NIF_COMM_TIMEOUT (seconds, default 300) bounds every wait.

The communicator object is pluggable (`install`): the CPU tests put a gloo-backed double with the same methods here to
run `Model.fit`'s real sharding logic on two processes without a GPU."""
import ctypes as C
import os
import tempfile
import time

import numpy as np

from . import _lib
from ._lib import check

_comm = [None]


class RcclComm(object):
    """One rank of the job.  Every Engine (= one nif_ctx, one HIP stream) joins its own RCCL communicator the first
    time it takes part in a collective; all ranks create their engines in the same order (SPMD), so the n-th
    communicator of every rank is the same one."""

    def __init__(self, rank, world, local_rank, key=None, directory=None, timeout=None):
        self.rank, self.world, self.local_rank = int(rank), int(world), int(local_rank)
        key = key or os.environ.get("NIF_RDZV_KEY")
        if not key:
            if self.world > 1 and "MASTER_PORT" not in os.environ:
                raise _lib.NifError("multi-process run without NIF_RDZV_KEY or MASTER_ADDR / MASTER_PORT: the ranks have nothing "
                                    "to find each other by (launch through torch.distributed.run, `bench.py --gpus N`, or export NIF_RDZV_KEY)")
            key = "%s_%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"))
        self._key = key
        base = directory or os.environ.get("NIF_RDZV_DIR") or os.path.join(tempfile.gettempdir(), "nif_rdzv_%d" % os.getuid())
        os.makedirs(base, mode=0o700, exist_ok=True)
        try:
            os.chmod(base, 0o700)
        except OSError:
            pass
        self._dir = base
        self._timeout = float(timeout if timeout is not None else os.environ.get("NIF_COMM_TIMEOUT", "300"))
        self._seq = 0

    # ---- host-side rendezvous of the 128-byte id -------------------------------------------------
    def _path(self, kind, seq, rank=None):
        safe = "".join(ch if ch.isalnum() or ch in "._-" else "_" for ch in self._key)
        return os.path.join(self._dir, "nif_rccl_%s_%s_%d%s" % (kind, safe, seq, "" if rank is None else "_%d" % rank))

    def _id_path(self, seq):
        return self._path("id", seq)

    @staticmethod
    def _publish(path, payload):
        tmp = "%s.%d.tmp" % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(payload)
        os.replace(tmp, path)     # atomic: a reader sees either the old file, nothing, or all of the new one

    @staticmethod
    def _read(path, nbytes):
        try:
            with open(path, "rb") as f:
                raw = f.read()
            return raw if len(raw) == nbytes else None
        except OSError:
            return None

    def _expired(self, t0, what):
        if time.time() - t0 > self._timeout:
            raise _lib.NifError("rank %d: %s after %.0f s (NIF_COMM_TIMEOUT)" % (self.rank, what, self._timeout))
        time.sleep(0.005)

    NONCE = 16

    def _exchange_id(self, lib):
        """-> (id, files_to_remove).  Collective over the ranks of the job.
        Rank 0 first publishes a random per-launch TOKEN (`open` file); every other rank answers with hello = [token | its own nonce]
        and rank 0 keeps re-reading a hello until it carries the CURRENT token -- a hello file a killed earlier launch left behind
        under the same key (torchrun's default port gives every launch the same key) is never taken for this launch's.
        The id file is [id | nonces]: a reader only accepts an answer that carries ITS nonce.  Every rank removes what it wrote
        when it fails; rank 0 removes everything in attach()'s finally."""
        seq = self._seq
        self._seq += 1
        idp, openp = self._path("id", seq), self._path("open", seq)
        nb = _lib.COMM_ID_BYTES + (self.world - 1) * self.NONCE
        if self.rank == 0:
            mine = [idp, openp] + [self._path("hello", seq, r) for r in range(1, self.world)]
            try:
                for q in (idp, openp):
                    try:
                        os.remove(q)    # whatever a crashed launch left under this key
                    except OSError:
                        pass
                token = os.urandom(self.NONCE)
                self._publish(openp, token)
                nonces = []
                t0 = time.time()
                for r in range(1, self.world):
                    hp = self._path("hello", seq, r)
                    while True:
                        n = self._read(hp, 2 * self.NONCE)
                        if n is not None and n[:self.NONCE] == token:
                            nonces.append(n[self.NONCE:])
                            break
                        self._expired(t0, "no hello of this launch from rank %d at %s" % (r, hp))
                buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
                check(lib.nif_comm_unique_id(buf))
                self._publish(idp, buf.raw + b"".join(nonces))
                return buf.raw, mine
            except BaseException:
                self._remove(mine)
                raise
        nonce = os.urandom(self.NONCE)
        hp = self._path("hello", seq, self.rank)
        lo = _lib.COMM_ID_BYTES + (self.rank - 1) * self.NONCE
        t0 = time.time()
        said = None
        try:
            while True:
                token = self._read(openp, self.NONCE)
                if token is not None and token != said:      # rank 0 of THIS launch is (now) listening: (re-)introduce ourselves
                    self._publish(hp, token + nonce)
                    said = token
                raw = self._read(idp, nb)
                if raw is not None and raw[lo:lo + self.NONCE] == nonce:
                    return raw[:_lib.COMM_ID_BYTES], []
                self._expired(t0, "no RCCL id from rank 0 at %s" % idp)
        except BaseException:
            self._remove([hp])
            raise

    @staticmethod
    def _remove(paths):
        for q in paths:
            try:
                os.remove(q)
            except OSError:
                pass

    def attach(self, engine):
        """Join `engine`'s context to a fresh communicator of all ranks (collective)."""
        if getattr(engine, "_comm_joined", False):
            return
        if self.world > 1 or os.environ.get("NIF_FORCE_RCCL") == "1":
            raw, mine = self._exchange_id(engine.lib)
            try:
                check(engine.lib.nif_comm_init_rank(engine.ctx, raw, self.rank, self.world))
            finally:                  # ncclCommInitRank returned (every rank has read the id) or failed: nothing stays behind
                self._remove(mine)
        engine._comm_joined = True

    # ---- collectives ---------------------------------------------------------------------------
    def all_reduce_grad(self, engine):
        """THE collective of the training step: SUM over ranks of [grad | loss], in place, on the engine's stream."""
        self.attach(engine)
        check(engine.lib.nif_allreduce_grad(engine.ctx))

    def selftest(self, engine):
        """rank + 1 through the training collective's own buffer and stream; raises unless every rank reads N (N + 1) / 2.
        Returns the number of ranks the sum accounts for."""
        self.attach(engine)
        n = C.c_int32(0)
        check(engine.lib.nif_comm_selftest(engine.ctx, C.byref(n)))
        return int(n.value)

    def zero_grad(self, engine):
        check(engine.lib.nif_zero_grad(engine.ctx))

    def _reduce_host(self, engine, arr, dtype, op):
        self.attach(engine)
        p = C.c_void_p()
        check(engine.lib.nif_dev_alloc(engine.ctx, arr.nbytes, C.byref(p)))
        try:
            check(engine.lib.nif_h2d(engine.ctx, p, _lib.ptr(arr), arr.nbytes))
            check(engine.lib.nif_comm_allreduce(engine.ctx, p, arr.size, dtype, op))
            check(engine.lib.nif_d2h(engine.ctx, _lib.ptr(arr), p, arr.nbytes))
        finally:
            engine.lib.nif_dev_free(engine.ctx, p)
        return arr

    def all_reduce_ints(self, engine, values, op="sum"):
        """Element-wise SUM (or MAX) over ranks of a short list of integers: ONE collective, one host read-back."""
        a = np.ascontiguousarray([int(v) for v in values], dtype=np.int64)
        self._reduce_host(engine, a, _lib.DT_I64, _lib.OP_MAX if op == "max" else _lib.OP_SUM)
        return [int(v) for v in a]

    def all_reduce_float(self, engine, value, op="max"):
        a = np.ascontiguousarray([float(value)], dtype=np.float64)
        self._reduce_host(engine, a, _lib.DT_F64, {"max": _lib.OP_MAX, "min": _lib.OP_MIN, "sum": _lib.OP_SUM}[op])
        return float(a[0])

    def barrier(self, engine):
        """all ranks are here and the engine's stream has drained"""
        self.attach(engine)
        check(engine.lib.nif_comm_barrier(engine.ctx))

    def shutdown(self):
        pass


def install(comm):
    """Make `comm` (an object with RcclComm's methods) the process's communicator; None = single process."""
    _comm[0] = comm
    return comm


def get():
    return _comm[0]


def init(rank=None, world=None, local_rank=None):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK (the launcher's environment).  Returns (rank, world)."""
    if _comm[0] is None:
        rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank))) if local_rank is None else int(local_rank)
        if not 0 <= rank < world:
            raise ValueError("RANK=%d outside WORLD_SIZE=%d" % (rank, world))
        install(RcclComm(rank, world, local_rank))
    return _comm[0].rank, _comm[0].world


def is_initialized():
    return _comm[0] is not None


def world_size():
    return _comm[0].world if _comm[0] is not None else 1


def rank():
    return _comm[0].rank if _comm[0] is not None else 0


def local_device():
    """HIP device of this process: LOCAL_RANK of the job, else 0."""
    return _comm[0].local_rank if _comm[0] is not None else 0


def cpus_of_numa_node(node, sysfs="/sys/devices/system/node"):
    """CPU ids of a NUMA node from its sysfs cpulist ("0-31,128-159"); [] when unknown"""
    try:
        txt = open(os.path.join(sysfs, "node%d" % node, "cpulist")).read().strip()
    except OSError:
        return []
    cpus = []
    for part in txt.split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_device_numa(device_id=None, lib=None, sysfs_pci="/sys/bus/pci/devices"):
    """Pin this process to the cores of the NUMA node its GPU hangs off (one process per GPU: host-side launches and the staging
    copies then stay on the socket next to the device).  Returns the node, or None when it cannot be determined (no GPU, no
    NUMA information, a container without sysfs): nothing is changed then."""
    try:
        lib = lib or _lib.load()
        dev = local_device() if device_id is None else int(device_id)
        buf = C.create_string_buffer(64)
        if lib.nif_device_pci_bus_id(dev, buf, 64) != 0:
            return None
        bdf = buf.value.decode().strip().lower()
        node = int(open(os.path.join(sysfs_pci, bdf, "numa_node")).read().strip())
        if node < 0:
            return None
        allowed = os.sched_getaffinity(0)
        cpus = [c for c in cpus_of_numa_node(node) if c in allowed]
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def shutdown():
    if _comm[0] is not None:
        _comm[0].shutdown()
    _comm[0] = None


def shard_bounds(n, world, r):
    """Contiguous, balanced split of n rows over `world` ranks: rank r gets [lo, hi)."""
    base, rem = divmod(n, world)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def plan_steps(n_local, batch_size, comm=None, engine=None):
    """Per-step (local, global) batch sizes of one epoch of `Model.fit` over a shard of n_local rows.  Every rank
    walks its own shard in batches of `batch_size` (last one partial); the global size of every step's batch is
    agreed ONCE per fit call (two small collectives, no host sync per step); a rank whose shard is a batch shorter
    than another's gets a trailing 0 and joins that step's all-reduce with a zero gradient."""
    sizes = [min(batch_size, n_local - b0) for b0 in range(0, n_local, batch_size)]
    if comm is None or comm.world == 1:
        return sizes, list(sizes)
    nb = comm.all_reduce_ints(engine, [len(sizes)], op="max")[0]
    sizes = sizes + [0] * (nb - len(sizes))
    return sizes, comm.all_reduce_ints(engine, sizes, op="sum")
