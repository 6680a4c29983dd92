"""Test doubles for the host logic of the N>1 path on a box without a GPU.

* `OracleEngine` offers the slice of `nif_amd.engine.Engine` that `Model.fit` drives (alloc / loss_grad_dev /
  zero_grad / adam_step_dev / metric_* / reserve ...) with the NumPy oracle as the per-shard compute and the same
  buffer conventions as libnif_hip.so: one flat float buffer [grad(P) | loss], both pre-scaled by 1/B_global; the
  weight-regulariser term is added once per Adam step and also when the gradient came from zero_grad (the behaviour
  ADVICE r1 asked for).
* `GlooComm` offers `nif_amd.distributed.RcclComm`'s methods over a torch.distributed gloo group.

Test infrastructure only: nothing under nif_amd/ imports this."""
import numpy as np

from oracle import nif_oracle as O


class _HostArray(object):
    def __init__(self, n):
        self.buf = np.zeros((int(n),), dtype=np.float32)

    def at(self, off):
        return (self.buf, int(off))

    def upload(self, host, float_offset=0):
        h = np.ascontiguousarray(host, dtype=np.float32).ravel()
        self.buf[float_offset:float_offset + h.size] = h

    def free(self):
        pass


class OracleEngine(object):
    def __init__(self, spec_oracle, weights, reg=(0.0, 0.0, 0, 0)):
        self.o = spec_oracle
        self.theta = O.flatten([np.asarray(w, dtype=np.float64) for w in weights])
        self.n_params = self.theta.size
        self.grad_buf = np.zeros((self.n_params + 1,), dtype=np.float64)
        self.m = np.zeros_like(self.theta); self.v = np.zeros_like(self.theta); self.t = 0
        self.reg = reg
        self.reg_applied = False
        self._metric = [0.0, 0.0]
        self.calls = []
        self._slot_free = [True, True]

    def forward(self, inputs):
        x = np.asarray(inputs, dtype=np.float64)
        return O.forward(self.o, O.unflatten(self.o, self.theta), x[:, :self.o.pi + self.o.si]).astype(np.float32)

    # ---- what Model.fit uses ------------------------------------------------------------------------
    def alloc(self, n):
        return _HostArray(n)

    def alloc_pinned(self, n):
        a = _HostArray(n)
        a.array = a.buf
        return a

    def gather_rows(self, src, d_perm, n, ncol, dst):
        perm = d_perm.buf[:n].view(np.int32)
        dst.buf[:n * ncol] = src.buf[:src.buf.size // ncol * ncol].reshape(-1, ncol)[perm].ravel()
        self.calls.append(("gather", int(n), int(ncol)))

    # shard streaming: the double copies at once and records the protocol
    def h2d_async(self, dst, pinned, n_floats, slot):
        assert self._slot_free[slot], "H2D into a slot whose steps were not released"
        dst.buf[:n_floats] = pinned.buf[:n_floats]
        self.calls.append(("h2d", int(slot), int(n_floats)))

    def copy_acquire(self, slot):
        self._slot_free[slot] = False
        self.calls.append(("acquire", int(slot)))

    def copy_release(self, slot):
        self._slot_free[slot] = True
        self.calls.append(("release", int(slot)))

    def copy_wait_host(self, slot):
        pass

    def reserve(self, b_max, n_tangents=0):
        assert b_max >= 1, "nif_reserve rejects B_max <= 0"
        self.calls.append(("reserve", int(b_max)))

    def set_jac_regularizer(self, l1):
        assert l1 == 0.0, "the double has no latent Jacobian regulariser"

    def set_opt_state(self, m, v, step):
        self.m = np.asarray(m, dtype=np.float64).copy(); self.v = np.asarray(v, dtype=np.float64).copy(); self.t = int(step)

    def loss_grad_dev(self, d_x, d_y, d_sw, b, bg):
        ncol = self.o.pi + self.o.si
        (xb, xo), (yb, yo) = d_x, d_y
        x = xb[xo:xo + b * ncol].reshape(b, ncol).astype(np.float64)
        y = yb[yo:yo + b * self.o.so].reshape(b, self.o.so).astype(np.float64)
        sw = None
        if d_sw is not None:
            sw = d_sw[0][d_sw[1]:d_sw[1] + b].astype(np.float64)
        loss, grads = O.loss_and_grad(self.o, O.unflatten(self.o, self.theta), x, y, sw, batch_global=bg)
        self.grad_buf[:-1] = O.flatten(grads); self.grad_buf[-1] = loss
        self.reg_applied = False
        self.calls.append(("loss_grad", int(b), int(bg)))

    def loss_and_grad(self, inputs, y, sample_weight=None, want_grad=True):
        """nif_loss_and_grad: host arrays in, (loss incl. the weight-regulariser term, flat gradient) out"""
        x = np.asarray(inputs, dtype=np.float64)[:, :self.o.pi + self.o.si]
        sw = None if sample_weight is None else np.asarray(sample_weight, dtype=np.float64)
        loss, grads = O.loss_and_grad(self.o, O.unflatten(self.o, self.theta), x, np.asarray(y, dtype=np.float64), sw)
        g = O.flatten(grads)
        l1, l2, lo, hi = self.reg
        if l1 or l2:
            w = self.theta[lo:hi]
            g[lo:hi] += 2.0 * l2 * w + l1 * np.sign(w)
            loss += l2 * np.sum(w * w) + l1 * np.sum(np.abs(w))
        return float(loss), g

    def zero_grad(self):
        self.grad_buf[:] = 0.0
        self.reg_applied = False
        self.calls.append(("zero_grad",))

    def adam_step_dev(self, adam):
        l1, l2, lo, hi = self.reg
        if (l1 or l2) and not self.reg_applied:
            w = self.theta[lo:hi]
            self.grad_buf[lo:hi] += 2.0 * l2 * w + l1 * np.sign(w)
            self.grad_buf[-1] += l2 * np.sum(w * w) + l1 * np.sum(np.abs(w))
        self.t += 1
        self.theta, self.m, self.v = O.adam_step(self.theta, self.grad_buf[:-1], self.m, self.v, self.t, lr=adam.lr,
                                                 b1=adam.beta1, b2=adam.beta2, eps=adam.eps)
        self.reg_applied = False

    def metric_accumulate(self, weight):
        self._metric[0] += weight * self.grad_buf[-1]; self._metric[1] += weight

    def metric_read(self, reset=True):
        s, n = self._metric
        if reset:
            self._metric = [0.0, 0.0]
        return s, n

    def sync(self):
        pass

    def last_loss(self):
        return float(self.grad_buf[-1])


class GlooComm(object):
    """RcclComm's interface on a gloo process group (CPU)."""

    def __init__(self):
        import torch.distributed as td
        self.td = td
        self.rank, self.world = td.get_rank(), td.get_world_size()
        self.local_rank = self.rank
        self.n_grad_reduces = 0

    def attach(self, engine):
        pass

    def all_reduce_grad(self, engine):
        import torch
        t = torch.from_numpy(engine.grad_buf)
        self.td.all_reduce(t, op=self.td.ReduceOp.SUM)
        self.n_grad_reduces += 1

    def zero_grad(self, engine):
        engine.zero_grad()

    def selftest(self, engine):
        """RcclComm.selftest on the double: rank + 1 through all_reduce_grad's buffer"""
        engine.grad_buf[:] = self.rank + 1
        self.all_reduce_grad(engine)
        self.n_grad_reduces -= 1
        s = float(engine.grad_buf[0])
        ok = np.all(engine.grad_buf == 0.5 * self.world * (self.world + 1))
        engine.grad_buf[:] = 0.0
        if not ok:
            raise RuntimeError("all-reduce self-check failed: %r" % s)
        return int(round(0.5 * (np.sqrt(8.0 * s + 1.0) - 1.0)))

    def all_reduce_ints(self, engine, values, op="sum"):
        import torch
        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX if op == "max" else self.td.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    def all_reduce_float(self, engine, value, op="max"):
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.td.all_reduce(t, op={"max": self.td.ReduceOp.MAX, "min": self.td.ReduceOp.MIN, "sum": self.td.ReduceOp.SUM}[op])
        return float(t.item())

    def barrier(self, engine):
        self.td.barrier()

    def shutdown(self):
        # without this a rank now and then aborts at interpreter exit ("terminate called without an active exception": gloo's
        # worker threads are still joinable) and the launcher reports the run as failed
        if self.td.is_initialized():
            self.td.destroy_process_group()


def bench_double(points):
    """NIF_BENCH_ENGINE=tests.doubles:bench_double -- what `bench.py` steps when the launcher / JSON contract is tested on a box
    without a GPU: the benchmark model on the oracle engine, a gloo communicator when the launcher's environment names more than
    one rank.  NIF_BENCH_DOUBLE_FAIL_RANK=r makes that rank die before it joins anything (the fast-failure test)."""
    import os
    import sys
    import bench
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("NIF_BENCH_DOUBLE_FAIL_RANK") == str(rank):
        sys.stderr.write("double: rank %d fails on purpose\n" % rank)
        sys.exit(3)
    comm = None
    if world > 1:
        import torch.distributed as td
        td.init_process_group("gloo")
        comm = GlooComm()
    spec = O.Spec("NIFMultiScale", bench.CFG_SHAPE, bench.CFG_PARAM)
    ws = O.init_weights(spec, np.random.default_rng(1))
    return OracleEngine(spec, ws), comm, (lambda engine, n: _HostArray(n))
