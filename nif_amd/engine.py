"""Engine: one libnif_hip context + the host-side conveniences around it (weights as a list of
NumPy arrays in Keras order, device-resident datasets, the train step)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class DeviceArray(object):
    """A float32 device buffer owned by an Engine (hipMalloc through the C-ABI)."""

    def __init__(self, engine, n_floats):
        self.engine = engine
        self.n = int(n_floats)
        p = C.c_void_p()
        check(engine.lib.nif_dev_alloc(engine.ctx, self.n * 4, C.byref(p)))
        self.ptr = p.value

    def at(self, float_offset):
        return C.c_void_p(self.ptr + 4 * int(float_offset))

    def upload(self, host, float_offset=0):
        host = _f32(host)
        assert host.size + float_offset <= self.n
        check(self.engine.lib.nif_h2d(self.engine.ctx, self.at(float_offset), ptr(host), host.size * 4))

    def download(self, n_floats=None, float_offset=0):
        n = self.n - float_offset if n_floats is None else int(n_floats)
        out = np.empty((n,), dtype=np.float32)
        check(self.engine.lib.nif_d2h(self.engine.ctx, ptr(out), self.at(float_offset), n * 4))
        return out

    def free(self):
        if self.ptr:
            self.engine.lib.nif_dev_free(self.engine.ctx, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray(object):
    """page-locked host staging buffer (hipHostMalloc through the C-ABI), exposed as a float32 NumPy view"""

    def __init__(self, engine, n_floats):
        self.engine = engine
        self.n = int(n_floats)
        p = C.c_void_p()
        check(engine.lib.nif_host_alloc(engine.ctx, self.n * 4, C.byref(p)))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_float * max(self.n, 1)).from_address(self.ptr))[:self.n]

    def free(self):
        if self.ptr:
            self.array = None
            self.engine.lib.nif_host_free(self.engine.ctx, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine(object):
    def __init__(self, spec, device_id=0):
        self.spec = spec
        self.lib = _lib.load()
        if self.lib.nif_device_count() <= 0:
            raise _lib.NifError("no HIP device visible: nif_amd runs on MI355X (gfx950) only, there is no CPU path")
        cfg = spec.to_cfg()
        ctx = C.c_void_p()
        check(self.lib.nif_create(C.byref(cfg), int(device_id), C.byref(ctx)))
        self.ctx = ctx
        n = C.c_int64()
        check(self.lib.nif_param_count(self.ctx, C.byref(n)))
        self.n_params = int(n.value)
        assert self.n_params == spec.n_params(), (self.n_params, spec.n_params())
        self.shapes = spec.param_shapes()

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.nif_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc(self, n_floats):
        """float32 device buffer on this context's GPU"""
        return DeviceArray(self, n_floats)

    def alloc_pinned(self, n_floats):
        return PinnedArray(self, n_floats)

    def gather_rows(self, src, d_perm, n, ncol, dst):
        """dst[i][:] = src[perm[i]][:] on the device (perm: int32 bit patterns in a DeviceArray)"""
        check(self.lib.nif_gather_rows_dev(self.ctx, src.at(0), d_perm.at(0), int(n), int(ncol), dst.at(0)))

    # shard streaming (include/nif_hip.h: nif_h2d_async / nif_copy_*)
    def h2d_async(self, dst, pinned, n_floats, slot):
        check(self.lib.nif_h2d_async(self.ctx, dst.at(0), C.c_void_p(pinned.ptr), int(n_floats) * 4, int(slot)))

    def copy_acquire(self, slot):
        check(self.lib.nif_copy_acquire(self.ctx, int(slot)))

    def copy_release(self, slot):
        check(self.lib.nif_copy_release(self.ctx, int(slot)))

    def copy_wait_host(self, slot):
        check(self.lib.nif_copy_wait_host(self.ctx, int(slot)))

    # ---- parameters -----------------------------------------------------------------------------
    def layout(self):
        n = C.c_int32(0)
        check(self.lib.nif_param_layout(self.ctx, None, C.byref(n)))
        descs = (_lib.nif_tensor_desc * n.value)()
        check(self.lib.nif_param_layout(self.ctx, descs, C.byref(n)))
        return [(d.name.decode(), int(d.offset), int(d.rows), int(d.cols)) for d in descs]

    def set_weights(self, weights):
        if len(weights) != len(self.shapes):
            raise ValueError("You called `set_weights(weights)` with a weight list of length %d, but the model "
                             "was expecting %d weights." % (len(weights), len(self.shapes)))
        for w, (nm, s) in zip(weights, self.shapes):
            if tuple(np.shape(w)) != tuple(s):
                raise ValueError("Layer weight shape %s not compatible with provided weight shape %s (%s)"
                                 % (tuple(s), tuple(np.shape(w)), nm))
        flat = np.concatenate([_f32(w).ravel() for w in weights])
        self.set_flat(flat)

    def set_flat(self, flat):
        flat = _f32(flat)
        check(self.lib.nif_set_params(self.ctx, ptr(flat), flat.size))

    def get_flat(self):
        out = np.empty((self.n_params,), dtype=np.float32)
        check(self.lib.nif_get_params(self.ctx, ptr(out), out.size))
        return out

    def get_weights(self):
        flat = self.get_flat()
        out, off = [], 0
        for _, s in self.shapes:
            k = int(np.prod(s))
            out.append(flat[off:off + k].reshape(s).copy())
            off += k
        return out

    def get_opt_state(self):
        m = np.empty((self.n_params,), dtype=np.float32)
        v = np.empty((self.n_params,), dtype=np.float32)
        step = C.c_int64()
        check(self.lib.nif_get_opt_state(self.ctx, ptr(m), ptr(v), m.size, C.byref(step)))
        return m, v, int(step.value)

    def set_opt_state(self, m, v, step):
        m, v = _f32(m), _f32(v)
        check(self.lib.nif_set_opt_state(self.ctx, ptr(m), ptr(v), m.size, int(step)))

    # ---- inference ------------------------------------------------------------------------------
    @staticmethod
    def _rows(a, ncol, what, allow_extra=False):
        """float32 C-contiguous [B, ncol] view of a caller array.  The C side copies B*ncol floats unconditionally, so
        the column count is checked here.  allow_extra: full-model inputs may carry more columns, the model reads the
        first pi+si (reference model.py:142-143 slices `inputs[:, :pi+si]`)."""
        x = np.asarray(a, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] < ncol or (x.shape[1] != ncol and not allow_extra):
            raise ValueError("%s: expected shape (batch, %d), got %s" % (what, ncol, x.shape))
        if x.shape[1] != ncol:
            x = x[:, :ncol]
        return np.ascontiguousarray(x)

    def _inputs(self, a):
        return self._rows(a, self.spec.pi_dim + self.spec.si_dim, "inputs", allow_extra=True)

    def forward(self, inputs):
        x = self._inputs(inputs)
        s = self.spec
        out = np.empty((x.shape[0], s.so_dim), dtype=np.float32)
        if x.shape[0]:
            check(self.lib.nif_forward(self.ctx, ptr(x), x.shape[0], ptr(out)))
        return out

    def p_to_lr(self, p):
        p = self._rows(p, self.spec.pi_dim, "parameter inputs")
        out = np.empty((p.shape[0], self.spec.pi_hidden), dtype=np.float32)
        if p.shape[0]:
            check(self.lib.nif_pnet_latent(self.ctx, ptr(p), p.shape[0], ptr(out)))
        return out

    def jacobian(self, inputs, y_index, x_index):
        y_index, x_index = list(y_index), list(x_index)
        if len(set(y_index)) != len(y_index) or len(set(x_index)) != len(x_index):
            # repeated entries (gradient.py:207-231 would simply repeat rows / columns): computed once, gathered
            uy, ux = sorted(set(y_index)), sorted(set(x_index))
            yv, d = self.jacobian(inputs, uy, ux)
            return yv, np.ascontiguousarray(d[:, [uy.index(i) for i in y_index]][:, :, [ux.index(j) for j in x_index]])
        x = self._inputs(inputs)
        s = self.spec
        yi = np.ascontiguousarray(list(y_index), dtype=np.int32)
        xi = np.ascontiguousarray(list(x_index), dtype=np.int32)
        y = np.empty((x.shape[0], s.so_dim), dtype=np.float32)
        d = np.empty((x.shape[0], yi.size, xi.size), dtype=np.float32)
        check(self.lib.nif_jacobian(self.ctx, ptr(x), x.shape[0], yi.ctypes.data_as(C.POINTER(C.c_int32)), yi.size,
                                    xi.ctypes.data_as(C.POINTER(C.c_int32)), xi.size, ptr(y), ptr(d)))
        return y, d

    def hessian(self, inputs, y_index, x_index):
        y_index, x_index = list(y_index), list(x_index)
        if len(set(y_index)) != len(y_index) or len(set(x_index)) != len(x_index):
            uy, ux = sorted(set(y_index)), sorted(set(x_index))       # (also lifts the C side's limit of 16 y_index entries)
            yv, d, h = self.hessian(inputs, uy, ux)
            iy, ix = [uy.index(i) for i in y_index], [ux.index(j) for j in x_index]
            return yv, np.ascontiguousarray(d[:, iy][:, :, ix]), np.ascontiguousarray(h[:, iy][:, :, ix][:, :, :, ix])
        x = self._inputs(inputs)
        s = self.spec
        yi = np.ascontiguousarray(list(y_index), dtype=np.int32)
        xi = np.ascontiguousarray(list(x_index), dtype=np.int32)
        y = np.empty((x.shape[0], s.so_dim), dtype=np.float32)
        d = np.empty((x.shape[0], yi.size, xi.size), dtype=np.float32)
        h = np.empty((x.shape[0], yi.size, xi.size, xi.size), dtype=np.float32)
        check(self.lib.nif_hessian(self.ctx, ptr(x), x.shape[0], yi.ctypes.data_as(C.POINTER(C.c_int32)), yi.size,
                                   xi.ctypes.data_as(C.POINTER(C.c_int32)), xi.size, ptr(y), ptr(d), ptr(h)))
        return y, d, h

    def hessian_dev(self, d_x, b, y_index, x_index, d_y, d_dydx, d_d2):
        """HessianLayer on device-resident inputs / outputs (nif_hessian_dev): [b, pi+si] -> y [b, so], dydx [b, ny, nx],
        d2 [b, ny, nx, nx]; asynchronous on the context's stream"""
        yi = np.ascontiguousarray(list(y_index), dtype=np.int32)
        xi = np.ascontiguousarray(list(x_index), dtype=np.int32)
        check(self.lib.nif_hessian_dev(self.ctx, d_x, int(b), yi.ctypes.data_as(C.POINTER(C.c_int32)), yi.size,
                                       xi.ctypes.data_as(C.POINTER(C.c_int32)), xi.size, d_y, d_dydx, d_d2))

    def x_to_phi(self, x):
        x = self._rows(x, self.spec.si_dim, "coordinates")
        s = self.spec
        out = np.empty((x.shape[0], s.so_dim, s.pi_hidden), dtype=np.float32)
        if x.shape[0]:
            check(self.lib.nif_x_to_phi(self.ctx, ptr(x), x.shape[0], ptr(out)))
        return out

    def lr_to_w(self, lr):
        lr = self._rows(lr, self.spec.pi_hidden, "latent")
        out = np.empty((lr.shape[0], self.spec.po_dim), dtype=np.float32)
        if lr.shape[0]:
            check(self.lib.nif_latent_to_w(self.ctx, ptr(lr), lr.shape[0], ptr(out)))
        return out

    def x_to_u_given_w(self, x, w):
        x, w = self._rows(x, self.spec.si_dim, "coordinates"), _f32(w)
        if w.shape != (x.shape[0], self.spec.po_dim):
            raise ValueError("expected w of shape (%d, %d), got %s" % (x.shape[0], self.spec.po_dim, w.shape))
        out = np.empty((x.shape[0], self.spec.so_dim), dtype=np.float32)
        if x.shape[0]:
            check(self.lib.nif_shapenet_given_w(self.ctx, ptr(x), ptr(w), x.shape[0], ptr(out)))
        return out

    # ---- training -------------------------------------------------------------------------------
    def _targets(self, y, n_rows):
        y = _f32(y)
        if y.ndim == 1:
            y = y[:, None]
        if y.shape != (n_rows, self.spec.so_dim):
            raise ValueError("targets: expected shape (%d, %d), got %s" % (n_rows, self.spec.so_dim, y.shape))
        return y

    def _weights(self, sw, n_rows):
        if sw is None:
            return None
        sw = _f32(sw).reshape(-1)
        if sw.shape != (n_rows,):
            raise ValueError("sample_weight: expected shape (%d,), got %s" % (n_rows, sw.shape))
        return sw

    def loss_and_grad(self, inputs, y, sample_weight=None, want_grad=True):
        """(total loss, flat gradient); want_grad=False: (total loss, None) -- only the scalar travels back (evaluation)"""
        x = self._inputs(inputs)
        y = self._targets(y, x.shape[0])
        sw = self._weights(sample_weight, x.shape[0])
        g = np.empty((self.n_params,), dtype=np.float32) if want_grad else None
        loss = C.c_float()
        check(self.lib.nif_loss_and_grad(self.ctx, ptr(x), ptr(y), ptr(sw), x.shape[0], C.byref(loss), ptr(g)))
        return float(loss.value), g

    def train_step(self, inputs, y, sample_weight, adam):
        x = self._inputs(inputs)
        y = self._targets(y, x.shape[0])
        sw = self._weights(sample_weight, x.shape[0])
        loss = C.c_float()
        check(self.lib.nif_train_step(self.ctx, ptr(x), ptr(y), ptr(sw), x.shape[0], C.byref(adam), C.byref(loss)))
        return float(loss.value)

    # device-pointer flavour (datasets resident in HBM; optional cross-rank gradient all-reduce)
    def loss_grad_dev(self, d_x, d_y, d_sw, b_local, b_global):
        check(self.lib.nif_loss_grad_dev(self.ctx, d_x, d_y, d_sw, int(b_local), int(b_global)))

    # Sobolev (two-output model u, du/dx; include/nif_hip.h nif_sobolev_*)
    def sobolev_loss_grad_dev(self, d_x, d_y, d_g, d_sw, b_local, b_global, x_index, w_jac, y_index=None):
        """y_index = None: the derivative term over every output; else over the listed outputs (d_g rows stay [so][nx])"""
        xi = (C.c_int32 * len(x_index))(*[int(i) for i in x_index])
        yi = None if y_index is None else (C.c_int32 * len(y_index))(*[int(i) for i in y_index])
        check(self.lib.nif_sobolev_loss_grad_dev_y(self.ctx, d_x, d_y, d_g, d_sw, int(b_local), int(b_global), xi, len(x_index),
                                                   yi, 0 if y_index is None else len(y_index), float(w_jac)))

    def sobolev_forward(self, inputs, x_index):
        x = self._inputs(inputs)
        B, nx, so = x.shape[0], len(x_index), self.spec.so_dim
        xi = (C.c_int32 * nx)(*[int(i) for i in x_index])
        d_x, d_u, d_j = DeviceArray(self, x.size), DeviceArray(self, B * so), DeviceArray(self, B * so * nx)
        try:
            d_x.upload(x)
            check(self.lib.nif_sobolev_forward_dev(self.ctx, d_x.at(0), B, xi, nx, d_u.at(0), d_j.at(0)))
            u = d_u.download().reshape(B, so)
            j = d_j.download().reshape(B, so, nx)
        finally:
            d_x.free(); d_u.free(); d_j.free()
        return u, j

    def sobolev_loss_and_grad(self, inputs, y, dydx, x_index, w_jac, sample_weight=None, want_grad=True, y_index=None):
        """(total loss incl. the weight regularisers, flat gradient) of the two-output model on host arrays"""
        x = self._inputs(inputs)
        B = x.shape[0]
        y, g = self._targets(y, B), _f32(dydx)
        if g.size != B * self.spec.so_dim * len(x_index):
            raise ValueError("dydx: expected %d x %d x %d values, got shape %s" % (B, self.spec.so_dim, len(x_index), g.shape))
        sample_weight = self._weights(sample_weight, B)
        d_x, d_y, d_g = DeviceArray(self, x.size), DeviceArray(self, y.size), DeviceArray(self, g.size)
        d_sw = DeviceArray(self, B) if sample_weight is not None else None
        try:
            d_x.upload(x); d_y.upload(y); d_g.upload(g)
            if d_sw is not None:
                d_sw.upload(_f32(sample_weight))
            self.sobolev_loss_grad_dev(d_x.at(0), d_y.at(0), d_g.at(0), d_sw.at(0) if d_sw is not None else None, B, B,
                                       x_index, w_jac, y_index)
            if want_grad:
                loss, grad = self.grad_read()           # adds the kernel / bias regularisers once, like nif_loss_and_grad
            else:
                loss, grad = self.grad_read_loss(), None
        finally:
            d_x.free(); d_y.free(); d_g.free()
            if d_sw is not None:
                d_sw.free()
        return loss, grad

    def adam_step_dev(self, adam):
        check(self.lib.nif_adam_step_dev(self.ctx, C.byref(adam)))

    # captured training steps (include/nif_hip.h nif_graph_*)
    def graph_begin(self):
        check(self.lib.nif_graph_begin(self.ctx))

    def graph_end(self):
        gid = C.c_int32(-1)
        check(self.lib.nif_graph_end(self.ctx, C.byref(gid)))
        return int(gid.value)

    def graph_launch(self, gid, adam):
        check(self.lib.nif_graph_launch(self.ctx, int(gid), C.byref(adam)))

    def graph_destroy(self, gid):
        check(self.lib.nif_graph_destroy(self.ctx, int(gid)))

    def zero_grad(self):
        check(self.lib.nif_zero_grad(self.ctx))

    def reserve(self, b_max, n_tangents=0):
        check(self.lib.nif_reserve(self.ctx, int(b_max), int(n_tangents)))

    def allreduce_grad(self):
        check(self.lib.nif_allreduce_grad(self.ctx))

    def set_regularizer(self, l1, l2, lo, hi):
        check(self.lib.nif_set_regularizer(self.ctx, float(l1), float(l2), int(lo), int(hi)))

    def set_shapenet_regularizer(self, l1, l2):
        """last-layer class: cfg_shape_net l1_reg / l2_reg over the shared ShapeNet's kernels and biases (model.py:1028-1039)"""
        check(self.lib.nif_set_shapenet_regularizer(self.ctx, float(l1), float(l2)))

    def set_loss(self, name):
        """compile(loss=name): 'mse' | 'mae' | 'huber' | 'log_cosh' (and Keras' aliases)"""
        check(self.lib.nif_set_loss(self.ctx, int(_lib.LOSS_IDS[name])))

    def set_jac_regularizer(self, l1):
        check(self.lib.nif_set_jac_regularizer(self.ctx, float(l1)))

    def set_activity_regularizer(self, l1, l2):
        check(self.lib.nif_set_activity_regularizer(self.ctx, float(l1), float(l2)))

    def metric_accumulate(self, weight):
        check(self.lib.nif_metric_accumulate(self.ctx, float(weight)))

    def metric_read(self, reset=True):
        s, n = C.c_double(), C.c_double()
        check(self.lib.nif_metric_read(self.ctx, C.byref(s), C.byref(n), 1 if reset else 0))
        return float(s.value), float(n.value)

    def set_option(self, key, value):
        check(self.lib.nif_set_option(self.ctx, key.encode(), int(value)))

    def grad_read(self):
        """(loss, flat gradient) of the last loss_grad_dev, weight regulariser included"""
        g = np.empty((self.n_params,), dtype=np.float32)
        loss = C.c_float()
        check(self.lib.nif_grad_read(self.ctx, C.byref(loss), ptr(g)))
        return float(loss.value), g

    def grad_read_loss(self):
        """total loss of the last loss_grad_dev (weight regulariser included); the gradient stays on the device"""
        loss = C.c_float()
        check(self.lib.nif_grad_read(self.ctx, C.byref(loss), None))
        return float(loss.value)

    def last_loss(self):
        loss = C.c_float()
        check(self.lib.nif_last_loss(self.ctx, C.byref(loss)))
        return float(loss.value)

    def sync(self):
        check(self.lib.nif_sync(self.ctx))

    def grad_dev_ptr(self):
        return self.lib.nif_grad_dev(self.ctx)

    def stream_ptr(self):
        return self.lib.nif_stream(self.ctx)

    # ---- measurement ----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        check(self.lib.nif_profile_enable(self.ctx, 1 if on else 0))

    def profile_read(self, reset=True):
        """{group: (total_ms, launches)} measured with HIP events on the context's stream."""
        n = len(_lib.PROF_NAMES)
        ms = (C.c_float * n)()
        cnt = (C.c_int64 * n)()
        check(self.lib.nif_profile_read(self.ctx, ms, cnt, n, 1 if reset else 0))
        return {nm: (float(ms[i]), int(cnt[i])) for i, nm in enumerate(_lib.PROF_NAMES)}

    def timer_start(self):
        check(self.lib.nif_timer_start(self.ctx))

    def timer_stop(self):
        ms = C.c_float()
        check(self.lib.nif_timer_stop(self.ctx, C.byref(ms)))
        return float(ms.value)
