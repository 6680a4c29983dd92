"""Point-wise data helpers on the host side of the path: the reference's normalisers
(nif/data/point_wise_data.py:50-114, NumPy) and a closed-form generator of the bundled
travelling-wave datasets (nif/demo/dataset/*.npz: u = exp(-1000 s^2) sin(omega s),
s = x - 0.2 - 0.006 t; omega = 4 / 400) used for synthetic benchmark batches."""
import numpy as np


class PointWiseData(object):
    """Container with the reference's column convention [parameters | coordinates | outputs (| weight)]
    (point_wise_data.py:16-48)."""

    def __init__(self, parameter_data, x_data, u_data, sample_weight=None):
        cols = [parameter_data, x_data, u_data] + ([sample_weight] if sample_weight is not None else [])
        self.data_raw = np.hstack(cols)
        self.data = None
        self.sample_weight = None
        self.n_p = parameter_data.shape[-1]
        self.n_x = x_data.shape[-1]
        self.n_o = u_data.shape[-1]

    @property
    def parameter(self):
        return self.data[:, : self.n_p]

    @property
    def x(self):
        return self.data[:, self.n_p: self.n_p + self.n_x]

    @property
    def u(self):
        return self.data[:, self.n_p + self.n_x: self.n_p + self.n_x + self.n_o]

    @staticmethod
    def standard_normalize(raw_data, area_weighted=False):
        """point_wise_data.py:50-78"""
        mean = raw_data.mean(axis=0)
        std = raw_data.std(axis=0)
        if area_weighted:
            mean[-1] = 0.0
            std[-1] = np.mean(raw_data[:, -1])
            nd = (raw_data - mean) / std
            return nd[:, :-1], mean, std, nd[:, -1]
        return (raw_data - mean) / std, mean, std

    @staticmethod
    def minmax_normalize(raw_data, n_para, n_x, n_target, area_weighted=False):
        """point_wise_data.py:80-114"""
        mean = raw_data.mean(axis=0)
        std = raw_data.std(axis=0)
        for i in range(n_para + n_x):
            lo, hi = np.min(raw_data[:, i]), np.max(raw_data[:, i])
            mean[i] = 0.5 * (lo + hi)
            std[i] = 0.5 * (hi - lo)
        for j in range(n_para + n_x, n_para + n_x + n_target):
            std[j] = np.max(np.abs(raw_data[:, j]))
        if area_weighted:
            mean[-1] = 0.0
            std[-1] = np.mean(raw_data[:, -1])
            nd = (raw_data - mean) / std
            return nd[:, :-1], mean, std, nd[:, -1]
        return (raw_data - mean) / std, mean, std


def traveling_wave(t, x, omega=4.0):
    s = x - 0.2 - 0.006 * t
    return np.exp(-1000.0 * s * s) * np.sin(omega * s)


def synthetic_wave_batch(n_points, seed=0, omega=4.0):
    """(inputs [N,2] = normalised (t, x), targets [N,1]) float32: t~U[0,90], x~U[0,1)."""
    rng = np.random.default_rng(seed)
    t = rng.uniform(0.0, 90.0, size=n_points)
    x = rng.uniform(0.0, 1.0, size=n_points)
    raw = np.stack([t, x, traveling_wave(t, x, omega)], axis=1)
    if omega <= 10:
        data, _, _ = PointWiseData.standard_normalize(raw)
    else:
        data, _, _ = PointWiseData.minmax_normalize(raw, 1, 1, 1)
    return np.ascontiguousarray(data[:, :2], dtype=np.float32), np.ascontiguousarray(data[:, 2:3], dtype=np.float32)


# ------------------------------------------------------------------------------------------------------------------
# Sharded point tables for datasets larger than HBM / host memory: the replacement of the reference's TFRecord
# converter and meta-dataset (nif/data/tfr_dataset.py:22-163, README.md:155-178).  Same workflow and method names:
#
#     fh = NPZShardDataset(n_feature=4, n_target=3)
#     fh.create_from_npz(num_pts_per_file, npz_path, npz_key, out_path, prefix)     # one big .npz -> column-major shards
#     meta = fh.get_meta_dataset(out_path, epoch, model=model)                      # (= get_tfr_meta_dataset)
#     for batch_file in meta:
#         model.fit(fh.gen_dataset_from_batch_file(batch_file, batch_size), epochs=1)
#
# A shard file holds ONE 1-D float32 array per column ("input_j", "output_j", "weight": the reference's feature names,
# tfr_dataset.py:64-78).  With `model=` the meta-dataset streams: a loader thread reads file i+1 into a pinned staging
# buffer and enqueues its H2D copy on the context's copy stream while file i trains (two staging slots on the host and
# in HBM; include/nif_hip.h nif_h2d_async / nif_copy_*), which is what tf.data's .prefetch(AUTOTUNE) does for the
# reference (tfr_dataset.py:161).  Rows are shuffled inside a file on the DEVICE by Model.fit (tfr_dataset.py:104-108).
# ------------------------------------------------------------------------------------------------------------------
import glob as _glob
import os as _os
import queue as _queue
import threading as _threading


class Shard(object):
    """one file of the dataset: what iterating the meta-dataset yields (`batch_file` in the reference's loop)"""

    def __init__(self, path, n_rows, n_feature, n_target, area_weight):
        self.path, self.n_rows = path, int(n_rows)
        self.n_feature, self.n_target, self.area_weight = n_feature, n_target, area_weight
        self.slot = None          # staging slot when streamed to a device
        self._on_release = None   # tells the loader thread that the slot's release has been RECORDED on the compute stream
        self.dev = None           # (x, y, w) DeviceArrays of that slot
        self.host = None          # (x [N,nf], y [N,nt], w [N] or None) when loaded without a device

    def load_host(self):
        if self.host is None:
            self.host = _read_shard(self.path, self.n_feature, self.n_target, self.area_weight)
        return self.host


def _read_shard(path, n_feature, n_target, area_weight, out=None):
    """columns of a shard file -> row-major float32 tables (into `out` = (x, y, w) views when given)"""
    with np.load(path) as f:
        n = f["input_0"].shape[0]
        x = np.empty((n, n_feature), np.float32) if out is None else out[0][:n * n_feature].reshape(n, n_feature)
        y = np.empty((n, n_target), np.float32) if out is None else out[1][:n * n_target].reshape(n, n_target)
        for j in range(n_feature):
            x[:, j] = f["input_%d" % j]
        for j in range(n_target):
            y[:, j] = f["output_%d" % j]
        w = None
        if area_weight:
            w = np.empty((n,), np.float32) if out is None else out[2][:n]
            w[:] = f["weight"]
    return x, y, w


class ShardBatches(object):
    """`gen_dataset_from_batch_file(batch_file, batch_size)`: the shuffled mini-batches of one file; Model.fit(x=this)"""

    def __init__(self, shard, batch_size, shuffle=True):
        self.shard, self.batch_size, self.shuffle = shard, int(batch_size), bool(shuffle)
        self.n_rows = shard.n_rows

    def widths(self, model):
        s = model._owner._spec
        if self.shard.n_feature != s.pi_dim + s.si_dim or self.shard.n_target != s.so_dim:
            raise ValueError("dataset has %d features / %d targets, the model takes %d / %d"
                             % (self.shard.n_feature, self.shard.n_target, s.pi_dim + s.si_dim, s.so_dim))
        if getattr(model, "_n_tangents", lambda: 0)() != 0:
            raise NotImplementedError("shard datasets feed the single-output model")
        return [self.shard.n_target]

    def device_tables(self, engine, model):
        sh = self.shard
        if sh.dev is None:                       # not streamed: stage it now
            x, y, w = sh.load_host()
            dx = engine.alloc(x.size); dx.upload(x)
            dy = engine.alloc(y.size); dy.upload(y)
            dw = None
            if w is not None:
                dw = engine.alloc(w.size); dw.upload(w)
            sh.dev, sh._owned = (dx, dy, dw), True
        else:
            engine.copy_acquire(sh.slot)         # the compute stream waits for this slot's H2D copies
        dx, dy, dw = sh.dev
        return dx, [dy], dw

    def release(self, engine):
        sh = self.shard
        if getattr(sh, "_owned", False):
            engine.sync()
            for a in sh.dev:
                if a is not None:
                    a.free()
            sh.dev, sh._owned = None, False
        elif sh.slot is not None and sh._on_release is not None:
            engine.copy_release(sh.slot)
            cb, sh._on_release = sh._on_release, None
            cb()


class MetaDataset(object):
    """iterable over the files of a sharded dataset, `epoch` passes, optional file-order shuffle buffer
    (tfr_dataset.py:150-163); with an engine: double-buffered streaming to the device"""

    def __init__(self, files, rows, epoch, shuffle_buffer_size, n_feature, n_target, area_weight, engine=None, seed=None):
        self.files, self.rows, self.epoch, self.buf = list(files), list(rows), int(epoch), int(shuffle_buffer_size)
        self.n_feature, self.n_target, self.area_weight = n_feature, n_target, area_weight
        self.engine, self.rng = engine, np.random.default_rng(seed)
        self.num_pts_per_file = len(self.files)      # the reference stores len(filenames) under this name (tfr_dataset.py:133)

    def __len__(self):
        return len(self.files) * self.epoch

    def _order(self):
        idx = list(range(len(self.files)))
        if self.buf > 1:            # tf.data shuffle(buffer_size): draw uniformly from a sliding buffer
            out, buf = [], []
            for i in idx:
                buf.append(i)
                if len(buf) >= self.buf:
                    out.append(buf.pop(int(self.rng.integers(len(buf)))))
            while buf:
                out.append(buf.pop(int(self.rng.integers(len(buf)))))
            idx = out
        return idx

    def _sequence(self):
        seq = []
        order = self._order()       # .shuffle() comes before .repeat(epoch) in the reference: reshuffled every pass
        for _ in range(self.epoch):
            seq += order
            order = self._order() if self.buf > 1 else order
        return seq

    def __iter__(self):
        seq = self._sequence()
        mk = lambda i: Shard(self.files[i], self.rows[i], self.n_feature, self.n_target, self.area_weight)
        if self.engine is None:
            for i in seq:
                yield mk(i)
            return
        e = self.engine
        nmax = max(self.rows)
        nf, nt = self.n_feature, self.n_target
        pinned = [tuple(e.alloc_pinned(nmax * k) for k in (nf, nt, 1)) for _ in range(2)]
        dev = [tuple(e.alloc(nmax * k) for k in (nf, nt, 1)) for _ in range(2)]
        e.copy_wait_host(0)         # creates the copy stream and its events before the loader thread uses them
        q = _queue.Queue(maxsize=1)
        stop = _threading.Event()
        # nif_h2d_async orders the copy behind the release event of the slot AS RECORDED AT THAT MOMENT: the loader may
        # only enqueue it once the consumer has recorded the release of the slot's previous file (host-side handshake)
        free = [_threading.Semaphore(1), _threading.Semaphore(1)]

        def loader():
            try:
                for pos, i in enumerate(seq):
                    if stop.is_set():
                        break
                    slot = pos & 1
                    e.copy_wait_host(slot)                       # the previous copy out of this pinned buffer has landed
                    views = tuple(p.array for p in pinned[slot])
                    x, y, w = _read_shard(self.files[i], nf, nt, self.area_weight, out=views)
                    n = x.shape[0]
                    while not free[slot].acquire(timeout=0.05):
                        if stop.is_set():
                            return
                    # from here the device side of the slot is protected by the stream: nif_h2d_async waits for nif_copy_release
                    e.h2d_async(dev[slot][0], pinned[slot][0], n * nf, slot)
                    e.h2d_async(dev[slot][1], pinned[slot][1], n * nt, slot)
                    if self.area_weight:
                        e.h2d_async(dev[slot][2], pinned[slot][2], n, slot)
                    sh = mk(i)
                    sh.slot, sh.dev = slot, (dev[slot][0], dev[slot][1], dev[slot][2] if self.area_weight else None)
                    sh._on_release = free[slot].release
                    q.put(sh)
                q.put(None)
            except BaseException as exc:     # surface loader errors in the consumer
                q.put(exc)

        th = _threading.Thread(target=loader, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield item
                if item._on_release is not None:       # the consumer skipped this file: hand the slot back ourselves
                    e.copy_release(item.slot)
                    cb, item._on_release = item._on_release, None
                    cb()
        finally:
            stop.set()
            while th.is_alive():
                try:
                    q.get(timeout=0.05)
                except _queue.Empty:
                    pass
            e.sync()
            e.copy_wait_host(0); e.copy_wait_host(1)
            for group in pinned + dev:
                for a in group:
                    a.free()


class NPZShardDataset(object):
    """nif/data/tfr_dataset.py:7-20 `TFRDataset(n_feature, n_target, area_weight=False)` on .npz shards"""

    def __init__(self, n_feature, n_target, area_weight=False):
        self.n_feature, self.n_target, self.area_weight = int(n_feature), int(n_target), bool(area_weight)

    def create_from_npz(self, num_pts_per_file, npz_path, npz_key, tfr_path, prefix, seed=None):
        """tfr_dataset.py:22-83: shuffle the rows of one big table, cut it into files of `num_pts_per_file` points, one
        array per column in every file"""
        num_pts_per_file = int(num_pts_per_file)
        data = np.load(npz_path)[npz_key]
        total, ncol = data.shape
        if self.area_weight:
            assert ncol == self.n_feature + self.n_target + 1
        else:
            assert ncol == self.n_feature + self.n_target
        n_files = int(np.ceil(total / num_pts_per_file))
        print("total number of shard files = ", n_files)
        data = data[np.random.default_rng(seed).permutation(total)]     # shuffle before distributing (tfr_dataset.py:49)
        _os.makedirs(tfr_path, exist_ok=True)
        for i in range(n_files):
            part = data[i * num_pts_per_file:(i + 1) * num_pts_per_file]
            cols = {}
            for j in range(self.n_feature):
                cols["input_%d" % j] = np.ascontiguousarray(part[:, j], dtype=np.float32)
            for j in range(self.n_target):
                cols["output_%d" % j] = np.ascontiguousarray(part[:, self.n_feature + j], dtype=np.float32)
            if self.area_weight:
                cols["weight"] = np.ascontiguousarray(part[:, -1], dtype=np.float32)
            np.savez(_os.path.join(tfr_path, "%s_%d.npz" % (prefix, i)), **cols)
        return n_files

    def get_meta_dataset(self, tfr_path, epoch, tfr_shuffle_buffer_size=1, model=None, seed=None):
        """tfr_dataset.py:120-163.  `model=` (the compiled nif_amd model that will train on it) turns on streaming."""
        files = sorted(_glob.glob(_os.path.join(tfr_path, "*.npz")))
        if not files:
            raise FileNotFoundError("no *.npz shard under %s" % tfr_path)
        rows = []
        for f in files:
            with np.load(f) as z:
                rows.append(int(z["input_0"].shape[0]))
        self.num_pts_per_file = len(files)
        engine = model._engine if model is not None else None
        return MetaDataset(files, rows, epoch, tfr_shuffle_buffer_size, self.n_feature, self.n_target, self.area_weight,
                           engine=engine, seed=seed)

    get_tfr_meta_dataset = get_meta_dataset

    def gen_dataset_from_batch_file(self, batch_file, batch_size, shuffle=True):
        """tfr_dataset.py:85-118: shuffled mini-batches of one file"""
        if not isinstance(batch_file, Shard):
            raise TypeError("expected an element of the meta-dataset")
        return ShardBatches(batch_file, batch_size, shuffle)


# `nif.data.TFRDataset` (tfr_dataset.py:22) under its reference name: same workflow, `.npz` column shards instead of TFRecords
# (`get_tfr_meta_dataset` is kept as an alias of `get_meta_dataset`)
TFRDataset = NPZShardDataset
