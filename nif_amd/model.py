"""Drop-in Python surface of reference nif/model.py for the point-wise training path:
NIF / NIFMultiScale / NIFMultiScaleLastLayerParameterized with build/model/compile/fit/predict and
the sub-model extractors.  Same constructor arguments, attribute names and error behaviour as the
reference (file:line cited per method); every number comes from libnif_hip.so (HIP, gfx950)."""
import contextlib
import json
import os
import time

import numpy as np

from . import _lib
from . import distributed as dist
from .data import ShardBatches
from .engine import Engine
from .optimizers import Adam, get as get_optimizer
from .spec import Spec

_global_rng = [np.random.default_rng()]


def set_seed(seed):
    """Seed for the initial weights of models constructed afterwards (tf.random.set_seed analogue)."""
    _global_rng[0] = np.random.default_rng(seed)


class History(object):
    def __init__(self):
        self.history = {}
        self.epoch = []


class Variable(object):
    """What `model.trainable_variables` lists: name, shape, numpy()."""

    def __init__(self, model, index, name, shape):
        self._model, self._index, self.name, self.shape = model, index, name, tuple(shape)

    def numpy(self):
        return self._model.get_weights()[self._index]


class Model(object):
    """The subset of tf.keras.Model the reference's README uses (README.md:23-37, :71-117, :179-195)."""

    def __init__(self, owner, role, n_inputs=1, jac_reg=0.0):
        self._owner = owner
        self._role = role  # 'full' | 'p_to_lr' | 'p_to_w' | 'lr_to_w' | 'x_to_u_given_w' | 'x_to_phi'
        # l1 of the latent Jacobian regulariser: only the model build() returns carries it (the reference wraps JacRegLatentLayer
        # in build() alone, model.py:353-375; .model() and the sub-models train / evaluate without it).  Pushed to the shared
        # engine whenever THIS model computes a loss.
        self._jac_reg = float(jac_reg or 0.0)
        self._po_l1 = 0.0      # layers.ParameterOutputL1ActReg: l1 * sum |pnet_output| (un-normalised, regularization.py:28-30)
        self.optimizer = None
        self.loss = None
        self.stop_training = False
        self.history = None
        self._n_inputs = n_inputs


    def _push_losses(self, e, bg):
        """the loss terms THIS model adds on top of its owner's configuration, for a step / evaluation over bg rows"""
        e.set_jac_regularizer(self._jac_reg)
        if hasattr(e, "set_loss"):
            e.set_loss(self.loss or "mse")          # compile(loss=...): per model, the engine is shared
        if self._po_l1:
            a1, a2 = getattr(self._owner, "_act_reg", (0.0, 0.0))
            if a2:
                raise NotImplementedError("ParameterOutputL1ActReg on a model configured with act_l2_reg")
            e.set_activity_regularizer(a1 + self._po_l1 * float(bg), 0.0)     # the engine's term is c / B * sum |out|

    def _pop_losses(self, e):
        """back to the owner's configuration: models that share the engine (.model(), the sub-models, SobolevModel, the L-BFGS
        closures) must not inherit this model's terms"""
        e.set_jac_regularizer(0.0)
        if hasattr(e, "set_loss"):
            e.set_loss("mse")
        if self._po_l1:
            e.set_activity_regularizer(*getattr(self._owner, "_act_reg", (0.0, 0.0)))

    @contextlib.contextmanager
    def _plain_loss(self, e, loss="mse"):
        """the bare `loss(model(x), y)` of the reference's L-BFGS closures (lbfgs.py:66-68, lbfgs_V2.py:63-66 evaluate the loss
        FUNCTION on the model's output: `model.losses` -- kernel / bias / activity regularisers, JacRegLatentLayer's add_loss --
        is never added there): every regularisation term of the engine is switched off for the evaluation and restored after"""
        o = self._owner
        reg, act, sreg = getattr(o, "_reg", (0.0, 0.0)), getattr(o, "_act_reg", (0.0, 0.0)), getattr(o, "_sreg", (0.0, 0.0))
        n_pnet = o._n_pnet_params() if reg != (0.0, 0.0) else 0
        e.set_jac_regularizer(0.0)
        if hasattr(e, "set_loss"):
            e.set_loss(loss)
        if reg != (0.0, 0.0):
            e.set_regularizer(0.0, 0.0, 0, n_pnet)
        if act != (0.0, 0.0):
            e.set_activity_regularizer(0.0, 0.0)
        if sreg != (0.0, 0.0):
            e.set_shapenet_regularizer(0.0, 0.0)
        try:
            yield
        finally:
            if hasattr(e, "set_loss"):
                e.set_loss("mse")
            if reg != (0.0, 0.0):
                e.set_regularizer(reg[0], reg[1], 0, n_pnet)
            if act != (0.0, 0.0):
                e.set_activity_regularizer(*act)
            if sreg != (0.0, 0.0):
                e.set_shapenet_regularizer(*sreg)

    # ---- weights ---------------------------------------------------------------------------------
    @property
    def _engine(self):
        return self._owner._engine

    def get_weights(self):
        return self._engine.get_weights()

    def set_weights(self, weights):
        self._engine.set_weights(weights)

    @property
    def trainable_variables(self):
        return [Variable(self, i, nm, s) for i, (nm, s) in enumerate(self._engine.shapes)]

    weights = trainable_variables

    def count_params(self):
        return self._engine.n_params

    def summary(self, print_fn=print):
        print_fn('Model: "%s" (%s)' % (self._owner.__class__.__name__, self._role))
        for nm, s in self._engine.shapes:
            print_fn("  %-24s %-16s %d" % (nm, str(tuple(s)), int(np.prod(s))))
        print_fn("Total params: %d" % self._engine.n_params)

    def save_weights(self, filepath):
        """Flat parameters in Keras variable order + Adam slots, as .npz (the reference's TF-checkpoint
        format, README.md:179-195, is a TensorFlow artefact and out of scope)."""
        ws = self.get_weights()
        m, v, step = self._engine.get_opt_state()
        arrs = {"w%03d" % i: w for i, w in enumerate(ws)}
        np.savez(filepath if str(filepath).endswith(".npz") else str(filepath) + ".npz",
                 names=np.array([nm for nm, _ in self._engine.shapes]), adam_m=m, adam_v=v,
                 adam_step=np.int64(step), **arrs)

    def load_weights(self, filepath):
        f = filepath if str(filepath).endswith(".npz") else str(filepath) + ".npz"
        d = np.load(f)
        n = len(self._engine.shapes)
        self.set_weights([d["w%03d" % i] for i in range(n)])
        if "adam_m" in d:
            self._engine.set_opt_state(d["adam_m"], d["adam_v"], int(d["adam_step"]))
            self._fresh_slots = False

    # ---- inference -------------------------------------------------------------------------------
    def _run(self, x):
        e = self._engine
        if self._role == "full":
            return e.forward(x)
        if self._role == "p_to_lr":
            return e.p_to_lr(x)
        if self._role == "p_to_w":
            if self._owner._spec.connectivity == "last_layer":   # pnet output IS the last-layer weight vector (model.py:583-585)
                return e.p_to_lr(x)
            return e.lr_to_w(e.p_to_lr(x))
        if self._role == "lr_to_w":
            return e.lr_to_w(x)
        if self._role == "x_to_u_given_w":
            if not (isinstance(x, (list, tuple)) and len(x) == 2):
                raise ValueError("model_x_to_u_given_w expects [x, w]")
            return e.x_to_u_given_w(x[0], x[1])
        if self._role == "x_to_phi":
            return e.x_to_phi(x)
        raise NotImplementedError(self._role)

    _PREDICT_CHUNK = 1 << 21

    def predict(self, x, batch_size=None, verbose=0, **kwargs):
        """Keras Model.predict: the result does not depend on batch_size; rows are staged through the device in chunks
        of max(batch_size, 2^21) so that an input table larger than the staging buffers never has to fit at once."""
        chunk = max(int(batch_size or 0), self._PREDICT_CHUNK)
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        n = int(np.shape(xs[0])[0])
        if n <= chunk:
            return self._run(x)
        outs = []
        for lo in range(0, n, chunk):
            part = [np.asarray(a)[lo:lo + chunk] for a in xs]
            outs.append(self._run(part if isinstance(x, (list, tuple)) else part[0]))
        if isinstance(outs[0], (list, tuple)):
            return [np.concatenate([o[i] for o in outs], axis=0) for i in range(len(outs[0]))]
        return np.concatenate(outs, axis=0)

    def __call__(self, x, training=False):
        return self._run(x)

    # ---- training --------------------------------------------------------------------------------
    # Keras' compile(metrics=[...]): names / metric objects -> (history key, kind).  Unweighted (Keras applies sample_weight to
    # `weighted_metrics` only); the value of an epoch is the running mean over every sample the epoch saw, evaluated with the weights
    # each batch's loss was evaluated with (before its update).  Computed on the host from a forward pass per batch: metrics are off
    # the hot path and cost nothing unless asked for
    _METRIC_KINDS = {"mse": "mse", "mean_squared_error": "mse", "MSE": "mse", "mae": "mae", "mean_absolute_error": "mae", "MAE": "mae",
                     "root_mean_squared_error": "rmse", "rmse": "rmse"}

    def _parse_metrics(self, metrics):
        out = []
        for mtr in ([metrics] if isinstance(metrics, str) else list(metrics or [])):
            key = mtr if isinstance(mtr, str) else (getattr(mtr, "name", None) or getattr(mtr, "__name__", None))
            if key not in self._METRIC_KINDS:
                raise NotImplementedError("compile(metrics=[%r]): built are 'mse' / 'mean_squared_error', 'mae' / 'mean_absolute_error' and "
                                          "RootMeanSquaredError (by name or Keras object)" % (mtr,))
            out.append((key, self._METRIC_KINDS[key]))
        return out

    @staticmethod
    def _metric_sums(kinds, u, y):
        """per-batch sums whose ratio to the sample count is the metric (rmse: the root is taken at the end)"""
        d = np.asarray(u, dtype=np.float64) - np.asarray(y, dtype=np.float64)
        return [float(np.abs(d).mean(axis=1).sum()) if k == "mae" else float((d * d).mean(axis=1).sum()) for _, k in kinds]

    @staticmethod
    def _metric_values(kinds, sums, count):
        return {key: (np.sqrt(s_ / max(count, 1)) if k == "rmse" else s_ / max(count, 1)) for (key, k), s_ in zip(kinds, sums)}

    def compile(self, optimizer="adam", loss="mse", metrics=None, **kwargs):
        if self._role != "full":
            raise ValueError("only the full model can be compiled for training")
        name = loss if isinstance(loss, str) else (getattr(loss, "name", None) or getattr(loss, "__name__", None))
        if not isinstance(loss, str):      # a loss OBJECT: only its defaults are built (huber: delta = 1 in the kernels; Keras' default reduction)
            if float(getattr(loss, "delta", 1.0)) != 1.0:
                raise NotImplementedError("loss=%r: huber with delta != 1" % (loss,))
            red = getattr(loss, "reduction", None)
            if red is not None and str(red).lower().rsplit(".", 1)[-1] not in ("auto", "sum_over_batch_size"):
                raise NotImplementedError("loss=%r: reduction %r (built: the default SUM_OVER_BATCH_SIZE)" % (loss, red))
        if name not in _lib.LOSS_IDS:
            raise NotImplementedError("loss=%r: built are 'mse' (README.md:33), 'mae', 'huber' (delta 1), 'log_cosh'" % (loss,))
        if kwargs:
            raise NotImplementedError("compile(%s): not on the built hot path" % ", ".join(sorted(kwargs)))
        self._metrics = self._parse_metrics(metrics)
        if self._metrics and self._n_tangents() != 0:
            raise NotImplementedError("compile(metrics=...) on the two-output model")
        new = get_optimizer(optimizer)
        if new is not self.optimizer:
            self._fresh_slots = True     # Keras: a newly compiled optimizer starts with zero slots and iteration 0
        self.optimizer = new
        self.loss = ("mse", "mae", "huber", "log_cosh")[_lib.LOSS_IDS[name]]

    _EVAL_CHUNK = 1 << 18
    # fit(): epochs of at least four batches no larger than _GRAPH_MAX_BATCH can be captured into a hipGraph (NIF_GRAPH=1 or
    # model._graph_epochs = True).  OFF by default: measured on configs[0] (10 k points, batch 512, tools/exp/small_batch.py) the
    # replayed epoch runs 83.3 us per step against 84.8 us eager -- the step is the sum of its 11 kernels' EXECUTION times (fixed
    # prologue / epilogue costs of persistent kernels built for 1e6 points: DESIGN 8.6), not host launch cost or dispatch gaps
    _GRAPH_MAX_BATCH = 16384
    _graph_epochs = os.environ.get("NIF_GRAPH", "0") == "1"

    def evaluate(self, x, y, sample_weight=None, verbose=0, **kwargs):
        """Keras Model.evaluate: the TOTAL loss -- sample-weighted mse plus every regularisation loss of the model (kernel / bias
        L1 / L2, the ParameterNet activity regulariser, the latent Jacobian regulariser of build()), as `fit` logs it; evaluated
        through the engine's loss kernels in chunks, each chunk's loss weighted by its rows (Keras' batch-weighted mean); only
        the scalar of a chunk travels back.  The sub-models (model_p_to_lr() ...) are never compiled -- in Keras their evaluate()
        raises, and so does this one."""
        if self._role != "full":
            raise RuntimeError("You must compile your model before training/testing. Use `model.compile(optimizer, loss)`. "
                               "(only the full model can be compiled; %s is an inference-only view of its variables)" % self._role)
        e = self._engine
        x = np.asarray(x); n = x.shape[0]
        if n == 0:
            return 0.0
        targets = self._targets(y, n)
        tot = 0.0
        try:
            for lo in range(0, n, self._EVAL_CHUNK):
                hi = min(n, lo + self._EVAL_CHUNK)
                sw = None if sample_weight is None else np.asarray(sample_weight)[lo:hi]
                self._push_losses(e, hi - lo)
                tot += (hi - lo) * self._loss_host(e, x[lo:hi], [t[lo:hi] for t in targets], sw)
        finally:
            self._pop_losses(e)
        kinds = getattr(self, "_metrics", [])
        if kinds:      # Keras: [loss, metric, ...] when metrics were compiled
            sums = self._metric_sums(kinds, self.predict(x), targets[0])
            vals = self._metric_values(kinds, sums, n)
            return [float(tot / n)] + [float(vals[k]) for k, _ in kinds]
        return float(tot / n)

    # hooks the two-output Sobolev model overrides
    def _targets(self, y, n_rows):
        y = np.ascontiguousarray(y, dtype=np.float32)
        if y.ndim == 1:
            y = y[:, None]
        if y.shape != (n_rows, self._owner._spec.so_dim):
            raise ValueError("y must have shape (N, so_dim=%d), got %r" % (self._owner._spec.so_dim, y.shape))
        return [y]

    def _loss_grad_dev(self, e, d_x, d_targets, d_sw, b, bg):
        e.loss_grad_dev(d_x, d_targets[0], d_sw, b, bg)

    def _loss_host(self, e, x, targets, sw):
        """total loss of one evaluation chunk (host arrays)"""
        return e.loss_and_grad(x, targets[0], sw, want_grad=False)[0]

    def _n_tangents(self):
        return 0

    def _scaled_weights(self, sw, n_rows):
        """sample weights as the engine sees them (hook of the two-output model: an overall loss weight rides on them)"""
        return sw

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None, shuffle=True,
            sample_weight=None, initial_epoch=0, validation_data=None, validation_split=0.0, steps_per_epoch=None, **kwargs):
        """Keras Model.fit semantics for in-memory arrays (or one file of a shard dataset, nif_amd.data): per epoch
        optionally shuffle, walk batches of `batch_size` (default 32, last one partial), one Adam step per batch; the
        epoch 'loss' is the sample-weighted mean of the batch losses.  The table is made resident in HBM once; a
        shuffled epoch uploads only its permutation and gathers on the device; batches are device-pointer slices.  Under `nif_amd.distributed` every rank walks its own
        shard and the flat gradient is SUM-all-reduced (RCCL) before the identical Adam update
        (tf.distribute.MirroredStrategy, README.md:39-49).
        validation_data = (x_val, y_val[, sample_weight_val]): Keras' per-epoch evaluation, logged as 'val_loss' (seen by the
        callbacks and History; evaluated through predict on this rank's device)."""
        if validation_data is not None and not (isinstance(validation_data, (tuple, list)) and len(validation_data) in (2, 3)):
            raise ValueError("validation_data = (x_val, y_val) or (x_val, y_val, sample_weight_val)")
        if validation_split and validation_data is None:
            # Keras: "the validation data is selected from the last samples in the x and y data provided, before shuffling"
            if not 0.0 < float(validation_split) < 1.0:
                raise ValueError("`validation_split` must be between 0 and 1, received: %r" % (validation_split,))
            if isinstance(x, ShardBatches):
                raise ValueError("`validation_split` is only supported for arrays")
            n_all = int(np.shape(x)[0])
            at = int(np.floor(n_all * (1.0 - float(validation_split))))
            if at == 0 or at == n_all:
                raise ValueError("Training data contains %d samples, which is not sufficient to split it into a validation and "
                                 "training set as specified by `validation_split=%r`" % (n_all, validation_split))
            cut = lambda a, lo, hi: [np.asarray(t)[lo:hi] for t in a] if isinstance(a, (list, tuple)) else np.asarray(a)[lo:hi]
            val = (cut(x, at, n_all), cut(y, at, n_all)) + (() if sample_weight is None else (np.asarray(sample_weight)[at:],))
            return self.fit(cut(x, 0, at), cut(y, 0, at), batch_size=batch_size, epochs=epochs, verbose=verbose, callbacks=callbacks,
                            shuffle=shuffle, sample_weight=None if sample_weight is None else np.asarray(sample_weight)[:at],
                            initial_epoch=initial_epoch, validation_data=val, steps_per_epoch=steps_per_epoch, **kwargs)
        if self.optimizer is None:
            raise RuntimeError("You must compile your model before training/testing. Use `model.compile(optimizer, loss)`.")
        if kwargs:
            raise NotImplementedError("fit(%s): not on the built hot path (in-memory x, y, sample_weight only)"
                                      % ", ".join(sorted(kwargs)))
        s = self._owner._spec
        e = self._engine
        kinds = getattr(self, "_metrics", [])
        if kinds and (isinstance(x, ShardBatches) or dist.get() is not None):
            raise NotImplementedError("compile(metrics=...): in-memory arrays on one GPU")
        self._push_losses(e, 1)                   # (raises here, not at first engine access, when the shape has no kernel for it)
        self.stop_training = False
        if getattr(self, "_fresh_slots", False):
            z = np.zeros((e.n_params,), dtype=np.float32)
            e.set_opt_state(z, z, 0)
            self._fresh_slots = False
        ncol = s.pi_dim + s.si_dim
        shard = x if isinstance(x, ShardBatches) else None
        if shard is not None:
            # one file of a sharded dataset (nif_amd.data.NPZShardDataset, the TFRDataset replacement): its columns are
            # already on their way to (or in) HBM through the double-buffered copy stream; fit() trains on the slot
            if y is not None or sample_weight is not None:
                raise ValueError("a shard dataset carries its own targets and weights")
            if batch_size is not None and int(batch_size) != shard.batch_size:
                raise ValueError("the batch size was fixed by gen_dataset_from_batch_file")
            batch_size = shard.batch_size
            shuffle = shard.shuffle
            N = shard.n_rows
            widths = shard.widths(self)
            if self._scaled_weights(None, 1) is not None:
                raise NotImplementedError("loss_weights[0] != 1 with a shard dataset: fold the factor into the shards' sample weights")
            src_x, src_t, src_sw = shard.device_tables(e, self)
            has_sw = src_sw is not None
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            targets = self._targets(y, x.shape[0])   # list of [N, width] tables that travel with x
            if x.shape[1] != ncol:
                x = np.ascontiguousarray(x[:, :ncol])
            N = x.shape[0]
            sw = None if sample_weight is None else np.ascontiguousarray(sample_weight, dtype=np.float32)
            sw = self._scaled_weights(sw, N)
            widths = [t.shape[1] for t in targets]
            has_sw = sw is not None
            # the table is made resident in HBM ONCE; a shuffled epoch uploads its permutation (4 bytes per row) and
            # gathers on the device
            src_x = e.alloc(N * ncol); src_x.upload(x)
            src_t = []
            for t in targets:
                dt = e.alloc(t.size); dt.upload(t); src_t.append(dt)
            src_sw = None
            if has_sw:
                src_sw = e.alloc(N); src_sw.upload(sw)
        bs = 32 if batch_size is None else int(batch_size)
        callbacks = list(callbacks or [])
        hist = History()
        owned = [] if shard is not None else [src_x] + src_t + ([src_sw] if has_sw else [])
        try:
            for cb in callbacks:
                if hasattr(cb, "set_model"):
                    cb.set_model(self)
            for cb in callbacks:
                if hasattr(cb, "on_train_begin"):
                    cb.on_train_begin({})
            # a shuffled epoch gathers the resident table into a second one on the device (4 bytes of permutation per row
            # travel); when HBM has no room for the second copy the epoch is permuted on the host and uploaded instead
            dev_shuffle = bool(shuffle) and N > 0
            if dev_shuffle:
                try:
                    d_x = e.alloc(N * ncol); owned.append(d_x)
                    d_t = []
                    for w in widths:
                        d_t.append(e.alloc(N * w)); owned.append(d_t[-1])
                    d_sw = None
                    if has_sw:
                        d_sw = e.alloc(N); owned.append(d_sw)
                    d_perm = e.alloc(N); owned.append(d_perm)
                except _lib.NifError:
                    if shard is not None:
                        raise
                    for arr in owned[1 + len(src_t) + (1 if has_sw else 0):]:
                        arr.free()
                    del owned[1 + len(src_t) + (1 if has_sw else 0):]
                    dev_shuffle = False
            if not dev_shuffle:
                d_x, d_t, d_sw = src_x, src_t, src_sw
            host_shuffle = bool(shuffle) and N > 0 and not dev_shuffle
            rng = np.random.default_rng(getattr(self, "_shuffle_seed", None))
            comm = dist.get()
            world = comm.world if comm is not None else 1
            # every rank walks its own shard; the global size of every step's batch is agreed ONCE per call, ranks whose
            # shard is a batch shorter (or empty) join the collectives of the steps they lack with a zero gradient
            sizes, gsizes = dist.plan_steps(N, bs, comm, e)
            e.reserve(max(1, max(sizes, default=0)), self._n_tangents())
            use_graph = (self._graph_epochs and world == 1 and shard is None and not self._po_l1 and not kinds and len(sizes) >= 4
                         and max(sizes) <= self._GRAPH_MAX_BATCH and hasattr(e, "graph_begin") and epochs - initial_epoch >= 2)
            graph_id = None
            stream = [-1, 0]          # steps_per_epoch: (pass, next batch of the pass) of the one iterator over `epochs` passes
            cur_perm = [None]         # the permutation of the pass the resident table holds (a pass may span epochs under steps_per_epoch:
                                      # the metric forward of a batch reads the host rows through it -- ADVICE r4)
            if steps_per_epoch is not None:
                if int(steps_per_epoch) < 1:
                    raise ValueError("steps_per_epoch must be a positive integer")
                if world > 1 or shard is not None:
                    raise NotImplementedError("steps_per_epoch: in-memory arrays on one GPU (ranks with uneven shards would disagree on the stream)")
            for epoch in range(initial_epoch, epochs):
                if self.stop_training:
                    break
                for cb in callbacks:
                    if hasattr(cb, "on_epoch_begin"):
                        cb.on_epoch_begin(epoch, {})
                t0 = time.time()

                msum, mcnt = [0.0] * len(kinds), [0]

                def new_pass():      # one pass over the table = one permutation (Keras: the adapter's dataset of `epochs` passes)
                    cur_perm[0] = None
                    if dev_shuffle:
                        perm = rng.permutation(N).astype(np.int32)
                        cur_perm[0] = perm
                        d_perm.upload(perm.view(np.float32))
                        e.gather_rows(src_x, d_perm, N, ncol, d_x)
                        for st_, dt, w in zip(src_t, d_t, widths):
                            e.gather_rows(st_, d_perm, N, w, dt)
                        if has_sw:
                            e.gather_rows(src_sw, d_perm, N, 1, d_sw)
                    elif host_shuffle:
                        perm = rng.permutation(N)
                        cur_perm[0] = perm
                        e.sync()                      # the previous pass's steps have read the table
                        src_x.upload(x[perm])
                        for dt, t in zip(src_t, targets):
                            dt.upload(t[perm])
                        if has_sw:
                            src_sw.upload(sw[perm])
                adam = self.optimizer.as_struct()
                self._push_losses(e, 1)           # (a callback of the previous epoch may have evaluated another model on the shared engine)
                e.metric_read(reset=True)

                def run_batches(lo=0, hi=None):
                    for ib in range(lo, len(sizes) if hi is None else hi):
                        b, bg = sizes[ib], gsizes[ib]
                        b0 = ib * bs
                        if self._po_l1:
                            self._push_losses(e, bg)
                        if kinds and b > 0:      # compile(metrics=...): the batch's predictions with the weights its loss sees
                            rows = slice(b0, b0 + b) if cur_perm[0] is None else cur_perm[0][b0:b0 + b]
                            for i_, v_ in enumerate(self._metric_sums(kinds, e.forward(x[rows]), targets[0][rows])):
                                msum[i_] += v_
                            mcnt[0] += b
                        if b > 0:
                            self._loss_grad_dev(e, d_x.at(b0 * ncol), [dt.at(b0 * w) for dt, w in zip(d_t, widths)],
                                                d_sw.at(b0) if d_sw is not None else None, b, bg)
                        else:
                            e.zero_grad()
                        if world > 1:
                            comm.all_reduce_grad(e)
                        e.adam_step_dev(adam)
                        e.metric_accumulate(bg)     # Keras' loss metric: sample-weighted mean over the batches,
                if steps_per_epoch is not None:
                    # Keras with array inputs and steps_per_epoch (TensorLikeDataAdapter: ONE iterator over `epochs` successive passes,
                    # never recreated): an epoch takes its steps from where the last one stopped, a new permutation whenever a pass is
                    # used up; when the passes run out Keras warns ("Your input ran out of data") and interrupts training
                    left = int(steps_per_epoch)
                    while left > 0 and not self.stop_training:
                        if stream[1] >= len(sizes) or stream[0] < 0:
                            stream[0] += 1; stream[1] = 0
                            if stream[0] >= epochs - initial_epoch:
                                import warnings
                                warnings.warn("Your input ran out of data; interrupting training. Make sure that your dataset can "
                                              "generate at least `steps_per_epoch * epochs` batches.")
                                self.stop_training = True
                                break
                            new_pass()
                        take = min(left, len(sizes) - stream[1])
                        run_batches(stream[1], stream[1] + take)
                        stream[1] += take; left -= take
                else:
                    new_pass()
                # launch-bound epochs (many small batches on one GPU: configs[0]'s batch 512 is 13 kernels of a few microseconds per
                # step) are recorded ONCE into a hipGraph -- the batch pointers into the resident table do not change between
                # epochs -- and replayed with one submission per epoch; Keras runs its steps from one traced graph as well
                if steps_per_epoch is not None:
                    pass
                elif use_graph and graph_id is None:
                    began = False
                    try:
                        e.graph_begin()           # (raises before anything is recorded: nothing to end then -- ADVICE r4)
                        began = True
                        run_batches()
                        graph_id = e.graph_end()
                        began = False
                    except _lib.NifError:
                        use_graph, graph_id = False, None       # (a workspace that had to grow, a call that cannot be captured): plain launches
                    finally:
                        if began:                 # ANY exception inside the capture (a NifError, a KeyboardInterrupt, an error of the
                            try:                  # communicator ...): end the capture (restores the step counter) and drop the graph, so
                                gid = e.graph_end()     # that the stream is usable again and the outer `finally`'s sync cannot mask the
                                if gid is not None:     # original error with "stream is capturing" 
                                    e.graph_destroy(gid)
                            except _lib.NifError:
                                pass
                if steps_per_epoch is not None:
                    pass
                elif use_graph and graph_id is not None:
                    e.graph_launch(graph_id, adam)
                else:
                    run_batches()
                tot, cnt = e.metric_read(reset=True)   # accumulated on the device: one host sync per epoch
                logs = {"loss": tot / max(cnt, 1.0)}
                logs.update({k: float(v) for k, v in self._metric_values(kinds, msum, mcnt[0]).items()})
                if validation_data is not None:
                    ev = self.evaluate(validation_data[0], validation_data[1],
                                       sample_weight=validation_data[2] if len(validation_data) == 3 else None)
                    if isinstance(ev, list):
                        logs["val_loss"] = ev[0]
                        logs.update({"val_" + k: v for (k, _), v in zip(kinds, ev[1:])})
                    else:
                        logs["val_loss"] = ev
                    self._push_losses(e, 1)       # evaluate() ends with _pop_losses: the next epoch trains with THIS model's loss / jac_reg again
                for cb in callbacks:
                    if hasattr(cb, "on_epoch_end"):
                        cb.on_epoch_end(epoch, logs)
                hist.epoch.append(epoch)          # Keras' History runs after the user callbacks: it records what they
                for k, v in logs.items():         # added to `logs` (LearningRateScheduler's 'lr')
                    hist.history.setdefault(k, []).append(v)
                if verbose:
                    print("Epoch %d/%d - %.2fs - loss: %.4e" % (epoch + 1, epochs, time.time() - t0, logs["loss"]))
        finally:
            self._pop_losses(e)
            if graph_id is not None:
                e.graph_destroy(graph_id)
            if shard is not None:
                shard.release(e)          # the slot's device buffers may be refilled once these steps have run
            e.sync()
            for arr in owned:
                arr.free()
        for cb in callbacks:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end({})
        self.history = hist
        return hist


class JacobianLayer(object):
    """reference nif/layers/gradient.py:4-49: `y, dys_dxs = JacobianLayer(model, y_index, x_index)(x)` with
    dys_dxs[a, i, j] = d y[a, y_index[i]] / d x[a, x_index[j]] w.r.t. the model *input vector* (parameter
    columns first, then coordinates).  Built as a forward-mode tangent HIP kernel for coordinate columns of
    the NIF / NIFMultiScale models (`k_jac`); the reference runs len(y_index) extra reverse sweeps."""

    def __init__(self, model, y_index, x_index, **kwargs):
        if not isinstance(model, Model) or model._role != "full":
            raise TypeError("JacobianLayer expects the model returned by NIF(...).build() / .model()")
        self.model = model
        self.y_index = [y_index] if isinstance(y_index, int) else list(y_index)
        self.x_index = [x_index] if isinstance(x_index, int) else list(x_index)

    def call(self, x, **kwargs):
        return self.model._engine.jacobian(x, self.y_index, self.x_index)

    __call__ = call


class HessianLayer(object):
    """reference nif/layers/gradient.py:130-180: `y, dys_dxs, dys2_dxs2 = HessianLayer(model, y_index, x_index)(x)` with
    dys2_dxs2[a, i, j, k] = d^2 y[a, y_index[i]] / d x[a, x_index[j]] d x[a, x_index[k]].  Second-order forward-mode
    tangents in one HIP kernel per coordinate pair (the reference nests two GradientTapes and batch_jacobian, :251-261);
    built for any input columns of all three classes."""

    def __init__(self, model, y_index, x_index, **kwargs):
        if not isinstance(model, Model) or model._role != "full":
            raise TypeError("HessianLayer expects the model returned by NIF(...).build() / .model()")
        self.model = model
        self.y_index = [y_index] if isinstance(y_index, int) else list(y_index)
        self.x_index = [x_index] if isinstance(x_index, int) else list(x_index)

    def call(self, x, **kwargs):
        return self.model._engine.hessian(x, self.y_index, self.x_index)

    __call__ = call


class SobolevModel(Model):
    """The Keras idiom  `tf.keras.Model(inp, JacobianLayer(nif_model, y_index, x_index)(inp))`  compiled with
    loss='mse' and loss_weights=[1, w]: a two-output model (u, du/dx) whose training differentiates through
    the Jacobian (reference nif/layers/gradient.py:36-49, SURVEY 3.4 / BASELINE config 5 "Sobolev training").
    Built for all three classes, any distinct outputs in y_index and any distinct input columns in x_index (coordinates and / or
    ParameterNet inputs, as gradient.py:207-231 allows; more than three columns run as passes over groups of three tangent streams),
    any positive loss_weights[0].
        m = SobolevModel(JacobianLayer(model, y_index, x_index)); m.compile("adam", "mse", loss_weights=[1, .1])
        m.fit(x, [y, dydx], ...);  u, dudx = m.predict(x)"""

    def __init__(self, jac_layer):
        if not isinstance(jac_layer, JacobianLayer):
            raise TypeError("SobolevModel wraps a JacobianLayer")
        base = jac_layer.model
        Model.__init__(self, base._owner, "full")
        so, ncol = base._owner._spec.so_dim, base._owner._spec.pi_dim + base._owner._spec.si_dim
        # any subset / order of outputs and input columns (gradient.py:207-231); a column or output listed twice would make the
        # reference build duplicated Jacobian entries -- not a training set-up, refused
        self.y_index = [int(i) for i in jac_layer.y_index]
        self.x_index = [int(i) for i in jac_layer.x_index]
        if len(set(self.y_index)) != len(self.y_index) or not all(0 <= i < so for i in self.y_index):
            raise ValueError("y_index: distinct outputs in [0, %d)" % so)
        if len(set(self.x_index)) != len(self.x_index) or not all(0 <= i < ncol for i in self.x_index):
            raise ValueError("x_index: distinct input columns in [0, %d)" % ncol)
        self._all_y = self.y_index == list(range(so))
        self.loss_weights = [1.0, 1.0]

    def compile(self, optimizer="adam", loss="mse", loss_weights=None, **kwargs):
        if isinstance(loss, (list, tuple)):
            if len(set(loss)) != 1:
                raise NotImplementedError("both outputs of the two-output model take the same loss")
            loss = loss[0]
        Model.compile(self, optimizer=optimizer, loss=loss, **kwargs)
        if loss_weights is not None:
            lw = [float(v) for v in loss_weights]
            if len(lw) != 2 or lw[0] <= 0.0:
                raise ValueError("loss_weights = [w_u > 0, w_dudx]")
            self.loss_weights = lw

    def _run(self, x):
        u, j = self._engine.sobolev_forward(x, self.x_index)
        return [u, j if self._all_y else np.ascontiguousarray(j[:, self.y_index, :])]

    def _loss_host(self, e, x, targets, sw):
        """Model.evaluate's chunk loss for the two-output model: w0 mse(u) + w1 mse(du/dx) + the regularisation losses -- the
        same total `fit` logs (r3 returned the data term alone, computed on the host)"""
        w0, w1 = self.loss_weights
        return e.sobolev_loss_and_grad(x, targets[0], targets[1], self.x_index, w1 / w0, self._scaled_weights(sw, x.shape[0]),
                                       want_grad=False, y_index=None if self._all_y else self.y_index)[0]

    def _targets(self, y, n_rows):
        if not (isinstance(y, (list, tuple)) and len(y) == 2):
            raise ValueError("the Sobolev model has two outputs: fit(x, [y, dydx])")
        so, nx, ny = self._owner._spec.so_dim, len(self.x_index), len(self.y_index)
        ty = Model._targets(self, y[0], n_rows)[0]
        tj = np.ascontiguousarray(y[1], dtype=np.float32).reshape(n_rows, ny, nx)
        if not self._all_y:       # the engine's table rows are [so][nx]: the listed outputs' rows in place, the others unused (zeros)
            full = np.zeros((n_rows, so, nx), dtype=np.float32)
            full[:, self.y_index, :] = tj
            tj = full
        return [ty, tj.reshape(n_rows, so * nx)]

    def _n_tangents(self):
        return len(self.x_index)

    def _scaled_weights(self, sw, n_rows):
        """Keras total loss = w0 mse(u) + w1 mse(du/dx) (+ regularisers, unscaled).  The kernels compute
        1/B sum_a sw_a (mse_a(u) + wj mse_a(du/dx)): w0 rides on the sample weights (sw_a <- w0 sw_a, a constant column when the
        caller gave none) and wj = w1 / w0 -- the data term and its gradient are scaled, the regularisation losses are not."""
        w0 = self.loss_weights[0]
        if w0 == 1.0:
            return sw
        if sw is None:
            return np.full((n_rows,), w0, dtype=np.float32)
        return np.ascontiguousarray(np.asarray(sw, dtype=np.float32) * np.float32(w0))

    def _loss_grad_dev(self, e, d_x, d_targets, d_sw, b, bg):
        w0, w1 = self.loss_weights
        e.sobolev_loss_grad_dev(d_x, d_targets[0], d_targets[1], d_sw, b, bg, self.x_index, w1 / w0,
                                None if self._all_y else self.y_index)


def _is_number(v):
    return isinstance(v, (float, int))        # the reference's test (model.py:109, :1031): True / False count as numbers there too


def shapenet_regularizer_coefficients(s_l1, s_l2, p_l1, p_l2):
    """(l1, l2) of the last-layer class's ShapeNet regulariser as the reference builds it (model.py:1031-1039)"""
    keras_default = 0.01
    if _is_number(s_l2):
        return (0.0, float(p_l2) if p_l2 is not None else keras_default)
    if _is_number(s_l1):
        return (float(p_l1) if p_l1 is not None else keras_default, 0.0)
    return (0.0, 0.0)


class NIF(object):
    """reference nif/model.py:48 `class NIF(object)`: a factory of Keras-like models that share one set
    of variables."""
    _KIND = "NIF"

    def __init__(self, cfg_shape_net, cfg_parameter_net, mixed_policy="float32"):
        self._spec = Spec(self._KIND, cfg_shape_net, cfg_parameter_net, mixed_policy)
        s = self._spec
        # attribute names of the reference (model.py:83-99)
        self.cfg_shape_net = cfg_shape_net
        self.cfg_parameter_net = cfg_parameter_net
        self.si_dim, self.so_dim, self.n_sx, self.l_sx = s.si_dim, s.so_dim, s.n_sx, s.l_sx
        self.pi_dim, self.pi_hidden, self.n_st, self.l_st = s.pi_dim, s.pi_hidden, s.n_st, s.l_st
        self.p_jac_reg = cfg_parameter_net.get("jac_reg", None)
        self.p_l1_reg = cfg_parameter_net.get("l1_reg", None)
        self.p_l2_reg = cfg_parameter_net.get("l2_reg", None)
        self.p_act_l1_reg = cfg_parameter_net.get("act_l1_reg", None)
        self.p_act_l2_reg = cfg_parameter_net.get("act_l2_reg", None)
        # activity regulariser of the ParameterNet output: L2 wins over L1 (model.py:118-125)
        self._act_reg = (0.0, 0.0)
        if isinstance(self.p_act_l2_reg, (float, int)):
            self._act_reg = (0.0, float(self.p_act_l2_reg))
        elif isinstance(self.p_act_l1_reg, (float, int)):
            self._act_reg = (float(self.p_act_l1_reg), 0.0)
        # kernel/bias regularisers of every ParameterNet layer: L2 wins over L1 (model.py:109-117)
        self._reg = (0.0, 0.0)
        if isinstance(self.p_l2_reg, (float, int)):
            self._reg = (0.0, float(self.p_l2_reg))
        elif isinstance(self.p_l1_reg, (float, int)):
            self._reg = (float(self.p_l1_reg), 0.0)
        # last-layer class only: cfg_shape_net['l2_reg' / 'l1_reg'] (model.py:1028-1039) -- read by the subclass constructor
        self._sreg = (0.0, 0.0)
        self.mixed_policy_name = mixed_policy
        self.variable_Dtype = "float32"
        self.compute_Dtype = {"mixed_bfloat16": "bfloat16", "mixed_float16": "float16"}.get(mixed_policy, "float32")
        self.po_dim = s.po_dim
        self.pnet_list = [nm for nm, _ in s.param_shapes() if nm.startswith("pnet_") and nm.endswith("_w")]
        self.__engine = None
        self._init_weights = s.initial_weights(_global_rng[0])

    @property
    def _engine(self):
        # the HIP context is created on first use (constructing the object needs no GPU, like the
        # reference's constructor needs no data)
        if self.__engine is None:
            self.__engine = Engine(self._spec, device_id=dist.local_device())
            self.__engine.set_weights(self._init_weights)
            if self._reg != (0.0, 0.0):
                self.__engine.set_regularizer(self._reg[0], self._reg[1], 0, self._n_pnet_params())
            if self._act_reg != (0.0, 0.0):
                self.__engine.set_activity_regularizer(*self._act_reg)
            if self._sreg != (0.0, 0.0):
                self.__engine.set_shapenet_regularizer(*self._sreg)
        return self.__engine

    def _n_pnet_params(self):
        return sum(int(np.prod(s)) for nm, s in self._spec.param_shapes() if nm.startswith("pnet_"))

    def call(self, inputs, training=None, mask=None):
        """model.py:130-154 / :510-539 / :1044-1068"""
        return self._engine.forward(inputs)

    def build(self):
        """model.py:345-377: with cfg_parameter_net['jac_reg'] the reference returns the model wrapped in JacRegLatentLayer
        (same outputs, + l1 * mean((d latent / d parameter)^2) in the loss); here the returned Model carries l1 and hands it to
        the engine when it trains or evaluates -- .model() and the sub-models stay without it, as in the reference."""
        l1 = float(self.p_jac_reg) if isinstance(self.p_jac_reg, (float, int)) and not isinstance(self.p_jac_reg, bool) else 0.0
        return Model(self, "full", jac_reg=l1)

    def model(self):
        """model.py:379-389"""
        return Model(self, "full")

    def model_p_to_w(self):
        """model.py:391-404"""
        return Model(self, "p_to_w")

    def model_p_to_lr(self):
        """model.py:406-420"""
        return Model(self, "p_to_lr")

    def model_lr_to_w(self):
        """model.py:422-433"""
        return Model(self, "lr_to_w")

    def model_x_to_u_given_w(self):
        """model.py:435-464 / :956-986"""
        return Model(self, "x_to_u_given_w", n_inputs=2)

    def save_config(self, filename="config.json"):
        """model.py:466-480"""
        config = {
            "cfg_shape_net": self.cfg_shape_net,
            "cfg_parameter_net": self.cfg_parameter_net,
            "mixed_policy": self.mixed_policy_name,
        }
        with open(filename, "w") as write_file:
            json.dump(config, write_file, indent=4)


class NIFMultiScale(NIF):
    """reference nif/model.py:483"""
    _KIND = "NIFMultiScale"


class NIFMultiScaleLastLayerParameterized(NIFMultiScale):
    """reference nif/model.py:989"""
    _KIND = "NIFMultiScaleLastLayerParameterized"

    def __init__(self, cfg_shape_net, cfg_parameter_net, mixed_policy="float32"):
        super(NIFMultiScaleLastLayerParameterized, self).__init__(cfg_shape_net, cfg_parameter_net, mixed_policy)
        # model.py:1028-1039: kernel AND bias regulariser of every layer of the shared ShapeNet (siren.py:266-269, :393-398).
        # WHICH one is chosen by cfg_shape_net's keys (L2 wins over L1); its COEFFICIENT is the reference's quirk: it passes
        # self.p_l2_reg / self.p_l1_reg -- cfg_parameter_net's number -- to regularizers.L2 / L1, and Keras 2.11 turns a None
        # there into its default 0.01 (keras/regularizers.py: `l2 = 0.01 if l2 is None else l2`).  Restated as is.
        self.s_l1_reg = cfg_shape_net.get("l1_reg", None)
        self.s_l2_reg = cfg_shape_net.get("l2_reg", None)
        self._sreg = shapenet_regularizer_coefficients(self.s_l1_reg, self.s_l2_reg, self.p_l1_reg, self.p_l2_reg)

    def model_lr_to_w(self):
        """model.py:1106-1115"""
        raise ValueError("In this class: NIFMultiScaleLastLayerParameterization, `w` is the same as `lr`")

    def model_x_to_phi(self):
        """model.py:1085-1104"""
        return Model(self, "x_to_phi")
