"""Optimizers accepted by Model.compile.  The hot path uses stock Keras-2.11 Adam (README.md:33;
SURVEY a-11); the update itself is the k_adam HIP kernel."""
from . import _lib


class Adam(object):
    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **kwargs):
        if "lr" in kwargs:
            learning_rate = kwargs.pop("lr")
        self.learning_rate = float(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)

    # Keras exposes optimizer.lr / optimizer.learning_rate; LearningRateScheduler sets it
    @property
    def lr(self):
        return self.learning_rate

    @lr.setter
    def lr(self, v):
        self.learning_rate = float(v)

    def as_struct(self):
        return _lib.nif_adam(self.learning_rate, self.beta_1, self.beta_2, self.epsilon)


def get(opt):
    if isinstance(opt, Adam):
        return opt
    if isinstance(opt, str):
        if opt.lower() == "adam":
            return Adam()
        raise NotImplementedError("optimizer %r: only Adam is on the built hot path" % (opt,))
    # duck-typed: anything with learning_rate/beta_1/beta_2/epsilon (e.g. a config object)
    if all(hasattr(opt, a) for a in ("learning_rate", "beta_1", "beta_2", "epsilon")):
        return Adam(float(opt.learning_rate), float(opt.beta_1), float(opt.beta_2), float(opt.epsilon))
    raise NotImplementedError("optimizer %r: only Adam is on the built hot path" % (opt,))


import numpy as np


class LBFGSMinimizer(object):
    """Host side of one `tfp.optimizer.lbfgs_minimize` call on a closure fun(x) -> (loss, gradient) (float64 NumPy):
    two-loop recursion over up to 20 correction pairs and a Hager-Zhang line search with approximate Wolfe
    conditions.  No device code: the closure is where the HIP kernels run."""
    NUM_CORRECTION_PAIRS = 20          # lbfgs.py:111
    MAX_LINE_SEARCH_ITERATIONS = 100   # lbfgs.py:117
    TOLERANCE = 1e-15                  # lbfgs.py:112-114 (gradient sup-norm, x and relative f tolerances)

    def __init__(self, fun):
        self.fun = fun

    # ---- Hager-Zhang line search along d from x: phi(a) = f(x + a d) -------------------------------------------
    def _line_search(self, x, d, f0, g0):
        delta, sigma, eps, theta, gamma, rho = 0.1, 0.9, 1e-6, 0.5, 0.66, 5.0
        dphi0 = float(g0.dot(d))
        flimit = f0 + eps * abs(f0)
        budget = [self.MAX_LINE_SEARCH_ITERATIONS]

        def phi(a):
            budget[0] -= 1
            fv, gv = self.fun(x + a * d)
            return {"a": a, "f": fv, "df": float(gv.dot(d)), "g": gv}

        def ok(p):   # Wolfe, or the approximate Wolfe conditions near the minimum
            if not np.isfinite(p["f"]):
                return False
            wolfe = p["f"] <= f0 + delta * p["a"] * dphi0 and p["df"] >= sigma * dphi0
            approx = p["f"] <= flimit and (2 * delta - 1) * dphi0 >= p["df"] >= sigma * dphi0
            return wolfe or approx

        p0 = {"a": 0.0, "f": f0, "df": dphi0, "g": g0}
        c = phi(1.0)
        while not np.isfinite(c["f"]) and budget[0] > 0:       # step into a non-finite region: shrink
            c = phi(c["a"] * 0.1)
        if ok(c):
            return c
        # bracket [lo, hi]: dphi(lo) < 0, phi(lo) <= flimit, dphi(hi) >= 0
        lo, hi = p0, None
        while budget[0] > 0:
            if c["df"] >= 0:
                hi = c
                break
            if c["f"] > flimit:            # went uphill with a negative slope: the minimum is in (lo, c): bisect
                a_, b_ = lo, c
                while budget[0] > 0:
                    m = phi((1 - theta) * a_["a"] + theta * b_["a"])
                    if ok(m):
                        return m
                    if m["df"] >= 0:
                        lo, hi = a_, m
                        break
                    if m["f"] <= flimit:
                        a_ = m
                    else:
                        b_ = m
                break
            lo = c
            c = phi(rho * c["a"])
            if ok(c):
                return c
        if hi is None:
            return c if np.isfinite(c["f"]) and c["f"] < f0 else None

        def update(a_, b_, m):       # HZ "update": keep a bracket with the sign conditions
            if not (a_["a"] < m["a"] < b_["a"]):
                return a_, b_
            if m["df"] >= 0:
                return a_, m
            if m["f"] <= flimit:
                return m, b_
            aa, bb = a_, m
            while budget[0] > 0:
                t = phi((1 - theta) * aa["a"] + theta * bb["a"])
                if t["df"] >= 0:
                    return aa, t
                if t["f"] <= flimit:
                    aa = t
                else:
                    bb = t
            return aa, bb

        def secant(a_, b_):
            den = b_["df"] - a_["df"]
            return (a_["a"] * b_["df"] - b_["a"] * a_["df"]) / den if den != 0 else 0.5 * (a_["a"] + b_["a"])

        while budget[0] > 0:
            width = hi["a"] - lo["a"]
            cs = secant(lo, hi)
            m = phi(cs) if lo["a"] < cs < hi["a"] else None
            if m is not None and ok(m):
                return m
            A, Bk = update(lo, hi, m) if m is not None else (lo, hi)
            if m is not None and budget[0] > 0:      # secant^2: a second secant step from the side that moved
                c2 = None
                if m is Bk:
                    c2 = secant(hi, Bk)
                elif m is A:
                    c2 = secant(lo, A)
                if c2 is not None and A["a"] < c2 < Bk["a"]:
                    m2 = phi(c2)
                    if ok(m2):
                        return m2
                    A, Bk = update(A, Bk, m2)
            if Bk["a"] - A["a"] > gamma * width and budget[0] > 0:
                m3 = phi(0.5 * (A["a"] + Bk["a"]))
                if ok(m3):
                    return m3
                A, Bk = update(A, Bk, m3)
            lo, hi = A, Bk
            if hi["a"] - lo["a"] <= 1e-16 * max(1.0, hi["a"]):
                break
        best = lo if lo["a"] > 0 and lo["f"] < f0 else None
        return best

    def run(self, x, max_iter):
        """one tfp.optimizer.lbfgs_minimize call: fresh memory, up to max_iter iterations"""
        x, f, _, _ = self.run_resumable(x, max_iter, None)
        return x, f

    def run_resumable(self, x, max_iter, state):
        """up to max_iter more iterations of a run whose correction pairs / last evaluation are carried in `state`
        (lbfgs_minimize(previous_optimizer_results=...)) -> (x, f, state, iterations done)"""
        if state is None:
            f, g = self.fun(x)
            S, Y = [], []
        else:
            f, g, S, Y = state
        done = 0
        for _ in range(max_iter):
            if np.abs(g).max() <= self.TOLERANCE:
                break
            q = g.copy()
            alphas = []
            for s_, y_ in zip(reversed(S), reversed(Y)):
                a = s_.dot(q) / y_.dot(s_)
                alphas.append(a)
                q -= a * y_
            if S:
                q *= S[-1].dot(Y[-1]) / Y[-1].dot(Y[-1])
            for (s_, y_), a in zip(zip(S, Y), reversed(alphas)):
                b = y_.dot(q) / y_.dot(s_)
                q += (a - b) * s_
            d = -q
            if g.dot(d) >= 0:        # not a descent direction (lost curvature): steepest descent, memory dropped
                S, Y, d = [], [], -g
            p = self._line_search(x, d, f, g)
            if p is None:
                break
            s_, y_ = p["a"] * d, p["g"] - g
            fprev = f
            x, f, g = x + s_, p["f"], p["g"]
            done += 1
            if s_.dot(y_) > 0:
                S.append(s_); Y.append(y_)
                if len(S) > self.NUM_CORRECTION_PAIRS:
                    S.pop(0); Y.pop(0)
            if np.abs(s_).max() <= self.TOLERANCE or abs(fprev - f) <= self.TOLERANCE * abs(fprev):
                break
        return x, f, (f, g, S, Y), done


class TFPLBFGS(object):
    """Second-stage fine-tuner of the reference (nif/optimizers/lbfgs.py:98-126, README.md:51-69):
    `TFPLBFGS(model, loss_fun, inps, outs, display_epoch).minimize(rounds, max_iter)` runs full-batch L-BFGS on the
    flat parameter vector.  As in the reference, every round is a FRESH `lbfgs_minimize` started from the model's
    current variables (correction pairs are dropped between rounds, lbfgs.py:106-118) with 20 correction pairs, up to
    `max_iter` iterations and up to 100 line-search evaluations each, and `history` lists the loss of EVERY closure
    evaluation (lbfgs.py:80-88, :123-126).  The loss+gradient closure (lbfgs.py:66-74) is the HIP training-step kernels
    on a dataset made resident in HBM once (`nif_loss_grad_dev` + `nif_grad_read`: per evaluation only the P parameters
    go up and P+1 floats come back).  The two-loop recursion and the Hager-Zhang line search (what
    tfp.optimizer.lbfgs_minimize uses; restated from the published algorithm, CG_DESCENT, Hager & Zhang 2005/2006:
    approximate Wolfe conditions, bracketing by expansion, secant^2 + bisection updates) run on the host in float64."""

    def __init__(self, model, loss_fun, inps, outs, display_epoch=1, sample_weight=None):
        import numpy as np
        self._np = np
        name = loss_fun if isinstance(loss_fun, str) else getattr(loss_fun, "name", None) or getattr(loss_fun, "__name__", None)
        if not isinstance(loss_fun, str) and loss_fun is not None:      # loss OBJECTS: only their defaults are built (as Model.compile)
            if float(getattr(loss_fun, "delta", 1.0)) != 1.0:
                raise NotImplementedError("TFPLBFGS: huber with delta != 1")
            red = getattr(loss_fun, "reduction", None)
            if red is not None and str(red).lower().rsplit(".", 1)[-1] not in ("auto", "sum_over_batch_size"):
                raise NotImplementedError("TFPLBFGS: loss reduction %r (built: the default SUM_OVER_BATCH_SIZE)" % (red,))
        from . import _lib
        if name is not None and name not in _lib.LOSS_IDS:
            raise NotImplementedError("TFPLBFGS: built losses are 'mse', 'mae', 'huber', 'log_cosh', got %r" % (loss_fun,))
        self._loss = "mse" if name is None else ("mse", "mae", "huber", "log_cosh")[_lib.LOSS_IDS[name]]
        self.model = model
        e = model._engine
        x = e._inputs(inps)
        self._B = x.shape[0]
        y = e._targets(outs, self._B)
        sw = e._weights(sample_weight, self._B)
        self._d_x, self._d_y = e.alloc(x.size), e.alloc(y.size)
        self._d_x.upload(x); self._d_y.upload(y)
        self._d_sw = None
        if sw is not None:
            self._d_sw = e.alloc(sw.size); self._d_sw.upload(sw)
        e.reserve(self._B, 0)
        self.display_epoch = max(int(display_epoch), 1)
        self._losses = []

    @property
    def history(self):
        """lbfgs.py:123-126"""
        return {"iteration": self._np.arange(1, len(self._losses) + 1), "loss": list(self._losses)}

    def _f(self, theta):
        """lbfgs.py:56-88: assign the parameters, loss and flat gradient; every call is counted and recorded"""
        np = self._np
        e = self.model._engine
        e.set_flat(theta.astype(np.float32))
        # the reference's closure is `loss(model(x), y)` -- the loss FUNCTION alone, model.losses is never added (lbfgs.py:66-68,
        # lbfgs_V2.py:63-66): no weight / activity / latent-Jacobian regulariser in the L-BFGS objective, whatever the last
        # fit() of a model sharing this engine left configured
        with self.model._plain_loss(e, **({} if getattr(self, "_loss", "mse") == "mse" else {"loss": self._loss})):
            e.loss_grad_dev(self._d_x.at(0), self._d_y.at(0), self._d_sw.at(0) if self._d_sw is not None else None, self._B, self._B)
            loss, g = e.grad_read()
        self._losses.append(loss)
        if len(self._losses) % self.display_epoch == 0:
            print("Epoch: %d loss: %.8e" % (len(self._losses), loss))
        return float(loss), g.astype(np.float64)

    def minimize(self, rounds=50, max_iter=50):
        """lbfgs.py:103-121: `rounds` independent lbfgs_minimize calls, each from the model's current variables"""
        np = self._np
        e = self.model._engine
        for _ in range(rounds):
            x0 = e.get_flat().astype(np.float64)
            x, _f = LBFGSMinimizer(self._f).run(x0, max_iter)
            e.set_flat(x.astype(np.float32))      # lbfgs.py:120 assign_new_model_parameters(results.position)
        return self.history


class MSEClosure(object):
    """What `LBFGSOptimizer` takes where the reference takes a Python function evaluated under a GradientTape
    (lbfgs_V2.py:77-85: `loss_closure` = "the model's loss on the training table"): the model, its full-batch table and
    optional sample weights -- resident in HBM once; calling it returns the current loss like the reference's closure does."""

    def __init__(self, model, x, y, sample_weight=None):
        self._t = TFPLBFGS(model, "mse", x, y, display_epoch=1 << 62, sample_weight=sample_weight)
        self.model = model

    def __call__(self):
        e = self.model._engine
        return self._t._f(e.get_flat().astype(self._t._np.float64))[0]


class LBFGSOptimizer(object):
    """nif/optimizers/lbfgs_V2.py:77-112: `opt = LBFGSOptimizer(loss_closure, trainable_variables, steps)`; every
    `opt.minimize()` continues the SAME L-BFGS run for `steps` more iterations (previous_optimizer_results: the correction
    pairs survive between calls, unlike TFPLBFGS); `.epoch` = iterations so far, `.loss` = the objective there.
    `loss_closure` is an `MSEClosure`; `trainable_variables` is accepted for signature parity (it is always the model's
    full variable list, which is what the reference passes)."""

    def __init__(self, loss_closure, trainable_variables=None, steps=1):
        if not isinstance(loss_closure, MSEClosure):
            raise TypeError("LBFGSOptimizer(loss_closure=nif_amd.optimizers.MSEClosure(model, x, y), ...): a Python loss function "
                            "cannot be differentiated here -- the closure names the model and its table, the HIP kernels do the rest")
        self._c = loss_closure
        self.steps = int(steps)
        self._it = 0
        self._loss = None
        self._state = None

    @property
    def epoch(self):
        return self._it

    @property
    def loss(self):
        return self._loss

    def minimize(self):
        import numpy as np
        t = self._c._t
        e = self._c.model._engine
        mz = LBFGSMinimizer(t._f)
        x0 = e.get_flat().astype(np.float64)
        x, f, self._state, done = mz.run_resumable(x0, self.steps, self._state)
        self._it += done
        self._loss = float(f)
        e.set_flat(x.astype(np.float32))           # lbfgs_V2.py:112 assign(results.position)
