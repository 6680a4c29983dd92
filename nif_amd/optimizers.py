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


class TFPLBFGS(object):
    """Second-stage fine-tuner of the reference (nif/optimizers/lbfgs.py:98-126, README.md:51-69):
    `TFPLBFGS(model, loss_fun, inps, outs, display_epoch).minimize(rounds, max_iter)` runs full-batch L-BFGS on the
    flat parameter vector.  The loss+gradient closure (lbfgs.py:66-74) is one `nif_loss_and_grad` call (HIP); the
    two-loop recursion and the line search run on the host in NumPy (10 correction pairs like
    tfp.optimizer.lbfgs_minimize; backtracking Armijo search with a curvature check)."""

    def __init__(self, model, loss_fun, inps, outs, display_epoch=1, sample_weight=None, history=10):
        import numpy as np
        self._np = np
        self.model = model
        self.inps = np.ascontiguousarray(inps, dtype=np.float32)
        outs = np.ascontiguousarray(outs, dtype=np.float32)
        self.outs = outs[:, None] if outs.ndim == 1 else outs
        self.sw = None if sample_weight is None else np.ascontiguousarray(sample_weight, dtype=np.float32)
        self.display_epoch = display_epoch
        self.history = []
        self.m = int(history)

    def _f(self, theta):
        e = self.model._engine
        e.set_flat(theta.astype(self._np.float32))
        loss, g = e.loss_and_grad(self.inps, self.outs, self.sw)
        return float(loss), g.astype(self._np.float64)

    def minimize(self, rounds=50, max_iter=50, verbose=False):
        np = self._np
        e = self.model._engine
        x = e.get_flat().astype(np.float64)
        f, g = self._f(x)
        S, Y = [], []
        it = 0
        for rnd in range(rounds):
            for _ in range(max_iter):
                q = g.copy()
                alphas = []
                for s, y in zip(reversed(S), reversed(Y)):
                    a = s.dot(q) / y.dot(s)
                    alphas.append(a)
                    q -= a * y
                if S:
                    q *= S[-1].dot(Y[-1]) / Y[-1].dot(Y[-1])
                for (s, y), a in zip(zip(S, Y), reversed(alphas)):
                    b = y.dot(q) / y.dot(s)
                    q += (a - b) * s
                d = -q
                gd = g.dot(d)
                if gd >= 0:           # not a descent direction: restart from steepest descent
                    S, Y, d = [], [], -g
                    gd = -g.dot(g)
                t = 1.0 if S else min(1.0, 1.0 / max(np.sqrt(g.dot(g)), 1e-12))
                for _ls in range(20):
                    fn, gn = self._f(x + t * d)
                    if np.isfinite(fn) and fn <= f + 1e-4 * t * gd:
                        break
                    t *= 0.5
                else:
                    break
                s, yv = t * d, gn - g
                x, f, g = x + s, fn, gn
                if s.dot(yv) > 1e-10 * np.sqrt(s.dot(s) * yv.dot(yv)):
                    S.append(s); Y.append(yv)
                    if len(S) > self.m:
                        S.pop(0); Y.pop(0)
                it += 1
                self.history.append(f)
                if np.sqrt(g.dot(g)) < 1e-10:
                    break
            if verbose and (rnd % max(self.display_epoch, 1) == 0):
                print("round %d  iter %d  loss %.6e" % (rnd, it, f))
        e.set_flat(x.astype(np.float32))
        return self.history
