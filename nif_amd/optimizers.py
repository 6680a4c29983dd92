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
