"""`from nif.layers import ...` counterpart (reference nif/layers/__init__.py:3-12) for the names that sit ON the built hot path.

The reference's layer classes are Keras graph nodes; here a "layer" that wraps a model is an object with the same constructor
arguments whose call / training behaviour is carried by the HIP engine:

* `JacobianLayer`, `HessianLayer` (gradient.py:4-49, :130-180): forward-mode tangent kernels, see nif_amd/model.py.
* `JacRegLatentLayer(model, y_index, x_index, l1)` (gradient.py:52-127): what `NIF.build()` wraps the model in when
  cfg_parameter_net['jac_reg'] is set -- the model with `l1 * mean((d latent / d parameter)^2)` added to its loss.
* `ParameterOutputL1ActReg(model, l1)` (regularization.py:4-32): the model with the UN-normalised `l1 * ||pnet_output||_1` added
  to its loss (tf.norm(po, ord=1) over the whole batch tensor: no division by the batch size, unlike Keras' activity regularisers).

The ParameterNet / ShapeNet building blocks of that file (SIREN, SIREN_ResNet, HyperLinearForSIREN, MLP_ResNet, MLP_SimpleShortCut,
EinsumLayer, BiasAddLayer, Dense) are not separate objects here: they are the layers of the fused kernels, selected by the same cfg
dictionaries (nif_amd/spec.py); asking for one of them by name raises with that explanation."""
from .model import HessianLayer, JacobianLayer, Model  # noqa: F401

__all__ = ["JacobianLayer", "HessianLayer", "JacRegLatentLayer", "ParameterOutputL1ActReg"]

_FUSED = ("SIREN", "SIREN_ResNet", "HyperLinearForSIREN", "MLP_ResNet", "MLP_SimpleShortCut", "EinsumLayer", "BiasAddLayer", "Dense")


def __getattr__(name):
    if name in _FUSED:
        raise AttributeError("nif_amd.layers.%s: the reference's Keras layer has no standalone counterpart -- it is a stage of the "
                             "fused HIP kernels, configured through cfg_shape_net / cfg_parameter_net (DESIGN.md 4)" % name)
    raise AttributeError(name)


def _full_model(model, who):
    if not isinstance(model, Model) or model._role != "full":
        raise TypeError("%s expects the model returned by NIF(...).build() / .model()" % who)
    return model


def JacRegLatentLayer(model, y_index, x_index, l1=1.0, **kwargs):
    """gradient.py:52-127.  The built form is the one `build()` uses (model.py:353-375): y_index = every latent component,
    x_index = every ParameterNet input.  Returns the trainable model that carries the term."""
    m = _full_model(model, "JacRegLatentLayer")
    s = m._owner._spec
    yi = [y_index] if isinstance(y_index, int) else list(y_index)
    xi = [x_index] if isinstance(x_index, int) else list(x_index)
    if yi != list(range(s.pi_hidden)) or xi != list(range(s.pi_dim)):
        raise NotImplementedError("JacRegLatentLayer is built for y_index = range(latent_dim), x_index = range(pi_dim) "
                                  "(what NIF.build() passes, model.py:353-375)")
    out = Model(m._owner, "full", jac_reg=float(l1))
    out.optimizer, out.loss = m.optimizer, m.loss
    return out


def ParameterOutputL1ActReg(model, l1=0.1, **kwargs):
    """regularization.py:4-32: loss += l1 * sum |pnet_output| over the batch tensor.  Returns the trainable model with the term."""
    m = _full_model(model, "ParameterOutputL1ActReg")
    out = Model(m._owner, "full", jac_reg=m._jac_reg)
    out._po_l1 = float(l1)
    out.optimizer, out.loss = m.optimizer, m.loss
    return out
