"""nif_amd -- MI355X (gfx950) native training step for Neural Implicit Flow.

Drop-in for the point-wise training path of pswpswpsw/nif (`from nif import NIF` ->
`from nif_amd import NIF`): same constructor cfg dicts, build/compile/fit/predict and the
sub-model extractors, on hand-written HIP kernels behind a C-ABI (include/nif_hip.h)."""
from .model import (NIF, NIFMultiScale, NIFMultiScaleLastLayerParameterized, Model, JacobianLayer,  # noqa: F401
                    HessianLayer, SobolevModel, set_seed)
from . import optimizers, callbacks, distributed, data, layers, demo  # noqa: F401
from .optimizers import Adam  # noqa: F401
from ._lib import NifError  # noqa: F401

__all__ = ["NIF", "NIFMultiScale", "NIFMultiScaleLastLayerParameterized", "Model", "JacobianLayer", "HessianLayer", "SobolevModel", "Adam", "NifError", "set_seed",
           "optimizers", "callbacks", "distributed", "data", "layers", "demo"]
