"""Configuration parsing: what NIF.__init__ / NIFMultiScale._initialize_pnet derive from the two
cfg dicts (reference nif/model.py:73-128, :156-231, :541-736, :1012-1042), plus the trainable-variable
order Keras would report and the reference's initial-value distributions
(nif/layers/siren.py:6-63, :178-204; TruncatedNormal(0.1) for Dense, model.py:181-182)."""
import math

import numpy as np

from . import _lib

KIND_NAMES = {"NIF": _lib.KIND_NIF, "NIFMultiScale": _lib.KIND_MULTISCALE,
              "NIFMultiScaleLastLayerParameterized": _lib.KIND_LASTLAYER}


class Spec(object):
    def __init__(self, kind, cfg_shape_net, cfg_parameter_net, mixed_policy="float32"):
        if not isinstance(cfg_parameter_net, dict):
            raise TypeError("cfg_parameter_net must be a dictionary")
        if not isinstance(cfg_shape_net, dict):
            raise TypeError("cfg_shape_net must be a dictionary")
        if mixed_policy not in _lib.POLICY_IDS:
            raise NotImplementedError("mixed_policy %r: 'float32', 'mixed_bfloat16' and 'mixed_float16' are built (fp32 variables; Keras' "
                                      "pure 'float16' / 'bfloat16' / 'float64' policies also change the variable dtype)" % (mixed_policy,))
        self.kind = kind
        self.cfg_shape_net = cfg_shape_net
        self.cfg_parameter_net = cfg_parameter_net
        self.mixed_policy = mixed_policy
        cs, cp = cfg_shape_net, cfg_parameter_net
        # model.py:84-91
        self.si_dim = cs["input_dim"]
        self.so_dim = cs["output_dim"]
        self.n_sx = cs["units"]
        self.l_sx = cs["nlayers"]
        self.pi_dim = cp["input_dim"]
        self.pi_hidden = cp["latent_dim"]
        self.n_st = cp["units"]
        self.l_st = cp["nlayers"]
        self.p_activation = cp["activation"]
        if kind == "NIF":
            self.s_activation = cs["activation"]
            if self.s_activation == "sine":
                # keras.activations.get('sine') raises in the reference too (model.py:303)
                raise ValueError("Unknown activation function: sine (class NIF takes Keras activations)")
            self.s_resblock = False
            self.s_omega0 = 1.0
            self.p_siren = False
            self.p_resblock = False
            self.p_omega0 = 1.0
            self.connectivity = "full"
        else:
            assert "use_resblock" in cs.keys(), "`use_resblock` should be in cfg_shape_net"  # model.py:559-566
            assert type(cs["use_resblock"]) == bool, "cfg_shape_net['use_resblock'] must be a bool"
            self.s_activation = "sine"
            self.s_resblock = cs["use_resblock"]
            self.s_omega0 = float(cs["omega_0"])
            self.p_siren = cp["activation"] == "sine"
            self.p_resblock = bool(cp["use_resblock"])
            self.p_omega0 = float(cp["omega_0"]) if self.p_siren else 1.0
            self.connectivity = cs["connectivity"]
            if self.connectivity not in ("full", "last_layer"):
                raise ValueError("cfg_shape_net missing correct `connectivity`")  # model.py:587
        if kind == "NIFMultiScaleLastLayerParameterized":
            assert self.connectivity == "last_layer", \
                "you should assign cfg_shape_net['connectivity'] == 'last_layer'"  # model.py:1024-1026
        for nm in (self.p_activation, self.s_activation):
            if nm not in _lib.ACT_IDS:
                raise ValueError("Unknown activation function: %s" % (nm,))
        self.n_hidden_mats = (2 * self.l_sx) if self.s_resblock else self.l_sx
        n, nh = self.n_sx, self.n_hidden_mats
        if self.connectivity == "full":  # model.py:169-173, :571-582
            self.po_dim = nh * n ** 2 + (self.si_dim + self.so_dim + 1 + nh) * n + self.so_dim
        else:
            self.po_dim = self.pi_hidden  # model.py:585

    def to_cfg(self):
        c = _lib.nif_cfg()
        c.abi_version = _lib.NIF_ABI_VERSION
        c.kind = KIND_NAMES[self.kind]
        c.pi_dim, c.si_dim, c.so_dim = self.pi_dim, self.si_dim, self.so_dim
        c.n_sx, c.l_sx, c.n_st, c.l_st = self.n_sx, self.l_sx, self.n_st, self.l_st
        c.latent_dim = self.pi_hidden
        c.s_act = _lib.ACT_IDS[self.s_activation]
        c.s_resblock = int(self.s_resblock)
        c.s_omega0 = self.s_omega0
        c.p_act = _lib.ACT_IDS[self.p_activation]
        c.p_resblock = int(self.p_resblock)
        c.p_omega0 = self.p_omega0
        c.mixed_policy = _lib.POLICY_IDS[self.mixed_policy]
        return c

    # ---- trainable variables, Keras order ---------------------------------------------------
    def param_shapes(self):
        pi, nst, r, po = self.pi_dim, self.n_st, self.pi_hidden, self.po_dim
        sh = [("pnet_first_w", (pi, nst)), ("pnet_first_b", (nst,))]
        for i in range(self.l_st):
            sh += [("pnet_h%d_w" % i, (nst, nst)), ("pnet_h%d_b" % i, (nst,))]
            if self.p_resblock:
                sh += [("pnet_h%d_w2" % i, (nst, nst)), ("pnet_h%d_b2" % i, (nst,))]
        sh += [("pnet_bottleneck_w", (nst, r)), ("pnet_bottleneck_b", (r,))]
        sh += [("pnet_last_w", (r, po)), ("pnet_last_b", (po,))]
        if self.kind == "NIFMultiScaleLastLayerParameterized":
            si, n, so = self.si_dim, self.n_sx, self.so_dim
            sh += [("snet_first_w", (si, n)), ("snet_first_b", (n,))]
            for i in range(self.l_sx):
                sh += [("snet_h%d_w" % i, (n, n)), ("snet_h%d_b" % i, (n,))]
                if self.s_resblock:
                    sh += [("snet_h%d_w2" % i, (n, n)), ("snet_h%d_b2" % i, (n,))]
            sh += [("snet_bottleneck_w", (n, po * so)), ("snet_bottleneck_b", (po * so,))]
            sh += [("last_layer_bias", (so,))]
        return sh

    def n_params(self):
        return int(sum(int(np.prod(s)) for _, s in self.param_shapes()))

    # ---- initial values ---------------------------------------------------------------------
    def initial_weights(self, rng):
        def tn(shape):  # Keras TruncatedNormal(stddev=0.1): resample outside 2 sigma
            out = rng.standard_normal(shape)
            bad = np.abs(out) > 2.0
            while bad.any():
                out[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(out) > 2.0
            return (0.1 * out).astype(np.float32)

        def un(shape, lim):
            return (rng.uniform(-1.0, 1.0, size=shape) * lim).astype(np.float32)

        def siren(nin, nout, pos, om):
            if pos == "first":  # siren.py:178-190
                return [un((nin, nout), 1.0 / nin), un((nout,), 1.0 / math.sqrt(nin))]
            return [un((nin, nout), math.sqrt(6.0 / nin) / om), un((nout,), 1.0 / math.sqrt(nin))]

        ws = []
        pi, nst, r, po = self.pi_dim, self.n_st, self.pi_hidden, self.po_dim
        if self.p_siren:
            ws += siren(pi, nst, "first", self.p_omega0)
            for _ in range(self.l_st):
                wb = siren(nst, nst, "hidden", self.p_omega0)
                ws += wb
                if self.p_resblock:  # w2, b2 are copies of w_init, b_init (siren.py:370-379)
                    ws += [wb[0].copy(), wb[1].copy()]
            ws += siren(nst, r, "bottleneck", self.p_omega0)
        else:
            ws += [tn((pi, nst)), tn((nst,))]
            for _ in range(self.l_st):
                ws += [tn((nst, nst)), tn((nst,))]
                if self.p_resblock:
                    ws += [tn((nst, nst)), tn((nst,))]
            ws += [tn((nst, r)), tn((r,))]
        if self.kind == "NIF":
            ws += [tn((r, po)), tn((po,))]
        else:  # gen_hypernetwork_weights_bias_for_siren_shapenet, siren.py:6-63
            wf = self.cfg_shape_net["weight_init_factor"]
            w = un((r, po), math.sqrt(6.0 / r) * wf)
            if self.connectivity == "full":
                nwf = self.si_dim * self.n_sx
                nwh = self.n_hidden_mats * self.n_sx ** 2
                nwl = self.so_dim * self.n_sx
            else:
                nwf, nwh, nwl = 0, 0, po
            scale = np.ones((po,))
            scale[:nwf] /= self.si_dim
            scale[nwf:nwf + nwh] *= math.sqrt(6.0 / self.n_sx) / self.s_omega0
            scale[nwf + nwh:nwf + nwh + nwl] *= math.sqrt(6.0 / (2 * self.n_sx))
            scale[nwf + nwh + nwl:] /= self.n_sx
            ws += [w, (rng.uniform(-1.0, 1.0, size=(po,)) * scale).astype(np.float32)]
        if self.kind == "NIFMultiScaleLastLayerParameterized":
            si, n, so = self.si_dim, self.n_sx, self.so_dim
            ws += siren(si, n, "first", self.s_omega0)
            for _ in range(self.l_sx):
                wb = siren(n, n, "hidden", self.s_omega0)
                ws += wb
                if self.s_resblock:
                    ws += [wb[0].copy(), wb[1].copy()]
            ws += siren(n, po * so, "bottleneck", self.s_omega0)
            ws += [tn((so,))]
        for w, (_, s) in zip(ws, self.param_shapes()):
            assert tuple(w.shape) == tuple(s)
        return ws
