"""CPU oracle for the NIF point-wise training-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement (fp64 by default, fp32 on request) of the arithmetic that
pswpswpsw/nif v1.0.3 asks TensorFlow/Keras to do on the path named by BASELINE.json.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product (`nif_amd/`) never does.

PARITY UNPINNED.  The reference has no tests, no golden vectors and cannot be imported here
(`import nif` needs tensorflow==2.11.1, which is not installed and not vendored).  The oracle
is therefore pinned only by (a) line-by-line restatement of the reference sources cited in
every function below, (b) an independent torch-autograd fp64 restatement used as a cross-check
in tests/test_oracle_autograd.py, (c) internal identities (3-stage factorisation, finite
differences) and (d) the reference's bundled travelling-wave dataset + its NumPy-only
normalisers, which *can* be executed here.

All citations are `file:line` under /root/reference/.

Deliberately reference-shaped: `pnet_output [B, po_dim]` is materialised and sliced exactly as
`nif/model.py:253-300` / `:769-846` / `:883-933` do, and the ShapeNet uses
`einsum('ai,aij->aj')` (`nif/layers/mlp.py:219`).  Use small B.
"""
from __future__ import annotations

import math
import numpy as np

# ----------------------------------------------------------------------------------------------
# activations (Keras names; `keras.activations.get`, nif/model.py:303, nif/layers/mlp.py:40-44)
# ----------------------------------------------------------------------------------------------


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _erf(x):
    try:
        from scipy.special import erf  # noqa
        return erf(x)
    except Exception:  # pragma: no cover
        return np.vectorize(math.erf)(x)


def act_fn(name):
    """Return (f, f') for a Keras activation name.  'sine' is the SIREN activation."""
    if name in (None, "linear"):
        return (lambda a: a), (lambda a: np.ones_like(a))
    if name in ("swish", "silu"):
        def f(a):
            return a * _sigmoid(a)

        def df(a):
            s = _sigmoid(a)
            return s * (1.0 + a * (1.0 - s))
        return f, df
    if name == "tanh":
        return np.tanh, (lambda a: 1.0 - np.tanh(a) ** 2)
    if name == "relu":
        return (lambda a: np.maximum(a, 0.0)), (lambda a: (a > 0).astype(a.dtype))
    if name == "sigmoid":
        return _sigmoid, (lambda a: _sigmoid(a) * (1.0 - _sigmoid(a)))
    if name == "elu":
        return (lambda a: np.where(a > 0, a, np.expm1(np.minimum(a, 0.0)))), (
            lambda a: np.where(a > 0, 1.0, np.exp(np.minimum(a, 0.0))))
    if name == "softplus":
        return (lambda a: np.logaddexp(a, 0.0)), _sigmoid
    if name == "gelu":  # keras default approximate=False
        def f(a):
            return 0.5 * a * (1.0 + _erf(a / math.sqrt(2.0)))

        def df(a):
            return 0.5 * (1.0 + _erf(a / math.sqrt(2.0))) + a * np.exp(-0.5 * a * a) / math.sqrt(2 * math.pi)
        return f, df
    if name == "selu":      # keras/activations.py selu: scale * elu(x, alpha)
        al, sc = 1.6732632423543772, 1.0507009873554805
        return (lambda a: sc * np.where(a > 0, a, al * np.expm1(np.minimum(a, 0.0)))), (
            lambda a: sc * np.where(a > 0, 1.0, al * np.exp(np.minimum(a, 0.0))))
    if name == "softsign":
        return (lambda a: a / (1.0 + np.abs(a))), (lambda a: 1.0 / (1.0 + np.abs(a)) ** 2)
    if name == "exponential":
        return np.exp, np.exp
    if name == "hard_sigmoid":      # Keras 2.11 backend.hard_sigmoid: clip(0.2 x + 0.5, 0, 1)
        return (lambda a: np.clip(0.2 * a + 0.5, 0.0, 1.0)), (lambda a: np.where((0.2 * a + 0.5 > 0) & (0.2 * a + 0.5 < 1), 0.2, 0.0))
    if name == "sine":
        return np.sin, np.cos
    raise ValueError("unknown activation %r" % (name,))


def loss_vd(name):
    """(v, v') of the per-element Keras regression losses keras.losses.get resolves for compile(loss=...): e = prediction - target.
    'mse' e^2; 'mae' |e|; 'huber' (delta = 1): e^2 / 2 for |e| <= 1, |e| - 1/2 beyond; 'log_cosh': log cosh e.  The reduction is the
    same for all of them: mean over the last axis, sample-weighted sum over the batch / batch size (README.md:33 uses 'mse')."""
    if name in ("mse", "mean_squared_error"):
        return (lambda e: e ** 2), (lambda e: 2.0 * e)
    if name in ("mae", "mean_absolute_error"):
        return np.abs, np.sign
    if name in ("huber", "huber_loss"):
        return (lambda e: np.where(np.abs(e) <= 1.0, 0.5 * e ** 2, np.abs(e) - 0.5)), (lambda e: np.clip(e, -1.0, 1.0))
    if name in ("log_cosh", "logcosh"):
        return (lambda e: np.logaddexp(e, -e) - math.log(2.0)), np.tanh
    raise ValueError("unknown loss %r" % (name,))


# ----------------------------------------------------------------------------------------------
# model description (what nif/model.py's three constructors derive from the cfg dicts)
# ----------------------------------------------------------------------------------------------

KIND_NIF = "NIF"
KIND_MS = "NIFMultiScale"
KIND_LL = "NIFMultiScaleLastLayerParameterized"


class Spec(object):
    """Shape bookkeeping for one model.  Follows NIF.__init__ (model.py:73-128),
    NIF._initialize_pnet po_dim (:169-173), NIFMultiScale._initialize_pnet (:569-587) and
    NIFMultiScaleLastLayerParameterized.__init__/_initialize_snet (:1012-1042,:1147-1217)."""

    def __init__(self, kind, cfg_shape_net, cfg_parameter_net):
        self.kind = kind
        cs, cp = cfg_shape_net, cfg_parameter_net
        self.cs, self.cp = dict(cs), dict(cp)
        self.si, self.so = cs["input_dim"], cs["output_dim"]
        self.n, self.L = cs["units"], cs["nlayers"]
        self.pi, self.r = cp["input_dim"], cp["latent_dim"]
        self.nst, self.lst = cp["units"], cp["nlayers"]
        self.p_act = cp["activation"]
        if kind == KIND_NIF:
            self.s_res = False
            self.s_act = cs["activation"]
            self.omega_s = 1.0
            self.p_siren = False
            self.p_res = False
            self.omega_p = 1.0
            self.po = self.L * self.n ** 2 + (self.si + self.so + 1 + self.L) * self.n + self.so
        else:
            assert "use_resblock" in cs, "`use_resblock` should be in cfg_shape_net"
            self.s_res = bool(cs["use_resblock"])
            self.s_act = "sine"
            self.omega_s = float(cs["omega_0"])
            self.p_siren = (cp["activation"] == "sine")
            self.p_res = bool(cp.get("use_resblock", False))
            self.omega_p = float(cp["omega_0"]) if self.p_siren else 1.0
            if cs["connectivity"] == "full":
                nh = 2 * self.L if self.s_res else self.L
                self.po = nh * self.n ** 2 + (self.si + self.so + 1 + nh) * self.n + self.so
            elif cs["connectivity"] == "last_layer":
                self.po = self.r
            else:
                raise ValueError("cfg_shape_net missing correct `connectivity`")
        if kind == KIND_LL:
            assert cs["connectivity"] == "last_layer"
        # number of hidden (n x n) hyper-matrices in the ShapeNet
        self.n_hidden_mats = (2 * self.L if self.s_res else self.L)

    # ---- pnet_output wire layout (model.py:253-300, :769-846, :883-933) -------------------
    def slices(self):
        """Offsets into pnet_output[:, po]: dict with 'w1','wh'(list),'wl','b1','bh'(list),'bl'."""
        si, so, n = self.si, self.so, self.n
        nh = self.n_hidden_mats
        off = 0
        out = {}
        out["w1"] = (off, off + si * n); off += si * n
        out["wh"] = []
        for _ in range(nh):
            out["wh"].append((off, off + n * n)); off += n * n
        out["wl"] = (off, off + n * so); off += n * so
        out["b1"] = (off, off + n); off += n
        out["bh"] = []
        for _ in range(nh):
            out["bh"].append((off, off + n)); off += n
        out["bl"] = (off, off + so); off += so
        assert off == self.po
        return out

    # ---- Keras variable order (model.py:178-231,:591-734,:1162-1215) ---
    def param_shapes(self):
        sh = []
        pi, nst, r, po = self.pi, self.nst, self.r, self.po
        sh += [("pnet_first_w", (pi, nst)), ("pnet_first_b", (nst,))]
        for i in range(self.lst):
            sh += [("pnet_h%d_w" % i, (nst, nst)), ("pnet_h%d_b" % i, (nst,))]
            if self.p_res:
                sh += [("pnet_h%d_w2" % i, (nst, nst)), ("pnet_h%d_b2" % i, (nst,))]
        sh += [("pnet_bottleneck_w", (nst, r)), ("pnet_bottleneck_b", (r,))]
        sh += [("pnet_last_w", (r, po)), ("pnet_last_b", (po,))]
        if self.kind == KIND_LL:
            si, n, so = self.si, self.n, self.so
            sh += [("snet_first_w", (si, n)), ("snet_first_b", (n,))]
            for i in range(self.L):
                sh += [("snet_h%d_w" % i, (n, n)), ("snet_h%d_b" % i, (n,))]
                if self.s_res:
                    sh += [("snet_h%d_w2" % i, (n, n)), ("snet_h%d_b2" % i, (n,))]
            sh += [("snet_bottleneck_w", (n, self.po * so)), ("snet_bottleneck_b", (self.po * so,))]
            sh += [("last_layer_bias", (so,))]
        return sh

    def n_params(self):
        return int(sum(int(np.prod(s)) for _, s in self.param_shapes()))


# ----------------------------------------------------------------------------------------------
# initialisers (distribution parity only; model.py:181-182, siren.py:6-63,:178-204, mlp.py:245)
# ----------------------------------------------------------------------------------------------


def _trunc_normal(rng, shape, std=0.1):
    """Keras TruncatedNormal(stddev): resample beyond 2 sigma."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * std


def _uniform(rng, shape, lim):
    return rng.uniform(-1.0, 1.0, size=shape) * lim


def _siren_init(rng, nin, nout, pos, omega):
    if pos == "first":  # siren.py:178-190
        return _uniform(rng, (nin, nout), 1.0 / nin), _uniform(rng, (nout,), 1.0 / math.sqrt(nin))
    # hidden / bottleneck, siren.py:192-204
    return (_uniform(rng, (nin, nout), math.sqrt(6.0 / nin) / omega),
            _uniform(rng, (nout,), 1.0 / math.sqrt(nin)))


def _hyper_init(rng, spec):
    """gen_hypernetwork_weights_bias_for_siren_shapenet (siren.py:6-63) as called from
    HyperLinearForSIREN.__init__ (siren.py:479-501)."""
    r, po = spec.r, spec.po
    wf = spec.cs["weight_init_factor"]
    w = _uniform(rng, (r, po), math.sqrt(6.0 / r) * wf)
    if spec.cs["connectivity"] == "full":
        nwf = spec.si * spec.n
        nwh = spec.n_hidden_mats * spec.n ** 2
        nwl = spec.so * spec.n
    else:  # last_layer: siren.py:485-486
        nwf, nwh, nwl = 0, 0, po
    scale = np.ones((po,))
    scale[:nwf] /= spec.si
    scale[nwf:nwf + nwh] *= math.sqrt(6.0 / spec.n) / spec.omega_s
    scale[nwf + nwh:nwf + nwh + nwl] *= math.sqrt(6.0 / (2 * spec.n))
    scale[nwf + nwh + nwl:] /= spec.n
    b = rng.uniform(-1.0, 1.0, size=(po,)) * scale
    return w, b


def init_weights(spec, rng, dtype=np.float64):
    """Draw one set of weights in Keras variable order with the reference's distributions."""
    ws = []
    if spec.kind == KIND_NIF or not spec.p_siren:
        # Dense layers, TruncatedNormal(0.1) for kernels AND biases (model.py:181-182,:671-672)
        def dense(nin, nout):
            return [_trunc_normal(rng, (nin, nout)), _trunc_normal(rng, (nout,))]
        ws += dense(spec.pi, spec.nst)
        for _ in range(spec.lst):
            ws += dense(spec.nst, spec.nst)
            if spec.p_res:
                ws += dense(spec.nst, spec.nst)
        ws += dense(spec.nst, spec.r)
    else:
        ws += list(_siren_init(rng, spec.pi, spec.nst, "first", spec.omega_p))
        for _ in range(spec.lst):
            w, b = _siren_init(rng, spec.nst, spec.nst, "hidden", spec.omega_p)
            ws += [w, b]
            if spec.p_res:  # w2,b2 start as copies (siren.py:370-379)
                ws += [w.copy(), b.copy()]
        ws += list(_siren_init(rng, spec.nst, spec.r, "bottleneck", spec.omega_p))
    if spec.kind == KIND_NIF:
        ws += [_trunc_normal(rng, (spec.r, spec.po)), _trunc_normal(rng, (spec.po,))]
    else:
        ws += list(_hyper_init(rng, spec))
    if spec.kind == KIND_LL:
        ws += list(_siren_init(rng, spec.si, spec.n, "first", spec.omega_s))
        for _ in range(spec.L):
            w, b = _siren_init(rng, spec.n, spec.n, "hidden", spec.omega_s)
            ws += [w, b]
            if spec.s_res:
                ws += [w.copy(), b.copy()]
        ws += list(_siren_init(rng, spec.n, spec.po * spec.so, "bottleneck", spec.omega_s))
        ws += [_trunc_normal(rng, (spec.so,))]
    shapes = spec.param_shapes()
    assert len(ws) == len(shapes)
    for w, (_, s) in zip(ws, shapes):
        assert tuple(w.shape) == tuple(s), (w.shape, s)
    return [np.asarray(w, dtype=dtype) for w in ws]


# ----------------------------------------------------------------------------------------------
# ParameterNet (model.py:326-343 driving the layers of model.py:178-231 / :591-734)
# ----------------------------------------------------------------------------------------------


def _pnet_split(spec, ws):
    """-> (first, hidden[list], bottleneck, last, rest) with each entry a tuple of arrays."""
    it = iter(ws)
    first = (next(it), next(it))
    hidden = []
    for _ in range(spec.lst):
        if spec.p_res:
            hidden.append((next(it), next(it), next(it), next(it)))
        else:
            hidden.append((next(it), next(it)))
    bott = (next(it), next(it))
    last = (next(it), next(it))
    rest = list(it)
    return first, hidden, bott, last, rest


def pnet_forward(spec, ws, p, keep=False):
    """p [B,pi] -> (pnet_out [B,po], latent [B,r]); `keep` also returns the tape for backward.

    Dense: act(x@K+b); MLP_SimpleShortCut: x + act(x@K+b) (mlp.py:148-160);
    MLP_ResNet: act(x + L2(act(L1 x))) (mlp.py:62-79); SIREN first/hidden sin(w0*(x@w)+b),
    bottleneck linear (siren.py:256-281); SIREN_ResNet 0.5*(x+sin(w0*(sin(w0*x@w+b))@w2+b2))
    (siren.py:381-410); last layer linear (siren.py:514-522 / Keras Dense model.py:220-230)."""
    first, hidden, bott, last, _ = _pnet_split(spec, ws)
    tape = []
    if spec.p_siren:
        om = spec.omega_p
        a = om * (p @ first[0]) + first[1]
        h = np.sin(a)
        tape.append(("first", p, a))
        for lay in hidden:
            if spec.p_res:
                a1 = om * (h @ lay[0]) + lay[1]
                t = np.sin(a1)
                a2 = om * (t @ lay[2]) + lay[3]
                hn = 0.5 * (h + np.sin(a2))
                tape.append(("sres", h, a1, t, a2))
            else:
                a1 = om * (h @ lay[0]) + lay[1]
                hn = np.sin(a1)
                tape.append(("siren", h, a1))
            h = hn
    else:
        f, _ = act_fn(spec.p_act)
        a = p @ first[0] + first[1]
        h = f(a)
        tape.append(("first", p, a))
        for lay in hidden:
            if spec.p_res:
                a1 = h @ lay[0] + lay[1]
                t = f(a1)
                a2 = h + (t @ lay[2] + lay[3])
                hn = f(a2)
                tape.append(("mres", h, a1, t, a2))
            else:
                a1 = h @ lay[0] + lay[1]
                hn = h + f(a1)
                tape.append(("short", h, a1))
            h = hn
    z = h @ bott[0] + bott[1]
    out = z @ last[0] + last[1]
    if keep:
        return out, z, (tape, h)
    return out, z


def pnet_backward(spec, ws, tape_h, g_out, g_z_extra=None, g_last=None):
    """Reverse sweep of pnet_forward.  g_out = dLoss/d pnet_out [B,po].  Returns list of grads
    for the pnet variables (Keras order).  (g_out None: the last layer's gradients `g_last` = [gW, gb] and the
    latent gradient `g_z_extra` come from the plane formulation below.)"""
    first, hidden, bott, last, _ = _pnet_split(spec, ws)
    tape, h_last = tape_h
    # last layer: out = z@Wl + bl
    z = h_last @ bott[0] + bott[1]
    if g_out is None:
        gz = g_z_extra
    else:
        g_last = [z.T @ g_out, g_out.sum(0)]
        gz = g_out @ last[0].T
        if g_z_extra is not None:
            gz = gz + g_z_extra
    g_bott = [h_last.T @ gz, gz.sum(0)]
    gh = gz @ bott[0].T
    g_hidden = []
    om = spec.omega_p
    f, df = act_fn(spec.p_act if not spec.p_siren else "sine")
    for lay, rec in zip(reversed(hidden), reversed(tape[1:])):
        kind = rec[0]
        if kind == "sres":
            _, hin, a1, t, a2 = rec
            ga2 = 0.5 * gh * np.cos(a2)
            gw2 = om * (t.T @ ga2); gb2 = ga2.sum(0)
            gt = om * (ga2 @ lay[2].T)
            ga1 = gt * np.cos(a1)
            gw1 = om * (hin.T @ ga1); gb1 = ga1.sum(0)
            gh = 0.5 * gh + om * (ga1 @ lay[0].T)
            g_hidden.append([gw1, gb1, gw2, gb2])
        elif kind == "siren":
            _, hin, a1 = rec
            ga1 = gh * np.cos(a1)
            g_hidden.append([om * (hin.T @ ga1), ga1.sum(0)])
            gh = om * (ga1 @ lay[0].T)
        elif kind == "mres":
            _, hin, a1, t, a2 = rec
            ga2 = gh * df(a2)
            gw2 = t.T @ ga2; gb2 = ga2.sum(0)
            gt = ga2 @ lay[2].T
            ga1 = gt * df(a1)
            gw1 = hin.T @ ga1; gb1 = ga1.sum(0)
            gh = ga2 + ga1 @ lay[0].T
            g_hidden.append([gw1, gb1, gw2, gb2])
        else:  # short
            _, hin, a1 = rec
            ga1 = gh * df(a1)
            g_hidden.append([hin.T @ ga1, ga1.sum(0)])
            gh = gh + ga1 @ lay[0].T
    _, p, a = tape[0]
    if spec.p_siren:
        ga = gh * np.cos(a)
        g_first = [om * (p.T @ ga), ga.sum(0)]
    else:
        ga = gh * df(a)
        g_first = [p.T @ ga, ga.sum(0)]
    grads = list(g_first)
    for g in reversed(g_hidden):
        grads += g
    grads += g_bott + g_last
    return grads


# ----------------------------------------------------------------------------------------------
# ShapeNet given per-sample weights (model.py:233-324 and :738-954) -- the einsum chain
# ----------------------------------------------------------------------------------------------


def _ein(u, w):
    """EinsumLayer('ai,aij->aj') (mlp.py:209-219): one vector-matrix product per sample (batched matmul: the same
    sums as the einsum, BLAS speed)."""
    return np.matmul(u[:, None, :], w)[:, 0, :]


def _ein_t(w, g):
    """its adjoint w.r.t. the vector: 'aij,aj->ai'"""
    return np.matmul(w, g[:, :, None])[:, :, 0]


def shapenet_given_w(spec, x, w, keep=False):
    """x [B,si], w = pnet_output [B,po] -> u [B,so].  NIF._call_shape_net (model.py:233-324)
    for KIND_NIF, NIFMultiScale._call_shape_net_mres (model.py:738-954) otherwise."""
    assert spec.kind in (KIND_NIF, KIND_MS)
    B = x.shape[0]
    si, so, n = spec.si, spec.so, spec.n
    sl = spec.slices()
    W1 = w[:, sl["w1"][0]:sl["w1"][1]].reshape(B, si, n)
    Wh = [w[:, a:b].reshape(B, n, n) for (a, b) in sl["wh"]]
    Wl = w[:, sl["wl"][0]:sl["wl"][1]].reshape(B, n, so)
    b1 = w[:, sl["b1"][0]:sl["b1"][1]]
    bh = [w[:, a:b] for (a, b) in sl["bh"]]
    bl = w[:, sl["bl"][0]:sl["bl"][1]]
    tape = {"W1": W1, "Wh": Wh, "Wl": Wl, "x": x}
    if spec.kind == KIND_NIF:
        f, _ = act_fn(spec.s_act)
        a0 = _ein(x, W1) + b1
        u = f(a0)
        acts = [(None, a0)]
        for i in range(spec.L):
            a = _ein(u, Wh[i]) + bh[i]
            acts.append((u, a))
            u = f(a) + u
    else:
        om = spec.omega_s
        a0 = om * _ein(x, W1) + b1
        u = np.sin(a0)
        acts = [(None, a0)]
        if spec.s_res:
            for i in range(spec.L):
                a1 = om * _ein(u, Wh[2 * i]) + bh[2 * i]
                t = np.sin(a1)
                a2 = om * _ein(t, Wh[2 * i + 1]) + bh[2 * i + 1]
                acts.append((u, a1, t, a2))
                u = 0.5 * (u + np.sin(a2))
        else:
            for i in range(spec.L):
                a = om * _ein(u, Wh[i]) + bh[i]
                acts.append((u, a))
                u = np.sin(a)
    out = _ein(u, Wl) + bl
    if keep:
        tape["acts"] = acts
        tape["hL"] = u
        return out, tape
    return out


def shapenet_given_w_backward(spec, tape, g_u):
    """Adjoint of shapenet_given_w w.r.t. w (returns g_w [B,po]) -- what GradientTape builds
    through the slices/reshapes/einsums (SURVEY a-10)."""
    B = g_u.shape[0]
    si, so, n = spec.si, spec.so, spec.n
    sl = spec.slices()
    gw = np.zeros((B, spec.po), dtype=g_u.dtype)
    hL = tape["hL"]
    gw[:, sl["wl"][0]:sl["wl"][1]] = (hL[:, :, None] * g_u[:, None, :]).reshape(B, -1)
    gw[:, sl["bl"][0]:sl["bl"][1]] = g_u
    gh = _ein_t(tape["Wl"], g_u)
    acts = tape["acts"]
    Wh = tape["Wh"]
    if spec.kind == KIND_NIF:
        _, df = act_fn(spec.s_act)
        for i in reversed(range(spec.L)):
            hin, a = acts[i + 1]
            ga = gh * df(a)
            gw[:, sl["wh"][i][0]:sl["wh"][i][1]] = (hin[:, :, None] * ga[:, None, :]).reshape(B, -1)
            gw[:, sl["bh"][i][0]:sl["bh"][i][1]] = ga
            gh = _ein_t(Wh[i], ga) + gh
        a0 = acts[0][1]
        ga0 = gh * df(a0)
        om = 1.0
    else:
        om = spec.omega_s
        if spec.s_res:
            for i in reversed(range(spec.L)):
                hin, a1, t, a2 = acts[i + 1]
                ga2 = 0.5 * gh * np.cos(a2)
                s2 = sl["wh"][2 * i + 1]
                gw[:, s2[0]:s2[1]] = (om * t[:, :, None] * ga2[:, None, :]).reshape(B, -1)
                gw[:, sl["bh"][2 * i + 1][0]:sl["bh"][2 * i + 1][1]] = ga2
                gt = om * _ein_t(Wh[2 * i + 1], ga2)
                ga1 = gt * np.cos(a1)
                s1 = sl["wh"][2 * i]
                gw[:, s1[0]:s1[1]] = (om * hin[:, :, None] * ga1[:, None, :]).reshape(B, -1)
                gw[:, sl["bh"][2 * i][0]:sl["bh"][2 * i][1]] = ga1
                gh = 0.5 * gh + om * _ein_t(Wh[2 * i], ga1)
        else:
            for i in reversed(range(spec.L)):
                hin, a = acts[i + 1]
                ga = gh * np.cos(a)
                gw[:, sl["wh"][i][0]:sl["wh"][i][1]] = (om * hin[:, :, None] * ga[:, None, :]).reshape(B, -1)
                gw[:, sl["bh"][i][0]:sl["bh"][i][1]] = ga
                gh = om * _ein_t(Wh[i], ga)
        a0 = acts[0][1]
        ga0 = gh * np.cos(a0)
    x = tape["x"]
    gw[:, sl["w1"][0]:sl["w1"][1]] = (om * x[:, :, None] * ga0[:, None, :]).reshape(B, -1)
    gw[:, sl["b1"][0]:sl["b1"][1]] = ga0
    return gw


# ----------------------------------------------------------------------------------------------
# The same ShapeNet in its PLANE formulation (what the HIP kernels execute): with zt = (z_1..z_r, 1) and
# M^(k) the k-th plane of pnet_output's affine map (k < r: row k of the hyper kernel, k = r: the hyper bias),
#     h . W(a) = sum_k zt_k(a) (h . M^(k))
# so nothing of size [B, po] exists.  Algebraically identical to shapenet_given_w (tests pin 1e-12 agreement); used
# (a) as a second, independent restatement and (b) to EMULATE the mixed_bfloat16 policy of the build, whose rounding
# points are those of the plane formulation: the operands of every hidden n x n product (activations and planes in
# the forward sweep, dL/da and planes in the data adjoint) are rounded to bfloat16 (`rnd`), accumulation, biases,
# activations, first/last layer, loss and the weight-gradient sums stay in full precision.
# ----------------------------------------------------------------------------------------------
def bf16_round(a):
    """round-to-nearest-even to bfloat16 of the float32 value of `a` (what v_cvt_pk_bf16_f32 does), as float64"""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).astype(np.float64).reshape(a32.shape)


def f16_round(a):
    """round-to-nearest-even to IEEE half precision of the float32 value of `a`, saturated at the largest finite half (65504):
    what v_med3_f32 + v_cvt_pk_f16_f32 do in k_snet4<.., PR = 2> / k_pack16b(f16) -- the build's mixed_float16 policy
    (model.py:101-105 hands 'mixed_float16' to tf.keras.mixed_precision.Policy).  As float64."""
    a32 = np.clip(np.ascontiguousarray(a, dtype=np.float32), -65504.0, 65504.0)
    return a32.astype(np.float16).astype(np.float64)


def _f16_round_grad(g):
    """dL/da [B, n] under mixed_float16: every point's row is rounded as half(s dL/da) / s with s the power of two that brings the
    row's largest |entry| (as float32) into [2^14, 2^15), at most 2^100 -- k_snet4<.., PR = 2>'s per-point loss scale (Keras holds
    one dynamic scale per step, 2^15 at first, and skips overflowing steps: keras/mixed_precision/loss_scale_optimizer.py).
    Exact powers of two: only the rounding -- half's subnormals below 6.1e-5, saturation above 65504 -- sees the scale"""
    g = np.asarray(g, dtype=np.float64)
    mx = np.ascontiguousarray(np.abs(g).max(axis=-1, keepdims=True), dtype=np.float32)
    ef = ((mx.view(np.uint32) >> 23) & 0xFF).astype(np.int64)
    sf = np.minimum(268 - ef, 227)
    s = np.ldexp(1.0, sf - 127)
    return f16_round(s * g) / s


f16_round.grad = _f16_round_grad     # the rounding of the data adjoint's dL/da operand where it differs from the forward operands'


def _grad_round(rnd):
    return (lambda a_: a_) if rnd is None else getattr(rnd, "grad", rnd)


def phase16(a):
    """the 16-bit phase stash of the build (nif_amd/csrc/k_snet3_dev.h, sine16_tag_ph): the argument a of a SIREN layer as the
    readers of the stash row rebuild it -- q = rint(65536 f), f = a / 2 pi - rint(a / 2 pi), a' = 2 pi q / 65536 (|a' - a| <= 4.8e-5
    modulo 2 pi).  Test infrastructure: restates the cast point of the policy step on the 128-wide kernels."""
    t = np.asarray(a, dtype=np.float64) / (2.0 * np.pi)
    f = t - np.rint(t)
    return 2.0 * np.pi * np.rint(65536.0 * f) / 65536.0


def planes_loss_and_grad(spec, ws, inputs, y, sample_weight=None, batch_global=None, rnd=None, stash_bf16=False, stash_ph16=False):
    """loss, grads (Keras order) and predictions of NIF / NIFMultiScale in the plane formulation.
    rnd=None: exact; rnd=bf16_round: the build's mixed_bfloat16 policy.  stash_bf16 (with rnd): the hidden layers' weight-gradient
    sums take dL/da as the bf16 rows the fused kernel stashed and bf16(zt_k h_in) as the other operand, one product each
    (k_gw_lds<.., DAB>: nets of 17..32 / 49..64 units on the bf16 kernels); biases sum the same bf16 dL/da.  stash_ph16 (plain SIREN,
    with stash_bf16: the 128-wide kernels, r5): the hidden matrices' inputs reach the adjoint sweep and the weight-gradient sums as
    sin / cos of the 16-bit phase (phase16) -- every cosine but the top layer's, every h_in of a hidden matrix."""
    assert spec.kind in (KIND_NIF, KIND_MS)
    assert not stash_ph16 or (spec.kind == KIND_MS and not spec.s_res and stash_bf16)
    R = (lambda a: a) if rnd is None else rnd
    Rg = _grad_round(rnd)
    B = inputs.shape[0]
    Bg = B if batch_global is None else batch_global
    p = inputs[:, :spec.pi]
    x = inputs[:, spec.pi:spec.pi + spec.si]
    _, z, ptape = pnet_forward(spec, ws, p, keep=True)
    first, hidden, bott, last, _ = _pnet_split(spec, ws)
    M = np.vstack([last[0], last[1][None, :]])          # [(r+1), po]
    zt = np.hstack([z, np.ones((B, 1), dtype=z.dtype)])   # [B, r+1]
    K = spec.r + 1
    si, so, n = spec.si, spec.so, spec.n
    sl = spec.slices()
    nif = spec.kind == KIND_NIF
    om = 1.0 if nif else spec.omega_s
    f, df = act_fn(spec.s_act) if nif else (np.sin, np.cos)
    mat = lambda k, ab, shape: M[k, ab[0]:ab[1]].reshape(shape)
    vec = lambda k, ab: M[k, ab[0]:ab[1]]

    def prod(u, ab, shape, rounded):      # sum_k zt_k (u . M_k) and the per-plane products
        T = [(R(u) if rounded else u) @ (R(mat(k, ab, shape)) if rounded else mat(k, ab, shape)) for k in range(K)]
        return sum(zt[:, k:k + 1] * T[k] for k in range(K)), T

    def prodw(u, ab, shape):              # hidden n x n products: sum_k zt_k (u . (om M_k)); the kernels' packed planes hold
        # om M_k (omega_0 folded in at pack time, round 3), so under the policy it is om M_k that is rounded to bf16
        return sum(zt[:, k:k + 1] * (R(u) @ R(om * mat(k, ab, shape))) for k in range(K))

    def bias(ab):
        return sum(zt[:, k:k + 1] * vec(k, ab)[None, :] for k in range(K))

    # ---- forward
    s0, T0 = prod(x, sl["w1"], (si, n), False)
    a0 = om * s0 + bias(sl["b1"])
    u = f(a0)
    acts = []
    nh = spec.n_hidden_mats
    if nif:
        for i in range(nh):
            a = prodw(u, sl["wh"][i], (n, n)) + bias(sl["bh"][i])
            acts.append((u, a)); u = f(a) + u
    elif spec.s_res:
        for i in range(spec.L):
            a1 = prodw(u, sl["wh"][2 * i], (n, n)) + bias(sl["bh"][2 * i])
            t = np.sin(a1)
            a2 = prodw(t, sl["wh"][2 * i + 1], (n, n)) + bias(sl["bh"][2 * i + 1])
            acts.append((u, a1, t, a2)); u = 0.5 * (u + np.sin(a2))
    else:
        for i in range(nh):
            a = prodw(u, sl["wh"][i], (n, n)) + bias(sl["bh"][i])
            acts.append((u, a)); u = np.sin(a)
    sL, TL = prod(u, sl["wl"], (n, so), False)
    out = sL + bias(sl["bl"])
    # ---- loss
    e = out - y
    w_a = np.ones((B,), dtype=out.dtype) if sample_weight is None else sample_weight
    loss = ((e ** 2).mean(axis=1) * w_a).sum() / Bg
    g_u = 2.0 * e * w_a[:, None] / (Bg * so)
    # ---- adjoint
    gM = np.zeros_like(M)
    gzt = np.zeros_like(zt)

    def wgrad(ab, hin, ga, scale):        # dL/dM_k[slot] += scale * sum_a zt_k h (x) ga   (full precision)
        for k in range(K):
            gM[k, ab[0]:ab[1]] += scale * ((zt[:, k:k + 1] * hin).T @ ga).ravel()

    def bgrad(ab, ga, hidden=False):
        gs = R(ga) if (hidden and stash_bf16) else ga
        for k in range(K):
            gM[k, ab[0]:ab[1]] += (zt[:, k:k + 1] * gs).sum(0)
            gzt[:, k] += ga @ vec(k, ab)

    def wgrad_h(ab, hin, ga, scale):      # hidden matrices
        if not stash_bf16:
            return wgrad(ab, hin, ga, scale)
        for k in range(K):
            gM[k, ab[0]:ab[1]] += scale * (R(zt[:, k:k + 1] * hin).T @ R(ga)).ravel()

    def back(ab, shape, ga, hin, scale, rounded):     # dL/dh_in and the latent part through the matrix
        if rounded:                                   # hidden matrices: the adjoint planes hold scale * M_k as well
            U = [Rg(ga) @ R(scale * mat(k, ab, shape)).T for k in range(K)]
            scale = 1.0
        else:
            U = [ga @ mat(k, ab, shape).T for k in range(K)]
        for k in range(K):
            gzt[:, k] += scale * (hin * U[k]).sum(1)
        return scale * sum(zt[:, k:k + 1] * U[k] for k in range(K))

    wgrad(sl["wl"], u, g_u, 1.0); bgrad(sl["bl"], g_u)
    gh = back(sl["wl"], (n, so), g_u, u, 1.0, False)
    if nif:
        for i in reversed(range(nh)):
            hin, a = acts[i]
            ga = gh * df(a)
            wgrad_h(sl["wh"][i], hin, ga, 1.0); bgrad(sl["bh"][i], ga, True)
            gh = back(sl["wh"][i], (n, n), ga, hin, 1.0, True) + gh
    elif spec.s_res:
        for i in reversed(range(spec.L)):
            hin, a1, t, a2 = acts[i]
            ga2 = 0.5 * gh * np.cos(a2)
            wgrad_h(sl["wh"][2 * i + 1], t, ga2, om); bgrad(sl["bh"][2 * i + 1], ga2, True)
            gt = back(sl["wh"][2 * i + 1], (n, n), ga2, t, om, True)
            ga1 = gt * np.cos(a1)
            wgrad_h(sl["wh"][2 * i], hin, ga1, om); bgrad(sl["bh"][2 * i], ga1, True)
            gh = 0.5 * gh + back(sl["wh"][2 * i], (n, n), ga1, hin, om, True)
    else:
        for i in reversed(range(nh)):
            hin, a = acts[i]
            if stash_ph16:      # slot i holds the phase of the layer below, slot i + 1 (i < nh - 1) this layer's
                hin = np.sin(phase16(a0 if i == 0 else acts[i - 1][1]))
                ga = gh * (np.cos(a) if i == nh - 1 else np.cos(phase16(a)))
            else:
                ga = gh * np.cos(a)
            wgrad_h(sl["wh"][i], hin, ga, om); bgrad(sl["bh"][i], ga, True)
            gh = back(sl["wh"][i], (n, n), ga, hin, om, True)
    ga0 = gh * (np.cos(phase16(a0)) if stash_ph16 else df(a0))
    wgrad(sl["w1"], x, ga0, om); bgrad(sl["b1"], ga0)
    back(sl["w1"], (si, n), ga0, x, om, False)
    grads = pnet_backward(spec, ws, ptape, None, g_z_extra=gzt[:, :spec.r], g_last=[gM[:spec.r], gM[spec.r]])
    return loss, grads, out


# ----------------------------------------------------------------------------------------------
# last-layer-parameterised ShapeNet (model.py:1219-1269)
# ----------------------------------------------------------------------------------------------


def _snet_split(spec, rest):
    it = iter(rest)
    first = (next(it), next(it))
    hidden = []
    for _ in range(spec.L):
        if spec.s_res:
            hidden.append((next(it), next(it), next(it), next(it)))
        else:
            hidden.append((next(it), next(it)))
    bott = (next(it), next(it))
    bias = next(it)
    return first, hidden, bott, bias


def snet_phi(spec, ws, x, keep=False, rnd=None):
    """x [B,si] -> phi [B,so,r] (model.py:1219-1238; layers siren.py:256-281,:381-410).
    rnd = bf16_round: the build's mixed_bfloat16 policy (both operands of the hidden n x n products rounded to bf16)."""
    *_, rest = _pnet_split(spec, ws)
    first, hidden, bott, _ = _snet_split(spec, rest)
    R = (lambda a_: a_) if rnd is None else rnd
    om = spec.omega_s
    a = om * (x @ first[0]) + first[1]
    h = np.sin(a)
    tape = [("first", x, a)]
    for lay in hidden:
        if spec.s_res:
            a1 = R(h) @ R(om * lay[0]) + lay[1]       # (the packed planes hold om W: that product is what gets rounded)
            t = np.sin(a1)
            a2 = R(t) @ R(om * lay[2]) + lay[3]
            tape.append(("sres", h, a1, t, a2))
            h = 0.5 * (h + np.sin(a2))
        else:
            a1 = R(h) @ R(om * lay[0]) + lay[1]
            tape.append(("siren", h, a1))
            h = np.sin(a1)
    phi = (h @ bott[0] + bott[1]).reshape(x.shape[0], spec.so, spec.r)
    if keep:
        return phi, (tape, h)
    return phi


def _snet_backward(spec, ws, tape_h, g_phi, rnd=None, stash_bf16=False, stash_ph16=False):
    *_, rest = _pnet_split(spec, ws)
    first, hidden, bott, _ = _snet_split(spec, rest)
    R = (lambda a_: a_) if rnd is None else rnd      # policy: dL/da and the weights rounded in the data adjoint; weight gradients fp32
    S = R if stash_bf16 else (lambda a_: a_)         # ... unless the dL/da stash rows are bf16: hidden sums bf16(h_in)^T bf16(dL/da)
    Rg = _grad_round(rnd)                            # (mixed_float16: dL/da is rounded under the loss scale)
    tape, hL = tape_h
    om = spec.omega_s
    g = g_phi.reshape(g_phi.shape[0], -1)
    g_bott = [hL.T @ g, g.sum(0)]
    gh = g @ bott[0].T
    g_hidden = []
    for li, lay, rec in zip(reversed(range(1, len(tape))), reversed(hidden), reversed(tape[1:])):
        if rec[0] == "sres":
            _, hin, a1, t, a2 = rec
            ga2 = 0.5 * gh * np.cos(a2)
            gw2 = om * (S(t).T @ S(ga2)); gb2 = S(ga2).sum(0)
            gt = Rg(ga2) @ R(om * lay[2]).T
            ga1 = gt * np.cos(a1)
            gw1 = om * (S(hin).T @ S(ga1)); gb1 = S(ga1).sum(0)
            gh = 0.5 * gh + Rg(ga1) @ R(om * lay[0]).T
            g_hidden.append([gw1, gb1, gw2, gb2])
        else:
            _, hin, a1 = rec
            if stash_ph16:      # (16-bit phase stash: planes_loss_and_grad)
                hin = np.sin(phase16(tape[li - 1][2]))
                ga1 = gh * (np.cos(a1) if li == len(tape) - 1 else np.cos(phase16(a1)))
            else:
                ga1 = gh * np.cos(a1)
            g_hidden.append([om * (S(hin).T @ S(ga1)), S(ga1).sum(0)])
            gh = Rg(ga1) @ R(om * lay[0]).T
    _, x, a = tape[0]
    ga = gh * (np.cos(phase16(a)) if stash_ph16 else np.cos(a))
    grads = [om * (x.T @ ga), ga.sum(0)]
    for g_ in reversed(g_hidden):
        grads += g_
    grads += g_bott
    return grads


# ----------------------------------------------------------------------------------------------
# full model, loss, gradient, optimiser
# ----------------------------------------------------------------------------------------------


def forward(spec, ws, inputs):
    """NIF.call (model.py:130-154) / NIFMultiScale.call (:510-539) /
    NIFMultiScaleLastLayerParameterized.call (:1044-1068).  inputs [B, pi+si] -> u [B,so]."""
    p = inputs[:, :spec.pi]
    x = inputs[:, spec.pi:spec.pi + spec.si]
    pout, _ = pnet_forward(spec, ws, p)
    if spec.kind == KIND_LL:
        phi = snet_phi(spec, ws, x)
        # Dot(axes=(2,1)) + BiasAddLayer (model.py:1267-1268)
        return np.einsum("bsj,bj->bs", phi, pout) + ws[-1]
    return shapenet_given_w(spec, x, pout)


def model_p_to_lr(spec, ws, p):
    """model.py:406-420; for the last-layer class it is the pnet *output* (model.py:1070-1083)."""
    out, z = pnet_forward(spec, ws, p)
    return out if spec.kind == KIND_LL else z


def model_lr_to_w(spec, ws, lr):
    """model.py:422-433 -- only the last pnet layer.  Raises for the last-layer class
    (model.py:1106-1115)."""
    if spec.kind == KIND_LL:
        raise ValueError("In this class: NIFMultiScaleLastLayerParameterization, `w` is the same as `lr`")
    _, _, _, last, _ = _pnet_split(spec, ws)
    return lr @ last[0] + last[1]


def model_x_to_phi(spec, ws, x):
    return snet_phi(spec, ws, x)


def mse_loss(u, y, sample_weight=None):
    """Keras 'mse' with SUM_OVER_BATCH_SIZE: mean_B( w_a * mean_so (u-y)^2 )."""
    per = ((u - y) ** 2).mean(axis=1)
    if sample_weight is not None:
        per = per * sample_weight
    return per.sum() / u.shape[0]


def ll_policy_loss_and_grad(spec, ws, inputs, y, sample_weight=None, batch_global=None, rnd=None, stash_bf16=False, stash_ph16=False):
    """last-layer class under the build's mixed_bfloat16 policy (rnd=bf16_round): -> (loss, grads, u), the counterpart of
    planes_loss_and_grad for the shared dense ShapeNet (hidden n x n products on rounded operands, everything else fp32)"""
    assert spec.kind == KIND_LL
    B = inputs.shape[0]
    Bg = B if batch_global is None else batch_global
    p = inputs[:, :spec.pi]
    x = inputs[:, spec.pi:spec.pi + spec.si]
    pout, z, ptape = pnet_forward(spec, ws, p, keep=True)
    phi, stape = snet_phi(spec, ws, x, keep=True, rnd=rnd)
    u = np.einsum("bsj,bj->bs", phi, pout) + ws[-1]
    e = u - y
    w_a = np.ones((B,), dtype=u.dtype) if sample_weight is None else sample_weight
    loss = ((e ** 2).mean(axis=1) * w_a).sum() / Bg
    g_u = 2.0 * e * w_a[:, None] / (Bg * spec.so)
    g_pout = np.einsum("bsj,bs->bj", phi, g_u)
    g_snet = _snet_backward(spec, ws, stape, g_u[:, :, None] * pout[:, None, :], rnd=rnd, stash_bf16=stash_bf16 and rnd is not None,
                            stash_ph16=stash_ph16 and stash_bf16 and rnd is not None)
    return loss, pnet_backward(spec, ws, ptape, g_pout) + g_snet + [g_u.sum(0)], u


def weight_regularizer_coefficients(cfg_shape_net, cfg_parameter_net, kind):
    """((p_l1, p_l2), (s_l1, s_l2)): the Keras kernel / bias regularisers the reference attaches.
    ParameterNet layers (model.py:109-117): L2(l2_reg) if cfg_parameter_net['l2_reg'] is a number, else L1(l1_reg) if that is.
    Shared ShapeNet of the last-layer class (model.py:1028-1039): L2 if cfg_shape_net['l2_reg'] is a number, else L1 if
    cfg_shape_net['l1_reg'] is -- but the coefficient handed to regularizers.L2 / L1 there is self.p_l2_reg / self.p_l1_reg, the
    PARAMETER net's value (:1032-1036), and Keras 2.11 replaces a None by its default 0.01 (keras/regularizers.py L1 / L2 __init__)."""
    num = lambda v: isinstance(v, (float, int))
    p_l1, p_l2 = cfg_parameter_net.get("l1_reg", None), cfg_parameter_net.get("l2_reg", None)
    preg = (0.0, float(p_l2)) if num(p_l2) else ((float(p_l1), 0.0) if num(p_l1) else (0.0, 0.0))
    sreg = (0.0, 0.0)
    if kind == KIND_LL:
        s_l1, s_l2 = cfg_shape_net.get("l1_reg", None), cfg_shape_net.get("l2_reg", None)
        if num(s_l2):
            sreg = (0.0, float(p_l2) if p_l2 is not None else 0.01)
        elif num(s_l1):
            sreg = (float(p_l1) if p_l1 is not None else 0.01, 0.0)
    return preg, sreg


def weight_regularizer_term(spec, ws, preg, sreg=(0.0, 0.0)):
    """loss term and per-variable gradient of the kernel / bias regularisers: Keras L2 = l2 * sum(w^2), L1 = l1 * sum(|w|), added
    once per layer variable (siren.py:266-269, :393-398; mlp.py Dense regularisers): every 'pnet_*' variable with `preg`, every
    'snet_*' variable (last-layer class: first / hidden / bottleneck kernels and biases) with `sreg`; last_layer_bias has none."""
    loss, grads = 0.0, []
    for (nm, _), w in zip(spec.param_shapes(), ws):
        l1, l2 = preg if nm.startswith("pnet_") else (sreg if nm.startswith("snet_") else (0.0, 0.0))
        loss += l2 * float((w ** 2).sum()) + l1 * float(np.abs(w).sum())
        grads.append(2.0 * l2 * w + l1 * np.sign(w))
    return loss, grads


def loss_and_grad(spec, ws, inputs, y, sample_weight=None, batch_global=None, act_reg=None, loss="mse"):
    """MSE loss and gradient w.r.t. every variable (Keras order), hand-derived adjoint
    (SURVEY a-10).  `batch_global` lets a shard compute its share of a larger batch's
    mean (loss and grads are scaled by 1/batch_global instead of 1/B).
    act_reg = (l1, l2): Keras activity_regularizer on the last ParameterNet layer (model.py:118-125, :226): L2 if l2 else L1
    of the materialised pnet_output, divided by the batch size."""
    B = inputs.shape[0]
    Bg = B if batch_global is None else batch_global
    p = inputs[:, :spec.pi]
    x = inputs[:, spec.pi:spec.pi + spec.si]
    pout, z, ptape = pnet_forward(spec, ws, p, keep=True)
    if spec.kind == KIND_LL:
        phi, stape = snet_phi(spec, ws, x, keep=True)
        u = np.einsum("bsj,bj->bs", phi, pout) + ws[-1]
    else:
        u, tape = shapenet_given_w(spec, x, pout, keep=True)
    e = u - y
    lv, ld = loss_vd(loss)
    per = lv(e).mean(axis=1)
    w_a = np.ones((B,), dtype=u.dtype) if sample_weight is None else sample_weight
    loss = (per * w_a).sum() / Bg
    g_u = ld(e) * w_a[:, None] / (Bg * spec.so)
    if spec.kind == KIND_LL:
        g_pout = np.einsum("bsj,bs->bj", phi, g_u)
        if act_reg is not None and (act_reg[0] or act_reg[1]):     # the ParameterNet output of this class is a [B, r]
            l1, l2 = act_reg
            if l2:
                loss = loss + l2 * (pout ** 2).sum() / Bg
                g_pout = g_pout + 2.0 * l2 * pout / Bg
            else:
                loss = loss + l1 * np.abs(pout).sum() / Bg
                g_pout = g_pout + l1 * np.sign(pout) / Bg
        g_phi = g_u[:, :, None] * pout[:, None, :]
        g_snet = _snet_backward(spec, ws, stape, g_phi)
        g_bias = g_u.sum(0)
        g_pnet = pnet_backward(spec, ws, ptape, g_pout)
        return loss, g_pnet + g_snet + [g_bias]
    g_pout = shapenet_given_w_backward(spec, tape, g_u)
    if act_reg is not None and (act_reg[0] or act_reg[1]):
        l1, l2 = act_reg
        if l2:
            loss = loss + l2 * (pout ** 2).sum() / Bg
            g_pout = g_pout + 2.0 * l2 * pout / Bg
        else:
            loss = loss + l1 * np.abs(pout).sum() / Bg
            g_pout = g_pout + l1 * np.sign(pout) / Bg
    return loss, pnet_backward(spec, ws, ptape, g_pout)


def sobolev_loss_and_grad(spec, ws, inputs, y, dydx, x_index, w_jac, sample_weight=None, batch_global=None, loss="mse"):
    """Sobolev training step (BASELINE config 5): the two-output Keras model
        y, dys_dxs = JacobianLayer(model, y_index=all, x_index)(inputs)      (nif/layers/gradient.py:36-49)
    compiled with loss='mse', loss_weights=[1, w_jac]:   loss = mse(y) + w_jac * mse(dys_dxs)   (Keras 'mse' =
    mean over the last axis, then the sample-weighted mean over the batch; for dys_dxs [B,so,nx] that is the
    mean over B*so*nx).  GradientTape differentiates through the inner tape; here: forward tangents
    (jacobian_analytic) and their adjoint, w.r.t. the materialised per-sample weights pnet_out [B,po], then
    pnet_backward -- the reference formulation.  NIFMultiScale (SIREN, plain / resblock) and class NIF (any Keras
    activation f with skip connections h_l = f(a_l) + h_{l-1}, model.py:309-320).
    With c = f'(a), -sn = f''(a):  nu = mu c ,  da = lambda c - sum mu sn a'.
    x_index may address ANY input column (gradient.py:207-231).  A parameter column j < pi seeds the ParameterNet: the
    per-sample weights then carry a tangent of their own, pnet_out' = z' Wh (pnet_tangents), every layer gets the product-rule
    term  a' = w0 (h' W + h W') + b' , and the adjoint yields dL/dpnet_out' next to dL/dpnet_out; both go back through the
    hyper layer and the (primal, tangent) ParameterNet (pnet_tangents_backward).
    Returns (loss, grads in Keras order, u, dudx)."""
    if spec.kind == KIND_LL:
        return _sobolev_ll(spec, ws, inputs, y, dydx, x_index, w_jac, sample_weight, batch_global, loss=loss)
    assert spec.kind in (KIND_MS, KIND_NIF)
    nif = spec.kind == KIND_NIF
    B = inputs.shape[0]
    Bg = B if batch_global is None else batch_global
    si, so, n, om = spec.si, spec.so, spec.n, (1.0 if nif else spec.omega_s)
    f_, df_ = act_fn(spec.s_act if nif else "sine")
    d2f_ = act_d2(spec.s_act if nif else "sine")
    x_index = list(x_index)
    assert all(0 <= j < spec.pi + si for j in x_index)
    nx = len(x_index)
    pcols = [j for j in x_index if j < spec.pi]          # parameter seeds (stream q is a parameter stream iff x_index[q] < pi)
    p = inputs[:, :spec.pi]
    x = inputs[:, spec.pi:spec.pi + si]
    pout, z, ptape = pnet_forward(spec, ws, p, keep=True)
    last = _pnet_split(spec, ws)[3]
    sl = spec.slices()

    def carve(po_):
        W_ = [po_[:, sl["w1"][0]:sl["w1"][1]].reshape(B, si, n)] + [po_[:, a:b].reshape(B, n, n) for (a, b) in sl["wh"]]
        b_ = [po_[:, sl["b1"][0]:sl["b1"][1]]] + [po_[:, a:b] for (a, b) in sl["bh"]]
        return W_, b_, po_[:, sl["wl"][0]:sl["wl"][1]].reshape(B, n, so), po_[:, sl["bl"][0]:sl["bl"][1]]
    W, bv, Wl, bl = carve(pout)
    tw = [None] * nx                                      # per stream: the weights' own tangent (parameter seeds) or None
    if pcols:
        _, zd, pctx = pnet_tangents(spec, ws, p, pcols)
        for q, j in enumerate(x_index):
            if j < spec.pi:
                tw[q] = carve(zd[pcols.index(j)] @ last[0])
    nl = len(W)                     # sine layers: first + hidden matrices
    # ---- forward: primal h and tangents hd[q] -----------------------------------------------------
    h = x
    hd = [np.zeros((B, si), dtype=x.dtype) if j < spec.pi else np.tile(np.eye(si, dtype=x.dtype)[j - spec.pi], (B, 1)) for j in x_index]
    tape = []
    blk_in = None
    for l in range(nl):
        a = om * _ein(h, W[l]) + bv[l]
        ad = [om * _ein(v, W[l]) for v in hd]
        for q in range(nx):
            if tw[q] is not None:
                ad[q] = ad[q] + om * _ein(h, tw[q][0][l]) + tw[q][1][l]
        fa, cs, sn = f_(a), df_(a), -d2f_(a)          # SIREN: sin, cos, sin
        tape.append((h, hd, sn, cs, ad))
        t, td = fa, [cs * v for v in ad]
        if nif and l >= 1:                            # class NIF: h = f(a) + h_in
            h, hd = t + h, [v + u0 for v, u0 in zip(td, hd)]
        elif spec.s_res and l >= 1:
            if (l - 1) % 2 == 0:
                blk_in = (h, hd)
                h, hd = t, td
            else:
                h = 0.5 * (blk_in[0] + t)
                hd = [0.5 * (u0 + v) for u0, v in zip(blk_in[1], td)]
        else:
            h, hd = t, td
    u = _ein(h, Wl) + bl
    ud = [_ein(v, Wl) for v in hd]
    for q in range(nx):
        if tw[q] is not None:
            ud[q] = ud[q] + _ein(h, tw[q][2]) + tw[q][3]
    J = np.stack(ud, axis=2)                                     # [B, so, nx]
    # ---- loss ------------------------------------------------------------------------------------
    w_a = np.ones((B,), dtype=u.dtype) if sample_weight is None else sample_weight
    e = u - y
    ej = J - np.asarray(dydx).reshape(B, so, nx)
    lv, ld = loss_vd(loss)
    loss = (lv(e).mean(axis=1) * w_a).sum() / Bg + w_jac * (lv(ej).mean(axis=(1, 2)) * w_a).sum() / Bg
    g_u = ld(e) * w_a[:, None] / (Bg * so)
    g_ud = [w_jac * ld(ej[:, :, k]) * w_a[:, None] / (Bg * so * nx) for k in range(nx)]
    # ---- adjoint ---------------------------------------------------------------------------------
    gw = np.zeros((B, spec.po), dtype=u.dtype)
    gwd = [np.zeros((B, spec.po), dtype=u.dtype) if tw[q] is not None else None for q in range(nx)]   # dL/dpnet_out'
    gWl = h[:, :, None] * g_u[:, None, :]
    for v, g in zip(hd, g_ud):
        gWl = gWl + v[:, :, None] * g[:, None, :]
    gw[:, sl["wl"][0]:sl["wl"][1]] = gWl.reshape(B, -1)
    gw[:, sl["bl"][0]:sl["bl"][1]] = g_u
    lam = _ein_t(Wl, g_u)
    mu = [_ein_t(Wl, g) for g in g_ud]
    for q in range(nx):
        if tw[q] is not None:
            gwd[q][:, sl["wl"][0]:sl["wl"][1]] = (h[:, :, None] * g_ud[q][:, None, :]).reshape(B, -1)
            gwd[q][:, sl["bl"][0]:sl["bl"][1]] = g_ud[q]
            lam = lam + _ein_t(tw[q][2], g_ud[q])
    wslices = [sl["w1"]] + list(sl["wh"])
    bslices = [sl["b1"]] + list(sl["bh"])
    skip = None
    for l in reversed(range(nl)):
        hin, hdin, sn, cs, ad = tape[l]
        if spec.s_res and l >= 1 and (l - 1) % 2 == 1:           # second layer of a block: h = .5 (u + t)
            lam = 0.5 * lam
            mu = [0.5 * m for m in mu]
            skip = (lam, mu)
        nu = [m * cs for m in mu]
        da = lam * cs
        for m, v in zip(mu, ad):
            da = da - m * sn * v
        gW = hin[:, :, None] * da[:, None, :]
        for v, g in zip(hdin, nu):
            gW = gW + v[:, :, None] * g[:, None, :]
        gw[:, wslices[l][0]:wslices[l][1]] = (om * gW).reshape(B, -1)
        gw[:, bslices[l][0]:bslices[l][1]] = da
        for q in range(nx):
            if tw[q] is not None:
                gwd[q][:, wslices[l][0]:wslices[l][1]] = (om * hin[:, :, None] * nu[q][:, None, :]).reshape(B, -1)
                gwd[q][:, bslices[l][0]:bslices[l][1]] = nu[q]
        if l > 0:
            lam_new = om * _ein_t(W[l], da)
            mu_new = [om * _ein_t(W[l], g) for g in nu]
            for q in range(nx):
                if tw[q] is not None:
                    lam_new = lam_new + om * _ein_t(tw[q][0][l], nu[q])
            if nif:                                              # skip connection of every hidden layer
                lam_new = lam_new + lam
                mu_new = [a_ + b_ for a_, b_ in zip(mu_new, mu)]
            lam, mu = lam_new, mu_new
            if spec.s_res and (l - 1) % 2 == 0:                  # first layer of a block: add the skip path
                lam = lam + skip[0]
                mu = [m + s_ for m, s_ in zip(mu, skip[1])]
    if not pcols:
        return loss, pnet_backward(spec, ws, ptape, gw), u, J
    # hyper layer: pnet_out = z Wh + bh, pnet_out'_c = z'_c Wh ; then the (primal, tangent) ParameterNet
    g_zd = [np.zeros_like(zd[0]) for _ in pcols]
    g_last_w = z.T @ gw
    for q in range(nx):
        if tw[q] is not None:
            ci = pcols.index(x_index[q])
            g_last_w = g_last_w + zd[ci].T @ gwd[q]
            g_zd[ci] = g_zd[ci] + gwd[q] @ last[0].T
    core = pnet_tangents_backward(spec, ws, pctx, gw @ last[0].T, g_zd)
    return loss, core + [g_last_w, gw.sum(0)], u, J


def sobolev_planes_loss_and_grad(spec, ws, inputs, y, dydx, x_index, w_jac, sample_weight=None, batch_global=None, rnd=None,
                                 stash_bf16=False):
    """The Sobolev step of NIF / NIFMultiScale with COORDINATE columns in x_index, in the plane formulation the kernel executes
    (k_sob / k_sobw):  h_q W(a) = sum_k zt_k (h_q M^(k))  for the primal (q = 0) and every tangent stream.  rnd=None:
    exact -- equal to sobolev_loss_and_grad (tests/test_oracle.py pins 1e-10), a second independent restatement.  rnd=bf16_round:
    the build's mixed_bfloat16 policy with the cast points of the kernels (the plain step's, planes_loss_and_grad): in the forward
    sweep the operands h_q and (w0 M^(k)) of every hidden n x n product are rounded to bfloat16 and the latent factor scales the
    product; in the data adjoint dL/da (and nu^d = mu^d c) and (w0 M^(k)) are rounded; first / last layer, biases, activations,
    loss, the dz dot products and the weight-gradient sums stay in full precision -- except with stash_bf16 (k_sobw<PR>: nets the
    bf16 kernels take), where the hidden layers' weight-gradient sums are bf16(zt_k h_in)^T bf16(dL/da) per stream, one product.
    Returns (loss, grads in Keras order, u, dudx)."""
    assert spec.kind in (KIND_NIF, KIND_MS)
    R = (lambda a: a) if rnd is None else rnd
    nif = spec.kind == KIND_NIF
    B = inputs.shape[0]
    Bg = B if batch_global is None else batch_global
    si, so, n = spec.si, spec.so, spec.n
    om = 1.0 if nif else spec.omega_s
    f_, df_ = act_fn(spec.s_act if nif else "sine")
    d2f_ = act_d2(spec.s_act if nif else "sine")
    x_index = list(x_index)
    assert all(spec.pi <= j < spec.pi + si for j in x_index), "coordinate columns only"
    seeds = [j - spec.pi for j in x_index]
    nx = len(seeds)
    p = inputs[:, :spec.pi]
    x = inputs[:, spec.pi:spec.pi + si]
    _, z, ptape = pnet_forward(spec, ws, p, keep=True)
    last = _pnet_split(spec, ws)[3]
    M = np.vstack([last[0], last[1][None, :]])          # [(r+1), po]
    zt = np.hstack([z, np.ones((B, 1), dtype=z.dtype)])
    K = spec.r + 1
    sl = spec.slices()
    mat = lambda k, ab, shape: M[k, ab[0]:ab[1]].reshape(shape)
    vec = lambda k, ab: M[k, ab[0]:ab[1]]
    ztk = lambda k: zt[:, k:k + 1]
    nh = spec.n_hidden_mats
    # ---- forward ---------------------------------------------------------------------------------------------------
    a = sum(ztk(k) * (om * (x @ mat(k, sl["w1"], (si, n))) + vec(k, sl["b1"])[None, :]) for k in range(K))
    ad = [sum(ztk(k) * (om * mat(k, sl["w1"], (si, n))[sd][None, :]) for k in range(K)) for sd in seeds]
    tape = []

    def act(a_, ad_):
        return f_(a_), df_(a_), -d2f_(a_), [df_(a_) * v for v in ad_]
    t, cs, sn, td = act(a, ad)
    tape.append((x, None, sn, cs, ad))
    h, hd = t, td
    blk = None
    for l in range(nh):
        ab = sl["wh"][l]
        prod = lambda v, k: ztk(k) * (R(v) @ R(om * mat(k, ab, (n, n))))
        a = sum(prod(h, k) for k in range(K)) + sum(ztk(k) * vec(k, sl["bh"][l])[None, :] for k in range(K))
        ad = [sum(prod(v, k) for k in range(K)) for v in hd]
        t, cs, sn, td = act(a, ad)
        tape.append((h, hd, sn, cs, ad))
        if nif:
            h, hd = t + h, [v + u0 for v, u0 in zip(td, hd)]
        elif spec.s_res:
            if l % 2 == 0:
                blk = (h, hd); h, hd = t, td
            else:
                h = 0.5 * (blk[0] + t); hd = [0.5 * (u0 + v) for u0, v in zip(blk[1], td)]
        else:
            h, hd = t, td
    u = sum(ztk(k) * (h @ mat(k, sl["wl"], (n, so)) + vec(k, sl["bl"])[None, :]) for k in range(K))
    ud = [sum(ztk(k) * (v @ mat(k, sl["wl"], (n, so))) for k in range(K)) for v in hd]
    J = np.stack(ud, axis=2)
    # ---- loss --------------------------------------------------------------------------------------------------------
    w_a = np.ones((B,), dtype=u.dtype) if sample_weight is None else sample_weight
    e = u - y
    ej = J - np.asarray(dydx).reshape(B, so, nx)
    lv, ld = loss_vd("mse")
    loss = (lv(e).mean(axis=1) * w_a).sum() / Bg + w_jac * (lv(ej).mean(axis=(1, 2)) * w_a).sum() / Bg
    g_u = ld(e) * w_a[:, None] / (Bg * so)
    g_ud = [w_jac * ld(ej[:, :, k]) * w_a[:, None] / (Bg * so * nx) for k in range(nx)]
    # ---- adjoint -----------------------------------------------------------------------------------------------------
    gM = np.zeros_like(M)
    gzt = np.zeros_like(zt)
    for k in range(K):
        Wl = mat(k, sl["wl"], (n, so))
        gM[k, sl["wl"][0]:sl["wl"][1]] += ((ztk(k) * h).T @ g_u + sum((ztk(k) * v).T @ g for v, g in zip(hd, g_ud))).ravel()
        gM[k, sl["bl"][0]:sl["bl"][1]] += (ztk(k) * g_u).sum(0)
        gzt[:, k] += ((h @ Wl + vec(k, sl["bl"])[None, :]) * g_u).sum(1) + sum(((v @ Wl) * g).sum(1) for v, g in zip(hd, g_ud))
    lam = sum(ztk(k) * (g_u @ mat(k, sl["wl"], (n, so)).T) for k in range(K))
    mu = [sum(ztk(k) * (g @ mat(k, sl["wl"], (n, so)).T) for k in range(K)) for g in g_ud]
    skip = None
    for l in reversed(range(nh)):
        hin, hdin, sn, cs, ad = tape[l + 1]
        if spec.s_res and l % 2 == 1:
            lam = 0.5 * lam; mu = [0.5 * m for m in mu]; skip = (lam, mu)
        if nif:
            skip = (lam, mu)
        nu = [m * cs for m in mu]
        da = lam * cs
        for m, v in zip(mu, ad):
            da = da - m * sn * v
        ab = sl["wh"][l]
        for k in range(K):
            if stash_bf16:
                gM[k, ab[0]:ab[1]] += (om * (R(ztk(k) * hin).T @ R(da) + sum(R(ztk(k) * v).T @ R(g) for v, g in zip(hdin, nu)))).ravel()
                gM[k, sl["bh"][l][0]:sl["bh"][l][1]] += (ztk(k) * R(da)).sum(0)
            else:
                gM[k, ab[0]:ab[1]] += (om * ((ztk(k) * hin).T @ da + sum((ztk(k) * v).T @ g for v, g in zip(hdin, nu)))).ravel()
                gM[k, sl["bh"][l][0]:sl["bh"][l][1]] += (ztk(k) * da).sum(0)
            gzt[:, k] += da @ vec(k, sl["bh"][l])
        U = [[R(v) @ R(om * mat(k, ab, (n, n))).T for k in range(K)] for v in [da] + nu]       # U[q][k]
        for k in range(K):
            gzt[:, k] += (hin * U[0][k]).sum(1) + sum((v * U[1 + q][k]).sum(1) for q, v in enumerate(hdin))
        lam = sum(ztk(k) * U[0][k] for k in range(K))
        mu = [sum(ztk(k) * U[1 + q][k] for k in range(K)) for q in range(nx)]
        if nif or (spec.s_res and l % 2 == 0):
            lam = lam + skip[0]; mu = [m + s_ for m, s_ in zip(mu, skip[1])]
    _, _, sn, cs, ad = tape[0]
    nu = [m * cs for m in mu]
    da = lam * cs
    for m, v in zip(mu, ad):
        da = da - m * sn * v
    for k in range(K):
        W1 = mat(k, sl["w1"], (si, n))
        g1 = (ztk(k) * x).T @ da
        for sd, g in zip(seeds, nu):
            g1[sd] += (ztk(k) * g).sum(0)
        gM[k, sl["w1"][0]:sl["w1"][1]] += (om * g1).ravel()
        gM[k, sl["b1"][0]:sl["b1"][1]] += (ztk(k) * da).sum(0)
        gzt[:, k] += ((om * (x @ W1) + vec(k, sl["b1"])[None, :]) * da).sum(1) + sum((om * W1[sd][None, :] * g).sum(1) for sd, g in zip(seeds, nu))
    grads = pnet_backward(spec, ws, ptape, None, g_z_extra=gzt[:, :spec.r], g_last=[gM[:spec.r], gM[spec.r]])
    return loss, grads, u, J


def _sobolev_ll(spec, ws, inputs, y, dydx, x_index, w_jac, sample_weight=None, batch_global=None, loss="mse"):
    """Sobolev step of the last-layer-parameterised class (model.py:1044-1068, :1219-1269 under JacobianLayer): the shared
    SIREN ShapeNet x -> phi [B,so,r] carries the coordinate tangents phi'_d (_mlp_tangents), u = Dot(phi, a) + bias and
    du/dx_d = Dot(phi'_d, a) with a = the ParameterNet output.  A parameter column c leaves phi alone and moves a:
    du/dp_c = Dot(phi, a'_c), a'_c = (dz/dp_c) last_w (pnet_tangents)."""
    B = inputs.shape[0]
    Bg = B if batch_global is None else batch_global
    si, so, r = spec.si, spec.so, spec.r
    x_index = list(x_index)
    nx = len(x_index)
    assert all(0 <= j < spec.pi + si for j in x_index)
    seeds = [j - spec.pi for j in x_index if j >= spec.pi]
    pcols = [j for j in x_index if j < spec.pi]
    p = inputs[:, :spec.pi]
    x = inputs[:, spec.pi:spec.pi + si]
    a, z, ptape = pnet_forward(spec, ws, p, keep=True)                 # po = r for this class (model.py:583-585)
    _, _, _, last, rest = _pnet_split(spec, ws)
    first, hidden, bott, bias = _snet_split(spec, rest)
    layers = (first, hidden, bott)
    phi_f, phid_f, sctx = _mlp_tangents(layers, True, spec.omega_s, "sine", spec.s_res, x, seeds)
    phi = phi_f.reshape(B, so, r)
    phid = [v.reshape(B, so, r) for v in phid_f]
    if pcols:
        _, zd, pctx = pnet_tangents(spec, ws, p, pcols)
        ad = [v @ last[0] for v in zd]                                  # a'_c [B, r]
    u = np.einsum("bsj,bj->bs", phi, a) + bias
    cols = []
    for j in x_index:
        if j >= spec.pi:
            cols.append(np.einsum("bsj,bj->bs", phid[seeds.index(j - spec.pi)], a))
        else:
            cols.append(np.einsum("bsj,bj->bs", phi, ad[pcols.index(j)]))
    J = np.stack(cols, axis=2)                                          # [B, so, nx]
    w_a = np.ones((B,), dtype=u.dtype) if sample_weight is None else sample_weight
    e = u - y
    ej = J - np.asarray(dydx).reshape(B, so, nx)
    lv, ld = loss_vd(loss)
    loss = (lv(e).mean(axis=1) * w_a).sum() / Bg + w_jac * (lv(ej).mean(axis=(1, 2)) * w_a).sum() / Bg
    g_u = ld(e) * w_a[:, None] / (Bg * so)
    g_J = [w_jac * ld(ej[:, :, k]) * w_a[:, None] / (Bg * so * nx) for k in range(nx)]
    g_a = np.einsum("bsj,bs->bj", phi, g_u)
    g_phi = g_u[:, :, None] * a[:, None, :]
    g_phid = [np.zeros((B, so * r), dtype=u.dtype) for _ in seeds]
    g_ad = [np.zeros((B, r), dtype=u.dtype) for _ in pcols]
    for k, j in enumerate(x_index):
        if j >= spec.pi:
            d = seeds.index(j - spec.pi)
            g_a = g_a + np.einsum("bsj,bs->bj", phid[d], g_J[k])
            g_phid[d] = g_phid[d] + (g_J[k][:, :, None] * a[:, None, :]).reshape(B, -1)
        else:
            ci = pcols.index(j)
            g_phi = g_phi + g_J[k][:, :, None] * ad[ci][:, None, :]
            g_ad[ci] = g_ad[ci] + np.einsum("bsj,bs->bj", phi, g_J[k])
    g_snet = _mlp_tangents_backward(layers, True, spec.omega_s, "sine", spec.s_res, sctx, g_phi.reshape(B, -1), g_phid)
    if not pcols:
        g_pnet = pnet_backward(spec, ws, ptape, g_a)
    else:
        g_last_w = z.T @ g_a + sum(zd[ci].T @ g_ad[ci] for ci in range(len(pcols)))
        core = pnet_tangents_backward(spec, ws, pctx, g_a @ last[0].T, [g @ last[0].T for g in g_ad])
        g_pnet = core + [g_last_w, g_a.sum(0)]
    return loss, g_pnet + g_snet + [g_u.sum(0)], u, J


def flatten(arrs):
    return np.concatenate([np.asarray(a).ravel() for a in arrs])


def unflatten(spec, flat):
    out, off = [], 0
    for _, s in spec.param_shapes():
        k = int(np.prod(s))
        out.append(np.asarray(flat[off:off + k]).reshape(s))
        off += k
    assert off == flat.shape[0]
    return out


def adam_step(theta, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """Keras-2.11 Adam (SURVEY a-11): t is the 1-based step index.  Returns (theta, m, v)."""
    m = m + (g - m) * (1.0 - b1)
    v = v + (g * g - v) * (1.0 - b2)
    alpha = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    theta = theta - alpha * m / (np.sqrt(v) + eps)
    return theta, m, v


# ----------------------------------------------------------------------------------------------
# JacobianLayer (gradient.py:36-49, :207-231): (y, dy_{y_index}/dx_{x_index}) w.r.t. the
# model *input vector* (parameter columns first, then coordinates).
# ----------------------------------------------------------------------------------------------


def jacobian(spec, ws, inputs, y_index, x_index, h=1e-6):
    """Central differences in fp64 -- the oracle for the analytic tangent kernels.  Returns
    (y [B,so], dys_dxs [B,len(y_index),len(x_index)])."""
    inputs = np.asarray(inputs, dtype=np.float64)
    ws64 = [np.asarray(w, dtype=np.float64) for w in ws]
    y = forward(spec, ws64, inputs)
    B = inputs.shape[0]
    J = np.zeros((B, len(y_index), len(x_index)))
    for jj, j in enumerate(x_index):
        d = np.zeros_like(inputs)
        d[:, j] = h
        yp = forward(spec, ws64, inputs + d)
        ym = forward(spec, ws64, inputs - d)
        J[:, :, jj] = ((yp - ym) / (2 * h))[:, list(y_index)]
    return y, J


def jacobian_analytic(spec, ws, inputs, y_index, x_index):
    """Forward-mode tangent for coordinate columns (x_index >= pi) of the
    hypernetwork classes (and, through the shared ShapeNet's tangents, of the last-layer class); used to cross-check the
    central-difference oracle."""
    if spec.kind == KIND_LL:
        seeds = [j - spec.pi for j in x_index]
        assert all(0 <= d < spec.si for d in seeds), "analytic tangent only for coordinate columns"
        a_out, _ = pnet_forward(spec, ws, inputs[:, :spec.pi])
        *_, rest = _pnet_split(spec, ws)
        first, hidden, bott, bias = _snet_split(spec, rest)
        B = inputs.shape[0]
        phi, phid, _ = _mlp_tangents((first, hidden, bott), True, spec.omega_s, "sine", spec.s_res,
                                     inputs[:, spec.pi:spec.pi + spec.si], seeds)
        u = np.einsum("bsj,bj->bs", phi.reshape(B, spec.so, spec.r), a_out) + bias
        J = np.stack([np.einsum("bsj,bj->bs", v.reshape(B, spec.so, spec.r), a_out) for v in phid], axis=2)
        return u, J[:, list(y_index), :]
    assert spec.kind in (KIND_NIF, KIND_MS)
    p = inputs[:, :spec.pi]
    x = inputs[:, spec.pi:spec.pi + spec.si]
    pout, _ = pnet_forward(spec, ws, p)
    u, tape = shapenet_given_w(spec, x, pout, keep=True)
    B = x.shape[0]
    J = np.zeros((B, len(y_index), len(x_index)), dtype=u.dtype)
    acts, Wh, W1, Wl = tape["acts"], tape["Wh"], tape["W1"], tape["Wl"]
    for jj, j in enumerate(x_index):
        assert j >= spec.pi, "analytic tangent only for coordinate columns"
        d = j - spec.pi
        if spec.kind == KIND_NIF:
            _, df = act_fn(spec.s_act)
            hd = df(acts[0][1]) * W1[:, d, :]
            for i in range(spec.L):
                _, a = acts[i + 1]
                hd = df(a) * _ein(hd, Wh[i]) + hd
        else:
            om = spec.omega_s
            hd = np.cos(acts[0][1]) * om * W1[:, d, :]
            if spec.s_res:
                for i in range(spec.L):
                    _, a1, t, a2 = acts[i + 1]
                    td = np.cos(a1) * om * _ein(hd, Wh[2 * i])
                    hd = 0.5 * (hd + np.cos(a2) * om * _ein(td, Wh[2 * i + 1]))
            else:
                for i in range(spec.L):
                    _, a = acts[i + 1]
                    hd = np.cos(a) * om * _ein(hd, Wh[i])
        ud = _ein(hd, Wl)
        J[:, :, jj] = ud[:, list(y_index)]
    return u, J


def pnet_tangents(spec, ws, p, cols):
    """ParameterNet up to the latent with forward-mode tangents w.r.t. the parameter columns `cols`:
    -> (z [B,r], zd = [dz/dp_c for c in cols], ctx for pnet_tangents_backward).  Same layer rules as pnet_forward
    (mlp.py:62-79, :148-160; siren.py:256-281, :381-410)."""
    first, hidden, bott, last, rest = _pnet_split(spec, ws)
    siren = spec.p_siren
    return _mlp_tangents((first, hidden, bott), siren, spec.omega_p if siren else 1.0, "sine" if siren else spec.p_act, spec.p_res,
                         p, cols)


def pnet_tangents_backward(spec, ws, ctx, g_z, g_zd):
    """Adjoint of pnet_tangents: g_z = dL/dz [B,r] (or None), g_zd[d] = dL/d(dz/dp_cols[d]) -> gradients of the first / hidden
    / bottleneck variables (Keras order)."""
    first, hidden, bott, last, rest = _pnet_split(spec, ws)
    siren = spec.p_siren
    return _mlp_tangents_backward((first, hidden, bott), siren, spec.omega_p if siren else 1.0, "sine" if siren else spec.p_act,
                                  spec.p_res, ctx, g_z, g_zd)


def _mlp_tangents(layers, siren, s, name, res, p, cols):
    """A shared-weight MLP of the reference's layer kinds (first / hidden / linear bottleneck) with forward-mode tangents
    w.r.t. its input columns `cols`: the ParameterNet (pnet_tangents) and the last-layer class's ShapeNet x -> phi."""
    first, hidden, bott = layers
    f, df = act_fn(name)
    D = range(len(cols))
    a0 = s * (p @ first[0]) + first[1]
    a0d = [np.broadcast_to(s * first[0][c], a0.shape) for c in cols]
    h = f(a0); hd = [df(a0) * a0d[d] for d in D]
    tape = []
    for lay in hidden:
        if not res:
            a = s * (h @ lay[0]) + lay[1]; ad = [s * (hd[d] @ lay[0]) for d in D]
            tape.append(("plain", h, hd, a, ad))
            if siren:
                h, hd = f(a), [df(a) * ad[d] for d in D]
            else:
                h, hd = h + f(a), [hd[d] + df(a) * ad[d] for d in D]
        else:
            a1 = s * (h @ lay[0]) + lay[1]; a1d = [s * (hd[d] @ lay[0]) for d in D]
            t = f(a1); td = [df(a1) * a1d[d] for d in D]
            a2 = s * (t @ lay[2]) + lay[3]; a2d = [s * (td[d] @ lay[2]) for d in D]
            if not siren:
                a2 = a2 + h; a2d = [a2d[d] + hd[d] for d in D]
            tape.append(("res", h, hd, a1, a1d, t, td, a2, a2d))
            if siren:
                h, hd = 0.5 * (h + f(a2)), [0.5 * (hd[d] + df(a2) * a2d[d]) for d in D]
            else:
                h, hd = f(a2), [df(a2) * a2d[d] for d in D]
    z = h @ bott[0] + bott[1]
    zd = [hd[d] @ bott[0] for d in D]                       # [B, r] each
    return z, zd, (p, list(cols), a0, a0d, tape, h, hd)


def _mlp_tangents_backward(layers, siren, s, name, res, ctx, g_z, g_zd):
    """Adjoint of _mlp_tangents.  With lambda = dL/dh, mu_d = dL/dh'_d:
    nu_d = mu_d f'(a), da = lambda f'(a) + sum_d mu_d f''(a) a'_d, dW = s (h^T da + sum_d h'_d^T nu_d)."""
    first, hidden, bott = layers
    p, cols, a0, a0d, tape, h, hd = ctx
    f, df = act_fn(name)
    d2f = act_d2(name)
    D = range(len(cols))
    gz = np.zeros((p.shape[0], bott[0].shape[1]), dtype=h.dtype) if g_z is None else g_z
    g_bott = [h.T @ gz + sum(hd[d].T @ g_zd[d] for d in D), gz.sum(0)]
    lam = gz @ bott[0].T
    mu = [g_zd[d] @ bott[0].T for d in D]

    def adj(a, ad, lam_, mu_, scale=1.0):
        da = scale * lam_ * df(a)
        for d in D:
            da = da + scale * mu_[d] * d2f(a) * ad[d]
        return da, [scale * mu_[d] * df(a) for d in D]
    g_hidden = []
    for lay, rec in zip(reversed(hidden), reversed(tape)):
        if rec[0] == "plain":
            _, hin, hind, a, ad = rec
            da, nu = adj(a, ad, lam, mu)
            gw = s * (hin.T @ da + sum(hind[d].T @ nu[d] for d in D))
            g_hidden.append([gw, da.sum(0)])
            keep = 0.0 if siren else 1.0
            lam = keep * lam + s * (da @ lay[0].T)
            mu = [keep * mu[d] + s * (nu[d] @ lay[0].T) for d in D]
        else:
            _, hin, hind, a1, a1d, t, td, a2, a2d = rec
            if siren:
                da2, nu2 = adj(a2, a2d, lam, mu, 0.5)
                lam_h, mu_h = 0.5 * lam, [0.5 * mu[d] for d in D]
            else:
                da2, nu2 = adj(a2, a2d, lam, mu)
                lam_h, mu_h = da2, list(nu2)
            gw2 = s * (t.T @ da2 + sum(td[d].T @ nu2[d] for d in D))
            lt = s * (da2 @ lay[2].T); mt = [s * (nu2[d] @ lay[2].T) for d in D]
            da1, nu1 = adj(a1, a1d, lt, mt)
            gw1 = s * (hin.T @ da1 + sum(hind[d].T @ nu1[d] for d in D))
            g_hidden.append([gw1, da1.sum(0), gw2, da2.sum(0)])
            lam = lam_h + s * (da1 @ lay[0].T)
            mu = [mu_h[d] + s * (nu1[d] @ lay[0].T) for d in D]
    da0, nu0 = adj(a0, a0d, lam, mu)
    gw0 = s * (p.T @ da0)
    for d in D:
        gw0[cols[d]] = gw0[cols[d]] + s * nu0[d].sum(0)
    grads = [gw0, da0.sum(0)]
    for g in reversed(g_hidden):
        grads += g
    return grads + g_bott


def jac_reg_loss_and_grad(spec, ws, p, l1, batch_global=None):
    """cfg_parameter_net["jac_reg"] (model.py:353-375 -> JacRegLatentLayer, gradient.py:52-127, :182-205):
    loss = l1 * mean_{a,c,d} (d z_c / d p_d)^2 with z the latent, and its gradient w.r.t. every variable (Keras order; zero
    for the layers downstream of z).  Forward-mode tangents of the ParameterNet + the hand-derived adjoint of the (primal,
    tangent) program (pnet_tangents / pnet_tangents_backward).  Pinned by torch double-backward in tests/test_oracle.py."""
    first, hidden, bott, last, rest = _pnet_split(spec, ws)
    B, pi = p.shape
    Bg = B if batch_global is None else batch_global
    _, zd, ctx = pnet_tangents(spec, ws, p, list(range(pi)))
    coef = l1 / (Bg * spec.r * pi)
    loss = coef * sum((v ** 2).sum() for v in zd)
    grads = pnet_tangents_backward(spec, ws, ctx, None, [2.0 * coef * v for v in zd])
    grads += [np.zeros_like(last[0]), np.zeros_like(last[1])] + [np.zeros_like(w) for w in rest]
    return loss, grads


def act_d2(name):
    """f'' for a Keras activation name / 'sine'"""
    if name in (None, "linear", "relu", "hard_sigmoid"):
        return lambda a: np.zeros_like(a)
    if name == "selu":
        return lambda a: np.where(a > 0, 0.0, 1.0507009873554805 * 1.6732632423543772 * np.exp(np.minimum(a, 0.0)))
    if name == "softsign":
        return lambda a: -2.0 * np.sign(a) / (1.0 + np.abs(a)) ** 3
    if name == "exponential":
        return np.exp
    if name in ("swish", "silu"):
        return lambda a: _sigmoid(a) * (1.0 - _sigmoid(a)) * (2.0 + a * (1.0 - 2.0 * _sigmoid(a)))
    if name == "tanh":
        return lambda a: -2.0 * np.tanh(a) * (1.0 - np.tanh(a) ** 2)
    if name == "sigmoid":
        return lambda a: _sigmoid(a) * (1.0 - _sigmoid(a)) * (1.0 - 2.0 * _sigmoid(a))
    if name == "elu":
        return lambda a: np.where(a > 0, 0.0, np.exp(np.minimum(a, 0.0)))
    if name == "softplus":
        return lambda a: _sigmoid(a) * (1.0 - _sigmoid(a))
    if name == "gelu":
        return lambda a: np.exp(-0.5 * a * a) / math.sqrt(2 * math.pi) * (2.0 - a * a)
    if name == "sine":
        return lambda a: -np.sin(a)
    raise ValueError("unknown activation %r" % (name,))


def _mlp_tangents2(layers, siren, s, name, res, p, cj, ck):
    """Second-order forward mode through a shared-weight MLP (the ParameterNet's first / hidden / bottleneck layers) for the
    pair of input columns (cj, ck): -> (z, dz/dp_cj, dz/dp_ck, d2z/dp_cj dp_ck).  h'' = f'(a) a'' + f''(a) a'_j a'_k."""
    first, hidden, bott = layers
    f, df = act_fn(name)
    d2f = act_d2(name)

    def act3(a, aj, ak, ajk):
        return f(a), df(a) * aj, df(a) * ak, df(a) * ajk + d2f(a) * aj * ak
    a0 = s * (p @ first[0]) + first[1]
    aj = np.broadcast_to(s * first[0][cj], a0.shape); ak = np.broadcast_to(s * first[0][ck], a0.shape)
    h, hj, hk, hjk = act3(a0, aj, ak, np.zeros_like(a0))

    def lin(W, b, v, vj, vk, vjk):
        return s * (v @ W) + b, s * (vj @ W), s * (vk @ W), s * (vjk @ W)
    for lay in hidden:
        if not res:
            t = act3(*lin(lay[0], lay[1], h, hj, hk, hjk))
            if siren:
                h, hj, hk, hjk = t
            else:
                h, hj, hk, hjk = h + t[0], hj + t[1], hk + t[2], hjk + t[3]
        else:
            t = act3(*lin(lay[0], lay[1], h, hj, hk, hjk))
            a2 = lin(lay[2], lay[3], *t)
            if not siren:
                a2 = (a2[0] + h, a2[1] + hj, a2[2] + hk, a2[3] + hjk)
            v = act3(*a2)
            if siren:
                h, hj, hk, hjk = 0.5 * (h + v[0]), 0.5 * (hj + v[1]), 0.5 * (hk + v[2]), 0.5 * (hjk + v[3])
            else:
                h, hj, hk, hjk = v
    return h @ bott[0] + bott[1], hj @ bott[0], hk @ bott[0], hjk @ bott[0]


def pnet_tangents2(spec, ws, p, cj, ck):
    first, hidden, bott, last, rest = _pnet_split(spec, ws)
    siren = spec.p_siren
    return _mlp_tangents2((first, hidden, bott), siren, spec.omega_p if siren else 1.0, "sine" if siren else spec.p_act, spec.p_res,
                          p, cj, ck)


def hessian_analytic(spec, ws, inputs, y_index, x_index):
    """HessianLayer (gradient.py:130-180, :234-261) for ANY columns of the model input, by second-order forward-mode tangents:
    returns (y [B, so], dy/dx [B, ny, nx], d2y/dx2 [B, ny, nx, nx]).  Per layer a = w0 h W + b with per-sample W, b = slices of
    pnet_output(p); for the pair (j, k), with ' the first- and '' the second-order tangents,
        a'_j  = w0 (h'_j W + h W'_j) + b'_j
        a''   = w0 (h'' W + h'_j W'_k + h'_k W'_j + h W'') + b''          W' = z' Wh , W'' = z'' Wh  (zero for coordinate columns)
        h'    = f'(a) a' ,  h'' = f'(a) a'' + f''(a) a'_j a'_k
    (the reference formulation with the weights' tangents materialised).  Pinned by central differences of jacobian_analytic
    and by torch autograd over all input columns."""
    if spec.kind == KIND_LL:
        return _hessian_ll(spec, ws, inputs, y_index, x_index)
    assert spec.kind in (KIND_NIF, KIND_MS)
    nif = spec.kind == KIND_NIF
    p = inputs[:, :spec.pi]
    x = inputs[:, spec.pi:spec.pi + spec.si]
    B = x.shape[0]
    si, so, n = spec.si, spec.so, spec.n
    pout, z = pnet_forward(spec, ws, p)
    last = _pnet_split(spec, ws)[3]
    sl = spec.slices()
    om = 1.0 if nif else spec.omega_s
    name = spec.s_act if nif else "sine"
    f, df = act_fn(name)
    d2f = act_d2(name)

    def carve(po_):
        W_ = [po_[:, sl["w1"][0]:sl["w1"][1]].reshape(B, si, n)] + [po_[:, a_:b_].reshape(B, n, n) for (a_, b_) in sl["wh"]]
        b_ = [po_[:, sl["b1"][0]:sl["b1"][1]]] + [po_[:, a_:b_] for (a_, b_) in sl["bh"]]
        return W_, b_, po_[:, sl["wl"][0]:sl["wl"][1]].reshape(B, n, so), po_[:, sl["bl"][0]:sl["bl"][1]]
    P0 = carve(pout)
    zero = carve(np.zeros_like(pout))
    u = shapenet_given_w(spec, x, pout)
    nx = len(x_index)
    yi = list(y_index)
    J = np.zeros((B, len(yi), nx), dtype=u.dtype)
    H = np.zeros((B, len(yi), nx, nx), dtype=u.dtype)

    def seed(col):
        if col >= spec.pi:
            return np.tile(np.eye(si, dtype=x.dtype)[col - spec.pi], (B, 1))
        return np.zeros((B, si), dtype=x.dtype)
    for jj in range(nx):
        for kk in range(jj, nx):
            cj, ck = x_index[jj], x_index[kk]
            assert 0 <= cj < spec.pi + si and 0 <= ck < spec.pi + si
            Pj = Pk = Pjk = zero
            if cj < spec.pi or ck < spec.pi:
                _, zj, zk, zjk = pnet_tangents2(spec, ws, p, min(cj, spec.pi - 1), min(ck, spec.pi - 1))
                if cj < spec.pi:
                    Pj = carve(zj @ last[0])
                if ck < spec.pi:
                    Pk = carve(zk @ last[0])
                if cj < spec.pi and ck < spec.pi:
                    Pjk = carve(zjk @ last[0])
            h, hj, hk, hjk = x, seed(cj), seed(ck), np.zeros((B, si), dtype=x.dtype)
            blk = None
            for l in range(len(P0[0])):
                W, Wj, Wk, Wjk = P0[0][l], Pj[0][l], Pk[0][l], Pjk[0][l]
                a = om * _ein(h, W) + P0[1][l]
                aj = om * (_ein(hj, W) + _ein(h, Wj)) + Pj[1][l]
                ak = om * (_ein(hk, W) + _ein(h, Wk)) + Pk[1][l]
                ajk = om * (_ein(hjk, W) + _ein(hj, Wk) + _ein(hk, Wj) + _ein(h, Wjk)) + Pjk[1][l]
                t, tj, tk, tjk = f(a), df(a) * aj, df(a) * ak, df(a) * ajk + d2f(a) * aj * ak
                if nif and l >= 1:
                    h, hj, hk, hjk = h + t, hj + tj, hk + tk, hjk + tjk
                elif spec.s_res and l >= 1:
                    if (l - 1) % 2 == 0:
                        blk = (h, hj, hk, hjk)
                        h, hj, hk, hjk = t, tj, tk, tjk
                    else:
                        h, hj, hk, hjk = 0.5 * (blk[0] + t), 0.5 * (blk[1] + tj), 0.5 * (blk[2] + tk), 0.5 * (blk[3] + tjk)
                else:
                    h, hj, hk, hjk = t, tj, tk, tjk
            uj = _ein(hj, P0[2]) + _ein(h, Pj[2]) + Pj[3]
            uk = _ein(hk, P0[2]) + _ein(h, Pk[2]) + Pk[3]
            ujk = _ein(hjk, P0[2]) + _ein(hj, Pk[2]) + _ein(hk, Pj[2]) + _ein(h, Pjk[2]) + Pjk[3]
            J[:, :, jj] = uj[:, yi]
            J[:, :, kk] = uk[:, yi]
            H[:, :, jj, kk] = H[:, :, kk, jj] = ujk[:, yi]
    return u, J, H


def _hessian_ll(spec, ws, inputs, y_index, x_index):
    """HessianLayer on the last-layer-parameterised class (model.py:1219-1269): u_i = sum_c phi[i,c](x) a_c(p) + bias_i: coordinate
    derivatives are those of the shared SIREN ShapeNet x -> phi (second-order forward mode through its plain / resblock layers,
    siren.py:256-281, :381-410), parameter derivatives those of a = z last_w + last_b (pnet_tangents2):
    d2u/dx dx' = phi''.a ,  d2u/dx dp = phi'_x.a'_p ,  d2u/dp dp' = phi.a''."""
    p = inputs[:, :spec.pi]
    x = inputs[:, spec.pi:spec.pi + spec.si]
    a_out, _ = pnet_forward(spec, ws, p)
    _, _, _, last, rest = _pnet_split(spec, ws)
    first, hidden, bott, bias = _snet_split(spec, rest)
    om = spec.omega_s
    B = x.shape[0]
    nx = len(x_index)
    yi = list(y_index)
    u = np.einsum("bsj,bj->bs", snet_phi(spec, ws, x), a_out) + bias
    J = np.zeros((B, len(yi), nx), dtype=u.dtype)
    H = np.zeros((B, len(yi), nx, nx), dtype=u.dtype)
    zeros_x = np.zeros((1, spec.n))
    for jj in range(nx):
        for kk in range(jj, nx):
            cj, ck = x_index[jj], x_index[kk]
            assert 0 <= cj < spec.pi + spec.si and 0 <= ck < spec.pi + spec.si
            # ShapeNet side: phi, phi'_j, phi'_k, phi''_jk (zero tangents for parameter columns)
            xs = lambda c: (first[0][c - spec.pi][None, :] if c >= spec.pi else zeros_x)
            phi, phj, phk, phjk = _mlp_tangents2_seeded((first, hidden, bott), om, spec.s_res, x, xs(cj), xs(ck))
            # ParameterNet side: a, a'_j, a'_k, a''_jk (zero for coordinate columns)
            aj = ak = ajk = np.zeros_like(a_out)
            if cj < spec.pi or ck < spec.pi:
                _, zj, zk, zjk = pnet_tangents2(spec, ws, p, min(cj, spec.pi - 1), min(ck, spec.pi - 1))
                if cj < spec.pi:
                    aj = zj @ last[0]
                if ck < spec.pi:
                    ak = zk @ last[0]
                if cj < spec.pi and ck < spec.pi:
                    ajk = zjk @ last[0]
            c = lambda ph, av: np.einsum("bsj,bj->bs", ph.reshape(B, spec.so, spec.r), av)[:, yi]
            J[:, :, jj] = c(phj, a_out) + c(phi, aj)
            J[:, :, kk] = c(phk, a_out) + c(phi, ak)
            H[:, :, jj, kk] = H[:, :, kk, jj] = c(phjk, a_out) + c(phj, ak) + c(phk, aj) + c(phi, ajk)
    return u, J, H


def _mlp_tangents2_seeded(layers, om, res, x, wj, wk):
    """second-order forward mode through the SIREN ShapeNet of the last-layer class for first-layer tangent rows wj, wk
    (= first_w[column] for a coordinate seed, zeros for none): -> (phi, phi'_j, phi'_k, phi''_jk), flat [B, so*r]"""
    first, hidden, bott = layers

    def act3(a, aj, ak, ajk):
        return np.sin(a), np.cos(a) * aj, np.cos(a) * ak, np.cos(a) * ajk - np.sin(a) * aj * ak

    def lin(W, b, v, vj, vk, vjk):
        return om * (v @ W) + b, om * (vj @ W), om * (vk @ W), om * (vjk @ W)
    a0 = om * (x @ first[0]) + first[1]
    h, hj, hk, hjk = act3(a0, np.broadcast_to(om * wj, a0.shape), np.broadcast_to(om * wk, a0.shape), np.zeros_like(a0))
    for lay in hidden:
        if res:
            t = act3(*lin(lay[0], lay[1], h, hj, hk, hjk))
            v = act3(*lin(lay[2], lay[3], *t))
            h, hj, hk, hjk = 0.5 * (h + v[0]), 0.5 * (hj + v[1]), 0.5 * (hk + v[2]), 0.5 * (hjk + v[3])
        else:
            h, hj, hk, hjk = act3(*lin(lay[0], lay[1], h, hj, hk, hjk))
    return h @ bott[0] + bott[1], hj @ bott[0], hk @ bott[0], hjk @ bott[0]


# ----------------------------------------------------------------------------------------------
# synthetic data = verified closed form of the bundled travelling-wave datasets
# (nif/demo/dataset/*.npz) + the reference normalisers
# (nif/data/point_wise_data.py:50-114)
# ----------------------------------------------------------------------------------------------


def traveling_wave(t, x, omega=4.0):
    s = x - 0.2 - 0.006 * t
    return np.exp(-1000.0 * s * s) * np.sin(omega * s)


def standard_normalize(raw):
    """point_wise_data.py:50-78 (area_weighted=False)."""
    mean = raw.mean(axis=0)
    std = raw.std(axis=0)
    return (raw - mean) / std, mean, std


def minmax_normalize(raw, n_para, n_x, n_target):
    """point_wise_data.py:80-114 (area_weighted=False)."""
    mean = raw.mean(axis=0)
    std = raw.std(axis=0)
    for i in range(n_para + n_x):
        mean[i] = 0.5 * (np.min(raw[:, i]) + np.max(raw[:, i]))
        std[i] = 0.5 * (-np.min(raw[:, i]) + np.max(raw[:, i]))
    for j in range(n_para + n_x, n_para + n_x + n_target):
        std[j] = np.max(np.abs(raw[:, j]))
    return (raw - mean) / std, mean, std


def synthetic_wave_batch(B, seed=0, omega=4.0, dtype=np.float32):
    """(inputs [B,2] = (t,x) normalised, y [B,1]) per SURVEY 8(d): t~U[0,90], x~U[0,1)."""
    rng = np.random.default_rng(seed)
    t = rng.uniform(0.0, 90.0, size=B)
    x = rng.uniform(0.0, 1.0, size=B)
    u = traveling_wave(t, x, omega)
    raw = np.stack([t, x, u], axis=1)
    if omega <= 10:
        data, _, _ = standard_normalize(raw)
    else:
        data, _, _ = minmax_normalize(raw, 1, 1, 1)
    return data[:, :2].astype(dtype), data[:, 2:3].astype(dtype)
