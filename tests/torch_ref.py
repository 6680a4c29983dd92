"""Independent fp64 torch restatement of the reference graph, used ONLY to cross-check the
NumPy oracle's forward and hand-derived adjoint with torch.autograd (CPU, test time).

Written separately from oracle/nif_oracle.py on purpose: it follows the reference's tensor
program literally (slice pnet_output, reshape, einsum 'ai,aij->aj': nif/model.py:253-324,
:769-954, :1219-1269) and lets autograd differentiate it, the way Keras' GradientTape does.
"""
import torch


def _act(name):
    return {
        "swish": lambda a: a * torch.sigmoid(a),
        "silu": lambda a: a * torch.sigmoid(a),
        "tanh": torch.tanh,
        "relu": torch.relu,
        "sigmoid": torch.sigmoid,
        "elu": torch.nn.functional.elu,
        "softplus": torch.nn.functional.softplus,
        "gelu": torch.nn.functional.gelu,
        "selu": torch.selu,
        "softsign": torch.nn.functional.softsign,
        "exponential": torch.exp,
        "hard_sigmoid": lambda a: torch.clamp(0.2 * a + 0.5, 0.0, 1.0),      # Keras 2.11 (torch's hardsigmoid is x / 6 + 0.5)
        "sine": torch.sin,
        "linear": lambda a: a,
        None: lambda a: a,
    }[name]


def forward(kind, cs, cp, ws, inputs):
    pi, r, nst, lst = cp["input_dim"], cp["latent_dim"], cp["units"], cp["nlayers"]
    si, so, n, L = cs["input_dim"], cs["output_dim"], cs["units"], cs["nlayers"]
    it = iter(ws)
    p = inputs[:, :pi]
    x = inputs[:, pi:pi + si]
    ms = kind != "NIF"
    p_siren = ms and cp["activation"] == "sine"
    p_res = ms and cp.get("use_resblock", False)
    if p_siren:
        om = cp["omega_0"]
        w, b = next(it), next(it)
        h = torch.sin(om * (p @ w) + b)
        for _ in range(lst):
            if p_res:
                w, b, w2, b2 = next(it), next(it), next(it), next(it)
                t = torch.sin(om * (h @ w) + b)
                h = 0.5 * (h + torch.sin(om * (t @ w2) + b2))
            else:
                w, b = next(it), next(it)
                h = torch.sin(om * (h @ w) + b)
    else:
        f = _act(cp["activation"])
        w, b = next(it), next(it)
        h = f(p @ w + b)
        for _ in range(lst):
            if p_res:
                w, b, w2, b2 = next(it), next(it), next(it), next(it)
                h = f(h + (f(h @ w + b) @ w2 + b2))
            else:
                w, b = next(it), next(it)
                h = h + f(h @ w + b)
    w, b = next(it), next(it)
    z = h @ w + b
    w, b = next(it), next(it)
    po = z @ w + b
    if kind == "NIFMultiScaleLastLayerParameterized":
        om = cs["omega_0"]
        w, b = next(it), next(it)
        h = torch.sin(om * (x @ w) + b)
        for _ in range(L):
            if cs["use_resblock"]:
                w, b, w2, b2 = next(it), next(it), next(it), next(it)
                t = torch.sin(om * (h @ w) + b)
                h = 0.5 * (h + torch.sin(om * (t @ w2) + b2))
            else:
                w, b = next(it), next(it)
                h = torch.sin(om * (h @ w) + b)
        w, b = next(it), next(it)
        phi = (h @ w + b).reshape(-1, so, r)
        bias = next(it)
        return torch.einsum("bsj,bj->bs", phi, po) + bias
    B = x.shape[0]
    res = ms and cs["use_resblock"]
    nh = 2 * L if res else L
    off = 0
    W1 = po[:, off:off + si * n].reshape(B, si, n); off += si * n
    Wh = []
    for _ in range(nh):
        Wh.append(po[:, off:off + n * n].reshape(B, n, n)); off += n * n
    Wl = po[:, off:off + n * so].reshape(B, n, so); off += n * so
    b1 = po[:, off:off + n]; off += n
    bh = []
    for _ in range(nh):
        bh.append(po[:, off:off + n]); off += n
    bl = po[:, off:]
    ein = lambda u, w: torch.einsum("ai,aij->aj", u, w)
    if not ms:
        f = _act(cs["activation"])
        u = f(ein(x, W1) + b1)
        for i in range(L):
            u = f(ein(u, Wh[i]) + bh[i]) + u
    else:
        om = cs["omega_0"]
        u = torch.sin(om * ein(x, W1) + b1)
        if res:
            for i in range(L):
                t = torch.sin(om * ein(u, Wh[2 * i]) + bh[2 * i])
                u = 0.5 * (u + torch.sin(om * ein(t, Wh[2 * i + 1]) + bh[2 * i + 1]))
        else:
            for i in range(L):
                u = torch.sin(om * ein(u, Wh[i]) + bh[i])
    return ein(u, Wl) + bl


def _loss_elem(name, e):
    if name == "mse":
        return e ** 2
    if name == "mae":
        return e.abs()
    if name == "huber":          # keras.losses.huber, delta = 1
        return torch.where(e.abs() <= 1.0, 0.5 * e ** 2, e.abs() - 0.5)
    if name == "log_cosh":
        return torch.log(torch.cosh(e))
    raise ValueError(name)


def loss_and_grad(kind, cs, cp, ws_np, inputs_np, y_np, sw_np=None, loss="mse"):
    ws = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in ws_np]
    inputs = torch.tensor(inputs_np, dtype=torch.float64)
    y = torch.tensor(y_np, dtype=torch.float64)
    u = forward(kind, cs, cp, ws, inputs)
    per = _loss_elem(loss, u - y).mean(dim=1)
    if sw_np is not None:
        per = per * torch.tensor(sw_np, dtype=torch.float64)
    loss = per.sum() / u.shape[0]
    grads = torch.autograd.grad(loss, ws, allow_unused=True)
    return (loss.item(), [g.numpy() if g is not None else None for g in grads], u.detach().numpy())


def jacobian(kind, cs, cp, ws_np, inputs_np):
    ws = [torch.tensor(w, dtype=torch.float64) for w in ws_np]
    inputs = torch.tensor(inputs_np, dtype=torch.float64, requires_grad=True)
    u = forward(kind, cs, cp, ws, inputs)
    rows = []
    for i in range(u.shape[1]):
        g, = torch.autograd.grad(u[:, i].sum(), inputs, retain_graph=True)
        rows.append(g)
    return u.detach().numpy(), torch.stack(rows, 1).numpy()


def sobolev_loss_and_grad(kind, cs, cp, ws_np, inputs_np, y_np, dydx_np, x_index, w_jac, sw_np=None):
    """What Keras does for Model(x, JacobianLayer(model, all, x_index)(x)) with loss_weights [1, w_jac]:
    autograd THROUGH the input-gradient (double backward)."""
    ws = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in ws_np]
    inputs = torch.tensor(inputs_np, dtype=torch.float64, requires_grad=True)
    y = torch.tensor(y_np, dtype=torch.float64)
    g_t = torch.tensor(dydx_np, dtype=torch.float64)
    u = forward(kind, cs, cp, ws, inputs)
    rows = []
    for i in range(u.shape[1]):
        g, = torch.autograd.grad(u[:, i].sum(), inputs, create_graph=True)
        rows.append(g[:, list(x_index)])
    J = torch.stack(rows, 1)
    per = ((u - y) ** 2).mean(dim=1) + w_jac * ((J - g_t.reshape(J.shape)) ** 2).mean(dim=(1, 2))
    if sw_np is not None:
        per = per * torch.tensor(sw_np, dtype=torch.float64)
    loss = per.sum() / u.shape[0]
    grads = torch.autograd.grad(loss, ws, allow_unused=True)
    return (loss.item(), [g.numpy() if g is not None else None for g in grads], u.detach().numpy(), J.detach().numpy())


def hessian(kind, cs, cp, ws_np, inputs_np):
    """d^2 u_i / d input_j d input_k for every output and every pair of input columns: [B, so, ncol, ncol]
    (what gradient.py:251-261 computes with two nested tapes and batch_jacobian)"""
    ws = [torch.tensor(w, dtype=torch.float64) for w in ws_np]
    inputs = torch.tensor(inputs_np, dtype=torch.float64, requires_grad=True)
    u = forward(kind, cs, cp, ws, inputs)
    B, so = u.shape
    ncol = inputs.shape[1]
    out = torch.zeros((B, so, ncol, ncol), dtype=torch.float64)
    for i in range(so):
        g, = torch.autograd.grad(u[:, i].sum(), inputs, create_graph=True)      # rows are independent: [B, ncol]
        for j in range(ncol):
            h, = torch.autograd.grad(g[:, j].sum(), inputs, retain_graph=True)
            out[:, i, j, :] = h
    return out.detach().numpy()


def latent(kind, cs, cp, ws, p):
    """the ParameterNet up to the bottleneck: p [B, pi] -> z [B, r] (same tensor program as forward())"""
    lst = cp["nlayers"]
    it = iter(ws)
    ms = kind != "NIF"
    p_siren = ms and cp["activation"] == "sine"
    p_res = ms and cp.get("use_resblock", False)
    if p_siren:
        om = cp["omega_0"]
        w, b = next(it), next(it)
        h = torch.sin(om * (p @ w) + b)
        for _ in range(lst):
            if p_res:
                w, b, w2, b2 = next(it), next(it), next(it), next(it)
                t = torch.sin(om * (h @ w) + b)
                h = 0.5 * (h + torch.sin(om * (t @ w2) + b2))
            else:
                w, b = next(it), next(it)
                h = torch.sin(om * (h @ w) + b)
    else:
        f = _act(cp["activation"])
        w, b = next(it), next(it)
        h = f(p @ w + b)
        for _ in range(lst):
            if p_res:
                w, b, w2, b2 = next(it), next(it), next(it), next(it)
                h = f(h + (f(h @ w + b) @ w2 + b2))
            else:
                w, b = next(it), next(it)
                h = h + f(h @ w + b)
    w, b = next(it), next(it)
    return h @ w + b


def jac_reg_loss_and_grad(kind, cs, cp, ws_np, p_np, l1):
    """JacRegLatentLayer (gradient.py:52-127): l1 * reduce_mean(square(d latent / d p)), differentiated through the inner
    Jacobian like Keras does"""
    ws = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in ws_np]
    p = torch.tensor(p_np, dtype=torch.float64, requires_grad=True)
    z = latent(kind, cs, cp, ws, p)
    rows = []
    for c in range(z.shape[1]):
        g, = torch.autograd.grad(z[:, c].sum(), p, create_graph=True)
        rows.append(g)
    J = torch.stack(rows, 1)                  # [B, r, pi]
    loss = l1 * (J ** 2).mean()
    grads = torch.autograd.grad(loss, ws, allow_unused=True)
    return loss.item(), [g.numpy() if g is not None else None for g in grads]
