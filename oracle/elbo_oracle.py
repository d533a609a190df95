"""CPU oracle for the spatial-VAE ELBO hot path (numpy, fp32, explicit backward).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
path (``spatial_vae_amd/``, ``spatial_vae/``, ``train_*.py``): only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker.

This is a restatement, from the mathematics, of what the reference computes on
its hot path; each function cites the
reference lines it follows.  It is PINNED against the reference itself: every
case in ``tests/golden/cases.py`` was run through the reference's own
``eval_minibatch`` / ``SpatialGenerator.forward`` by ``tests/golden/gen_golden.py``
and ``tests/test_oracle_golden.py`` checks this file against those fixtures.

The backward pass is written out by hand (no autograd) because it is the
specification the HIP kernels implement: each gradient formula here has a
kernel that computes the same sum.

Layout conventions: coords (B, N, 2); z (B, Zd); targets (B, N) or (B, N, C);
parameters keyed by the reference's state-dict names.
"""
import numpy as np

F32 = np.float32
LEAKY_SLOPE = F32(0.01)          # nn.LeakyReLU default negative_slope


# --------------------------------------------------------------------------
# model description
# --------------------------------------------------------------------------
class DecoderSpec(object):
    """Constructor arguments of SpatialGenerator (/root/reference/spatial_vae/models.py:58-59)."""

    def __init__(self, latent_dim, hidden_dim, n_out=1, num_layers=1, activation="tanh",
                 softplus=False, resid=False, expand_coords=False, bilinear=False):
        self.latent_dim = latent_dim
        self.hidden_dim = hidden_dim
        self.n_out = n_out
        self.num_layers = num_layers
        self.activation = activation
        self.softplus = softplus
        self.resid = resid
        self.expand_coords = expand_coords
        self.bilinear = bilinear and latent_dim > 0
        self.in_dim = 5 if expand_coords else 2

    @staticmethod
    def from_case(case):
        return DecoderSpec(case["z_dim"], case["H"], n_out=case["n_out"], num_layers=case["L"],
                           activation=case["act"], softplus=case["softplus"], resid=case["resid"],
                           expand_coords=case["expand_coords"], bilinear=case["bilinear"])

    def hidden_names(self):
        """State-dict prefixes of the (L-1) hidden Linear layers and of the output Linear
        (index arithmetic of models.py:77-85: slot 0 is the first activation)."""
        names, idx = [], 1
        for _ in range(1, self.num_layers):
            if self.resid:
                names.append("layers.%d.linear" % idx)
                idx += 1
            else:
                names.append("layers.%d" % idx)
                idx += 2
        return names, "layers.%d" % idx


def act_forward(name, h):
    if name == "tanh":
        return np.tanh(h, dtype=F32)
    if name == "leakyrelu":
        return np.where(h > 0, h, h * LEAKY_SLOPE).astype(F32)
    if name == "relu":
        return np.maximum(h, F32(0)).astype(F32)
    if name == "sigmoid":
        return sigmoid(h)
    raise ValueError(name)


def act_grad_from_output(name, a):
    """d act / d pre-activation expressed through the activation OUTPUT a (what the
    backward kernels have in HBM).  relu'(0) = 0 and leaky'(0) = slope, as ATen's
    threshold_backward / leaky_relu_backward (x > 0 ? g : g*slope)."""
    if name == "tanh":
        return (F32(1) - a * a).astype(F32)
    if name == "leakyrelu":
        return np.where(a > 0, F32(1), LEAKY_SLOPE).astype(F32)
    if name == "relu":
        return (a > 0).astype(F32)
    if name == "sigmoid":
        return (a * (F32(1) - a)).astype(F32)
    raise ValueError(name)


def sigmoid(x):
    """fp32 1/(1+exp(-x)); rounds to exactly 1.0 for x >~ 16.64 and to 0.0 once exp(-x)
    overflows (x <~ -88.7), which is what makes BCE hit its -100 clamp."""
    with np.errstate(over="ignore"):
        return (F32(1) / (F32(1) + np.exp(-x.astype(F32), dtype=F32))).astype(F32)


# --------------------------------------------------------------------------
# A1: pose (train_mnist.py:42-74; train_galaxy.py:73-110; train_particles.py:60-97)
# --------------------------------------------------------------------------
def pose_forward(grid, theta=None, dx=None):
    """coords[b,i] = grid[i] @ [[c,s],[-s,c]] + dx[b]   (train_mnist.py:54-59, 70-74)."""
    N = grid.shape[0]
    if theta is None and dx is None:
        raise ValueError("need at least one of theta, dx (else use the grid itself)")
    B = (theta if theta is not None else dx).shape[0]
    x0 = np.broadcast_to(grid[None, :, 0], (B, N)).astype(F32)
    x1 = np.broadcast_to(grid[None, :, 1], (B, N)).astype(F32)
    if theta is not None:
        c = np.cos(theta, dtype=F32)[:, None]
        s = np.sin(theta, dtype=F32)[:, None]
        x0, x1 = c * x0 - s * x1, s * x0 + c * x1
    if dx is not None:
        x0 = x0 + dx[:, 0:1]
        x1 = x1 + dx[:, 1:2]
    return np.stack([x0, x1], 2).astype(F32)


def pose_backward(grid, theta, dcoords, has_dx):
    """Adjoint of pose_forward: dtheta[b] = sum_i <dcoords[b,i], d(rot)/dtheta applied to grid[i]>,
    ddx[b] = sum_i dcoords[b,i]."""
    dtheta = None
    if theta is not None:
        c = np.cos(theta, dtype=F32)[:, None]
        s = np.sin(theta, dtype=F32)[:, None]
        g0, g1 = grid[None, :, 0], grid[None, :, 1]
        dtheta = (dcoords[:, :, 0] * (-s * g0 - c * g1) + dcoords[:, :, 1] * (c * g0 - s * g1)).sum(1).astype(F32)
    ddx = dcoords.sum(1).astype(F32) if has_dx else None
    return dtheta, ddx


# --------------------------------------------------------------------------
# A2-A4: SpatialGenerator.forward (models.py:90-132) and its adjoint (A8)
# --------------------------------------------------------------------------
def coord_features(coords2, expand):
    """(M,2) -> (M,in_dim): [x0, x1, x0^2, x1^2, x0*x1] when expand_coords (models.py:99-102)."""
    if not expand:
        return coords2
    x0, x1 = coords2[:, 0], coords2[:, 1]
    return np.stack([x0, x1, x0 * x0, x1 * x1, x0 * x1], 1).astype(F32)


def decoder_forward(spec, P, coords, z):
    """Returns a cache with every intermediate the backward needs.

    h0 = feat W_c^T + b_c + z W_z^T [+ bilinear(feat, z)]   models.py:104-123
    a0 = act(h0); a_l = act(a_{l-1} W_l^T + b_l [+ a_{l-1}])   models.py:13-21, 77-83, 126
    o  = a_{L-1} W_o^T + b_o  (the logits);  y = sigmoid(o)     models.py:84-85
    y[...,0] = softplus(y[...,0]) when spec.softplus           models.py:129-130
    """
    B, N = coords.shape[0], coords.shape[1]
    M = B * N
    feat = coord_features(coords.reshape(M, 2).astype(F32), spec.expand_coords)
    h = feat @ P["coord_linear.weight"].T + P["coord_linear.bias"]
    if spec.latent_dim > 0:
        hz = z.astype(F32) @ P["latent_linear.weight"].T                      # (B,H), no bias
        h = (h.reshape(B, N, -1) + hz[:, None, :]).reshape(M, -1)
    if spec.bilinear:
        # nn.Bilinear: out[m,k] = sum_{p,q} feat[m,p] W[k,p,q] z[b(m),q]    models.py:114-121
        Wb = P["bilinear.weight"]
        weff = np.einsum("kpq,bq->bkp", Wb, z.astype(F32)).astype(F32)        # (B,H,in)
        hb = np.einsum("bnp,bkp->bnk", feat.reshape(B, N, -1), weff).astype(F32)
        h = h + hb.reshape(M, -1)
    acts = [act_forward(spec.activation, h.astype(F32))]
    hidden, outname = spec.hidden_names()
    for nm in hidden:
        pre = acts[-1] @ P[nm + ".weight"].T + P[nm + ".bias"]
        if spec.resid:
            pre = pre + acts[-1]
        acts.append(act_forward(spec.activation, pre.astype(F32)))
    logits = (acts[-1] @ P[outname + ".weight"].T + P[outname + ".bias"]).astype(F32)
    sig = sigmoid(logits)
    y = sig.copy()
    if spec.softplus:
        y[:, 0] = np.log1p(np.exp(sig[:, 0], dtype=F32), dtype=F32)          # softplus(beta=1) of a value in (0,1)
    C = spec.n_out
    return dict(B=B, N=N, feat=feat, coords=coords.reshape(M, 2).astype(F32), z=z, acts=acts,
                logits=logits.reshape(B, N, C), sig=sig.reshape(B, N, C), y=y.reshape(B, N, C))


def decoder_backward(spec, P, cache, dy):
    """Given dL/dy (B,N,C): gradients for every parameter, for the coordinates and for z."""
    B, N = cache["B"], cache["N"]
    M = B * N
    C = spec.n_out
    hidden, outname = spec.hidden_names()
    sig = cache["sig"].reshape(M, C)
    dsig = dy.reshape(M, C).astype(F32).copy()
    if spec.softplus:
        dsig[:, 0] = dsig[:, 0] * sigmoid(sig[:, 0])                          # softplus'(s) = sigmoid(s)
    do = (dsig * sig * (F32(1) - sig)).astype(F32)                            # sigmoid backward
    g = {}
    acts = cache["acts"]
    g[outname + ".weight"] = (do.T @ acts[-1]).astype(F32)
    g[outname + ".bias"] = do.sum(0).astype(F32)
    da = (do @ P[outname + ".weight"]).astype(F32)
    for li in range(len(hidden) - 1, -1, -1):
        nm = hidden[li]
        dh = (da * act_grad_from_output(spec.activation, acts[li + 1])).astype(F32)
        g[nm + ".weight"] = (dh.T @ acts[li]).astype(F32)
        g[nm + ".bias"] = dh.sum(0).astype(F32)
        da = (dh @ P[nm + ".weight"]).astype(F32)
        if spec.resid:
            da = da + dh
    dh0 = (da * act_grad_from_output(spec.activation, acts[0])).astype(F32)
    feat = cache["feat"]
    g["coord_linear.weight"] = (dh0.T @ feat).astype(F32)
    g["coord_linear.bias"] = dh0.sum(0).astype(F32)
    dfeat = (dh0 @ P["coord_linear.weight"]).astype(F32)
    dz = None
    if spec.latent_dim > 0:
        S = dh0.reshape(B, N, -1).sum(1).astype(F32)                          # (B,H)
        g["latent_linear.weight"] = (S.T @ cache["z"].astype(F32)).astype(F32)
        dz = (S @ P["latent_linear.weight"]).astype(F32)
    if spec.bilinear:
        Wb = P["bilinear.weight"]
        z = cache["z"].astype(F32)
        G = np.einsum("bnk,bnp->bkp", dh0.reshape(B, N, -1), feat.reshape(B, N, -1)).astype(F32)   # (B,H,in)
        g["bilinear.weight"] = np.einsum("bkp,bq->kpq", G, z).astype(F32)
        dz = dz + np.einsum("bkp,kpq->bq", G, Wb).astype(F32)
        weff = np.einsum("kpq,bq->bkp", Wb, z).astype(F32)
        dfeat = dfeat + np.einsum("bnk,bkp->bnp", dh0.reshape(B, N, -1), weff).reshape(M, -1).astype(F32)
    x = cache["coords"]
    if spec.expand_coords:
        d0 = dfeat[:, 0] + F32(2) * x[:, 0] * dfeat[:, 2] + x[:, 1] * dfeat[:, 4]
        d1 = dfeat[:, 1] + F32(2) * x[:, 1] * dfeat[:, 3] + x[:, 0] * dfeat[:, 4]
        dcoords = np.stack([d0, d1], 1)
    else:
        dcoords = dfeat
    return g, dcoords.reshape(B, N, 2).astype(F32), dz


# --------------------------------------------------------------------------
# A5: per-pixel log-likelihoods; A6: CTF application
# --------------------------------------------------------------------------
def bce_loglik(y_hat, target):
    """sum over pixels/channels of -bce with torch's clamps (F.binary_cross_entropy,
    train_mnist.py:78-81, train_galaxy.py:116-119; semantics SURVEY A.4).
    Returns per-image log-likelihood (B,) and d(loglik_b)/d(y_hat)."""
    B = y_hat.shape[0]
    s = y_hat.reshape(B, -1).astype(F32)
    t = target.reshape(B, -1).astype(F32)
    with np.errstate(divide="ignore"):
        log_s = np.maximum(np.log(s, dtype=F32), F32(-100))
        log_1s = np.maximum(np.log1p(-s, dtype=F32), F32(-100))
    ll = (t * log_s + (F32(1) - t) * log_1s).astype(F32)                      # = -bce_e
    dll = (-(s - t) / np.maximum((F32(1) - s) * s, F32(1e-12))).astype(F32)
    return ll.sum(1).astype(F32), dll.reshape(y_hat.shape)


def ctf_apply(y_mu, ctf):
    """Depth-wise cross-correlation of each image with its own filter, zero padding k//2
    (F.conv2d(..., groups=B), train_particles.py:112-119).  y_mu (B, n*n); ctf (B,1,k,k)."""
    B = y_mu.shape[0]
    n = int(np.sqrt(y_mu.shape[1]))
    k = ctf.shape[2]
    pad = k // 2
    img = np.zeros((B, n + 2 * pad, n + 2 * pad), F32)
    img[:, pad:pad + n, pad:pad + n] = y_mu.reshape(B, n, n)
    n_out = n + 2 * pad - k + 1
    out = np.zeros((B, n_out, n_out), F32)
    for u in range(k):
        for v in range(k):
            out += img[:, u:u + n_out, v:v + n_out] * ctf[:, 0, u, v][:, None, None]
    return out.reshape(B, n_out * n_out).astype(F32)


def ctf_apply_backward(dout, ctf, n):
    """Adjoint of ctf_apply with respect to y_mu."""
    B = dout.shape[0]
    k = ctf.shape[2]
    pad = k // 2
    n_out = n + 2 * pad - k + 1
    d = dout.reshape(B, n_out, n_out)
    dimg = np.zeros((B, n + 2 * pad, n + 2 * pad), F32)
    for u in range(k):
        for v in range(k):
            dimg[:, u:u + n_out, v:v + n_out] += d * ctf[:, 0, u, v][:, None, None]
    return dimg[:, pad:pad + n, pad:pad + n].reshape(B, n * n).astype(F32)


def gaussian_loglik(y_params, target, mask=None, ctf=None):
    """train_particles.py:102-139.  y_params (B, N*C) is the decoder output flattened with the
    channel index fastest; with C = 2 the reference takes the FIRST N entries as y_mu and the
    LAST N as y_logvar (so both halves interleave the two channels; quirk A.3).
    Returns per-image log-likelihood and d(loglik_b)/d(y_params)."""
    B, N = target.shape
    fit_noise = y_params.shape[1] > N
    y_mu = y_params[:, :N].astype(F32)
    n = int(np.sqrt(N))
    if ctf is not None:
        if fit_noise:
            raise RuntimeError("CTF with fit-noise is broken in the reference (SURVEY A.5): no parity target")
        y_mu = ctf_apply(y_mu, ctf)
    t = target.astype(F32)
    sel = slice(None) if mask is None else np.asarray(mask, bool)
    diff = (y_mu[:, sel] - t[:, sel]).astype(F32)
    dparams = np.zeros_like(y_params, dtype=F32)
    if fit_noise:
        y_logvar = y_params[:, N:].astype(F32)
        inv_var = np.exp(-y_logvar[:, sel], dtype=F32)
        y_var = np.exp(y_logvar[:, sel], dtype=F32)
        ll = (F32(-0.5) * (diff * diff / y_var + y_logvar[:, sel]).sum(1)).astype(F32)
        dmu_sel = (-diff / y_var).astype(F32)
        dlv_sel = (F32(-0.5) * (F32(1) - diff * diff * inv_var)).astype(F32)
        dlv = np.zeros((B, N), F32)
        dlv[:, sel] = dlv_sel
        dparams[:, N:] = dlv
    else:
        ll = (F32(-0.5) * (diff * diff).sum(1)).astype(F32)
        dmu_sel = (-diff).astype(F32)
    dmu = np.zeros((B, N), F32)
    dmu[:, sel] = dmu_sel
    if ctf is not None:
        dmu = ctf_apply_backward(dmu, ctf, n)
    dparams[:, :N] = dmu
    return ll, dparams


# --------------------------------------------------------------------------
# A7: KL terms, ELBO, and the whole minibatch (the three eval_minibatch functions)
# --------------------------------------------------------------------------
def elbo_minibatch(script, spec, P, grid, y, q_out, r, rotate=True, translate=True, dx_scale=0.1,
                   theta_prior=np.pi, z_scale=1.0, mask=None, ctf=None, theta_offset=None):
    """One ELBO minibatch from the encoder output onward, with d(-elbo)/d(everything).

    script: 'mnist' (train_mnist.py:24-90), 'galaxy' (train_galaxy.py:27-128) or
    'particles' (train_particles.py:22-148).  q_out (B, 2*inf_dim) is the encoder's raw
    output [z_mu | z_logstd] (models.py:50-52); r (B, inf_dim) the N(0,1) draw
    (train_mnist.py:38-39).  Gradients are those of loss = -elbo (train_mnist.py:147-148).
    theta_offset (B,) float64: the augmentation angles added back to theta (train_galaxy.py:84-87); q_out must then
    be the encoder's output for the ROTATED images (oracle/pil_rotate.py).
    """
    B = y.shape[0]
    inf = q_out.shape[1] // 2
    z_mu, z_logstd = q_out[:, :inf].astype(F32), q_out[:, inf:].astype(F32)
    z_std = np.exp(z_logstd, dtype=F32)
    z = (z_std * r + z_mu).astype(F32)

    off = 0
    theta = None
    kl = np.zeros(B, F32)
    s = F32(theta_prior)
    if rotate:
        theta = z[:, 0]
        if theta_offset is not None and np.any(theta_offset > 0):
            theta = theta + np.asarray(theta_offset).astype(F32)
        kl = -z_logstd[:, 0] + np.log(s, dtype=F32) + z_std[:, 0] ** 2 / F32(2) / s ** 2 - F32(0.5)
        if script == "mnist":                                                 # train_mnist.py:63 keeps the mu^2 term
            kl = kl + z_mu[:, 0] ** 2 / F32(2) / s ** 2
        off = 1
    dxv = None
    if translate:
        dxv = (z[:, off:off + 2] * F32(dx_scale)).astype(F32)
    zc0 = off + (2 if translate else 0)
    zc = (z[:, zc0:] * F32(z_scale)).astype(F32) if script != "mnist" else z[:, zc0:]
    if rotate or translate:
        coords = pose_forward(grid, theta, dxv)
    else:
        coords = np.broadcast_to(grid[None], (B,) + grid.shape).astype(F32)

    cache = decoder_forward(spec, P, coords, zc)
    y_hat = cache["y"]

    if script in ("mnist", "galaxy"):
        ll_b, dll = bce_loglik(y_hat, y)
        dll = dll.reshape(y_hat.shape)
    else:
        ll_b, dparams = gaussian_loglik(y_hat.reshape(B, -1), y.reshape(B, -1), mask=mask, ctf=ctf)
        dll = dparams.reshape(y_hat.shape)
    log_p = ll_b.mean(dtype=F32)

    # unit-normal prior on every remaining latent, translation included (train_mnist.py:84-86)
    klz = (-z_logstd[:, off:] + F32(0.5) * z_std[:, off:] ** 2 + F32(0.5) * z_mu[:, off:] ** 2 - F32(0.5)).sum(1)
    kl_div = (kl + klz).mean(dtype=F32)
    elbo = log_p - kl_div

    # ---- backward of loss = -elbo
    invB = F32(1.0 / B)
    dy = (-invB * dll).astype(F32)
    gP, dcoords, dzc = decoder_backward(spec, P, cache, dy)
    dz = np.zeros_like(z)
    if rotate or translate:
        dtheta, ddx = pose_backward(grid, theta, dcoords, translate)
        if rotate:
            dz[:, 0] = dtheta
        if translate:
            dz[:, off:off + 2] = ddx * F32(dx_scale)
    if dzc is not None:
        dz[:, zc0:] = dzc * (F32(z_scale) if script != "mnist" else F32(1))
    dmu = dz.copy()
    dlogstd = (dz * r * z_std).astype(F32)
    if rotate:
        dlogstd[:, 0] += invB * (F32(-1) + z_std[:, 0] ** 2 / s ** 2)
        if script == "mnist":
            dmu[:, 0] += invB * z_mu[:, 0] / s ** 2
    dlogstd[:, off:] += invB * (F32(-1) + z_std[:, off:] ** 2)
    dmu[:, off:] += invB * z_mu[:, off:]
    g_q_out = np.concatenate([dmu, dlogstd], 1).astype(F32)
    return dict(elbo=elbo, log_p=log_p, kl=kl_div, y_hat=y_hat, logits=cache["logits"], gP=gP,
                g_q_out=g_q_out, coords=coords, z_content=zc, loglik_b=ll_b)
