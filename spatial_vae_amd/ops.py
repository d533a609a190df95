"""torch.autograd plumbing around the C-ABI (include/svae.h).

PyTorch is used here for device memory, the current HIP stream and the autograd graph only; all
arithmetic of the decoder path happens in the HIP library.  There is deliberately NO fallback:
CPU tensors or a missing library raise.
"""
import ctypes
import math
from collections import namedtuple

import numpy as np
import torch

from . import _lib

DecoderSpec = namedtuple("DecoderSpec", "latent_dim hidden_dim n_out num_layers act softplus resid expand_coords bilinear")

_ws_cache = {}


def _require_hip(t, what):
    if not t.is_cuda:
        raise RuntimeError("spatial_vae_amd: %s must live on a HIP device (got %s); the MI355X path has no CPU "
                           "fallback" % (what, t.device))


def _buf(device, nbytes, key):
    """Grow-only scratch buffer per (device, stream, key); 256-byte aligned by the caching allocator.  Keyed by the current
    stream: the library's scratch may be reused as soon as the call's work has completed ON ITS STREAM, so two streams must
    not share one."""
    k = (device, torch.cuda.current_stream(device).cuda_stream, key)
    t = _ws_cache.get(k)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[k] = t
    return t


def _f32(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def make_desc(spec, B, N):
    d = _lib.Desc()
    d.B, d.N, d.H, d.L = int(B), int(N), int(spec.hidden_dim), int(spec.num_layers)
    d.Zd, d.C = int(spec.latent_dim), int(spec.n_out)
    d.in_dim = 5 if spec.expand_coords else 2
    d.act = _lib.ACT[spec.act]
    bil = bool(spec.bilinear) and spec.latent_dim > 0
    d.flags = (_lib.FLAG_RESID if spec.resid else 0) | (_lib.FLAG_BILINEAR if bil else 0) | \
              (_lib.FLAG_SOFTPLUS if spec.softplus else 0)
    return d


def _fill_params(struct, coord_w, coord_b, latent_w, bilinear_w, out_w, out_b, hidden):
    struct.coord_w, struct.coord_b = _p(coord_w), _p(coord_b)
    struct.latent_w, struct.bilinear_w = _p(latent_w), _p(bilinear_w)
    struct.out_w, struct.out_b = _p(out_w), _p(out_b)
    for l in range(len(hidden) // 2):
        struct.hidden_w[l] = _p(hidden[2 * l])
        struct.hidden_b[l] = _p(hidden[2 * l + 1])
    return struct


class _Decoder(torch.autograd.Function):
    """y, logits[, loglik] = decoder(coords | grid+theta+dx, z; parameters)   (svae_decoder_forward[_bce] / _backward).

    With `target` (a Bernoulli observation, same shape as y) the per-image log-likelihood of train_mnist.py:78-81 comes out
    of the same call (svae_decoder_forward_bce): no bce kernel, and the backward pass hands the kept d(loglik)/d(y) plus
    the upstream gradient of loglik to svae_decoder_backward as (dy, dy_scale) -- no elementwise multiply in between."""

    # positions of the tensor arguments in forward(): spec, B, sinks, target come first
    _ARG0 = 4

    @staticmethod
    def forward(ctx, spec, B, sinks, target, coords, grid, theta, dx, z, coord_w, coord_b, latent_w, bilinear_w, out_w, out_b, *hidden):
        L = _lib.lib()
        ref = coords if coords is not None else grid
        _require_hip(ref, "coordinates")
        device = ref.device
        coords, grid, theta, dx, z = _f32(coords), _f32(grid), _f32(theta), _f32(dx), _f32(z)
        coord_w, coord_b, latent_w, bilinear_w = _f32(coord_w), _f32(coord_b), _f32(latent_w), _f32(bilinear_w)
        out_w, out_b = _f32(out_w), _f32(out_b)
        hidden = tuple(_f32(h) for h in hidden)
        for t in (z, coord_w, out_w) + hidden:
            if t is not None:
                _require_hip(t, "decoder inputs and parameters")
        N = coords.shape[1] if coords is not None else grid.shape[0]
        if len(hidden) != 2 * (spec.num_layers - 1):
            raise RuntimeError("expected %d hidden tensors, got %d" % (2 * (spec.num_layers - 1), len(hidden)))
        if spec.latent_dim > 0 and (z is None or tuple(z.shape) != (B, spec.latent_dim)):
            raise RuntimeError("z must be (%d, %d), got %s" % (B, spec.latent_dim, None if z is None else tuple(z.shape)))
        desc = make_desc(spec, B, N)
        params = _fill_params(_lib.Params(), coord_w, coord_b, latent_w if spec.latent_dim > 0 else None,
                              bilinear_w if (desc.flags & _lib.FLAG_BILINEAR) else None, out_w, out_b, hidden)
        pose = _lib.Pose()
        pose.coords, pose.grid, pose.theta, pose.dx = _p(coords), _p(grid), _p(theta), _p(dx)
        need_grad = any(ctx.needs_input_grad)
        ws_bytes = L.svae_workspace_bytes(ctypes.byref(desc))
        ws = _buf(device, ws_bytes, "ws")
        saved = None
        if need_grad:
            saved = torch.empty(max(L.svae_saved_bytes(ctypes.byref(desc)), 256), dtype=torch.uint8, device=device)
        y = torch.empty((B, N, spec.n_out), dtype=torch.float32, device=device)
        logits = torch.empty_like(y)
        loglik = dll = None
        with torch.cuda.device(device):
            if target is None:
                _lib.check(L.svae_decoder_forward(ctypes.byref(desc), ctypes.byref(params), ctypes.byref(pose), _p(z),
                                                  y.data_ptr(), logits.data_ptr(), _p(saved), ws.data_ptr(), ws.numel(),
                                                  _stream(device)))
            else:
                target = _f32(target)
                _require_hip(target, "target")
                if target.numel() != y.numel():
                    raise RuntimeError("target shape %s does not match the decoder output %s" % (tuple(target.shape), tuple(y.shape)))
                loglik = torch.empty(B, dtype=torch.float32, device=device)
                dll = torch.empty_like(y) if need_grad else None
                _lib.check(L.svae_decoder_forward_bce(ctypes.byref(desc), ctypes.byref(params), ctypes.byref(pose), _p(z),
                                                      target.data_ptr(), y.data_ptr(), logits.data_ptr(), loglik.data_ptr(),
                                                      _p(dll), _p(saved), ws.data_ptr(), ws.numel(), _stream(device)))
        ctx.spec, ctx.B, ctx.N = spec, B, N
        ctx.sinks = sinks
        # everything the backward call reads goes through save_for_backward, so the buffers have the lifetime the reference's
        # autograd graph gives its activations (spatial_vae/models.py:90-132 is plain autograd): released after the first
        # backward(), kept under retain_graph=True (any number of backward passes), and a second backward() without it raises
        # torch's own "backward through the graph a second time" error
        ctx.save_for_backward(coords, grid, theta, dx, z, coord_w, coord_b, latent_w, bilinear_w, out_w, out_b, logits, saved,
                              dll, *hidden)
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)      # an unused output (y when only loglik feeds the loss) arrives as None
        if loglik is None:
            return y, logits
        return y, logits, loglik

    @staticmethod
    def backward(ctx, dy, _dlogits, g_loglik=None):
        L = _lib.lib()
        spec, B, N = ctx.spec, ctx.B, ctx.N
        (coords, grid, theta, dx, z, coord_w, coord_b, latent_w, bilinear_w, out_w, out_b, logits, saved_buf, dll,
         *hidden) = ctx.saved_tensors
        hidden = tuple(hidden)
        device = logits.device
        # what reaches the kernels: dy (B, N, C) and an optional per-image factor dy_scale (B)
        dy_scale = None
        if dll is not None and g_loglik is not None:
            if dy is None:
                dy, dy_scale = dll, _f32(g_loglik).reshape(-1)          # the fused loss: d(loglik_b)/dy times its upstream gradient
            else:
                dy = _f32(dy) + dll * g_loglik.reshape(-1, 1, 1)        # y_hat is ALSO used downstream: add the two paths
        elif dy is None:
            dy = torch.zeros((B, N, spec.n_out), dtype=torch.float32, device=device)
        dy = _f32(dy)
        desc = make_desc(spec, B, N)
        bil = bool(desc.flags & _lib.FLAG_BILINEAR)
        params = _fill_params(_lib.Params(), coord_w, coord_b, latent_w if spec.latent_dim > 0 else None,
                              bilinear_w if bil else None, out_w, out_b, hidden)
        pose = _lib.Pose()
        pose.coords, pose.grid, pose.theta, pose.dx = _p(coords), _p(grid), _p(theta), _p(dx)
        # (spec, B, sinks, target, coords, grid, theta, dx, z, coord_w, coord_b, latent_w, bilinear_w, out_w, out_b, *hidden)
        ng = ctx.needs_input_grad[_Decoder._ARG0:]
        sinks = ctx.sinks or {}

        def new(t, want, key=None):
            """Gradient buffer: the caller's sink (a view of a flat gradient buffer the kernels write
            into directly, see dp.FlatGrads) when one is registered for this parameter, else fresh."""
            if t is None or not want:
                return None
            sk = sinks.get(key) if key is not None else None
            if sk is not None and sk.shape == t.shape and sk.dtype == torch.float32 and sk.is_contiguous():
                return sk
            return torch.empty_like(t)

        g_coords, g_theta, g_dx, g_z = new(coords, ng[0]), new(theta, ng[2]), new(dx, ng[3]), new(z, ng[4] and spec.latent_dim > 0)
        g_cw, g_cb = new(coord_w, ng[5], "coord_w"), new(coord_b, ng[6], "coord_b")
        g_lw = new(latent_w, ng[7] and spec.latent_dim > 0, "latent_w")
        g_bw = new(bilinear_w, ng[8] and bil, "bilinear_w")
        g_ow, g_ob = new(out_w, ng[9], "out_w"), new(out_b, ng[10], "out_b")
        g_hidden = tuple(new(h, ng[11 + i], "hidden%d" % i) for i, h in enumerate(hidden))
        grads = _fill_params(_lib.Params(), g_cw, g_cb, g_lw, g_bw, g_ow, g_ob, g_hidden)
        pg = _lib.PoseGrads()
        pg.dcoords, pg.dtheta, pg.ddx = _p(g_coords), _p(g_theta), _p(g_dx)
        ws_bytes = L.svae_workspace_bytes(ctypes.byref(desc))
        ws = _buf(device, ws_bytes, "ws")
        with torch.cuda.device(device):
            _lib.check(L.svae_decoder_backward(ctypes.byref(desc), ctypes.byref(params), ctypes.byref(pose), _p(z),
                                               logits.data_ptr(), dy.data_ptr(), _p(dy_scale), saved_buf.data_ptr(),
                                               ctypes.byref(grads), _p(g_z), ctypes.byref(pg), ws.data_ptr(), ws.numel(),
                                               _stream(device)))
        ready = sinks.get("__ready__")   # dp.TrainStep: every decoder gradient is now enqueued -> start its all-reduce
        if ready is not None:
            ready()

        def ret(t, key):
            """A gradient the kernels wrote straight into the caller's sink is not handed to autograd again
            (it would clone the view and, were .grad the same view, add it to itself)."""
            return None if (t is not None and sinks.get(key) is t) else t

        return (None, None, None, None, g_coords, None, g_theta, g_dx, g_z, ret(g_cw, "coord_w"), ret(g_cb, "coord_b"),
                ret(g_lw, "latent_w"), ret(g_bw, "bilinear_w"), ret(g_ow, "out_w"), ret(g_ob, "out_b")) + \
            tuple(ret(h, "hidden%d" % i) for i, h in enumerate(g_hidden))


def decoder(spec, B, coords, grid, theta, dx, z, coord_w, coord_b, latent_w, bilinear_w, out_w, out_b, hidden, sinks=None,
            bce_target=None):
    """Returns (y, logits), each (B, N, n_out) -- and, with bce_target (B, N, n_out), also the per-image Bernoulli
    log-likelihood (B) computed inside the same call.  Exactly one of coords / grid is given.
    sinks: optional {name: tensor} of preallocated parameter-gradient buffers (names coord_w, coord_b,
    latent_w, bilinear_w, out_w, out_b, hidden0, hidden1, ...) the backward kernels write into; the entry
    "__ready__", if present, is a callable invoked once the backward launch sequence has been enqueued."""
    return _Decoder.apply(spec, B, sinks, bce_target, coords, grid, theta, dx, z, coord_w, coord_b, latent_w, bilinear_w, out_w,
                          out_b, *hidden)


class _Latent(torch.autograd.Function):
    """theta, dx, z_content, kl = latent_head(q_out, r)   (svae_latent_forward/backward)."""

    @staticmethod
    def forward(ctx, q_out, r, rotate, translate, mu_penalty, dx_scale, z_scale, theta_prior):
        L = _lib.lib()
        _require_hip(q_out, "encoder output")
        q_out, r = _f32(q_out), _f32(r)
        B, inf = r.shape
        if q_out.shape != (B, 2 * inf):
            raise RuntimeError("q_out must be (%d, %d), got %s" % (B, 2 * inf, tuple(q_out.shape)))
        d = _lib.LatentDesc(B, inf, int(bool(rotate)), int(bool(translate)), int(bool(mu_penalty)), float(dx_scale),
                            float(z_scale), float(theta_prior))
        dev = q_out.device
        zd = inf - (1 if rotate else 0) - (2 if translate else 0)
        theta = torch.empty(B, dtype=torch.float32, device=dev) if rotate else None
        dx = torch.empty(B, 2, dtype=torch.float32, device=dev) if translate else None
        zc = torch.empty(B, zd, dtype=torch.float32, device=dev)
        kl = torch.empty(B, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.svae_latent_forward(ctypes.byref(d), q_out.data_ptr(), r.data_ptr(), _p(theta), _p(dx),
                                             zc.data_ptr() if zd > 0 else None, kl.data_ptr(), _stream(dev)))
        ctx.desc, ctx.q_out, ctx.r = d, q_out, r
        return theta, dx, zc, kl

    @staticmethod
    def backward(ctx, g_theta, g_dx, g_zc, g_kl):
        L = _lib.lib()
        q_out, r = ctx.q_out, ctx.r
        g_theta, g_dx, g_zc, g_kl = _f32(g_theta), _f32(g_dx), _f32(g_zc), _f32(g_kl)
        if g_zc is not None and g_zc.numel() == 0:
            g_zc = None
        gq = torch.empty_like(q_out)
        with torch.cuda.device(q_out.device):
            _lib.check(L.svae_latent_backward(ctypes.byref(ctx.desc), q_out.data_ptr(), r.data_ptr(), _p(g_theta), _p(g_dx),
                                              _p(g_zc), _p(g_kl), gq.data_ptr(), _stream(q_out.device)))
        return gq, None, None, None, None, None, None, None


def latent_head(q_out, r, rotate, translate, mu_penalty, dx_scale, z_scale, theta_prior):
    """(theta | None, dx | None, z_content, kl_per_image) from the encoder output [z_mu | z_logstd] and the noise r."""
    return _Latent.apply(q_out, r, rotate, translate, mu_penalty, dx_scale, z_scale, theta_prior)


class _ElboHead(torch.autograd.Function):
    """elbo, log_p, kl = mean(loglik) - mean(kl_b), mean(loglik), mean(kl_b)  (svae_elbo_head_forward/backward)."""

    @staticmethod
    def forward(ctx, loglik, kl_b):
        L = _lib.lib()
        _require_hip(loglik, "loglik")
        loglik, kl_b = _f32(loglik), _f32(kl_b)
        B = loglik.numel()
        if kl_b.numel() != B:
            raise RuntimeError("loglik has %d entries, kl %d" % (B, kl_b.numel()))
        ctx.set_materialize_grads(False)     # unused outputs (log_p, kl) arrive as None, not as zero tensors autograd fills
        out = torch.empty(3, dtype=torch.float32, device=loglik.device)
        with torch.cuda.device(loglik.device):
            _lib.check(L.svae_elbo_head_forward(loglik.data_ptr(), kl_b.data_ptr(), B, out.data_ptr(), _stream(loglik.device)))
        ctx.B, ctx.shapes = B, (loglik.shape, kl_b.shape)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_elbo, g_logp, g_kl):
        L = _lib.lib()
        ref = next(g for g in (g_elbo, g_logp, g_kl) if g is not None)
        g_elbo, g_logp, g_kl = _f32(g_elbo), _f32(g_logp), _f32(g_kl)
        dl = torch.empty(ctx.B, dtype=torch.float32, device=ref.device)
        dk = torch.empty(ctx.B, dtype=torch.float32, device=ref.device)
        with torch.cuda.device(ref.device):
            _lib.check(L.svae_elbo_head_backward(_p(g_elbo), _p(g_logp), _p(g_kl), ctx.B, dl.data_ptr(), dk.data_ptr(),
                                                 _stream(ref.device)))
        return dl.view(ctx.shapes[0]), dk.view(ctx.shapes[1])


def elbo_head(loglik, kl_b):
    """(elbo, log_p_x_g_z, kl_div) of a minibatch from its per-image log-likelihoods and KL terms."""
    return _ElboHead.apply(loglik, kl_b)


ENC_ACT = {None: -1, "tanh": 0, "leakyrelu": 1, "relu": 2, "sigmoid": 3}
ENC_LINEAR_MAX_WEIGHT = 4 * 1024 * 1024     # elements: the hand-written layer is for weights of a few MB (see svae.h)


def enc_linear_applies(x, weight):
    """The small-batch Linear kernels take fp32 CUDA operands whose weight matrix has at most 4 M elements."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2 and
            weight.numel() <= ENC_LINEAR_MAX_WEIGHT)


class _EncLinear(torch.autograd.Function):
    """out = act(x W^T + b) in ONE launch (svae_linear_forward); the backward pass is ONE launch too (svae_linear_backward):
    dW, db and dx with act' applied to the upstream gradient as it is loaded.  Replaces an nn.Linear + activation pair of
    InferenceNetwork.layers (models.py:31-43) -- addmm, tanh, tanh_backward, two mm and a column sum otherwise.  With sinks
    (views of dp.FlatGrads' flat buffer) dW / db are written in place and not handed to autograd."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, sink_w, sink_b):
        L = _lib.lib()
        x, weight, bias = _f32(x), _f32(weight), _f32(bias)
        rows, k = x.shape
        n = weight.shape[0]
        out = torch.empty(rows, n, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.svae_linear_forward(x.data_ptr(), weight.data_ptr(), _p(bias), out.data_ptr(), rows, k, n, ENC_ACT[act],
                                             _stream(x.device)))
        ctx.act = act
        ctx.sinks = (sink_w, sink_b)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        x, weight, out = ctx.saved_tensors
        sink_w, sink_b = ctx.sinks
        dout = _f32(dout)
        rows, k = x.shape
        n = weight.shape[0]
        want_x, want_w, want_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if want_x else None
        dw = (sink_w if sink_w is not None else torch.empty_like(weight)) if want_w else None
        db = (sink_b if sink_b is not None else torch.empty(n, dtype=torch.float32, device=x.device)) if want_b else None
        with torch.cuda.device(x.device):
            _lib.check(L.svae_linear_backward(x.data_ptr(), weight.data_ptr(), out.data_ptr(), dout.data_ptr(), rows, k, n,
                                              ENC_ACT[ctx.act], _p(dw), _p(db), _p(dx), _stream(x.device)))
        return (dx, None if (dw is None or dw is sink_w) else dw, None if (db is None or db is sink_b) else db, None, None, None)


def enc_linear(x, weight, bias, act=None, sink_w=None, sink_b=None):
    """act(x W^T + b), act in (None, "tanh", "leakyrelu", "relu", "sigmoid") -- see _EncLinear."""
    return _EncLinear.apply(x, weight, bias, act, sink_w, sink_b)


class _SinkLinear(torch.autograd.Function):
    """y = x W^T + b; the backward pass writes dW and db straight into caller-owned gradient views (slices of the flat
    buffer of dp.FlatGrads) with torch.mm(out=) / torch.sum(out=) instead of returning tensors that autograd would then
    ADD into those views: one kernel less per parameter per step.  The arithmetic is torch's (hipBLASLt).

    With a `collector` (data-parallel runs: dp.LowRankExchange) the backward pass does not form dW at all: it hands its two
    factors (x, dy) to the collector, and the step all-gathers the factors of every rank and forms the GLOBAL dW = dy_all^T
    x_all locally -- rank <= global batch, a few MB on the wire instead of the weight's size."""

    @staticmethod
    def forward(ctx, x, weight, bias, sink_w, sink_b, collector=None, key=None):
        ctx.save_for_backward(x, weight)
        ctx.sinks = (sink_w, sink_b)
        ctx.collector, ctx.key = collector, key
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        sink_w, sink_b = ctx.sinks
        dx = dy.mm(weight) if ctx.needs_input_grad[0] else None
        if ctx.collector is not None:
            ctx.collector.add(ctx.key, x, dy)
            return dx, None, None, None, None, None, None
        torch.mm(dy.t(), x, out=sink_w)
        if dy.is_cuda and dy.dtype == torch.float32 and dy.is_contiguous() and sink_b.is_contiguous():
            with torch.cuda.device(dy.device):   # column sums in ~3 us (ATen's reduce kernel takes 13 us for 256 x 500)
                _lib.check(_lib.lib().svae_colsum(dy.data_ptr(), dy.size(0), dy.size(1), sink_b.data_ptr(), _stream(dy.device)))
        else:
            torch.sum(dy, 0, out=sink_b)
        return dx, None, None, None, None, None, None


def sink_linear(x, weight, bias, sink_w, sink_b, collector=None, key=None):
    return _SinkLinear.apply(x, weight, bias, sink_w, sink_b, collector, key)


class _BceLoglik(torch.autograd.Function):
    """loglik[b] = -sum_j bce(y_hat[b, j], target[b, j])  (svae_bce_loglik)."""

    @staticmethod
    def forward(ctx, y_hat, target):
        L = _lib.lib()
        _require_hip(y_hat, "y_hat")
        y_hat, target = _f32(y_hat), _f32(target)
        B = y_hat.shape[0]
        n = y_hat.numel() // B
        if target.numel() != y_hat.numel():
            raise RuntimeError("target shape %s does not match y_hat %s" % (tuple(target.shape), tuple(y_hat.shape)))
        loglik = torch.empty(B, dtype=torch.float32, device=y_hat.device)
        dll = torch.empty_like(y_hat) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(y_hat.device):
            _lib.check(L.svae_bce_loglik(B, n, y_hat.data_ptr(), target.data_ptr(), loglik.data_ptr(), _p(dll),
                                         _stream(y_hat.device)))
        ctx.save_for_backward(dll)
        return loglik

    @staticmethod
    def backward(ctx, g):
        dll, = ctx.saved_tensors
        return dll * g.reshape((-1,) + (1,) * (dll.dim() - 1)), None


def bce_loglik(y_hat, target):
    return _BceLoglik.apply(y_hat, target)


class _GaussianLoglik(torch.autograd.Function):
    """Per-image Gaussian log-likelihood of train_particles.py:102-139 (svae_gaussian_loglik)."""

    @staticmethod
    def forward(ctx, y_params, target, mask, ctf):
        L = _lib.lib()
        _require_hip(y_params, "y_params")
        y_params, target, ctf = _f32(y_params), _f32(target), _f32(ctf)
        B, N = target.shape
        C = y_params.shape[1] // N
        if mask is not None:
            mask = mask.to(device=y_params.device, dtype=torch.uint8).contiguous()
        k = 0 if ctf is None else int(ctf.shape[-1])
        ws_bytes = L.svae_gaussian_workspace_bytes(B, N) if ctf is not None else 0
        ws = _buf(y_params.device, ws_bytes, "gauss") if ws_bytes else None
        loglik = torch.empty(B, dtype=torch.float32, device=y_params.device)
        dll = torch.zeros_like(y_params) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(y_params.device):
            _lib.check(L.svae_gaussian_loglik(B, N, C, y_params.data_ptr(), target.data_ptr(), _p(mask), _p(ctf), k,
                                              loglik.data_ptr(), _p(dll), _p(ws), ws_bytes, _stream(y_params.device)))
        ctx.save_for_backward(dll)
        return loglik

    @staticmethod
    def backward(ctx, g):
        dll, = ctx.saved_tensors
        return dll * g[:, None], None, None, None


def gaussian_loglik(y_params, target, mask=None, ctf=None):
    return _GaussianLoglik.apply(y_params, target, mask, ctf)


def rotation_matrices(offset, rows, cols):
    """Host part of Image.rotate for a batch: per image the six inverse-affine coefficients PIL/Image.py derives
    from the angle (cos/sin of -radians(angle) rounded to 15 decimals about the centre (cols/2, rows/2)), and the
    quarter-turn code of Pillow's exact fast paths (-1 = general angle).  offset: radians, as drawn by
    train_galaxy.py:44; the reference passes 360*offset/2/pi degrees (train_galaxy.py:51)."""
    B = len(offset)
    mat = np.zeros((B, 6), np.float64)
    quarter = np.full(B, -1, np.int32)
    cx, cy = cols / 2, rows / 2
    for i in range(B):
        angle = (360 * float(offset[i]) / 2 / np.pi) % 360.0
        if angle in (0.0, 180.0) or (angle in (90.0, 270.0) and rows == cols):
            quarter[i] = int(angle // 90)
            continue
        a = -math.radians(angle)
        c, s = round(math.cos(a), 15), round(math.sin(a), 15)
        ms, mc = round(-math.sin(a), 15), round(math.cos(a), 15)
        tx = c * (-cx) + s * (-cy) + 0.0
        ty = ms * (-cx) + mc * (-cy) + 0.0
        mat[i] = (c, s, tx + cx, ms, mc, ty + cy)
    return mat, quarter


def rotate_augment(y, offset, rows, cols, quantize_u8):
    """Rotate image i of the batch by offset[i] radians the way the reference does with Pillow
    (train_galaxy.py:41-54 quantize_u8=True; train_particles.py:31-43 quantize_u8=False), on the device
    (svae_rotate_bicubic).  y: (B, rows*cols[, C]) fp32 CUDA tensor; returns a new tensor of the same shape."""
    _require_hip(y, "y")
    B = y.size(0)
    yc = y.contiguous().float()
    C = yc.numel() // (B * rows * cols)
    if C * B * rows * cols != yc.numel() or len(offset) != B:
        raise RuntimeError("rotate_augment: y %s does not match %d images of %dx%d" % (tuple(y.shape), len(offset), rows, cols))
    mat, quarter = rotation_matrices(offset, rows, cols)
    mat_d = torch.from_numpy(mat).to(y.device, non_blocking=True)
    q_d = torch.from_numpy(quarter).to(y.device, non_blocking=True)
    out = torch.empty_like(yc)
    with torch.cuda.device(y.device):
        _lib.check(_lib.lib().svae_rotate_bicubic(yc.data_ptr(), out.data_ptr(), mat_d.data_ptr(), q_d.data_ptr(), B, rows,
                                                  cols, C, 1 if quantize_u8 else 0, _stream(y.device)))
    return out.view_as(y)


def ctf_filter(table, n, m, scale=1.0, device=None):
    """(P, n, m) real-space CTF filters on the device from the (P, 8) parameter table of spatial_vae/ctf.py:26-30
    (svae_ctf_filter; the reference builds them one particle at a time with numpy, ctf.py:33-56)."""
    tab = torch.as_tensor(np.ascontiguousarray(table, dtype=np.float64))
    if tab.dim() != 2 or tab.size(1) != 8:
        raise RuntimeError("ctf_filter: expected a (P, 8) parameter table, got %s" % (tuple(tab.shape),))
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    tab = tab.to(dev)
    _require_hip(tab, "CTF table")
    out = torch.empty(tab.size(0), n, m, dtype=torch.float32, device=dev)
    L = _lib.lib()
    ws_bytes = L.svae_ctf_filter_workspace_bytes(tab.size(0), n, m)    # 0 while a filter's transform fits the LDS (~80 x 80)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    with torch.cuda.device(dev):
        _lib.check(L.svae_ctf_filter(tab.data_ptr(), out.data_ptr(), tab.size(0), n, m, float(scale), _p(ws), ws_bytes,
                                     _stream(dev)))
    return out


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam's update (amsgrad off, no weight decay) for ONE flat fp32 CUDA parameter, executed by
    svae_adam_step.  Same defaults and state names (step, exp_avg, exp_avg_sq) as torch.optim.Adam."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, zero_grad=False):
        """zero_grad=True: step() also clears each parameter's .grad buffer in the same kernel (the loop's
        optim.zero_grad(), train_mnist.py:150, without a second pass)."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.zero_grad_in_step = bool(zero_grad)

    @torch.no_grad()
    def step(self, closure=None):
        L = _lib.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                _require_hip(p, "FlatAdam parameter")
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FlatAdam needs contiguous fp32 parameters and gradients")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                with torch.cuda.device(p.device):
                    _lib.check(L.svae_adam_step(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                                st["exp_avg_sq"].data_ptr(), p.numel(), group["lr"], b1, b2, group["eps"],
                                                st["step"], 1 if self.zero_grad_in_step else 0, _stream(p.device)))
