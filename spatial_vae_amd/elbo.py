"""ELBO of one minibatch -- the three eval_minibatch functions of the reference, host side.

  eval_minibatch_mnist      <- /root/reference/train_mnist.py:24-90
  eval_minibatch_galaxy     <- /root/reference/train_galaxy.py:27-128
  eval_minibatch_particles  <- /root/reference/train_particles.py:22-148

Same positional arguments and return tuples.  The encoder is the caller's torch module; from its output
onward everything -- reparameterisation, pose split and KL terms (one small kernel), pose, decoder and
log-likelihood (everything that touches B*N rows) -- goes through the HIP library (ops.py).  Differences from the
reference, all additive: `noise=` lets a caller supply the N(0,1) draw (parity tests; data-parallel
runs that slice one global draw), `return_logits=` also returns the pre-Sigmoid output, `offset=` supplies the
augmentation angles (radians) instead of the np.random draw.
`augment_rotation` (train_galaxy.py:41-54, train_particles.py:31-43) rotates the observed images before inference
with the device restatement of Pillow's bicubic Image.rotate (ops.rotate_augment) instead of a per-image PIL loop on
the host; the angles come from np.random exactly as in the reference.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops


_ENC_ACT_NAMES = {nn.Tanh: "tanh", nn.LeakyReLU: "leakyrelu", nn.ReLU: "relu", nn.Sigmoid: "sigmoid"}


def _fusable_act(m):
    """Name of an activation module the small-batch Linear kernel can fold into its epilogue, else None."""
    name = _ENC_ACT_NAMES.get(type(m))
    if name == "leakyrelu" and m.negative_slope != 0.01:
        return None
    return name


def _encode(q_net, y2d):
    """Raw encoder output [z_mu | z_logstd] (B, 2*inf_dim): InferenceNetwork.forward is layers(x) split in two
    (models.py:46-54); any other encoder's two outputs are concatenated back.

    On the device the plain Linear layers of InferenceNetwork.layers run through ops.enc_linear -- one launch per layer
    forward (bias + activation in the epilogue) and one backward (dW, db, dx) -- with the gradient sinks of dp.TrainStep when
    they are set (views into its flat buffer: no AccumulateGrad adds).  Layers the kernel does not cover (ResidLinear, weights
    above 4 M elements such as the galaxy encoder's 49 152 x 5 000 first layer) stay torch ops (ops.sink_linear / the module)."""
    if hasattr(q_net, "layers") and hasattr(q_net, "latent_dim"):
        sinks = getattr(q_net, "_grad_sinks", None) or {}
        sinks = sinks if torch.is_grad_enabled() else {}
        if not y2d.is_cuda:
            return q_net.layers(y2d)
        mods = list(q_net.layers)
        h = y2d
        idx = 0
        while idx < len(mods):
            m = mods[idx]
            if isinstance(m, nn.Linear) and m.bias is not None:
                sw, sb = sinks.get("layers.%d.weight" % idx), sinks.get("layers.%d.bias" % idx)
                if ops.enc_linear_applies(h, m.weight):
                    act = _fusable_act(mods[idx + 1]) if idx + 1 < len(mods) else None
                    h = ops.enc_linear(h, m.weight, m.bias, act, sw, sb)
                    idx += 2 if act is not None else 1
                    continue
                if sw is not None:
                    # data parallel: a layer this large exchanges the two factors of its weight gradient, not the gradient
                    coll = sinks.get("__lowrank__")
                    key = "layers.%d" % idx
                    h = ops.sink_linear(h, m.weight, m.bias, sw, sb, coll if (coll is not None and coll.has(key)) else None, key)
                    idx += 1
                    continue
            h = m(h)
            idx += 1
        return h
    z_mu, z_logstd = q_net(y2d)
    return torch.cat([z_mu, z_logstd], 1)


def draw_offsets(rotate, B):
    """The augmentation angles of train_galaxy.py:43-46 / train_particles.py:33-36 from np.random, in the reference's
    draw order."""
    offset = np.random.uniform(0, 2 * np.pi, size=B)
    if rotate < 1:
        offset = offset * np.random.binomial(1, p=rotate, size=B)
    return offset


def _augment(script, y, rotate, offset):
    """The reference's augmentation block: offset ~ U(0, 2 pi) per image from np.random (times a Bernoulli(rotate) when
    rotate is a probability < 1), image i rotated by offset[i] -- through uint8 for galaxy images
    (train_galaxy.py:44-54), as float32 for particles (train_particles.py:35-43)."""
    B = y.size(0)
    if offset is None:
        offset = draw_offsets(rotate, B)
    offset = np.asarray(offset, np.float64)
    side = int(np.sqrt(y.size(1)))
    return ops.rotate_augment(y, offset, side, side, quantize_u8=(script == "galaxy")), offset


def _core(script, x, y, p_net, q_net, rotate, translate, dx_scale, theta_prior, z_scale, mask, ctf, noise, use_cuda,
          augment_rotation=False, offset=None):
    B = y.size(0)
    if use_cuda:
        y = y.cuda()
    y_in = y
    if rotate and augment_rotation:
        y_in, offset = _augment(script, y, rotate, offset)
    else:
        offset = None
    q_out = _encode(q_net, y_in.view(B, -1))
    inf_dim = q_out.size(1) // 2
    # E_q[log p(x|z)] by one reparameterised sample (train_mnist.py:36-39); the draw itself stays a torch call
    r = noise if noise is not None else torch.empty(B, inf_dim, device=x.device, dtype=q_out.dtype).normal_()
    # reparameterise + pose split + KL terms in one kernel (train_mnist.py:33-39, 42-72, 61-63, 83-85)
    theta, dx, zc, kl_b = ops.latent_head(q_out, r, rotate, translate, script == "mnist", dx_scale, z_scale, theta_prior)
    if offset is not None and np.any(offset > 0):
        # invert the random rotation: reconstruct the original with the offset added (train_galaxy.py:84-87)
        theta = theta + torch.from_numpy(offset).float().to(theta.device)

    loglik = None
    if hasattr(p_net, "forward_posed"):
        if script == "particles" or getattr(p_net, "softplus", False):
            y_hat, logits = p_net.forward_posed(x, B, theta=theta, dx=dx, z=zc, return_logits=True)
        else:   # Bernoulli likelihood: scored inside the decoder call (no second pass over y_hat, no scaling pass backward)
            y_hat, logits, loglik = p_net.forward_posed(x, B, theta=theta, dx=dx, z=zc, bce_target=y)
    else:                                           # --vanilla baseline: ignores coordinates
        y_hat, logits = p_net(x, zc), None

    if script == "particles":
        loglik = ops.gaussian_loglik(y_hat.reshape(B, -1), y.view(B, -1), mask=mask, ctf=ctf)
    elif loglik is None:
        loglik = ops.bce_loglik(y_hat.reshape(B, -1), y.reshape(B, -1))
    elbo, log_p_x_g_z, kl_div = ops.elbo_head(loglik, kl_b)     # the two batch means and their difference, one kernel
    return elbo, log_p_x_g_z, kl_div, y_hat, logits


def eval_minibatch_mnist(x, y, p_net, q_net, rotate=True, translate=True, dx_scale=0.1, theta_prior=math.pi,
                         use_cuda=False, noise=None, return_logits=False):
    elbo, log_p, kl, y_hat, logits = _core("mnist", x, y, p_net, q_net, rotate, translate, dx_scale, theta_prior, 1,
                                           None, None, noise, use_cuda)
    out = (elbo, log_p, kl, y_hat.view(y.size(0), -1))
    return out + (logits,) if return_logits else out


def eval_minibatch_galaxy(x, y, p_net, q_net, rotate=True, translate=True, dx_scale=0.1, theta_prior=math.pi,
                          augment_rotation=False, z_scale=1, use_cuda=False, noise=None, return_logits=False, offset=None):
    channels = y.size(2)
    elbo, log_p, kl, y_hat, logits = _core("galaxy", x, y, p_net, q_net, rotate, translate, dx_scale, theta_prior,
                                           z_scale, None, None, noise, use_cuda, augment_rotation, offset)
    out = (elbo, log_p, kl, y_hat.view(y.size(0), -1, channels))
    return out + (logits,) if return_logits else out


def eval_minibatch_particles(x, y, mask, ctf, p_net, q_net, rotate=True, translate=True, dx_scale=0.1,
                             theta_prior=math.pi, augment_rotation=False, z_scale=1, use_cuda=False, noise=None,
                             return_logits=False, offset=None):
    elbo, log_p, kl, y_hat, logits = _core("particles", x, y, p_net, q_net, rotate, translate, dx_scale, theta_prior,
                                           z_scale, mask, ctf, noise, use_cuda, augment_rotation, offset)
    out = (elbo, log_p, kl)
    return out + (logits,) if return_logits else out


# ---------------------------------------------------------------- forward-only paths (image dumps of the training scripts)
def _decode_unposed(x, y, p_net, q_net, rotate, translate, z_scale, use_cuda, noise):
    B = y.size(0)
    if use_cuda:
        y = y.cuda()
    q_out = _encode(q_net, y.view(B, -1))
    inf_dim = q_out.size(1) // 2
    r = noise if noise is not None else torch.empty(B, inf_dim, device=x.device, dtype=q_out.dtype).normal_()
    # sample z, then drop the rotation and translation slots: the image is drawn on the UNposed grid
    _, _, zc, _ = ops.latent_head(q_out, r, rotate, translate, False, 1.0, z_scale, math.pi)
    if hasattr(p_net, "forward_posed"):
        return p_net.forward_posed(x, B, z=zc)
    return p_net(x, zc)


@torch.no_grad()
def minibatch_for_display(x, y, p_net, q_net, rotate=True, translate=True, z_scale=1, use_cuda=False, noise=None):
    """train_mnist.py:93-124: reconstruct from the content latents only (pose removed) -> (B, N)."""
    return _decode_unposed(x, y, p_net, q_net, rotate, translate, z_scale, use_cuda, noise).view(y.size(0), -1)


@torch.no_grad()
def minibatch_for_display_galaxy(x, y, q_net, p_net, rotate=True, translate=True, z_scale=1, use_cuda=False, noise=None):
    """train_galaxy.py:131-163 (note the reference's argument order: q_net before p_net) -> (B, N, C)."""
    return _decode_unposed(x, y, p_net, q_net, rotate, translate, z_scale, use_cuda, noise).view(y.size(0), -1, y.size(2))


@torch.no_grad()
def random_minibatch_generator(x, y, p_net, z_dim, z_scale=1, use_cuda=False, noise=None):
    """train_galaxy.py:166-183: decode z ~ N(0, 1) * z_scale on the unposed grid -> (B, N, C)."""
    B = y.size(0)
    z = noise if noise is not None else torch.empty(B, z_dim, device=x.device, dtype=torch.float32).normal_()
    z = z * z_scale
    out = p_net.forward_posed(x, B, z=z) if hasattr(p_net, "forward_posed") else p_net(x, z)
    return out.view(B, -1, y.size(2))
