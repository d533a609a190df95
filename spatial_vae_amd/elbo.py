"""ELBO of one minibatch -- the three eval_minibatch functions of the reference, host side.

  eval_minibatch_mnist      <- /root/reference/train_mnist.py:24-90
  eval_minibatch_galaxy     <- /root/reference/train_galaxy.py:27-128
  eval_minibatch_particles  <- /root/reference/train_particles.py:22-148

Same positional arguments and return tuples.  The encoder call, the reparameterisation and the
(B, inf_dim)-sized KL terms are a handful of tiny torch ops; everything that touches B*N rows --
pose, decoder, log-likelihood -- goes through the HIP library (ops.py).  Differences from the
reference, all additive: `noise=` lets a caller supply the N(0,1) draw (parity tests; data-parallel
runs that slice one global draw), `return_logits=` also returns the pre-Sigmoid output.
PIL-based rotation augmentation (`augment_rotation`) is host preprocessing outside the hot path
and is not implemented here.
"""
import math

import torch

from . import ops


def _core(script, x, y, p_net, q_net, rotate, translate, dx_scale, theta_prior, z_scale, mask, ctf, noise, use_cuda):
    B = y.size(0)
    if use_cuda:
        y = y.cuda()
    z_mu, z_logstd = q_net(y.view(B, -1))
    z_std = torch.exp(z_logstd)
    inf_dim = z_mu.size(1)
    # E_q[log p(x|z)] by one reparameterised sample (train_mnist.py:36-39)
    r = noise if noise is not None else torch.empty(B, inf_dim, device=x.device, dtype=z_mu.dtype).normal_()
    z = z_std * r + z_mu

    kl = 0
    theta = None
    off = 0
    if rotate:
        theta = z[:, 0]
        var_ratio = z_std[:, 0] ** 2
        if script == "mnist":                       # train_mnist.py:63 penalises the mean too; the others do not
            var_ratio = var_ratio + z_mu[:, 0] ** 2
        kl = -z_logstd[:, 0] + math.log(theta_prior) + var_ratio / 2 / theta_prior ** 2 - 0.5
        off = 1
    dx = None
    if translate:
        dx = z[:, off:off + 2] * dx_scale
    c0 = off + (2 if translate else 0)
    zc = z[:, c0:] * z_scale

    if hasattr(p_net, "forward_posed"):
        y_hat, logits = p_net.forward_posed(x, B, theta=theta, dx=dx, z=zc, return_logits=True)
    else:                                           # --vanilla baseline: ignores coordinates
        y_hat, logits = p_net(x, zc), None

    if script == "particles":
        loglik = ops.gaussian_loglik(y_hat.reshape(B, -1), y.view(B, -1), mask=mask, ctf=ctf)
    else:
        loglik = ops.bce_loglik(y_hat.reshape(B, -1), y.reshape(B, -1))
    log_p_x_g_z = loglik.mean()

    # unit-normal prior on every remaining latent, translation included (train_mnist.py:83-86)
    z_kl = -z_logstd[:, off:] + 0.5 * z_std[:, off:] ** 2 + 0.5 * z_mu[:, off:] ** 2 - 0.5
    kl_div = (kl + z_kl.sum(1)).mean()
    elbo = log_p_x_g_z - kl_div
    return elbo, log_p_x_g_z, kl_div, y_hat, logits


def eval_minibatch_mnist(x, y, p_net, q_net, rotate=True, translate=True, dx_scale=0.1, theta_prior=math.pi,
                         use_cuda=False, noise=None, return_logits=False):
    elbo, log_p, kl, y_hat, logits = _core("mnist", x, y, p_net, q_net, rotate, translate, dx_scale, theta_prior, 1,
                                           None, None, noise, use_cuda)
    out = (elbo, log_p, kl, y_hat.view(y.size(0), -1))
    return out + (logits,) if return_logits else out


def eval_minibatch_galaxy(x, y, p_net, q_net, rotate=True, translate=True, dx_scale=0.1, theta_prior=math.pi,
                          augment_rotation=False, z_scale=1, use_cuda=False, noise=None, return_logits=False):
    if augment_rotation:
        raise NotImplementedError("augment_rotation is host-side PIL preprocessing; rotate the batch before calling")
    channels = y.size(2)
    elbo, log_p, kl, y_hat, logits = _core("galaxy", x, y, p_net, q_net, rotate, translate, dx_scale, theta_prior,
                                           z_scale, None, None, noise, use_cuda)
    out = (elbo, log_p, kl, y_hat.view(y.size(0), -1, channels))
    return out + (logits,) if return_logits else out


def eval_minibatch_particles(x, y, mask, ctf, p_net, q_net, rotate=True, translate=True, dx_scale=0.1,
                             theta_prior=math.pi, augment_rotation=False, z_scale=1, use_cuda=False, noise=None,
                             return_logits=False):
    if augment_rotation:
        raise NotImplementedError("augment_rotation is host-side PIL preprocessing; rotate the batch before calling")
    elbo, log_p, kl, y_hat, logits = _core("particles", x, y, p_net, q_net, rotate, translate, dx_scale, theta_prior,
                                           z_scale, mask, ctf, noise, use_cuda)
    out = (elbo, log_p, kl)
    return out + (logits,) if return_logits else out
