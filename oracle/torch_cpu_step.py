"""CPU restatement of one reference training step in plain PyTorch ops (autograd + Adam).

TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/elbo_oracle.py's header): imported by tests/ and by
the ``cpu_baseline`` leg of bench.py, never by the product path.

The reference's Python files cannot travel to the GPU box, so the CPU number reported next to the
GPU number is this file timed on the box's host cores: it issues the same ATen op sequence as
train_mnist.py:24-90 + :147-150 (expand, cos/sin, bmm, add, addmm, tanh, addmm, tanh, addmm,
sigmoid, binary_cross_entropy, KL, backward, Adam.step) -- SURVEY.md section 2.2, K1-K16 -- and of its
siblings train_galaxy.py:27-128 (multi-channel BCE) and train_particles.py:22-148 (Gaussian likelihood,
fit-noise halves, grouped-conv CTF, mask), written functionally over a parameter dict instead of nn.Modules.  Pinned against the golden vectors by
tests/test_oracle_golden.py::test_torch_cpu_step_matches_reference.
"""
import math

import torch
import torch.nn.functional as F

_ACT = {"tanh": torch.tanh, "leakyrelu": lambda h: F.leaky_relu(h, 0.01), "relu": torch.relu,
        "sigmoid": torch.sigmoid}


def _hidden_names(state, resid):
    names = sorted({k.rsplit(".", 1)[0] for k in state if k.startswith("layers.")},
                   key=lambda s: int(s.split(".")[1]))
    return names[:-1], names[-1]


def encoder(qp, y, act, resid):
    """InferenceNetwork.forward (models.py:46-54)."""
    hidden, last = _hidden_names(qp, resid)
    h = _ACT[act](F.linear(y, qp[hidden[0] + ".weight"], qp[hidden[0] + ".bias"]))
    for nm in hidden[1:]:
        pre = F.linear(h, qp[nm + ".weight"], qp[nm + ".bias"])
        h = _ACT[act](pre + h) if resid else _ACT[act](pre)
    out = F.linear(h, qp[last + ".weight"], qp[last + ".bias"])
    d = out.shape[1] // 2
    return out[:, :d], out[:, d:]


def decoder(pp, x, z, act, resid=False, softplus=False, expand_coords=False):
    """SpatialGenerator.forward (models.py:90-132); x (B,N,2), z (B,Zd)."""
    b, n = x.shape[0], x.shape[1]
    xf = x.reshape(b * n, -1)
    if expand_coords:
        xf = torch.cat([xf, xf ** 2, (xf[:, 0] * xf[:, 1]).unsqueeze(1)], 1)
    h = F.linear(xf, pp["coord_linear.weight"], pp["coord_linear.bias"]).view(b, n, -1)
    if "latent_linear.weight" in pp:
        h = h + F.linear(z, pp["latent_linear.weight"]).unsqueeze(1)
    if "bilinear.weight" in pp:
        zz = z.unsqueeze(1).expand(b, n, z.shape[1]).contiguous()
        h = h + F.bilinear(xf.view(b, n, -1), zz, pp["bilinear.weight"])
    h = _ACT[act](h.view(b * n, -1))
    hidden, last = _hidden_names(pp, resid)
    for nm in hidden:
        pre = F.linear(h, pp[nm + ".weight"], pp[nm + ".bias"])
        h = _ACT[act](pre + h) if resid else _ACT[act](pre)
    y = torch.sigmoid(F.linear(h, pp[last + ".weight"], pp[last + ".bias"])).view(b, n, -1)
    if softplus:
        y = torch.cat([F.softplus(y[:, :, :1]), y[:, :, 1:]], 2)
    return y


def _posterior_and_pose(qp, x, y_flat, r, act, resid, rotate, translate, dx_scale, theta_prior, mu_penalty, z_scale):
    """Inference, reparameterised sample, pose applied to the coordinates and the KL terms -- the block the three
    eval_minibatch functions share (train_mnist.py:29-74, train_galaxy.py:60-112, train_particles.py:48-99).
    mu_penalty: train_mnist.py:62-63 keeps the mu^2 term in the KL of theta; the other two scripts drop it."""
    B = y_flat.shape[0]
    xb = x.expand(B, x.shape[0], x.shape[1])
    z_mu, z_logstd = encoder(qp, y_flat, act, resid)
    z_std = torch.exp(z_logstd)
    z = z_std * r + z_mu
    kl = 0
    if rotate:
        theta = z[:, 0]
        rot = torch.stack([torch.stack([torch.cos(theta), torch.sin(theta)], 1),
                           torch.stack([-torch.sin(theta), torch.cos(theta)], 1)], 1)
        xb = torch.bmm(xb, rot)
        second = z_std[:, 0] ** 2 + z_mu[:, 0] ** 2 if mu_penalty else z_std[:, 0] ** 2
        kl = -z_logstd[:, 0] + math.log(theta_prior) + second / 2 / theta_prior ** 2 - 0.5
        z, z_mu, z_std, z_logstd = z[:, 1:], z_mu[:, 1:], z_std[:, 1:], z_logstd[:, 1:]
    if translate:
        xb = xb + (z[:, :2] * dx_scale).unsqueeze(1)
        z = z[:, 2:]
    z = z * z_scale if z_scale != 1 else z
    kl_b = kl + (-z_logstd + 0.5 * z_std ** 2 + 0.5 * z_mu ** 2 - 0.5).sum(1)
    return xb.contiguous(), z, kl_b


def elbo_mnist(pp, qp, x, y, r, act="tanh", rotate=True, translate=True, dx_scale=0.1, theta_prior=math.pi):
    """eval_minibatch of train_mnist.py:24-90 with the noise r supplied."""
    B = y.shape[0]
    xb, z, kl_b = _posterior_and_pose(qp, x, y, r, act, False, rotate, translate, dx_scale, theta_prior, True, 1)
    y_hat = decoder(pp, xb, z, act).view(B, -1)
    log_p = -F.binary_cross_entropy(y_hat, y) * y.shape[1]
    kl = kl_b.mean()
    return log_p - kl, log_p, kl, y_hat


def elbo_galaxy(pp, qp, x, y, r, act="tanh", rotate=True, translate=True, dx_scale=0.1, theta_prior=math.pi, z_scale=1,
                resid=False):
    """eval_minibatch of train_galaxy.py:27-128 (no augmentation) with the noise r supplied; y is (B, N, channels)."""
    B, channels = y.shape[0], y.shape[2]
    xb, z, kl_b = _posterior_and_pose(qp, x, y.reshape(B, -1), r, act, resid, rotate, translate, dx_scale, theta_prior,
                                      False, z_scale)
    y_hat = decoder(pp, xb, z, act, resid=resid).view(B, -1, channels)
    log_p = -F.binary_cross_entropy(y_hat, y) * (y.shape[1] * channels)
    kl = kl_b.mean()
    return log_p - kl, log_p, kl, y_hat


def elbo_particles(pp, qp, x, y, r, mask=None, ctf=None, act="tanh", rotate=True, translate=True, dx_scale=0.1,
                   theta_prior=math.pi, z_scale=1, resid=False, softplus=False, expand_coords=False):
    """eval_minibatch of train_particles.py:22-148 (no augmentation) with the noise r supplied; y is (B, N); the decoder
    has 1 output (unit-variance Gaussian) or 2 (--fit-noise; the reference's channel-interleaved halves);
    ctf (B, 1, k, k) is applied as a grouped cross-correlation (train_particles.py:112-119)."""
    B, N = y.shape
    n = int(math.sqrt(N))
    xb, z, kl_b = _posterior_and_pose(qp, x, y, r, act, resid, rotate, translate, dx_scale, theta_prior, False, z_scale)
    y_params = decoder(pp, xb, z, act, resid=resid, softplus=softplus, expand_coords=expand_coords).view(B, -1)
    y_mu, y_var, y_logvar = y_params, None, None
    if y_params.shape[1] > N:
        y_mu, y_logvar = y_params[:, :N], y_params[:, N:]
        y_var = torch.exp(y_logvar)
    if ctf is not None:
        if y_var is not None:
            raise RuntimeError("the reference's CTF + fit-noise path raises (train_particles.py:121-124: conv2d without groups)")
        y_mu = F.conv2d(y_mu.reshape(1, -1, n, n), ctf, padding=ctf.shape[2] // 2, groups=ctf.shape[0]).view(-1, N)
    if mask is not None:
        y, y_mu = y[:, mask], y_mu[:, mask]
        if y_var is not None:
            y_var, y_logvar = y_var[:, mask], y_logvar[:, mask]
    if y_var is not None:
        log_p = -0.5 * torch.sum((y_mu - y) ** 2 / y_var + y_logvar, 1).mean()
    else:
        log_p = -0.5 * torch.sum((y_mu - y) ** 2, 1).mean()
    kl = kl_b.mean()
    return log_p - kl, log_p, kl


class CpuTrainer(object):
    """Parameters + Adam; step() = forward, backward, optimiser step (train_mnist.py:143-150 and its siblings
    train_galaxy.py:204-210, train_particles.py:170-181).  script: "mnist" | "galaxy" | "particles"."""

    def __init__(self, p_state, q_state, x_coord, lr=1e-4, script="mnist", **cfg):
        self.pp = {k: torch.tensor(v).requires_grad_(True) for k, v in p_state.items()}
        self.qp = {k: torch.tensor(v).requires_grad_(True) for k, v in q_state.items()}
        self.x = torch.as_tensor(x_coord)
        self.cfg = cfg
        self.fn = {"mnist": elbo_mnist, "galaxy": elbo_galaxy, "particles": elbo_particles}[script]
        self.optim = torch.optim.Adam(list(self.pp.values()) + list(self.qp.values()), lr=lr)

    def step(self, y, r, **batch):
        out = self.fn(self.pp, self.qp, self.x, y, r, **dict(self.cfg, **batch))
        (-out[0]).backward()
        self.optim.step()
        self.optim.zero_grad()
        return out[0].detach(), out[1].detach(), out[2].detach()
