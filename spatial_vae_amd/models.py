"""The reference's model classes, with the decoder routed to the MI355X kernels.

Same constructor signatures, attribute names and state-dict keys as
/root/reference/spatial_vae/models.py (ResidLinear :13-21, InferenceNetwork :24-54,
SpatialGenerator :57-132, VanillaGenerator :135-172), so checkpoints and calling code carry
over.  ``SpatialGenerator.forward`` does not run torch ops: it hands the parameters to
``ops.decoder`` (HIP, see csrc/).  The encoder and the vanilla baseline stay ordinary
PyTorch-ROCm modules, as the north star prescribes.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

_ACT_NAMES = {nn.Tanh: "tanh", nn.LeakyReLU: "leakyrelu", nn.ReLU: "relu", nn.Sigmoid: "sigmoid"}


def _act_name(activation):
    try:
        return _ACT_NAMES[activation]
    except KeyError:
        raise NotImplementedError("SpatialGenerator on MI355X supports activations %s, got %r"
                                  % (sorted(a.__name__ for a in _ACT_NAMES), activation))


class ResidLinear(nn.Module):
    """act(linear(x) + x)."""

    def __init__(self, n_in, n_out, activation=nn.Tanh):
        super().__init__()
        self.linear = nn.Linear(n_in, n_out)
        self.act = activation()

    def forward(self, x):
        return self.act(self.linear(x) + x)


def _mlp_trunk(n_in, hidden_dim, num_layers, activation, resid, first_is_linear):
    """[Linear(n_in,H)?, act, (Linear, act | ResidLinear) x (num_layers-1)] -- the slot order every
    class of the reference shares (and that the state-dict indices depend on)."""
    mods = [nn.Linear(n_in, hidden_dim)] if first_is_linear else []
    mods.append(activation())
    for _ in range(num_layers - 1):
        if resid:
            mods.append(ResidLinear(hidden_dim, hidden_dim, activation=activation))
        else:
            mods += [nn.Linear(hidden_dim, hidden_dim), activation()]
    return mods


class InferenceNetwork(nn.Module):
    """Encoder MLP; returns (z_mu, z_logstd), the two halves of the last layer's output."""

    def __init__(self, n, latent_dim, hidden_dim, num_layers=1, activation=nn.Tanh, resid=False):
        super().__init__()
        self.latent_dim = latent_dim
        self.n = n
        mods = _mlp_trunk(n, hidden_dim, num_layers, activation, resid, first_is_linear=True)
        mods.append(nn.Linear(hidden_dim, 2 * latent_dim))
        self.layers = nn.Sequential(*mods)
        print(self)

    def forward(self, x):
        out = self.layers(x)
        return out[:, :self.latent_dim], out[:, self.latent_dim:]


class SpatialGenerator(nn.Module):
    """Coordinate-conditioned MLP decoder; forward runs in the HIP library."""

    def __init__(self, latent_dim, hidden_dim, n_out=1, num_layers=1, activation=nn.Tanh,
                 softplus=False, resid=False, expand_coords=False, bilinear=False):
        super().__init__()
        self.softplus = softplus
        self.expand_coords = expand_coords
        in_dim = 5 if expand_coords else 2
        self.coord_linear = nn.Linear(in_dim, hidden_dim)
        self.latent_dim = latent_dim
        if latent_dim > 0:
            self.latent_linear = nn.Linear(latent_dim, hidden_dim, bias=False)
            if bilinear:
                self.bilinear = nn.Bilinear(in_dim, latent_dim, hidden_dim, bias=False)
        mods = _mlp_trunk(None, hidden_dim, num_layers, activation, resid, first_is_linear=False)
        mods += [nn.Linear(hidden_dim, n_out), nn.Sigmoid()]
        self.layers = nn.Sequential(*mods)
        self._spec = ops.DecoderSpec(latent_dim=latent_dim, hidden_dim=hidden_dim, n_out=n_out, num_layers=num_layers,
                                     act=_act_name(activation), softplus=bool(softplus), resid=bool(resid),
                                     expand_coords=bool(expand_coords),
                                     bilinear=bool(bilinear) and latent_dim > 0)
        slope = [m.negative_slope for m in self.layers if isinstance(m, nn.LeakyReLU)]
        if any(s != 0.01 for s in slope):
            raise NotImplementedError("LeakyReLU slope other than the default 0.01")
        print(self)

    # -- parameter plumbing ---------------------------------------------------------------
    def _linears(self):
        lin = []
        for m in self.layers:
            if isinstance(m, ResidLinear):
                lin.append(m.linear)
            elif isinstance(m, nn.Linear):
                lin.append(m)
        return lin[:-1], lin[-1]

    def _decode(self, B, coords, grid, theta, dx, z, bce_target=None):
        hidden_lin, out_lin = self._linears()
        hidden = []
        for m in hidden_lin:
            hidden += [m.weight, m.bias]
        latent_w = self.latent_linear.weight if self.latent_dim > 0 else None
        bil_w = self.bilinear.weight if hasattr(self, "bilinear") else None
        return ops.decoder(self._spec, B, coords, grid, theta, dx, z, self.coord_linear.weight, self.coord_linear.bias,
                           latent_w, bil_w, out_lin.weight, out_lin.bias, hidden, sinks=getattr(self, "_grad_sinks", None),
                           bce_target=bce_target)

    def decoder_parameters(self):
        """{sink name: parameter} in the naming ops.decoder uses for gradient sinks (see dp.FlatGrads)."""
        hidden_lin, out_lin = self._linears()
        named = {"coord_w": self.coord_linear.weight, "coord_b": self.coord_linear.bias, "out_w": out_lin.weight,
                 "out_b": out_lin.bias}
        if self.latent_dim > 0:
            named["latent_w"] = self.latent_linear.weight
        if hasattr(self, "bilinear"):
            named["bilinear_w"] = self.bilinear.weight
        for i, m in enumerate(hidden_lin):
            named["hidden%d" % (2 * i)] = m.weight
            named["hidden%d" % (2 * i + 1)] = m.bias
        return named

    # -- the reference's entry point ----------------------------------------------------------
    def forward(self, x, z):
        """x (batch, num_coords, 2) [or (num_coords, 2)], z (batch, latent_dim) -> (batch, num_coords, n_out)."""
        if x.dim() < 3:
            x = x.unsqueeze(0)
        if self.latent_dim > 0 and z.dim() < 2:
            z = z.unsqueeze(0)
        y, _ = self._decode(x.size(0), x.contiguous(), None, None, None, z if self.latent_dim > 0 else None)
        return y

    # -- fused entry used by eval_minibatch -----------------------------------------------------
    def forward_posed(self, grid, batch_size, theta=None, dx=None, z=None, return_logits=False, bce_target=None):
        """Decode on the shared grid (N, 2) rotated by theta (B) and shifted by dx (B, 2), without ever
        materialising the (B, N, 2) coordinates (replaces x.expand/bmm/+dx of eval_minibatch).  With bce_target
        (the observed images, (B, N[, C])) the per-image Bernoulli log-likelihood is computed in the same call and
        returned last: (y, logits, loglik)."""
        out = self._decode(batch_size, None, grid.contiguous(), theta, dx, z if self.latent_dim > 0 else None,
                           bce_target=bce_target)
        if bce_target is not None:
            return out
        return out if return_logits else out[0]


class VanillaGenerator(nn.Module):
    """Plain MLP decoder z -> all pixels (the --vanilla baseline); ordinary PyTorch."""

    def __init__(self, n, latent_dim, hidden_dim, n_out=1, num_layers=1, activation=nn.Tanh,
                 softplus=False, resid=False):
        super().__init__()
        self.n_out = n_out
        self.softplus = softplus
        mods = _mlp_trunk(latent_dim, hidden_dim, num_layers, activation, resid, first_is_linear=True)
        mods += [nn.Linear(hidden_dim, n * n_out), nn.Sigmoid()]
        if softplus:
            mods.append(nn.Softplus())
        self.layers = nn.Sequential(*mods)
        print(self)

    def forward(self, x, z):
        y = self.layers(z).view(z.size(0), -1, self.n_out)
        if self.softplus:
            y = torch.cat([F.softplus(y[:, :, :1]), y[:, :, 1:]], 2)
        return y
