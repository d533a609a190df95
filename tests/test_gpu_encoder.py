"""svae_linear_forward / svae_linear_backward (the inference network's Linear + activation layers on the fp32 MFMA,
spatial_vae/models.py:31-43) against the same layer in float64 on the CPU: every activation, ragged tile edges in all three
dimensions (rows not a multiple of 16, outputs not a multiple of 16, inputs not a multiple of 4), with and without dx, with
and without gradient sinks, and the whole encoder of the BASELINE shapes through elbo._encode."""
import contextlib
import io

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import rel_err

pytestmark = pytest.mark.gpu
ACTS = {None: lambda v: v, "tanh": torch.tanh, "leakyrelu": lambda v: F.leaky_relu(v, 0.01), "relu": torch.relu,
        "sigmoid": torch.sigmoid}
SHAPES = [(256, 784, 500), (256, 500, 10), (5, 49, 24), (3, 37, 10), (17, 130, 33), (64, 16, 16), (1, 1, 1), (512, 1600, 500)]


@pytest.mark.parametrize("act", list(ACTS))
@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
def test_linear_layer_matches_float64(shape, act):
    from spatial_vae_amd import ops
    M, K, N = shape
    g = torch.Generator().manual_seed(M * 1000 + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    dy = torch.randn(M, N, generator=g)
    dev = torch.device("cuda:0")
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    out = ops.enc_linear(xg, wg, bg, act)
    out.backward(dy.to(dev))
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = ACTS[act](F.linear(x64, w64, b64))
    ref.backward(dy.double())
    errs = dict(out=rel_err(out.detach().cpu().numpy(), ref.detach().numpy()), dw=rel_err(wg.grad.cpu().numpy(), w64.grad.numpy()),
                db=rel_err(bg.grad.cpu().numpy(), b64.grad.numpy()), dx=rel_err(xg.grad.cpu().numpy(), x64.grad.numpy()))
    # fp32 accumulation over up to 1600 terms against float64: a few 1e-7 relative to the largest entry
    assert all(v < 5e-6 for v in errs.values()), errs


def test_linear_layer_without_dx_and_with_sinks():
    from spatial_vae_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(40, 70, generator=g).to(dev)                      # data: no gradient wanted
    w = nn.Parameter((torch.randn(33, 70, generator=g) / 8).to(dev))
    b = nn.Parameter(torch.randn(33, generator=g).to(dev))
    sink_w, sink_b = torch.full_like(w, 7.0), torch.full_like(b, 7.0)
    out = ops.enc_linear(x, w, b, "tanh", sink_w, sink_b)
    dy = torch.randn(40, 33, generator=g).to(dev)
    out.backward(dy)
    assert w.grad is None and b.grad is None                         # written into the sinks, not handed to autograd
    ref = torch.tanh(F.linear(x.double(), w.detach().double().requires_grad_(True), b.detach().double()))
    w64 = w.detach().double().requires_grad_(True)
    b64 = b.detach().double().requires_grad_(True)
    torch.tanh(F.linear(x.double(), w64, b64)).backward(dy.double())
    assert rel_err(sink_w.cpu().numpy(), w64.grad.cpu().numpy()) < 5e-6
    assert rel_err(sink_b.cpu().numpy(), b64.grad.cpu().numpy()) < 5e-6
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 5e-6


@pytest.mark.parametrize("cfg", [dict(n_in=784, inf=5, hq=500, lq=2, act=nn.Tanh), dict(n_in=1600, inf=11, hq=500, lq=2, act=nn.Tanh),
                                 dict(n_in=49, inf=3, hq=24, lq=3, act=nn.LeakyReLU), dict(n_in=36, inf=6, hq=24, lq=2, act=nn.ELU)],
                         ids=["cfg2", "cfg5", "leaky_L3", "elu_not_fusable"])
def test_whole_encoder_matches_the_torch_module(cfg):
    """elbo._encode routes InferenceNetwork.layers through the kernels (activation fused where it is one of the four the
    decoder knows, applied as a torch module otherwise) and must reproduce the module itself, values and gradients."""
    import spatial_vae.models as models
    from spatial_vae_amd import elbo as E
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    with contextlib.redirect_stdout(io.StringIO()):
        q = models.InferenceNetwork(cfg["n_in"], cfg["inf"], cfg["hq"], num_layers=cfg["lq"], activation=cfg["act"]).to(dev)
    y = torch.rand(48, cfg["n_in"], device=dev)
    got = E._encode(q, y)
    got.square().sum().backward()
    mine = {k: p.grad.clone() for k, p in q.named_parameters()}
    q.zero_grad(set_to_none=True)
    ref = q.layers(y)
    ref.square().sum().backward()
    assert rel_err(got.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 5e-6
    for k, p in q.named_parameters():
        assert rel_err(mine[k].cpu().numpy(), p.grad.cpu().numpy()) < 2e-5, k
