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
    # Error budget of the 5e-6 bound (max |error| relative to the largest entry of the float64 result).
    #   * accumulation: a K-term fp32 dot product summed in a fixed order rounds each partial sum to u = 2^-24 = 6e-8 relative;
    #     with terms of random sign the roundings add in quadrature to ~ sqrt(K) u of the result's scale: 1.7e-6 at K = 784,
    #     2.4e-6 at K = 1600 (x W^T contracts over the inputs; dW over <= 512 rows and dx over <= 500 outputs are shorter);
    #   * activation: the kernel's tanh is 1 - 2 / (exp2(2 x log2 e) + 1) on v_exp_f32 / v_rcp_f32 (1 ulp each): <= 2e-7
    #     absolute on values up to 1, entering dW / dx twice through act' = 1 - a^2.
    #   Sum ~ 2-3e-6 for the largest shapes; observed maximum over this matrix 2.4e-6 (256 x 784 x 500 with tanh, r02).  The
    #   bound is twice that, and 20x below the 1e-4 the north star asks of the decoder.
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


def test_galaxy_encoder_takes_the_sink_path_and_matches_float64(monkeypatch):
    """BASELINE configs[3] builds InferenceNetwork(49152, 23, 5000, num_layers=2) (train_galaxy.py:306, :461-470: 271 M
    parameters).  Its 49 152 x 5 000 and 5 000 x 5 000 layers are above the enc_linear kernels' 4 M-weight limit, so under
    dp.TrainStep they run through ops.sink_linear -- hipBLASLt GEMMs writing dW straight into the flat gradient buffer and
    svae_colsum for db -- and only the 5 000 -> 46 head through the hand-written kernel.  This is the branch bench.py --config 4
    executes; here it is checked numerically at B = 4: output and every parameter gradient against the same MLP in float64
    on the CPU, with spies asserting that the sink path and svae_colsum really ran."""
    import spatial_vae.models as models
    from spatial_vae_amd import _lib, dp, elbo as E, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(20, 32, n_out=3, num_layers=2).to(dev)
        q = models.InferenceNetwork(128 * 128 * 3, 23, 5000, num_layers=2).to(dev)
    step = dp.TrainStep(p, q, E.eval_minibatch_galaxy, lr=1e-4, rotate=True, translate=True)
    assert step._bucketed                                            # >= 32 M encoder parameters: the two-bucket rule
    assert set(q._grad_sinks) == {"layers.%d.%s" % (i, t) for i in (0, 2, 4) for t in ("weight", "bias")}
    calls = {"sink": 0, "colsum": 0, "enc": 0}
    orig_sink, orig_enc = ops.sink_linear, ops.enc_linear
    L = _lib.lib()
    orig_colsum = L.svae_colsum

    def spy_sink(*a, **k):
        calls["sink"] += 1
        return orig_sink(*a, **k)

    def spy_enc(*a, **k):
        calls["enc"] += 1
        return orig_enc(*a, **k)

    def spy_colsum(*a):
        calls["colsum"] += 1
        return orig_colsum(*a)

    monkeypatch.setattr(ops, "sink_linear", spy_sink)
    monkeypatch.setattr(ops, "enc_linear", spy_enc)
    monkeypatch.setattr(L, "svae_colsum", spy_colsum)
    g = torch.Generator().manual_seed(5)
    y = torch.rand(4, 128 * 128 * 3, generator=g)
    dout = torch.randn(4, 46, generator=g)
    out = E._encode(q, y.to(dev))
    out.backward(dout.to(dev))
    torch.cuda.synchronize()
    assert calls == {"sink": 2, "colsum": 2, "enc": 1}, calls
    got = {k: step.grads.sinks[k].detach().cpu().numpy() for k in q._grad_sinks}
    assert all(p_.grad is None for p_ in q.parameters())             # written into the flat buffer, not handed to autograd
    st = {k: v.detach().cpu().double().requires_grad_(True) for k, v in q.state_dict().items()}
    h = y.double()
    for i in (0, 2, 4):
        h = F.linear(h, st["layers.%d.weight" % i], st["layers.%d.bias" % i])
        h = torch.tanh(h) if i < 4 else h
    h.backward(dout.double())
    # K = 49 152 inputs in [0, 1): fp32 GEMM accumulation against float64, relative to the largest entry
    assert rel_err(out.detach().cpu().numpy(), h.detach().numpy()) < 2e-5
    for k in got:
        assert rel_err(got[k], st[k].grad.numpy()) < 2e-5, (k, rel_err(got[k], st[k].grad.numpy()))
