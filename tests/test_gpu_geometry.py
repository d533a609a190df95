"""Odd geometries (GPU): the HIP path against the CPU oracle (itself pinned by the reference's golden
vectors) where no golden exists: single image, fewer pixels than one 32-row tile, tiny and non-multiple
widths, the deepest stack the ABI allows, 4 output channels, every activation with residual layers,
expand_coords + bilinear on the generic (non-fused) first-layer path, rotation-only with z_dim 0."""
import contextlib
import io

import numpy as np
import pytest
import torch
import torch.nn as nn

import cases as C
from helpers import rel_err
from oracle import elbo_oracle as O

pytestmark = pytest.mark.gpu
ACT = {"tanh": nn.Tanh, "leakyrelu": nn.LeakyReLU, "relu": nn.ReLU, "sigmoid": nn.Sigmoid}

GEOMS = [
    dict(name="one_image_tiny", n=3, m=5, B=1, H=7, L=2),
    dict(name="sub_tile_pixels", n=4, m=4, B=5, H=40, L=2, z_dim=1),
    dict(name="deepest_stack", n=6, m=6, B=3, H=36, L=8, act="leakyrelu"),
    dict(name="deep_resid_sigmoid", n=6, m=6, B=3, H=36, L=5, resid=True, act="sigmoid"),
    dict(name="deep_resid_relu", n=7, m=5, B=2, H=65, L=4, resid=True, act="relu"),
    dict(name="four_channels", script="galaxy", n=6, m=6, B=3, H=33, L=3, n_out=4, z_dim=3),
    dict(name="expand_bilinear_deep", n=7, m=7, B=4, H=48, L=3, expand_coords=True, bilinear=True, z_dim=4),
    dict(name="rotate_only_z0", n=9, m=9, B=6, H=128, L=2, z_dim=0, translate=False),
    dict(name="width_160_three_blocks", n=10, m=10, B=4, H=160, L=3),
    dict(name="exact_tiles", n=8, m=8, B=8, H=256, L=2),
    dict(name="one_layer_wide", n=8, m=8, B=4, H=300, L=1),
    dict(name="particles_noise_mask_resid", script="particles", n=8, m=8, B=5, H=70, L=3, n_out=2, mask=True, resid=True,
         act="leakyrelu", z_scale=0.3),
    # widths that are multiples of 64 with bounded activations: the shapes the fp16x3 GEMM mode takes (tests/test_gpu_split.py
    # runs this file with the mode on) -- residual stack, two-channel fused logits, sigmoid chain, 8-tile width
    dict(name="w64_resid_tanh_deep", n=6, m=6, B=3, H=64, L=4, resid=True),
    dict(name="w128_two_channels_noise", script="particles", n=8, m=8, B=4, H=100, L=2, n_out=2),
    dict(name="w128_sigmoid_three", n=5, m=7, B=3, H=128, L=3, act="sigmoid", z_dim=3),
    dict(name="w256_rgb_two_layers", script="galaxy", n=6, m=6, B=3, H=250, L=2, n_out=3, z_dim=5),
]


@pytest.mark.parametrize("geom", GEOMS, ids=[g["name"] for g in GEOMS])
def test_geometry_against_oracle(geom):
    import spatial_vae.models as models
    from spatial_vae_amd import elbo as E
    kw = dict(geom)
    kw.setdefault("theta_prior", np.pi / 3)
    case = C._case(seed=200 + GEOMS.index(geom), q_hidden=16, **kw)
    inp = C.build_inputs(case)
    dev = torch.device("cuda:0")
    with contextlib.redirect_stdout(io.StringIO()):
        p_net = models.SpatialGenerator(case["z_dim"], case["H"], n_out=case["n_out"], num_layers=case["L"],
                                        activation=ACT[case["act"]], softplus=case["softplus"], resid=case["resid"],
                                        expand_coords=case["expand_coords"], bilinear=case["bilinear"])
        n_in = case["n"] * case["m"] * (case["n_out"] if case["script"] == "galaxy" else 1)
        q_net = models.InferenceNetwork(n_in, C.inf_dim(case), case["q_hidden"], activation=ACT[case["act"]], resid=case["resid"])
    p_net.load_state_dict({k: torch.from_numpy(v) for k, v in inp["p_state"].items()})
    q_net.load_state_dict({k: torch.from_numpy(v) for k, v in inp["q_state"].items()})
    p_net.to(dev)
    q_net.to(dev)
    x = torch.from_numpy(inp["x_coord"]).to(dev)
    y = torch.from_numpy(inp["y"]).to(dev)
    r = torch.from_numpy(inp["r"]).to(dev)
    call = dict(rotate=case["rotate"], translate=case["translate"], dx_scale=case["dx_scale"], theta_prior=case["theta_prior"],
                noise=r, return_logits=True)
    if case["script"] == "mnist":
        elbo, log_p, kl, _, logits = E.eval_minibatch_mnist(x, y, p_net, q_net, **call)
    elif case["script"] == "galaxy":
        elbo, log_p, kl, _, logits = E.eval_minibatch_galaxy(x, y, p_net, q_net, z_scale=case["z_scale"], **call)
    else:
        mask = torch.from_numpy(inp["mask"]).to(dev) if inp["mask"] is not None else None
        elbo, log_p, kl, logits = E.eval_minibatch_particles(x, y, mask, None, p_net, q_net, z_scale=case["z_scale"], **call)
    (-elbo).backward()
    torch.cuda.synchronize()
    with torch.no_grad():
        q_out = q_net.layers(y.view(y.size(0), -1))
    ref = O.elbo_minibatch(case["script"], O.DecoderSpec.from_case(case), inp["p_state"], inp["x_coord"], inp["y"],
                           q_out.cpu().numpy(), inp["r"], rotate=case["rotate"], translate=case["translate"],
                           dx_scale=case["dx_scale"], theta_prior=case["theta_prior"], z_scale=case["z_scale"], mask=inp["mask"])
    assert abs(elbo.item() - float(ref["elbo"])) <= 2e-5 * abs(float(ref["elbo"]))
    assert abs(kl.item() - float(ref["kl"])) <= 2e-5 * max(abs(float(ref["kl"])), 1.0)
    assert rel_err(logits.detach().cpu().numpy(), ref["logits"]) < 2e-5
    for k, p in p_net.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), ref["gP"][k]) < 1e-4, (geom["name"], k)
    # the gradient reaching the encoder: the last Linear's bias gradient is d(-elbo)/d(q_out) summed over the batch
    gq = {k: p.grad.cpu().numpy() for k, p in q_net.named_parameters()}
    last = [k for k in gq if k.endswith("bias")][-1]
    assert rel_err(gq[last], ref["g_q_out"].sum(0)) < 1e-4


def test_second_backward_follows_autograd_semantics():
    """The reference's decoder is plain autograd (spatial_vae/models.py:90-132): backward(retain_graph=True) may be followed by
    further backward passes through the same graph, and a second pass without it raises torch's "backward through the graph a
    second time" error.  The HIP decoder keeps what its backward call reads in autograd's saved tensors, so it behaves the same."""
    import contextlib
    import io
    import spatial_vae.models as models
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(3, 40, num_layers=2).to(dev)
    x = (torch.rand(5, 49, 2, device=dev) * 2 - 1).requires_grad_(True)
    z = torch.randn(5, 3, device=dev, requires_grad=True)
    up = torch.randn(5, 49, 1, device=dev)

    def grads():
        g = [q.grad.clone() for q in p.parameters()] + [x.grad.clone(), z.grad.clone()]
        p.zero_grad(set_to_none=True)
        x.grad = z.grad = None
        return g

    y = p(x, z)
    y.backward(up, retain_graph=True)
    first = grads()
    y.backward(up, retain_graph=True)              # the saved buffers are still there
    second = grads()
    for a, b in zip(first, second):
        assert torch.equal(a, b)
    y.backward(2.0 * up)                           # last pass: releases them
    third = grads()
    for a, b in zip(first, third):
        assert rel_err(b.cpu().numpy(), 2.0 * a.cpu().numpy()) < 1e-6
    with pytest.raises(RuntimeError, match="second time|already been freed"):
        y.backward(up)


def test_backward_refuses_a_saved_buffer_planned_under_another_mode(monkeypatch):
    """svae_decoder_backward re-derives the kernel plan from the descriptor, the GEMM mode and SVAE_FUSE_OUT; the forward call
    baked some of those decisions into `saved` (row-scaled packed weights for the rank-1 output backward).  Changing
    SVAE_FUSE_OUT between the two calls used to mix the forms silently; now the backward call fails with SVAE_E_INVALID."""
    import contextlib
    import io
    import spatial_vae.models as models
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(2, 40, num_layers=2).to(dev)       # one output channel, tanh: the rank-1 form applies
    x = torch.rand(4, 49, 2, device=dev) * 2 - 1
    z = torch.randn(4, 2, device=dev)
    monkeypatch.delenv("SVAE_FUSE_OUT", raising=False)
    y = p(x, z)
    monkeypatch.setenv("SVAE_FUSE_OUT", "0")
    with pytest.raises(RuntimeError, match="planned differently"):
        y.sum().backward()
    monkeypatch.delenv("SVAE_FUSE_OUT", raising=False)
    p.zero_grad(set_to_none=True)
    p(x, z).sum().backward()                                          # and a consistent pair still works
    assert all(torch.isfinite(q.grad).all() for q in p.parameters())
