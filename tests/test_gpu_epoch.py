"""The training LOOP against the reference's own train_epoch: cli.run_epoch + dp.TrainStep on the GPU, fed
the fixtures' minibatches and noise, must reproduce what /root/reference/train_mnist.py:127-171 and
train_particles.py:151-202 produced with torch.optim.Adam on the CPU -- every step's (elbo, log_p, kl), the running means
each epoch returns, and every parameter after the last update.  Covers a ragged last minibatch, the --z-delay schedule
(z_scale 0 then 1), CTF + mask, and the --vanilla generator through the same loop."""
import contextlib
import io

import numpy as np
import pytest
import torch
import torch.nn as nn

import cases as C
from helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL_STEP, TOL_PARAM = 2e-5, 1e-4
ACT = {"tanh": nn.Tanh, "leakyrelu": nn.LeakyReLU, "relu": nn.ReLU, "sigmoid": nn.Sigmoid}


def _nets(case, inp, dev):
    import spatial_vae.models as models
    act = ACT[case["act"]]
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(case["z_dim"], case["H"], n_out=case["n_out"], num_layers=case["L"], activation=act,
                                    softplus=case["softplus"], resid=case["resid"], expand_coords=case["expand_coords"],
                                    bilinear=case["bilinear"])
        q = models.InferenceNetwork(case["n"] * case["m"], C.inf_dim(case), case["q_hidden"], num_layers=case["q_layers"],
                                    activation=act, resid=case["resid"])
    p.load_state_dict({k: torch.from_numpy(v) for k, v in inp["p_state"].items()})
    q.load_state_dict({k: torch.from_numpy(v) for k, v in inp["q_state"].items()})
    return p.to(dev), q.to(dev)


@pytest.mark.parametrize("name", [e["name"] for e in C.EPOCH_CASES])
def test_run_epoch_matches_reference_train_epoch(name):
    from spatial_vae_amd import cli, dp, elbo as E
    ec = C.EPOCH_CASES_BY_NAME[name]
    case = ec["case"]
    inp = C.build_epoch_inputs(ec)
    gold = load_golden(name)
    dev = torch.device("cuda:0")
    p_net, q_net = _nets(case, inp, dev)
    script = case["script"]
    fn = {"mnist": E.eval_minibatch_mnist, "particles": E.eval_minibatch_particles}[script]
    step = dp.TrainStep(p_net, q_net, fn, lr=ec["lr"], rotate=case["rotate"], translate=case["translate"],
                        dx_scale=case["dx_scale"], theta_prior=case["theta_prior"])
    x = torch.from_numpy(inp["x_coord"]).to(dev)
    data = {"y": torch.from_numpy(inp["y"]).to(dev), "ctf": torch.from_numpy(inp["ctf"]).to(dev) if inp["ctf"] is not None else None}
    mask = torch.from_numpy(inp["mask"]).to(dev) if inp["mask"] is not None else None
    bounds = np.cumsum((0,) + ec["batches"])
    batches = [torch.arange(int(bounds[i]), int(bounds[i + 1]), device=dev) for i in range(len(ec["batches"]))]
    per_step = []
    orig_call = step._step

    def recording(x_, batch, weight, kw):
        out = orig_call(x_, batch, weight, kw)
        per_step.append(step.metrics.detach().cpu().numpy().astype(np.float64))
        return out

    step._step = recording
    means = []
    for e, zs in enumerate(ec["z_scales"]):
        kw = {} if script == "mnist" else {"z_scale": zs}
        noise = [torch.from_numpy(r).to(dev) for r in inp["r"][e]]
        means.append(cli.run_epoch(script, step, x, batches, True, int(bounds[-1]), e, len(ec["z_scales"]), 0, 1, 0,
                                   dict(data=data, mask=mask, kw=kw, inf_dim=C.inf_dim(case), noise=noise)))
    got = np.array(per_step)
    assert got.shape == gold["steps"].shape
    assert np.abs(got - gold["steps"]).max() <= TOL_STEP * np.abs(gold["steps"]).max(), (got, gold["steps"])
    # train_epoch returns (elbo_accum, gen_loss_accum = running mean of -log_p, kl_accum)
    assert np.abs(np.array(means) - gold["means"]).max() <= TOL_STEP * np.abs(gold["means"]).max()
    for k, v in p_net.state_dict().items():
        assert rel_err(v.cpu().numpy(), gold["p." + k]) < TOL_PARAM, k
    for k, v in q_net.state_dict().items():
        assert rel_err(v.cpu().numpy(), gold["q." + k]) < TOL_PARAM, k
    assert step.aliased()
    # the evaluation pass (eval_model: forward only, its own noise) over the same minibatches with the trained networks;
    # the parameters agree with the reference's to 1e-4, so do the metrics
    kw = {} if script == "mnist" else {"z_scale": ec["z_scales"][-1]}
    noise = [torch.from_numpy(r).to(dev) for r in inp["r_eval"]]
    ev = cli.run_epoch(script, step, x, batches, False, int(bounds[-1]), 0, 1, 0, 1, 0,
                       dict(data=data, mask=mask, kw=kw, inf_dim=C.inf_dim(case), noise=noise))
    assert np.abs(np.array(ev) - gold["eval_means"]).max() <= 1e-4 * np.abs(gold["eval_means"]).max(), (ev, gold["eval_means"])
    assert not p_net.training and not q_net.training                       # eval_model leaves the networks in eval mode


@pytest.mark.parametrize("name", [c["name"] for c in C.VANILLA_CASES])
def test_vanilla_generator_through_eval_minibatch(name):
    """--vanilla (models.py:135-172 via train_mnist.py:351-357 / train_particles.py:446-452): the generator is a plain
    PyTorch MLP, but reparameterisation, KL, the likelihood kernels and the ELBO head are the HIP library's
    (elbo._core's else-branch)."""
    import spatial_vae.models as models
    from spatial_vae_amd import elbo as E
    case = C.VANILLA_CASES_BY_NAME[name]
    inp = C.build_vanilla_inputs(case)
    gold = load_golden(name)
    dev = torch.device("cuda:0")
    act = ACT[case["act"]]
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.VanillaGenerator(case["n"] * case["m"], case["z_dim"], case["H"], n_out=case["n_out"], num_layers=case["L"],
                                    activation=act, softplus=case["softplus"], resid=case["resid"])
        q = models.InferenceNetwork(case["n"] * case["m"], case["z_dim"], case["q_hidden"], num_layers=case["q_layers"],
                                    activation=act, resid=case["resid"])
    p.load_state_dict({k: torch.from_numpy(v) for k, v in inp["p_state"].items()})
    q.load_state_dict({k: torch.from_numpy(v) for k, v in inp["q_state"].items()})
    p, q = p.to(dev), q.to(dev)
    x, y, r = (torch.from_numpy(inp[k]).to(dev) for k in ("x_coord", "y", "r"))
    kw = dict(rotate=False, translate=False, dx_scale=case["dx_scale"], theta_prior=case["theta_prior"], noise=r)
    if case["script"] == "mnist":
        elbo, log_p, kl, y_hat = E.eval_minibatch_mnist(x, y, p, q, **kw)
        assert rel_err(y_hat.detach().cpu().numpy(), gold["y_hat"]) < TOL_STEP
    else:
        mask = torch.from_numpy(inp["mask"]).to(dev) if inp["mask"] is not None else None
        elbo, log_p, kl = E.eval_minibatch_particles(x, y, mask, None, p, q, **kw)
    (-elbo).backward()
    for got, key in ((elbo, "elbo"), (log_p, "log_p"), (kl, "kl")):
        assert abs(got.item() - float(gold[key])) <= TOL_STEP * max(abs(float(gold[key])), 1.0), key
    for k, v in p.named_parameters():
        assert rel_err(v.grad.cpu().numpy(), gold["gp." + k]) < 5 * TOL_STEP, k
    for k, v in q.named_parameters():
        assert rel_err(v.grad.cpu().numpy(), gold["gq." + k]) < 5 * TOL_STEP, k


def test_vanilla_trains_through_trainstep():
    """dp.TrainStep with a generator that has no decoder sinks: one bucket, plain autograd gradients into the flat buffer."""
    import math
    import spatial_vae.models as models
    from spatial_vae_amd import cli, dp, elbo as E
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.VanillaGenerator(100, 3, 32, num_layers=2).to(dev)
        q = models.InferenceNetwork(100, 3, 32, num_layers=1).to(dev)
    step = dp.TrainStep(p, q, E.eval_minibatch_mnist, lr=1e-2, rotate=False, translate=False, theta_prior=math.pi / 4)
    assert not step._bucketed and step.aliased()
    x = cli.coord_grid(10, 10).to(dev)
    y = torch.from_numpy(np.random.RandomState(0).uniform(size=(16, 100)).astype(np.float32)).to(dev)
    step(x, y)
    first = float(step.metrics[0])
    for _ in range(30):
        step(x, y)
    assert float(step.metrics[0]) > first
