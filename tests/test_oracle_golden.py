"""Pin the CPU oracle (oracle/elbo_oracle.py) against the golden vectors the reference
itself produced (tests/golden/*.npz, made by tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

import cases as C
from helpers import check_grads, load_golden, rel_err
from oracle import elbo_oracle as O

# fp32 restatement vs fp32 reference: the reference's own fp32-vs-fp64 noise is ~3e-7
# (SURVEY.md section 6); 2e-5 leaves room for summation-order differences at H=500.
TOL = 2e-5


@pytest.mark.parametrize("name", [c["name"] for c in C.CASES])
def test_elbo_minibatch_matches_reference(name):
    case = C.CASES_BY_NAME[name]
    inp = C.build_inputs(case)
    gold = load_golden(name)
    spec = O.DecoderSpec.from_case(case)
    res = O.elbo_minibatch(case["script"], spec, inp["p_state"], inp["x_coord"], inp["y"], gold["q_out"], inp["r"],
                           rotate=case["rotate"], translate=case["translate"], dx_scale=case["dx_scale"],
                           theta_prior=case["theta_prior"], z_scale=case["z_scale"], mask=inp["mask"], ctf=inp["ctf"],
                           theta_offset=inp["offset"])
    assert abs(float(res["elbo"]) - float(gold["elbo"])) <= TOL * abs(float(gold["elbo"]))
    assert abs(float(res["log_p"]) - float(gold["log_p"])) <= TOL * abs(float(gold["log_p"]))
    assert abs(float(res["kl"]) - float(gold["kl"])) <= TOL * max(abs(float(gold["kl"])), 1.0)
    assert rel_err(res["logits"], gold["logits"]) < TOL
    if "y_hat" in gold:
        assert rel_err(res["y_hat"].reshape(gold["y_hat"].shape), gold["y_hat"]) < TOL
    check_grads("gp.", res["gP"], gold, case, 5 * TOL)
    assert rel_err(res["g_q_out"], gold["q_out_grad"]) < 5 * TOL


@pytest.mark.parametrize("name", [c["name"] for c in C.CASES])
def test_decoder_entry_matches_reference(name):
    """SpatialGenerator.forward(x, z) on explicit coordinates with an explicit upstream gradient."""
    case = C.CASES_BY_NAME[name]
    inp = C.build_inputs(case)
    gold = load_golden(name)
    spec = O.DecoderSpec.from_case(case)
    cache = O.decoder_forward(spec, inp["p_state"], inp["dec_x"], inp["dec_z"])
    assert rel_err(cache["logits"], gold["dec.logits"]) < TOL
    assert rel_err(cache["y"], gold["dec.y"]) < TOL
    g, dcoords, dz = O.decoder_backward(spec, inp["p_state"], cache, inp["dec_dy"])
    check_grads("dec.gp.", g, gold, case, 5 * TOL)
    assert rel_err(dcoords, gold["dec.dx"]) < 5 * TOL
    if case["z_dim"] > 0:
        assert rel_err(dz, gold["dec.dz"]) < 5 * TOL


def test_saturated_case_hits_the_bce_clamp():
    """The saturated fixture must actually exercise sigmoid == 1.0 / the -100 clamp."""
    gold = load_golden("mnist_saturated")
    assert (gold["y_hat"] == 1.0).any() or (gold["y_hat"] == 0.0).any()
    assert float(gold["log_p"]) < -1000


@pytest.mark.parametrize("name", ["mnist_rt", "mnist_r", "mnist_t", "mnist_none", "mnist_leaky", "mnist_dxscale"])
def test_torch_cpu_step_matches_reference(name):
    """The torch-CPU restatement timed as cpu_baseline computes the reference's numbers."""
    import torch
    from oracle import torch_cpu_step as T
    case = C.CASES_BY_NAME[name]
    inp = C.build_inputs(case)
    gold = load_golden(name)
    pp = {k: torch.tensor(v).requires_grad_(True) for k, v in inp["p_state"].items()}
    qp = {k: torch.tensor(v).requires_grad_(True) for k, v in inp["q_state"].items()}
    elbo, log_p, kl, y_hat = T.elbo_mnist(pp, qp, torch.from_numpy(inp["x_coord"]), torch.from_numpy(inp["y"]),
                                          torch.from_numpy(inp["r"]), act=case["act"], rotate=case["rotate"],
                                          translate=case["translate"], dx_scale=case["dx_scale"],
                                          theta_prior=case["theta_prior"])
    (-elbo).backward()
    assert abs(elbo.item() - float(gold["elbo"])) <= TOL * abs(float(gold["elbo"]))
    assert rel_err(y_hat.detach().numpy(), gold["y_hat"]) < TOL
    check_grads("gp.", {k: v.grad.numpy() for k, v in pp.items()}, gold, case, 5 * TOL)
    check_grads("gq.", {k: v.grad.numpy() for k, v in qp.items()}, gold, case, 5 * TOL)


@pytest.mark.parametrize("name", ["galaxy_rgb", "galaxy_rgb_relu", "particles_gauss", "particles_fit_noise", "particles_mask",
                                  "particles_mask_fit_noise", "particles_ctf", "particles_ctf_mask", "particles_resid_leaky",
                                  "particles_softplus_noise", "particles_h500_noise", "mnist_L3_resid"])
def test_torch_cpu_step_siblings_match_reference(name):
    """elbo_galaxy / elbo_particles of the timed CPU port (bench.py --config 3/4/5) against the reference's outputs."""
    import torch
    from oracle import torch_cpu_step as T
    case = C.CASES_BY_NAME[name]
    inp = C.build_inputs(case)
    gold = load_golden(name)
    pp = {k: torch.tensor(v).requires_grad_(True) for k, v in inp["p_state"].items()}
    qp = {k: torch.tensor(v).requires_grad_(True) for k, v in inp["q_state"].items()}
    x, y, r = torch.from_numpy(inp["x_coord"]), torch.from_numpy(inp["y"]), torch.from_numpy(inp["r"])
    kw = dict(act=case["act"], rotate=case["rotate"], translate=case["translate"], dx_scale=case["dx_scale"],
              theta_prior=case["theta_prior"])
    if case["script"] == "mnist":
        if case["resid"]:
            pytest.skip("train_mnist.py has no --resid flag; the decoder's resid form is covered through particles")
        out = T.elbo_mnist(pp, qp, x, y, r, **kw)
    elif case["script"] == "galaxy":
        out = T.elbo_galaxy(pp, qp, x, y, r, z_scale=case["z_scale"], resid=case["resid"], **kw)
    else:
        mask = torch.from_numpy(inp["mask"]) if inp["mask"] is not None else None
        ctf = torch.from_numpy(inp["ctf"]) if inp["ctf"] is not None else None
        out = T.elbo_particles(pp, qp, x, y, r, mask=mask, ctf=ctf, z_scale=case["z_scale"], resid=case["resid"],
                               softplus=case["softplus"], expand_coords=case["expand_coords"], **kw)
    (-out[0]).backward()
    assert abs(out[0].item() - float(gold["elbo"])) <= TOL * abs(float(gold["elbo"]))
    assert abs(out[1].item() - float(gold["log_p"])) <= TOL * abs(float(gold["log_p"]))
    check_grads("gp.", {k: v.grad.numpy() for k, v in pp.items()}, gold, case, 5 * TOL)
    check_grads("gq.", {k: v.grad.numpy() for k, v in qp.items()}, gold, case, 5 * TOL)


@pytest.mark.parametrize("name", [e["name"] for e in C.EPOCH_CASES])
def test_torch_cpu_trainer_reproduces_reference_epochs(name):
    """CpuTrainer (what bench.py times as cpu_baseline) stepped over the reference's train_epoch fixtures: every step's
    (elbo, log_p, kl) and every parameter after the last Adam update (train_mnist.py:127-171, train_particles.py:151-202,
    incl. the --z-delay schedule)."""
    import torch
    from oracle import torch_cpu_step as T
    ec = C.EPOCH_CASES_BY_NAME[name]
    case = ec["case"]
    inp = C.build_epoch_inputs(ec)
    gold = load_golden(name)
    kw = dict(act=case["act"], rotate=case["rotate"], translate=case["translate"], dx_scale=case["dx_scale"],
              theta_prior=case["theta_prior"])
    tr = T.CpuTrainer(inp["p_state"], inp["q_state"], inp["x_coord"], lr=ec["lr"], script=case["script"], **kw)
    bounds = np.cumsum((0,) + ec["batches"])
    y = torch.from_numpy(inp["y"])
    got = []
    for e, zs in enumerate(ec["z_scales"]):
        for i in range(len(ec["batches"])):
            sl = slice(int(bounds[i]), int(bounds[i + 1]))
            batch = {}
            if case["script"] == "particles":
                batch = dict(z_scale=zs, mask=torch.from_numpy(inp["mask"]) if inp["mask"] is not None else None,
                             ctf=torch.from_numpy(inp["ctf"][sl]) if inp["ctf"] is not None else None)
            got.append([float(v) for v in tr.step(y[sl], torch.from_numpy(inp["r"][e][i]), **batch)])
    got = np.array(got)
    assert np.abs(got - gold["steps"]).max() <= TOL * np.abs(gold["steps"]).max()
    for k, v in tr.pp.items():
        assert rel_err(v.detach().numpy(), gold["p." + k]) < 1e-4, k
    for k, v in tr.qp.items():
        assert rel_err(v.detach().numpy(), gold["q." + k]) < 1e-4, k


@pytest.mark.parametrize("name", [c["name"] for c in C.VANILLA_CASES if c["script"] == "mnist"])
def test_vanilla_generator_mirror_matches_reference_on_cpu(name):
    """spatial_vae.models.VanillaGenerator (plain torch, so it runs here): state-dict names and y_hat of the reference's
    --vanilla eval_minibatch (models.py:135-172)."""
    import torch
    import torch.nn as nn
    import spatial_vae.models as models
    case = C.VANILLA_CASES_BY_NAME[name]
    inp = C.build_vanilla_inputs(case)
    gold = load_golden(name)
    act = {"tanh": nn.Tanh, "leakyrelu": nn.LeakyReLU}[case["act"]]
    p = models.VanillaGenerator(case["n"] * case["m"], case["z_dim"], case["H"], n_out=case["n_out"], num_layers=case["L"],
                                activation=act, softplus=case["softplus"], resid=case["resid"])
    q = models.InferenceNetwork(case["n"] * case["m"], case["z_dim"], case["q_hidden"], num_layers=case["q_layers"], activation=act,
                                resid=case["resid"])
    assert list(p.state_dict().keys()) == list(inp["p_state"].keys())
    p.load_state_dict({k: torch.from_numpy(v) for k, v in inp["p_state"].items()})
    q.load_state_dict({k: torch.from_numpy(v) for k, v in inp["q_state"].items()})
    with torch.no_grad():
        mu, logstd = q(torch.from_numpy(inp["y"]))
        z = torch.exp(logstd) * torch.from_numpy(inp["r"]) + mu
        y_hat = p(torch.from_numpy(inp["x_coord"]), z).view(case["B"], -1)
    assert rel_err(y_hat.numpy(), gold["y_hat"]) < TOL
