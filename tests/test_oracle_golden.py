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
