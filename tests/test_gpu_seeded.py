"""Seeded-run equivalence (coverage row R; BASELINE.json north_star: "identical inputs/seeds"): the three command lines, given
only `--seed s --synthetic N`, must follow the trajectory of the REFERENCE's CPU path run under torch.manual_seed(s) /
np.random.seed(s) with nothing patched (tests/golden/gen_seeded_golden.py: default initialisation, real
DataLoader(shuffle=True), x.data.new(B, z).normal_() on the CPU generator, np.random augmentation angles through Pillow, the
display helpers' draws on dump epochs).  Compared: every training step's (elbo, log_p, kl), the rows the scripts print (training
and validation running means per epoch) and every parameter after the last epoch.  Tolerances as for the loop fixtures
(tests/test_gpu_epoch.py): steps and rows 2e-5 relative to the largest entry, parameters 1e-4."""
import importlib

import numpy as np
import pytest
import torch

import cases as C
from helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL_STEP, TOL_PARAM = 2e-5, 1e-4


@pytest.mark.parametrize("name", [c["name"] for c in C.SEEDED_CASES])
def test_command_line_follows_the_reference_from_the_seed_alone(name, tmp_path, monkeypatch, capsys):
    from spatial_vae_amd import cli, dp
    case = C.SEEDED_CASES_BY_NAME[name]
    gold = load_golden(name)
    script = case["script"]
    mod = importlib.import_module("train_" + script)
    args = getattr(mod, {"mnist": "mnist_arguments", "galaxy": "galaxy_arguments", "particles": "particle_arguments"}[script])(case["argv"])
    monkeypatch.chdir(tmp_path)
    made, per_step = [], []
    orig_init, orig_step = dp.TrainStep.__init__, dp.TrainStep._step

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        made.append(self)

    def step(self, x, batch, weight, kw):
        out = orig_step(self, x, batch, weight, kw)
        per_step.append(self.metrics.detach().clone())
        return out

    monkeypatch.setattr(dp.TrainStep, "__init__", init)
    monkeypatch.setattr(dp.TrainStep, "_step", step)
    assert cli.train_main(script, args, mod.build) == 0
    torch.cuda.synchronize()
    out = capsys.readouterr().out
    table = [l.split("\t") for l in out.splitlines() if "\t" in l][1:]
    if script == "particles":                                    # Epoch, Split, ELBO, Error, KL
        rows = np.array([[float(v) for v in table[2 * e][2:]] + [float(v) for v in table[2 * e + 1][2:]] for e in range(case["epochs"])])
    else:                                                        # Epoch, ELBO, BCE loss, KL; a training then a validation line
        rows = np.array([[float(v) for v in table[2 * e][1:]] + [float(v) for v in table[2 * e + 1][1:]] for e in range(case["epochs"])])
    # the fixture lists every minibatch in order: per epoch the training steps, then the evaluation steps
    ntr = -(-case["count"] // case["bs"])
    nev = -(-max(case["count"] // 4, 1) // case["bs"])
    train_rows = [e * (ntr + nev) + i for e in range(case["epochs"]) for i in range(ntr)]
    want = gold["steps"][train_rows]
    got = torch.stack(per_step).cpu().double().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= TOL_STEP * np.abs(want).max(), (got, want)
    assert np.abs(rows - gold["rows"]).max() <= TOL_STEP * np.abs(gold["rows"]).max(), (rows, gold["rows"])
    (ts,) = made
    for k, v in ts.p_net.state_dict().items():
        assert rel_err(v.cpu().numpy(), gold["p." + k]) < TOL_PARAM, k
    for k, v in ts.q_net.state_dict().items():
        assert rel_err(v.cpu().numpy(), gold["q." + k]) < TOL_PARAM, k
