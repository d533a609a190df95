#!/usr/bin/env python3
"""Training-loop and --vanilla golden vectors, produced by running the REFERENCE on CPU (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_epoch_golden.py

* epoch_*.npz -- the reference's own train_epoch (train_mnist.py:127-171; train_particles.py:151-202) driven with a list of
  minibatches instead of a DataLoader, a fresh torch.optim.Adam over p_net then q_net parameters exactly as main() builds it
  (train_mnist.py:389-392), and the noise of every step supplied by patching Tensor.normal_ (as gen_golden.py does).  Stored:
  the (elbo, log_p, kl) of every step, the running means train_epoch returns per epoch, every parameter after the last
  step, and the running means eval_model (train_mnist.py:174-226, train_particles.py:205-245) then returns over the same
  minibatches with the trained networks.  One case runs two epochs with z_scale 0 then 1 (the --z-delay schedule of train_particles.py:500-504).
* vanilla_*.npz -- eval_minibatch with VanillaGenerator (models.py:135-172) as the scripts build it for --vanilla
  (rotate = translate = False; train_mnist.py:351-357, train_particles.py:446-452): elbo, log_p, kl, y_hat and the gradient of
  -elbo with respect to every parameter.
Data only: inputs are regenerated from cases.py's seeded streams.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (loads the reference modules read-only, stubs torchvision / skimage)
import cases as C  # noqa: E402


class _Noise(object):
    """Make x.data.new(B, z).normal_() (train_mnist.py:38) return the prepared draws, one per call."""

    def __init__(self, draws):
        self.draws, self.i, self.orig = list(draws), 0, torch.Tensor.normal_

    def __enter__(self):
        outer = self

        def fake(t, *a, **k):
            r = torch.from_numpy(outer.draws[outer.i])
            outer.i += 1
            assert tuple(t.shape) == tuple(r.shape), (t.shape, r.shape)
            return t.copy_(r)
        torch.Tensor.normal_ = fake
        return self

    def __exit__(self, *exc):
        torch.Tensor.normal_ = self.orig
        assert exc[0] is not None or self.i == len(self.draws)


def run_epoch_case(ec):
    case = ec["case"]
    inp = C.build_epoch_inputs(ec)
    p_net, q_net = G.build_nets(case, inp)
    params = list(p_net.parameters()) + list(q_net.parameters())        # train_mnist.py:389-391
    optim = torch.optim.Adam(params, lr=ec["lr"])
    x = torch.from_numpy(inp["x_coord"])
    y = torch.from_numpy(inp["y"])
    ctf = torch.from_numpy(inp["ctf"]) if inp["ctf"] is not None else None
    mask = torch.from_numpy(inp["mask"]) if inp["mask"] is not None else None
    bounds = np.cumsum((0,) + ec["batches"])
    mod = G.ref_mnist if case["script"] == "mnist" else G.ref_particles
    steps = []
    orig_eval = mod.eval_minibatch

    def recording_eval(*a, **k):
        res = orig_eval(*a, **k)
        steps.append([float(res[0]), float(res[1]), float(res[2])])
        return res

    mod.eval_minibatch = recording_eval
    means = []
    devnull = open(os.devnull, "w")
    stderr, sys.stderr = sys.stderr, devnull                             # the progress line
    try:
        for e, zs in enumerate(ec["z_scales"]):
            if case["script"] == "mnist":
                it = [(y[bounds[i]:bounds[i + 1]],) for i in range(len(ec["batches"]))]
                with _Noise(inp["r"][e]):
                    acc = mod.train_epoch(it, x, p_net, q_net, optim, rotate=case["rotate"], translate=case["translate"],
                                          dx_scale=case["dx_scale"], theta_prior=case["theta_prior"], epoch=e,
                                          num_epochs=len(ec["z_scales"]), N=int(bounds[-1]), use_cuda=False)
            else:
                it = [((y[bounds[i]:bounds[i + 1]], ctf[bounds[i]:bounds[i + 1]]) if ctf is not None else (y[bounds[i]:bounds[i + 1]],))
                      for i in range(len(ec["batches"]))]
                with _Noise(inp["r"][e]):
                    acc = mod.train_epoch(it, x, mask, p_net, q_net, optim, rotate=case["rotate"], translate=case["translate"],
                                          dx_scale=case["dx_scale"], theta_prior=case["theta_prior"], augment_rotation=False,
                                          z_scale=zs, epoch=e, num_epochs=len(ec["z_scales"]), N=int(bounds[-1]), use_cuda=False)
            means.append([float(v) for v in acc])                       # (elbo_accum, gen/bce_loss_accum, kl_loss_accum)
        # the evaluation pass of the same script over the same minibatches with the trained networks (eval_model:
        # train_mnist.py:174-226, train_particles.py:205-245): forward only, its own noise draws
        nsteps = len(steps)
        with _Noise(inp["r_eval"]), torch.no_grad():
            if case["script"] == "mnist":
                ev = mod.eval_model(it, x, p_net, q_net, rotate=case["rotate"], translate=case["translate"],
                                    dx_scale=case["dx_scale"], theta_prior=case["theta_prior"], use_cuda=False)
            else:
                ev = mod.eval_model(it, x, mask, p_net, q_net, rotate=case["rotate"], translate=case["translate"],
                                    dx_scale=case["dx_scale"], theta_prior=case["theta_prior"], z_scale=ec["z_scales"][-1],
                                    use_cuda=False)
        eval_steps = steps[nsteps:]
        del steps[nsteps:]
    finally:
        sys.stderr = stderr
        mod.eval_minibatch = orig_eval
    out = {"steps": np.array(steps, np.float64), "means": np.array(means, np.float64),
           "eval_steps": np.array(eval_steps, np.float64), "eval_means": np.array([float(v) for v in ev], np.float64)}
    for k, p in p_net.named_parameters():
        out["p." + k] = p.detach().numpy()
    for k, p in q_net.named_parameters():
        out["q." + k] = p.detach().numpy()
    return out


def run_vanilla_case(case):
    inp = C.build_vanilla_inputs(case)
    act = G.ACT[case["act"]]
    devnull = open(os.devnull, "w")
    stdout, sys.stdout = sys.stdout, devnull
    try:
        p_net = G.ref_models.VanillaGenerator(case["n"] * case["m"], case["z_dim"], case["H"], n_out=case["n_out"],
                                              num_layers=case["L"], activation=act, softplus=case["softplus"], resid=case["resid"])
        q_net = G.ref_models.InferenceNetwork(case["n"] * case["m"], case["z_dim"], case["q_hidden"], num_layers=case["q_layers"],
                                              activation=act, resid=case["resid"])
    finally:
        sys.stdout = stdout
    assert list(p_net.state_dict().keys()) == list(inp["p_state"].keys()), (list(p_net.state_dict().keys()), list(inp["p_state"].keys()))
    p_net.load_state_dict({k: torch.from_numpy(v) for k, v in inp["p_state"].items()})
    q_net.load_state_dict({k: torch.from_numpy(v) for k, v in inp["q_state"].items()})
    x, y = torch.from_numpy(inp["x_coord"]), torch.from_numpy(inp["y"])
    kw = dict(rotate=False, translate=False, dx_scale=case["dx_scale"], theta_prior=case["theta_prior"], use_cuda=False)
    out = {}
    with _Noise([inp["r"]]):
        if case["script"] == "mnist":
            elbo, log_p, kl, y_hat = G.ref_mnist.eval_minibatch(x, y, p_net, q_net, **kw)
            out["y_hat"] = y_hat.detach().numpy()
        else:
            mask = torch.from_numpy(inp["mask"]) if inp["mask"] is not None else None
            elbo, log_p, kl = G.ref_particles.eval_minibatch(x, y, mask, None, p_net, q_net, **kw)
    (-elbo).backward()
    out.update(elbo=elbo.detach().numpy(), log_p=log_p.detach().numpy(), kl=kl.detach().numpy())
    for k, p in p_net.named_parameters():
        out["gp." + k] = p.grad.numpy()
    for k, p in q_net.named_parameters():
        out["gq." + k] = p.grad.numpy()
    return out


def main():
    torch.set_num_threads(4)
    for ec in C.EPOCH_CASES:
        out = run_epoch_case(ec)
        np.savez_compressed(os.path.join(HERE, ec["name"] + ".npz"), **out)
        print("%-28s steps %d  means %s" % (ec["name"], len(out["steps"]), np.round(out["means"], 4).tolist()))
    for case in C.VANILLA_CASES:
        out = run_vanilla_case(case)
        np.savez_compressed(os.path.join(HERE, case["name"] + ".npz"), **out)
        print("%-34s elbo=% .6f log_p=% .6f kl=%.6f" % (case["name"], out["elbo"], out["log_p"], out["kl"]))


if __name__ == "__main__":
    main()
