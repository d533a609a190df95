#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation on CPU.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  The reference modules are loaded by file path, read-only; the
two third-party imports they pull in that are absent from this image
(torchvision, skimage -- neither is on the hot path) are
satisfied with empty stub modules.  Outputs go to tests/golden/<case>.npz and
contain data only: the reference's outputs (and gradients) for the inputs that
tests/golden/cases.py regenerates deterministically.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [case ...]

What is captured per case:
  * eval_minibatch(...) of the script the case names
    (/root/reference/train_mnist.py:24-90, train_galaxy.py:27-128,
    train_particles.py:22-148): elbo, log_p_x_g_z, kl_div, y_hat, the
    pre-Sigmoid logits (forward hook on p_net.layers[-2]), the encoder outputs
    and d(-elbo)/d(every parameter of p_net and q_net);
  * the decoder alone, SpatialGenerator.forward(x, z)
    (/root/reference/spatial_vae/models.py:90-132) on explicit coordinates with
    an explicit upstream gradient: y, logits, d/d(params), d/dx, d/dz.
"""
import importlib.util
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# the repo root also holds a package called spatial_vae; make sure the reference wins
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

for name in ("torchvision", "torchvision.utils", "torchvision.datasets", "skimage", "skimage.transform"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
sys.modules["skimage.transform"].resize = lambda *a, **k: None

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


import spatial_vae.models as ref_models  # noqa: E402  (the reference's, see sys.path above)

assert os.path.abspath(ref_models.__file__).startswith(REF), ref_models.__file__
ref_mnist = _load("ref_train_mnist", os.path.join(REF, "train_mnist.py"))
ref_galaxy = _load("ref_train_galaxy", os.path.join(REF, "train_galaxy.py"))
ref_particles = _load("ref_train_particles", os.path.join(REF, "train_particles.py"))

sys.path.insert(0, HERE)
import cases as C  # noqa: E402

ACT = {"tanh": nn.Tanh, "leakyrelu": nn.LeakyReLU, "relu": nn.ReLU, "sigmoid": nn.Sigmoid}


def build_nets(case, inp):
    act = ACT[case["act"]]
    n_pix = case["n"] * case["m"]
    devnull = open(os.devnull, "w")
    stdout, sys.stdout = sys.stdout, devnull        # the constructors print(self)
    try:
        p_net = ref_models.SpatialGenerator(case["z_dim"], case["H"], n_out=case["n_out"], num_layers=case["L"],
                                            activation=act, softplus=case["softplus"], resid=case["resid"],
                                            expand_coords=case["expand_coords"], bilinear=case["bilinear"])
        n_in = n_pix * (case["n_out"] if case["script"] == "galaxy" else 1)
        q_net = ref_models.InferenceNetwork(n_in, C.inf_dim(case), case["q_hidden"], num_layers=case["q_layers"],
                                            activation=act, resid=case["resid"])
    finally:
        sys.stdout = stdout
    p_net.load_state_dict({k: torch.from_numpy(v) for k, v in inp["p_state"].items()})
    q_net.load_state_dict({k: torch.from_numpy(v) for k, v in inp["q_state"].items()})
    return p_net, q_net


def run_case(case):
    inp = C.build_inputs(case)
    p_net, q_net = build_nets(case, inp)
    x = torch.from_numpy(inp["x_coord"])
    y = torch.from_numpy(inp["y"])
    r = torch.from_numpy(inp["r"])
    out = {}

    # --- make the reference's noise draw return our r: eval_minibatch draws
    # x.data.new(B, z_dim).normal_() (train_mnist.py:38) from the global CPU
    # generator; patch Tensor.normal_ for the duration of the call.
    captured = {}
    hook_logits = p_net.layers[-2].register_forward_hook(lambda m, i, o: captured.__setitem__("logits", o.detach().clone()))
    def _q_hook(mod, i, o):
        o.retain_grad()
        captured["q_out"] = o            # returns None: a hook's return value would replace the output

    hook_q = q_net.layers.register_forward_hook(_q_hook)
    hook_in = q_net.layers.register_forward_pre_hook(lambda mod, i: captured.__setitem__("q_in", i[0].detach().clone()))
    if case["augment"]:
        np.random.seed(case["seed"])                 # the reference draws offset from the global numpy generator
    orig_normal = torch.Tensor.normal_

    def fake_normal(self, *a, **k):
        assert self.shape[0] == r.shape[0] and self.shape[1] <= r.shape[1]
        return self.copy_(r[:, :self.shape[1]])

    torch.Tensor.normal_ = fake_normal
    try:
        kw = dict(rotate=case["rotate"], translate=case["translate"], dx_scale=case["dx_scale"],
                  theta_prior=case["theta_prior"], use_cuda=False)
        if case["script"] == "mnist":
            elbo, log_p, kl, y_hat = ref_mnist.eval_minibatch(x, y, p_net, q_net, **kw)
        elif case["script"] == "galaxy":
            elbo, log_p, kl, y_hat = ref_galaxy.eval_minibatch(x, y, p_net, q_net, augment_rotation=case["augment"],
                                                               z_scale=case["z_scale"], **kw)
        else:
            mask = torch.from_numpy(inp["mask"]) if inp["mask"] is not None else None
            ctf = torch.from_numpy(inp["ctf"]) if inp["ctf"] is not None else None
            elbo, log_p, kl = ref_particles.eval_minibatch(x, y, mask, ctf, p_net, q_net, augment_rotation=case["augment"],
                                                           z_scale=case["z_scale"], **kw)
            y_hat = None
    finally:
        torch.Tensor.normal_ = orig_normal
    (-elbo).backward()                               # train_mnist.py:147-148
    hook_logits.remove()
    hook_q.remove()
    hook_in.remove()
    if case["augment"]:
        out["y_rot"] = captured["q_in"].numpy().reshape(inp["y"].shape)      # what Pillow made of y

    out["elbo"] = elbo.detach().numpy()
    out["log_p"] = log_p.detach().numpy()
    out["kl"] = kl.detach().numpy()
    if y_hat is not None:
        out["y_hat"] = y_hat.detach().numpy()
    out["logits"] = captured["logits"].numpy().reshape(case["B"], case["n"] * case["m"], case["n_out"])
    out["q_out"] = captured["q_out"].detach().numpy()
    out["q_out_grad"] = captured["q_out"].grad.numpy()

    def put_grad(prefix, name, g):
        g = g.detach().numpy().astype(np.float32)
        if case["store"] == "sampled" and g.size > 4096:
            idx = C.sample_idx(g.size)
            out[prefix + name + "@sample"] = g.reshape(-1)[idx]
            out[prefix + name + "@norm"] = np.array(np.linalg.norm(g.astype(np.float64)))
        else:
            out[prefix + name] = g

    for k, p in p_net.named_parameters():
        put_grad("gp.", k, p.grad)
    for k, p in q_net.named_parameters():
        put_grad("gq.", k, p.grad)

    # --- forward-only display paths (train_mnist.py:93-124; train_galaxy.py:131-183) ---------------
    if case["name"] in C.DISPLAY_CASES:
        torch.Tensor.normal_ = fake_normal
        try:
            with torch.no_grad():
                if case["script"] == "mnist":
                    out["display.y_hat"] = ref_mnist.minibatch_for_display(
                        x, y, p_net, q_net, rotate=case["rotate"], translate=case["translate"], use_cuda=False).numpy()
                else:
                    out["display.y_hat"] = ref_galaxy.minibatch_for_display(
                        x, y, q_net, p_net, rotate=case["rotate"], translate=case["translate"], z_scale=case["z_scale"],
                        use_cuda=False).numpy()
                    out["random.y_hat"] = ref_galaxy.random_minibatch_generator(
                        x, y, p_net, case["z_dim"], z_scale=case["z_scale"], use_cuda=False).numpy()
        finally:
            torch.Tensor.normal_ = orig_normal

    # --- decoder-only entry -------------------------------------------------
    p_net.zero_grad()
    dx = torch.from_numpy(inp["dec_x"]).requires_grad_(True)
    dz = torch.from_numpy(inp["dec_z"]).requires_grad_(True)
    hook_logits = p_net.layers[-2].register_forward_hook(lambda m, i, o: captured.__setitem__("dlogits", o.detach().clone()))
    yd = p_net(dx, dz)
    hook_logits.remove()
    yd.backward(torch.from_numpy(inp["dec_dy"]))
    out["dec.y"] = yd.detach().numpy()
    out["dec.logits"] = captured["dlogits"].numpy().reshape(yd.shape)
    out["dec.dx"] = dx.grad.numpy()
    if case["z_dim"] > 0:
        out["dec.dz"] = dz.grad.numpy()
    for k, p in p_net.named_parameters():
        put_grad("dec.gp.", k, p.grad)
    return out


def main():
    if "--ctf" in sys.argv:
        return
    names = sys.argv[1:] or [c["name"] for c in C.CASES]
    torch.set_num_threads(4)
    total = 0
    for nm in names:
        case = C.CASES_BY_NAME[nm]
        out = run_case(case)
        path = os.path.join(HERE, nm + ".npz")
        np.savez_compressed(path, **out)
        sz = os.path.getsize(path)
        total += sz
        print("%-28s elbo=% .6f log_p=% .6f kl=%.6f  (%d bytes)" % (nm, out["elbo"], out["log_p"], out["kl"], sz))
    print("total bytes:", total)


if __name__ == "__main__":
    main()


def gen_ctf_fixture():
    """Reference CTF filter bank (spatial_vae/ctf.py:33-56) on a small seeded parameter table."""
    ref_ctf = _load("ref_ctf", os.path.join(REF, "spatial_vae", "ctf.py"))
    rs = np.random.RandomState(5)
    tab = np.stack([rs.uniform(1, 3, 6), np.full(6, 2.7), np.full(6, 300.0), rs.uniform(1.0, 2.5, 6), np.full(6, 100.0),
                    np.full(6, 10.0), np.zeros(6), rs.uniform(0, 180, 6)], 1)
    path = os.path.join(HERE, "ctf_table.txt")
    np.savetxt(path, tab)
    out = {"table": tab}
    for n, m, scale in ((9, 9, 1), (15, 13, 2), (39, 39, 1)):
        out["filt_%dx%d_s%d" % (n, m, scale)] = ref_ctf.ctf_filter(ref_ctf.parse_ctf(path), n, m, scale=scale)
    # two particles at sizes beyond the device kernel's in-LDS limit (the reference's numpy code has none)
    first2 = os.path.join(HERE, "_ctf_first2.txt")
    np.savetxt(first2, tab[:2])
    for n, m, scale in ((129, 129, 1), (100, 96, 2)):
        out["big_%dx%d_s%d" % (n, m, scale)] = ref_ctf.ctf_filter(ref_ctf.parse_ctf(first2), n, m, scale=scale)
    os.remove(first2)
    np.savez_compressed(os.path.join(HERE, "ctf_golden.npz"), **out)
    print("ctf fixture written")


if __name__ == "__main__" and "--ctf" in sys.argv:
    gen_ctf_fixture()
