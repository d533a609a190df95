#!/usr/bin/env python3
"""Do the two GEMM modes train the same model?  Runs K Adam steps of BASELINE cfg 2 from identical weights, data and noise in
a child process per mode and compares the ELBO trajectories and the final parameters.

    python tools/mode_drift.py [--steps 200]          (needs the GPU)
"""
import argparse
import contextlib
import io
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mode, steps, out):
    import numpy as np
    import torch
    import torch.nn as nn
    import spatial_vae.models as models
    from spatial_vae_amd import _lib, dp, elbo as E
    _lib.set_gemm_mode(mode)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(2, 500, n_out=1, num_layers=2, activation=nn.Tanh).to(dev)
        q = models.InferenceNetwork(784, 5, 500, num_layers=2, activation=nn.Tanh).to(dev)
    x0, x1 = np.meshgrid(np.linspace(-1, 1, 28), np.linspace(1, -1, 28))
    x = torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1).astype(np.float32)).to(dev)
    rs = np.random.RandomState(7)
    data = torch.from_numpy((np.floor(rs.uniform(size=(8, 256, 784)) * (rs.uniform(size=(8, 256, 784)) > 0.8) * 255) / 255)
                            .astype(np.float32)).to(dev)
    noise = torch.from_numpy(rs.normal(size=(steps, 256, 5)).astype(np.float32)).to(dev)
    step = dp.TrainStep(p, q, E.eval_minibatch_mnist, lr=1e-3, rotate=True, translate=True, dx_scale=0.1, theta_prior=math.pi / 4)
    elbos = []
    for i in range(steps):
        elbos.append(step(x, data[i % 8], noise=noise[i])[0].detach())
    torch.cuda.synchronize()
    flat = step.grads.flat_param.detach().cpu().numpy()
    np.savez(out, elbo=torch.stack(elbos).cpu().numpy(), params=flat)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--child", default=None)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args.child, args.steps, args.out)
    import numpy as np
    res = {}
    for mode in ("fp32", "fp16x3"):
        out = "/tmp/mode_drift_%s.npz" % mode
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, "--steps", str(args.steps), "--out", out],
                       check=True, cwd=ROOT)
        with np.load(out) as f:
            res[mode] = (f["elbo"], f["params"])
    (e0, p0), (e1, p1) = res["fp32"], res["fp16x3"]
    rel = np.abs(e1 - e0) / np.abs(e0)
    print(json.dumps({"steps": args.steps, "elbo_first": float(e0[0]), "elbo_last_fp32": float(e0[-1]), "elbo_last_fp16x3": float(e1[-1]),
                      "max_rel_elbo_diff": float(rel.max()), "rel_elbo_diff_at_last_step": float(rel[-1]),
                      "param_rel_l2_diff": float(np.linalg.norm(p1 - p0) / np.linalg.norm(p0)),
                      "note": "lr 1e-3, identical weights / data / noise in both runs"}))


if __name__ == "__main__":
    main()
