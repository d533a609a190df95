#!/usr/bin/env python3
"""Per-kernel time of one training step at a BASELINE config (2..5), through dp.TrainStep:  python tools/step_profile.py --cfg 5
cfg 2: MNIST 28x28 z=2 H=500x2 B=256 (BCE) | cfg 3: particles 40x40 z=2 H=500x2 C=2 (fit-noise) B=512 |
cfg 4: galaxy 128x128 RGB z=20 H=1024x3 B=128 (BCE) | cfg 5: particles 40x40 z=8 H=500x2 B=256 + CTF 39x39."""
import argparse
import contextlib
import io
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

CFGS = {2: dict(script="mnist", B=256, n=28, z=2, H=500, L=2, C=1, qh=500),
        3: dict(script="particles", B=512, n=40, z=2, H=500, L=2, C=2, qh=500),
        4: dict(script="galaxy", B=128, n=128, z=20, H=1024, L=3, C=3, qh=5000),
        5: dict(script="particles", B=256, n=40, z=8, H=500, L=2, C=1, qh=500, ctf=True)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=5, choices=sorted(CFGS))
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    c = CFGS[args.cfg]
    import spatial_vae.models as models
    from spatial_vae_amd import _lib, dp, elbo as E
    dev = torch.device("cuda:0")
    B, n, z = c["B"], c["n"], c["z"]
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(z, c["H"], n_out=c["C"], num_layers=c["L"], activation=nn.Tanh).to(dev)
        n_in = n * n * (c["C"] if c["script"] == "galaxy" else 1)
        q = models.InferenceNetwork(n_in, z + 3, c["qh"], num_layers=2, activation=nn.Tanh).to(dev)
    x0, x1 = np.meshgrid(np.linspace(-1, 1, n), np.linspace(1, -1, n))
    x = torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1).astype(np.float32)).to(dev)
    fn = {"mnist": E.eval_minibatch_mnist, "galaxy": E.eval_minibatch_galaxy, "particles": E.eval_minibatch_particles}[c["script"]]
    if c["script"] == "particles":
        batch = (torch.randn(B, n * n, device=dev), None, torch.randn(B, 1, n - 1, n - 1, device=dev) / n if c.get("ctf") else None)
    elif c["script"] == "galaxy":
        batch = (torch.rand(B, n * n, c["C"], device=dev),)
    else:
        batch = (torch.rand(B, n * n, device=dev),)
    step = dp.TrainStep(p, q, fn, lr=1e-4, rotate=True, translate=True, dx_scale=0.1, theta_prior=math.pi)
    for _ in range(3):
        step(x, *batch)
    torch.cuda.synchronize()
    _lib.profile_enable(2)
    _lib.profile_read()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(x, *batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    prof = _lib.profile_read()
    _lib.profile_enable(0)
    print("cfg %d: step %.3f ms = %.0f images/s (%s)" % (args.cfg, ms, B / ms * 1e3, _lib.gemm_mode()))
    for k, (t, cnt) in sorted(prof.items()):
        print("  %-14s %9.4f ms/step" % (k, t / args.steps))
    print("  %-14s %9.4f ms/step (everything else: encoder, autograd glue)" % ("outside", ms - sum(t for t, _ in prof.values()) / args.steps))


if __name__ == "__main__":
    main()
