#!/usr/bin/env python3
"""Per-kernel time of one BASELINE cfg 5 training step (particles 40x40, z=8, H=500x2, B=256, CTF 39x39, mask)."""
import contextlib
import io
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402


def main():
    import spatial_vae.models as models
    from spatial_vae_amd import _lib, dp, elbo as E
    dev = torch.device("cuda:0")
    B, n, z = 256, 40, 8
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(z, 500, n_out=1, num_layers=2, activation=nn.Tanh).to(dev)
        q = models.InferenceNetwork(n * n, z + 3, 500, num_layers=2, activation=nn.Tanh).to(dev)
    x0, x1 = np.meshgrid(np.linspace(-1, 1, n), np.linspace(1, -1, n))
    x = torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1).astype(np.float32)).to(dev)
    y = torch.randn(B, n * n, device=dev)
    ctf = torch.randn(B, 1, 39, 39, device=dev) / 39
    step = dp.TrainStep(p, q, E.eval_minibatch_particles, lr=1e-4, rotate=True, translate=True, dx_scale=0.1, theta_prior=math.pi)
    for _ in range(5):
        step(x, y, None, ctf)
    torch.cuda.synchronize()
    _lib.profile_enable(2)
    _lib.profile_read()
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        step(x, y, None, ctf)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    prof = _lib.profile_read()
    _lib.profile_enable(0)
    print("step %.3f ms (%s)" % (ms, _lib.gemm_mode()))
    for k, (t, c) in sorted(prof.items()):
        print("  %-14s %8.4f ms/step" % (k, t / 10))


if __name__ == "__main__":
    main()
