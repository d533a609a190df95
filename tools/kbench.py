#!/usr/bin/env python3
"""Kernel-level timing of the decoder path at BASELINE cfg 2 (or --B/--n/--H/--L), through the C-ABI.

    python tools/kbench.py [--iters 10] [--fwd-only]

Prints the per-kernel HIP-event times (svae_profile_*).  Under rocprofv3 --pmc it is the small,
quiet workload the counter passes are taken on.
"""
import argparse
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--n", type=int, default=28)
    ap.add_argument("--H", type=int, default=500)
    ap.add_argument("--L", type=int, default=2)
    ap.add_argument("--z", type=int, default=2)
    ap.add_argument("--C", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--fwd-only", action="store_true")
    args = ap.parse_args()
    import spatial_vae.models as models
    from spatial_vae_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(args.z, args.H, n_out=args.C, num_layers=args.L, activation=nn.Tanh).to(dev)
    N = args.n * args.n
    x0, x1 = np.meshgrid(np.linspace(-1, 1, args.n), np.linspace(1, -1, args.n))
    grid = torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1).astype(np.float32)).to(dev)
    theta = torch.randn(args.B, device=dev, requires_grad=True)
    dx = (0.1 * torch.randn(args.B, 2, device=dev)).requires_grad_(True)
    z = torch.randn(args.B, args.z, device=dev, requires_grad=True)
    dy = torch.randn(args.B, N, args.C, device=dev) / N

    def it():
        if args.fwd_only:
            with torch.no_grad():
                p.forward_posed(grid, args.B, theta=theta, dx=dx, z=z)
        else:
            y = p.forward_posed(grid, args.B, theta=theta, dx=dx, z=z)
            y.backward(dy)
            p.zero_grad(set_to_none=True)

    for _ in range(3):
        it()
    torch.cuda.synchronize()
    _lib.profile_enable(2)
    _lib.profile_read()
    for _ in range(args.iters):
        it()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(0)
    M = args.B * N
    gf = 2.0 * M * args.H * args.H / 1e9
    for k, (ms, cnt) in sorted(prof.items()):
        avg = ms / cnt
        extra = "  %.1f TFLOP/s algorithmic" % (gf / avg) if k in ("dense_fwd", "dense_dgrad", "wgrad") else ""
        print("%-14s %8.4f ms x %d%s" % (k, avg, cnt, extra))
    print("total decoder kernel time per fwd+bwd: %.4f ms" % (sum(ms for ms, _ in prof.values()) / args.iters))


if __name__ == "__main__":
    main()
