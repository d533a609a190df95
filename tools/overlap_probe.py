#!/usr/bin/env python3
"""Does an HBM-bound pass (coordinate layer, output-layer backward) hide under an MFMA-bound GEMM of ANOTHER row chunk?
Times the decoder forward+backward at BASELINE cfg 2 as one 256-image call on one stream against two 128-image calls on two
streams (each with its own scratch), which lets the hardware overlap chunk B's streaming passes with chunk A's GEMMs.

    python tools/overlap_probe.py [--iters 20]
"""
import argparse
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--chunks", type=int, default=2)
    args = ap.parse_args()
    import spatial_vae.models as models
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(2, 500, n_out=1, num_layers=2, activation=nn.Tanh).to(dev)
    n, B = 28, args.B
    N = n * n
    x0, x1 = np.meshgrid(np.linspace(-1, 1, n), np.linspace(1, -1, n))
    grid = torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1).astype(np.float32)).to(dev)
    theta = torch.randn(B, device=dev)
    dx = 0.1 * torch.randn(B, 2, device=dev)
    z = torch.randn(B, 2, device=dev)
    dy = torch.randn(B, N, 1, device=dev) / N
    streams = [torch.cuda.Stream() for _ in range(args.chunks)]
    per = B // args.chunks

    def full():
        y = p.forward_posed(grid, B, theta=theta.requires_grad_(True), dx=dx.requires_grad_(True), z=z.requires_grad_(True))
        y.backward(dy)
        p.zero_grad(set_to_none=True)

    def split():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            sl = slice(i * per, (i + 1) * per)
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                y = p.forward_posed(grid, per, theta=theta[sl].detach().requires_grad_(True), dx=dx[sl].detach().requires_grad_(True),
                                    z=z[sl].detach().requires_grad_(True))
                y.backward(dy[sl])
        for s in streams:
            cur.wait_stream(s)
        p.zero_grad(set_to_none=True)

    for name, fn in (("one call, one stream", full), ("%d chunks on %d streams" % (args.chunks, args.chunks), split),
                     ("one call, one stream", full)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        print("%-28s %.4f ms per decoder fwd+bwd" % (name, (time.perf_counter() - t0) * 1e3 / args.iters))


if __name__ == "__main__":
    main()
