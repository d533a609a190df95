"""Child process of tests/test_gpu_split.py (the GEMM mode is fixed per process).  Prints one JSON line.

    python tests/split_child.py dispatch            which kernel families each geometry dispatched (svae_path_counts)
    python tests/split_child.py adversarial         decoder forward/backward errors against a float64 CPU reference
"""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

ACT = {"tanh": nn.Tanh, "relu": nn.ReLU, "sigmoid": nn.Sigmoid, "leakyrelu": nn.LeakyReLU}


def build(H, L, C, act, zd=2, seed=0, wscale=1.0):
    import spatial_vae.models as models
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        p = models.SpatialGenerator(zd, H, n_out=C, num_layers=L, activation=ACT[act])
    if wscale != 1.0:
        with torch.no_grad():
            for q in p.parameters():
                q.mul_(wscale)
    return p


def run_gpu(p, x, z, dy):
    dev = torch.device("cuda:0")
    p = p.to(dev)
    xg = x.to(dev).requires_grad_(True)
    zg = z.to(dev).requires_grad_(True)
    y = p(xg, zg)
    y.backward(dy.to(dev))
    torch.cuda.synchronize()
    out = {"y": y.detach().cpu().double(), "dx": xg.grad.cpu().double(), "dz": zg.grad.cpu().double()}
    for k, q in p.named_parameters():
        out["g." + k] = q.grad.detach().cpu().double()
    p.zero_grad(set_to_none=True)
    return out


def run_ref64(p, x, z, dy):
    """The same decoder in float64 on the CPU (oracle/torch_cpu_step.decoder: plain torch ops of models.py:90-132)."""
    from oracle import torch_cpu_step as T
    pp = {k: v.detach().cpu().double().requires_grad_(True) for k, v in p.state_dict().items()}
    x64 = x.double().requires_grad_(True)
    z64 = z.double().requires_grad_(True)
    act = {nn.Tanh: "tanh", nn.ReLU: "relu", nn.Sigmoid: "sigmoid", nn.LeakyReLU: "leakyrelu"}[type(p.layers[0])]
    y = T.decoder(pp, x64, z64, act)
    y.backward(dy.double())
    out = {"y": y.detach(), "dx": x64.grad, "dz": z64.grad}
    for k, v in pp.items():
        out["g." + k] = v.grad
    return out


def rel(a, b):
    d = float(b.abs().max())
    return float((a - b).abs().max()) / d if d > 0 else float((a - b).abs().max())


def dispatch():
    from spatial_vae_amd import _lib
    cases = {"H20_tanh": (20, 2, 1, "tanh", 49, 3), "H33_L3_tanh": (33, 3, 1, "tanh", 49, 3), "H70_tanh": (70, 2, 1, "tanh", 49, 3), "H64_tanh": (64, 2, 1, "tanh", 64, 4),
             "H128_tanh": (128, 2, 1, "tanh", 64, 4), "H256_sigmoid": (256, 2, 1, "sigmoid", 64, 4),
             "H500_tanh_cfg2": (500, 2, 1, "tanh", 784, 8), "H500_tanh_C2_cfg3": (500, 2, 2, "tanh", 1600, 4),
             "H1024_L3_C3_cfg4": (1024, 3, 3, "tanh", 1024, 2), "H512_relu": (512, 2, 1, "relu", 64, 4),
             "H500_leaky": (500, 2, 1, "leakyrelu", 64, 4)}
    res = {"mode": _lib.gemm_mode()}
    for name, (H, L, C, act, N, B) in cases.items():
        p = build(H, L, C, act)
        x = torch.rand(B, N, 2) * 2 - 1
        z = torch.randn(B, 2)
        dy = torch.randn(B, N, C) / N
        _lib.path_counts(reset=True)
        run_gpu(p, x, z, dy)
        res[name] = {k: v for k, v in _lib.path_counts(reset=True).items() if v}
    print(json.dumps(res))


def adversarial():
    """Operands the scaling of the split mode was NOT tuned on: weights 40x the default init (max |W| >> 1, tanh saturated
    over most of the plane), an upstream gradient spanning 2^-20 .. 2^0 across pixels, an all-zero upstream gradient
    (amax = 0), and the plain case for reference.  Errors are against float64."""
    from spatial_vae_amd import _lib
    H, L, C, N, B = 512, 2, 1, 784, 4
    res = {"mode": _lib.gemm_mode()}
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, N, 2, generator=g) * 2 - 1
    z = torch.randn(B, 2, generator=g)
    dy_plain = torch.randn(B, N, C, generator=g) / N
    ramp = torch.pow(2.0, -20.0 * torch.rand(B, N, C, generator=g))          # per-pixel scale in [2^-20, 1]
    cases = {"plain": (1.0, dy_plain), "weights_x40": (40.0, dy_plain), "weights_x8": (8.0, dy_plain),
             "dy_range_2^-20": (1.0, dy_plain * ramp * N), "dy_zero": (1.0, torch.zeros(B, N, C)),
             "dy_huge": (1.0, dy_plain * 1e6), "dy_tiny": (1.0, dy_plain * 1e-12)}
    for name, (ws, dy) in cases.items():
        p = build(H, L, C, "tanh", wscale=ws)
        _lib.path_counts(reset=True)
        got = run_gpu(p, x, z, dy)
        paths = {k: v for k, v in _lib.path_counts(reset=True).items() if v}
        ref = run_ref64(p.cpu(), x, z, dy)
        errs = {k: rel(got[k], ref[k]) for k in got}
        finite = all(bool(torch.isfinite(v).all()) for v in got.values())
        res[name] = {"errs": errs, "finite": finite, "paths": paths,
                     "all_zero": all(float(v.abs().max()) == 0.0 for k, v in got.items() if k != "y")}
    print(json.dumps(res))


if __name__ == "__main__":
    {"dispatch": dispatch, "adversarial": adversarial}[sys.argv[1]]()
