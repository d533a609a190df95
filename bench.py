#!/usr/bin/env python3
"""Benchmark of the spatial-VAE ELBO training step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = forward + backward + Adam step of the ELBO on one minibatch per GPU
(train_mnist.py:143-150 of the reference), on BASELINE.json configs[1]: rotated+translated MNIST
shape 28x28, z=2, p/q hidden 500 x 2 layers, tanh, batch 256 PER GPU (weak scaling: the global
batch is 256*N, sharded as spatial_vae_amd/dp.py describes, one RCCL all-reduce of the flat
gradient per step).  Inputs are synthetic (there are no datasets here) and resident in HBM before
the timed region.  Rank 0 prints ONE JSON line.

The line also carries
  roofline     : the dominant kernel (an fp32-MFMA GEMM of the decoder's 500x500 layer) --
                 algorithmic FLOPs per launch / its average launch duration, the durations taken
                 live from HIP events recorded on the launch stream during the timed steps
                 (svae_profile_* in include/svae.h); peak = 157.3 TFLOP/s fp32 MFMA.
  cpu_baseline : the torch-CPU restatement of the reference's step (oracle/torch_cpu_step.py),
                 timed on this box's host cores on the same workload (rank 0, N=1 only).
  fp16x3_mode  : (N=1 only) the same workload measured in a child process with SVAE_GEMM=fp16x3 -- the hidden-layer
                 GEMMs on the f16 matrix pipe with split (hi + lo) operands, fp32-accurate (it passes the same parity
                 tests; spatial_vae_amd/csrc/split.h).  Reported beside the headline, which stays the fp32-MFMA path:
                 `value`, `dtype` and `roofline` at the top level are that path's.
--gemm fp16x3 makes that mode the measured one (its roofline then counts EXECUTED f16 FLOPs, 3 per algorithmic one,
against the 2.5 PF f16 peak).
"""
import argparse
import contextlib
import io
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # same table, "Peak BF16/FP16 MFMA ~2.5 PF dense"

CFG = dict(name="mnist-rotated-translated 28x28 z=2 H=500x2 tanh B=256/GPU (BASELINE configs[1])",
           n=28, m=28, B=256, z_dim=2, H=500, L=2, q_hidden=500, q_layers=2, dx_scale=0.1, theta_prior=math.pi / 4,
           lr=1e-4)


def synthetic_batch(rs, B, N):
    """MNIST-like sparse uniform targets, uint8-quantised /255."""
    u = rs.uniform(size=(B, N))
    keep = rs.uniform(size=(B, N)) > 0.8
    return (np.floor(u * keep * 255.0) / 255.0).astype(np.float32)


def coord_grid(n, m):
    x0, x1 = np.meshgrid(np.linspace(-1, 1, m), np.linspace(1, -1, n))
    return np.stack([x0.ravel(), x1.ravel()], 1).astype(np.float32)


def build_nets(cfg):
    import spatial_vae.models as models
    torch.manual_seed(0)  # p_net before q_net, default nn.Linear init (train_mnist.py:370-375)
    with contextlib.redirect_stdout(io.StringIO()):
        p_net = models.SpatialGenerator(cfg["z_dim"], cfg["H"], n_out=1, num_layers=cfg["L"], activation=nn.Tanh)
        q_net = models.InferenceNetwork(cfg["n"] * cfg["m"], cfg["z_dim"] + 3, cfg["q_hidden"],
                                        num_layers=cfg["q_layers"], activation=nn.Tanh)
    return p_net, q_net


def decoder_flops(cfg):
    M = cfg["B"] * cfg["n"] * cfg["m"]
    H = cfg["H"]
    fwd = 2.0 * M * (2 * H + (cfg["L"] - 1) * H * H + H * 1) + 2.0 * cfg["B"] * cfg["z_dim"] * H
    return fwd, 3.0 * fwd, 2.0 * M * H * H


def host_cores():
    """Cores this process may actually use: affinity mask, cgroup quota, and -- when neither restricts a
    big host -- the 16-core CPU share a 1-GPU box of this pool gets (SVAE_CPU_THREADS overrides)."""
    if os.environ.get("SVAE_CPU_THREADS"):
        return max(1, int(os.environ["SVAE_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def cpu_baseline(cfg, seconds):
    """Time the torch-CPU restatement of the reference step on the host cores."""
    from oracle import torch_cpu_step as T
    cores = host_cores()
    torch.set_num_threads(cores)
    p_net, q_net = build_nets(cfg)
    rs = np.random.RandomState(123)
    N = cfg["n"] * cfg["m"]
    tr = T.CpuTrainer({k: v.detach().numpy() for k, v in p_net.state_dict().items()},
                      {k: v.detach().numpy() for k, v in q_net.state_dict().items()}, coord_grid(cfg["n"], cfg["m"]),
                      lr=cfg["lr"], act="tanh", rotate=True, translate=True, dx_scale=cfg["dx_scale"],
                      theta_prior=cfg["theta_prior"])
    y = torch.from_numpy(synthetic_batch(rs, cfg["B"], N))
    r = torch.randn(cfg["B"], cfg["z_dim"] + 3)
    tr.step(y, r)  # warm-up
    t0 = time.perf_counter()
    steps = 0
    while steps < 3 or (time.perf_counter() - t0 < seconds and steps < 50):
        tr.step(y, r)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": cfg["B"] * steps / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d full steps (fwd+bwd+Adam) at batch %d after 1 warm-up, %.1f s, torch %s CPU, %d threads"
                      % (steps, cfg["B"], dt, torch.__version__, cores)}


def measured_traffic(kind):
    """HBM bytes per launch of the dominant GEMM kernel from the committed counter passes (profiles/r01_traffic.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 correction applied); None if the file is absent."""
    split = os.environ.get("SVAE_GEMM") == "fp16x3"
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic_fp16x3.json" if split else "r01_traffic.json")))["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    if split:
        want = {"dense_fwd": "svae::dense_split_kernel<4, 0", "dense_dgrad": "svae::dense_split_kernel<4, 2",
                "wgrad": "svae::split_wgrad_kernel"}[kind]
    else:
        want = {"dense_fwd": "svae::dense_kernel<4, false", "dense_dgrad": "svae::dense_kernel<4, true",
                "wgrad": "svae::wgrad_kernel"}[kind]
    for name, d in prof.items():
        if name.startswith(want):
            return d.get("hbm_bytes_corrected")
    return None


def secondary_mode(args):
    """The same workload in a child process with --gemm fp16x3 (the mode is fixed per process)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gemm", "fp16x3", "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--no-cpu-baseline", "--no-secondary"]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:  # the headline must not depend on the secondary measurement
        return {"error": "%s: %s" % (type(e).__name__, e)}
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
            "roofline": d["roofline"],
            "parity": "passes the fp32 path's parity tests (tests/test_gpu_split.py); GEMM error vs fp64 equals an fp32 "
                      "GEMM's (tools/split_numerics.py)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--graph", action="store_true", help="replay the step from a HIP graph (single GPU; implies --no-profile)")
    ap.add_argument("--gemm", choices=["fp32", "fp16x3"], default="fp32",
                    help="hidden-layer GEMM path: fp32 MFMA (headline) or the fp32-accurate split-operand f16 MFMA path")
    ap.add_argument("--no-secondary", action="store_true", help="do not also measure the fp16x3 mode in a child process")
    args = ap.parse_args()
    if args.gemm == "fp16x3":      # read once by the library when it plans its first call
        os.environ["SVAE_GEMM"] = "fp16x3"
    else:
        os.environ.pop("SVAE_GEMM", None)
    split = args.gemm == "fp16x3"

    from spatial_vae_amd import _lib, dp
    from spatial_vae_amd import elbo as E
    _lib.set_gemm_mode(args.gemm)  # explicit (svae_gemm_mode_set); the environment variable covers child processes

    rank, world, local = dp.init_process_group(device_is_gpu=True)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cfg = dict(CFG)
    N = cfg["n"] * cfg["m"]

    p_net, q_net = build_nets(cfg)
    p_net.to(dev)
    q_net.to(dev)
    step = dp.TrainStep(p_net, q_net, E.eval_minibatch_mnist, lr=cfg["lr"], fused_adam=True if args.graph else None,
                        rotate=True, translate=True, dx_scale=cfg["dx_scale"], theta_prior=cfg["theta_prior"])
    x = torch.from_numpy(coord_grid(cfg["n"], cfg["m"])).to(dev)
    rs = np.random.RandomState(1000 + rank)
    pool = [torch.from_numpy(synthetic_batch(rs, cfg["B"], N)).to(dev) for _ in range(4)]

    def run(k):
        for i in range(k):  # each rank holds 1/world of the global minibatch: its mean gradient enters the all-reduce with that weight
            step(x, pool[i % len(pool)], weight=1.0 / world)

    if args.graph:
        if world > 1:
            raise SystemExit("--graph is single-GPU")
        args.no_profile = True
        step.capture(x, pool[0])
    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    profile = not args.no_profile
    if profile:
        _lib.profile_enable(1)  # HIP events around the three GEMM kernels only: ~6 event pairs per step
        _lib.profile_read()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = _lib.profile_read() if profile else {}
    _lib.profile_enable(0)
    breakdown = {}
    if profile:  # untimed extra steps with every kernel bracketed, for the per-kernel breakdown only
        _lib.profile_enable(2)
        run(min(args.steps, 10))
        torch.cuda.synchronize()
        n = min(args.steps, 10)
        breakdown = {k: round(v[0] / n, 4) for k, v in sorted(_lib.profile_read().items())}
        _lib.profile_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        f_fwd, f_step, f_gemm = decoder_flops(cfg)
        ms = 1e3 * elapsed / args.steps
        gemm = {k: prof[k] for k in ("dense_fwd", "dense_dgrad", "wgrad") if k in prof}
        roofline = None
        if gemm:
            dom = max(gemm, key=lambda k: gemm[k][0])
            avg_ms = gemm[dom][0] / gemm[dom][1]
            alg = f_gemm / (avg_ms * 1e-3) / 1e12                       # algorithmic TFLOP/s of that launch
            ach, peak = (3.0 * alg, PEAK_F16_MFMA_TFLOPS) if split else (alg, PEAK_FP32_MFMA_TFLOPS)
            roofline = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": measured_traffic(dom),
                        "avg_launch_ms": round(avg_ms, 4), "flops_per_launch": f_gemm * (3 if split else 1),
                        "gemm_kernels_avg_ms": {k: round(v[0] / v[1], 4) for k, v in sorted(gemm.items())},
                        "kernels_ms_per_step": breakdown}
            if split:
                roofline["note"] = ("executed f16 FLOPs (3 per algorithmic FLOP: hi*hi + hi*lo + lo*hi) against the f16 "
                                    "dense peak; algorithmic rate %.1f TFLOP/s = %.2f of the 157.3 TF fp32-MFMA peak"
                                    % (alg, alg / PEAK_FP32_MFMA_TFLOPS))
        out = {"metric": "images/sec (ELBO fwd+bwd+step), MNIST 28x28 batch=256, 1/2/4/8 GPU",
               "value": round(cfg["B"] * world * args.steps / elapsed, 1), "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None,
               "dtype": "f32 (fp16x3 split-operand MFMA, fp32 accumulate)" if split else "f32", "data": "synthetic",
               "config": {"workload": cfg["name"], "global_batch": cfg["B"] * world, "per_gpu_batch": cfg["B"],
                          "pixels": N, "parallelism": "dp%d" % world,
                          "decoder_step_gflop_per_gpu": round(f_step / 1e9, 1),
                          "decoder_mfma_frac_of_step": round(f_step / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not split and not args.no_secondary:
            out["fp16x3_mode"] = secondary_mode(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
