#!/usr/bin/env python3
"""Benchmark of the spatial-VAE ELBO training step on MI355X (BASELINE.json metric).

    python bench.py [--config 2] [--gpus N] [--scaling weak|strong] [--steps K] [--warmup W]

A "step" = forward + backward + Adam step of the ELBO on one minibatch (train_mnist.py:143-150 of the reference and its
siblings in train_galaxy.py / train_particles.py).  --config picks the BASELINE.json configuration (default 2, the one
the metric is quoted on: rotated+translated MNIST shape 28x28, z=2, p/q hidden 500 x 2, tanh, batch 256); 1, 3, 4, 5 are
the other BASELINE configs at their full sizes.  Inputs are synthetic (there are no datasets here) and resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.

--gpus N > 1: one process per GPU.  Launched plainly (`python bench.py --gpus N`) this process starts the N ranks itself
-- fresh interpreters, started BEFORE anything here touches the GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set --
and exits with their status; launched under torch.distributed.run it is one of the ranks.  --scaling weak (default): the
config's batch PER GPU (global batch = N x batch); --scaling strong: the config's batch is the GLOBAL minibatch, sharded
contiguously over the ranks.  Either way each rank's gradient enters the all-reduce weighted by
local/global rows (spatial_vae_amd/dp.py: one all-reduce per step; two buckets, the decoder's on a side stream, for heavy encoders).

The line also carries
  roofline     : the dominant kernel (an fp32-MFMA GEMM of a decoder hidden layer) -- algorithmic FLOPs per launch / its
                 average launch duration, the durations taken live from HIP events recorded on the launch stream during
                 the timed steps (svae_profile_* in include/svae.h); peak = 157.3 TFLOP/s fp32 MFMA.  `traffic` is the
                 HBM bytes per launch from the rocprofv3 --pmc passes named in `traffic_profile` (tools/profile_round.sh
                 writes them; they cannot be collected inside a timed run), or null.
  cpu_baseline : the torch-CPU restatement of the reference's step (oracle/torch_cpu_step.py) timed on this box's host
                 cores on a bounded sample of the same workload (rank 0, N=1 only): 2 warm-up steps, median of >= 5,
                 all usable cores, plus a 1-thread figure.
  allreduce    : (N > 1) bytes per step and the time the compute stream spends waiting for the collectives.  SVAE_DP_SOLO=1
                 at N=1 issues the same collectives over a one-rank RCCL group: the per-step cost of the calls themselves.
  fp16x3_mode  : (config 2, N=1 only) the same workload measured in a child process with SVAE_GEMM=fp16x3 -- the hidden-layer
                 GEMMs on the f16 matrix pipe with split (hi + lo) operands, fp32-accurate.  Reported beside the headline,
                 which stays the fp32-MFMA path: `value`, `dtype` and `roofline` at the top level are that path's.
--gemm fp16x3 makes that mode the measured one (its roofline then counts EXECUTED f16 FLOPs, 3 per algorithmic one,
against the 2.5 PF f16 peak).
"""
import argparse
import contextlib
import io
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # same table, "Peak BF16/FP16 MFMA ~2.5 PF dense"
METRIC = "images/sec (ELBO fwd+bwd+step), MNIST 28x28 batch=256, 1/2/4/8 GPU"

# BASELINE.json configs (SURVEY.md appendix A.7 maps them to the reference's command lines).  cpu_B = batch of the bounded
# CPU sample (per-image throughput is reported; config 4 at B=128 would need > 60 GB unfused on the host).
CONFIGS = {
    1: dict(name="cfg1 mnist-rotated 28x28 z=2 H=500x2 tanh B=64, rotate only (BASELINE configs[0])", script="mnist", n=28,
            B=64, z_dim=2, H=500, L=2, C=1, q_hidden=500, q_layers=2, rotate=True, translate=False, theta_prior=math.pi / 4,
            cpu_B=64),
    2: dict(name="mnist-rotated-translated 28x28 z=2 H=500x2 tanh B=256 (BASELINE configs[1])", script="mnist", n=28, B=256,
            z_dim=2, H=500, L=2, C=1, q_hidden=500, q_layers=2, rotate=True, translate=True, theta_prior=math.pi / 4,
            cpu_B=256),
    3: dict(name="cfg3 5HDB-like particles 40x40 z=2 H=500x2 --fit-noise B=512 (BASELINE configs[2])", script="particles",
            n=40, B=512, z_dim=2, H=500, L=2, C=2, q_hidden=500, q_layers=2, rotate=True, translate=True,
            theta_prior=math.pi, cpu_B=128),
    4: dict(name="cfg4 galaxy-zoo 128x128x3 z=20 H=1024x3 q=5000x2 B=128 (BASELINE configs[3])", script="galaxy", n=128,
            B=128, z_dim=20, H=1024, L=3, C=3, q_hidden=5000, q_layers=2, rotate=True, translate=True, theta_prior=math.pi,
            cpu_B=8),
    5: dict(name="cfg5 CODH/ACS-like particles 40x40 + CTF 39x39, z=8 H=500x2 B=256 (BASELINE configs[4])",
            script="particles", n=40, B=256, z_dim=8, H=500, L=2, C=1, q_hidden=500, q_layers=2, rotate=True, translate=True,
            theta_prior=math.pi, ctf=True, cpu_B=128),
}
DX_SCALE = 0.1
LR = 1e-4


def inf_dim(cfg):
    return cfg["z_dim"] + (1 if cfg["rotate"] else 0) + (2 if cfg["translate"] else 0)


def coord_grid(n, m):
    import numpy as np
    x0, x1 = np.meshgrid(np.linspace(-1, 1, m), np.linspace(1, -1, n))
    return np.stack([x0.ravel(), x1.ravel()], 1).astype(np.float32)


def synthetic_targets(cfg, rs, B):
    """MNIST-like sparse uniform targets, uint8-quantised /255; particles ~ N(0,1) (the scripts
    standardise them); galaxy ~ U[0,1] with 3 channels."""
    import numpy as np
    N = cfg["n"] * cfg["n"]
    if cfg["script"] == "mnist":
        u = rs.uniform(size=(B, N))
        keep = rs.uniform(size=(B, N)) > 0.8
        return (np.floor(u * keep * 255.0) / 255.0).astype(np.float32)
    if cfg["script"] == "particles":
        return rs.normal(size=(B, N)).astype(np.float32)
    return rs.uniform(size=(B, N, cfg["C"])).astype(np.float32)


def ctf_table(rs, B):
    """defocus U[1,3] um, cs 2.7 mm, 300 kV, apix 1.7 A, bfactor 100, ampcont 10 %, dfang U[0,180)."""
    import numpy as np
    return np.stack([rs.uniform(1, 3, B), np.full(B, 2.7), np.full(B, 300.0), np.full(B, 1.7), np.full(B, 100.0),
                     np.full(B, 10.0), np.zeros(B), rs.uniform(0, 180, B)], 1)


def build_nets(cfg):
    import torch
    import torch.nn as nn
    import spatial_vae.models as models
    torch.manual_seed(0)  # p_net before q_net, default nn.Linear init (train_mnist.py:370-375)
    n_in = cfg["n"] * cfg["n"] * (cfg["C"] if cfg["script"] == "galaxy" else 1)
    with contextlib.redirect_stdout(io.StringIO()):
        p_net = models.SpatialGenerator(cfg["z_dim"], cfg["H"], n_out=cfg["C"], num_layers=cfg["L"], activation=nn.Tanh)
        q_net = models.InferenceNetwork(n_in, inf_dim(cfg), cfg["q_hidden"], num_layers=cfg["q_layers"], activation=nn.Tanh)
    return p_net, q_net


def decoder_flops(cfg, B):
    """F_fwd = 2 M (D_in H + (L-1) H^2 + H C) + 2 B Zd H, F_step = 3 F_fwd; one hidden-layer GEMM
    launch = 2 M H^2."""
    M = B * cfg["n"] * cfg["n"]
    H = cfg["H"]
    fwd = 2.0 * M * (2 * H + (cfg["L"] - 1) * H * H + H * cfg["C"]) + 2.0 * B * cfg["z_dim"] * H
    return fwd, 3.0 * fwd, 2.0 * M * H * H


def host_cores():
    """Cores this process may actually use: the affinity mask, cut by a cgroup CPU quota if there is one
    (SVAE_CPU_THREADS overrides)."""
    if os.environ.get("SVAE_CPU_THREADS"):
        return max(1, int(os.environ["SVAE_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_trainer(cfg, B, ctf_filters):
    import numpy as np
    import torch
    from oracle import torch_cpu_step as T
    p_net, q_net = build_nets(cfg)
    rs = np.random.RandomState(123)
    kw = dict(lr=LR, script=cfg["script"], act="tanh", rotate=cfg["rotate"], translate=cfg["translate"], dx_scale=DX_SCALE,
              theta_prior=cfg["theta_prior"])
    tr = T.CpuTrainer({k: v.detach().numpy() for k, v in p_net.state_dict().items()},
                      {k: v.detach().numpy() for k, v in q_net.state_dict().items()}, coord_grid(cfg["n"], cfg["n"]), **kw)
    y = torch.from_numpy(synthetic_targets(cfg, rs, B))
    r = torch.randn(B, inf_dim(cfg))
    batch = {}
    if cfg.get("ctf"):
        batch["ctf"] = ctf_filters[:B].contiguous()
    return tr, y, r, batch


def _time_cpu(cfg, B, threads, warm, min_steps, seconds, ctf_filters):
    import torch
    torch.set_num_threads(threads)
    tr, y, r, batch = _cpu_trainer(cfg, B, ctf_filters)
    for _ in range(warm):
        tr.step(y, r, **batch)
    times = []
    t_start = time.perf_counter()
    while len(times) < min_steps or (time.perf_counter() - t_start < seconds and len(times) < 25):
        t0 = time.perf_counter()
        tr.step(y, r, **batch)
        times.append(time.perf_counter() - t0)
    return statistics.median(times), len(times)


def cpu_baseline(cfg, seconds, ctf_filters):
    """SURVEY.md section 8d protocol: the torch-CPU port on every usable host core, 2 warm-up steps, median of >= 5 timed
    steps, the same synthetic workload (at cfg['cpu_B'] images per step where the full batch would take minutes or too
    much memory: throughput is per image), and a 1-thread figure on a smaller sample."""
    import torch
    cores = host_cores()
    B = cfg["cpu_B"]
    # config 4: a step at batch 8 takes several seconds on 16 cores (0.6 GB of activations per image; at the config's own 128
    # the port would hold ~80 GB of autograd state), so fewer timed steps; its per-step fixed cost (Adam over the
    # 271 M-parameter encoder) is amortised over 8 images instead of 128, which still UNDERSTATES the CPU's per-image rate at
    # the real batch: the line says so (`note`) rather than letting a GPU/CPU ratio be read off it
    heavy = cfg["script"] == "galaxy"
    med, steps = _time_cpu(cfg, B, cores, 1 if heavy else 2, 3 if heavy else 5, seconds, ctf_filters)
    B1 = max(1, min(B, 32 if cfg["script"] != "galaxy" else 1))
    med1, steps1 = _time_cpu(cfg, B1, 1, 1, 3, seconds / 4, ctf_filters)
    torch.set_num_threads(cores)
    out_note = {}
    if B != cfg["B"]:
        out_note["note"] = ("timed at batch %d of the config's %d: per-step fixed costs (optimizer and weight traffic of the "
                            "encoder) are amortised over fewer images, so the per-image rate is pessimistic and a GPU/CPU "
                            "ratio from this line is not comparable with the other configs'" % (B, cfg["B"]))
    return {"value": round(B / med, 2), "unit": "images/s", "cores": cores, "kind": "port", **out_note,
            "sample": "median of %d full steps (fwd+bwd+Adam) at batch %d%s after %d warm-up(s), torch %s CPU, %d threads"
                      % (steps, B, "" if B == cfg["B"] else " (of the config's %d; per-image rate)" % cfg["B"],
                         1 if heavy else 2, torch.__version__, cores),
            "one_thread": {"value": round(B1 / med1, 2), "unit": "images/s",
                           "sample": "median of %d steps at batch %d after 1 warm-up, 1 thread" % (steps1, B1)}}


def traffic_for(kind, cfg_id, split):
    """HBM bytes per launch of `kind` from the newest committed counter passes for this config (profiles/rNN_traffic*.json,
    written by tools/profile_round.sh + tools/traffic_json.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs,
    gfx950 correction applied).  Returns (bytes | None, profile file | None): the value is NOT measured in this run, so
    the line names the profile it came from."""
    import glob
    suffix = "_cfg%d" % cfg_id + ("_fp16x3" if split else "")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic%s.json" % suffix)))
    if not files and cfg_id == 2:       # r01 / r02 named the headline config's file without a suffix
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic%s.json" % ("_fp16x3" if split else ""))))
    if not files:
        return None, None
    try:
        prof = json.load(open(files[-1]))["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    if split:
        want = {"dense_fwd": "svae::dense_split_kernel<4, 0", "dense_dgrad": "svae::dense_split_kernel<4, 2",
                "wgrad": "svae::split_wgrad_kernel"}[kind]
    else:   # the hidden-layer GEMMs run on dense4_kernel at the BASELINE sizes (dense_kernel for small launches)
        want = {"dense_fwd": ("svae::dense4_dual_kernel<false", "svae::dense4_kernel<2, false", "svae::dense4_kernel<1, false",
                              "svae::dense_kernel<4, false"),
                "dense_dgrad": ("svae::dense4_dual_kernel<true", "svae::dense4_kernel<2, true", "svae::dense4_kernel<1, true",
                                "svae::dense_kernel<4, true"),
                "wgrad": ("svae::wgrad2_kernel", "svae::wgrad_kernel")}[kind]
    if isinstance(want, str):
        want = (want,)
    for w in want:
        hits = [d for name, d in prof.items() if name.startswith(w)]
        if hits:    # several instantiations of one role (L = 3: plain and fused forms): the launch with the most bytes
            return max(h.get("hbm_bytes_corrected") or 0 for h in hits), os.path.relpath(files[-1], ROOT)
    return None, None


def secondary_mode(args):
    """The same workload in a child process with --gemm fp16x3 (the mode is fixed per process)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gemm", "fp16x3", "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--config", str(args.config), "--no-cpu-baseline", "--no-secondary", "--sustained", "0"]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:  # the headline must not depend on the secondary measurement
        return {"error": "%s: %s" % (type(e).__name__, e)}
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
            "roofline": d["roofline"],
            "parity": "passes the fp32 path's parity tests with the split kernels asserted dispatched (tests/test_gpu_split.py); "
                      "GEMM error vs fp64 equals an fp32 GEMM's (tools/split_numerics.py)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configuration (default 2 = headline)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = the config's batch per GPU, strong = the config's batch sharded over the GPUs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--graph", action="store_true", help="replay the step from a HIP graph (single GPU; implies --no-profile)")
    ap.add_argument("--gemm", choices=["fp32", "fp16x3"], default="fp32",
                    help="hidden-layer GEMM path: fp32 MFMA (headline) or the fp32-accurate split-operand f16 MFMA path")
    ap.add_argument("--no-secondary", action="store_true", help="do not also measure the fp16x3 mode in a child process")
    ap.add_argument("--sustained", type=float, default=10.0,
                    help="N=1: after the timed window keep stepping for at least this many seconds and report the rate and the "
                         "GEMM averages under sustained load (0 = skip)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks from here.  Nothing above or in dp's imports touches the GPU.
        from spatial_vae_amd import dp
        sys.exit(dp.launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    if args.gemm == "fp16x3":      # read once by the library when it plans its first call
        os.environ["SVAE_GEMM"] = "fp16x3"
    else:
        os.environ.pop("SVAE_GEMM", None)
    split = args.gemm == "fp16x3"

    import numpy as np
    import torch
    import torch.distributed as dist
    from spatial_vae_amd import _lib, dp, ops
    from spatial_vae_amd import elbo as E
    _lib.set_gemm_mode(args.gemm)  # explicit (svae_gemm_mode_set); the environment variable covers child processes

    # RCCL prints a version banner on stdout when a communicator is created; stdout carries exactly one JSON line, so file
    # descriptor 1 points at stderr until the process group (and the first collective: TrainStep's broadcast) exist
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    have = torch.cuda.device_count()        # counts devices without touching them
    if have < (1 if os.environ.get("SVAE_SHARE_GPU") == "1" else args.gpus):
        raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s); one process per GPU" % (args.gpus, have))
    rank, world, local = dp.init_process_group(device_is_gpu=True)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cfg = dict(CONFIGS[args.config])
    N = cfg["n"] * cfg["n"]
    strong = args.scaling == "strong" and world > 1
    global_B = cfg["B"] if strong or world == 1 else cfg["B"] * world
    lo, hi = dp.shard_bounds(global_B, rank, world)
    local_B = hi - lo
    weight = local_B / global_B

    p_net, q_net = build_nets(cfg)
    p_net.to(dev)
    q_net.to(dev)
    fn = {"mnist": E.eval_minibatch_mnist, "galaxy": E.eval_minibatch_galaxy, "particles": E.eval_minibatch_particles}[cfg["script"]]
    step = dp.TrainStep(p_net, q_net, fn, lr=LR, fused_adam=True if args.graph else None, rotate=cfg["rotate"],
                        translate=cfg["translate"], dx_scale=DX_SCALE, theta_prior=cfg["theta_prior"])
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    x = torch.from_numpy(coord_grid(cfg["n"], cfg["n"])).to(dev)
    rs = np.random.RandomState(1000 + rank)
    npool = 4 if args.config != 4 else 2
    pool = [torch.from_numpy(synthetic_targets(cfg, rs, local_B)).to(dev) for _ in range(npool)]
    ctf_filters = None
    if cfg.get("ctf"):
        ctf_filters = ops.ctf_filter(ctf_table(rs, max(local_B, cfg["cpu_B"])), cfg["n"] - 1, cfg["n"] - 1, device=dev).unsqueeze(1)

    def batch(i):
        y = pool[i % len(pool)]
        if cfg["script"] == "particles":
            return (y, None, ctf_filters[:local_B] if ctf_filters is not None else None)
        return (y,)

    comm_pairs = []

    def run(k, timed=False):
        for i in range(k):  # each rank holds local_B of the global minibatch: its gradient enters the all-reduce with that weight
            if timed and dp.collectives_on():
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                step.comm_events = pair
                comm_pairs.append(pair)
            step(x, *batch(i), weight=weight, global_batch=global_B)
        step.comm_events = None

    if args.graph:
        if world > 1:
            raise SystemExit("--graph is single-GPU")
        args.no_profile = True
        step.capture(x, *batch(0))
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
    run(args.steps, timed=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = _lib.profile_read() if profile else {}
    _lib.profile_enable(0)
    comm_ms = [a.elapsed_time(b) for a, b in comm_pairs]
    breakdown = {}
    if profile:  # untimed extra steps with every kernel bracketed, for the per-kernel breakdown only
        n = min(args.steps, 10)
        _lib.profile_enable(2)
        run(n)
        torch.cuda.synchronize()
        breakdown = {k: round(v[0] / n, 4) for k, v in sorted(_lib.profile_read().items())}
        _lib.profile_enable(0)
    sustained = None
    if world == 1 and args.sustained > 0 and not args.graph:
        # the timed window above is tens of milliseconds; this shows the rate the chip HOLDS: >= --sustained seconds of
        # back-to-back steps (synchronised every chunk so that the host clock is the device's), then one more chunk with the
        # GEMM events on
        chunk = max(10, int(0.25 / max(elapsed / args.steps, 1e-6)))
        n_sus = 0
        t_sus = time.perf_counter()
        while time.perf_counter() - t_sus < args.sustained:
            run(chunk)
            torch.cuda.synchronize()
            n_sus += chunk
        dt = time.perf_counter() - t_sus
        sustained = {"seconds": round(dt, 2), "steps": n_sus, "value": round(global_B * n_sus / dt, 1), "unit": "images/s",
                     "ms_per_step": round(1e3 * dt / n_sus, 4)}
        if profile:
            _lib.profile_enable(1)
            _lib.profile_read()
            run(chunk)
            torch.cuda.synchronize()
            sp = _lib.profile_read()
            _lib.profile_enable(0)
            sustained["gemm_kernels_avg_ms"] = {k: round(v[0] / v[1], 4) for k, v in sorted(sp.items())
                                                if k in ("dense_fwd", "dense_dgrad", "wgrad")}
    joined = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # proof, in the record itself, that `world` ranks on distinct devices took part in the collectives
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        ids = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([rank, dev.index], dtype=torch.int64, device=dev))
        joined = {"world": int(ones.item()), "ranks_devices": [[int(v[0]), int(v[1])] for v in ids]}

    if rank == 0:
        f_fwd, f_step, f_gemm = decoder_flops(cfg, local_B)
        ms = 1e3 * elapsed / args.steps
        gemm = {k: prof[k] for k in ("dense_fwd", "dense_dgrad", "wgrad") if k in prof}
        roofline = None
        if gemm:
            dom = max(gemm, key=lambda k: gemm[k][0])               # the kernel with the most time in the step
            avg_ms = gemm[dom][0] / gemm[dom][1]
            alg = f_gemm / (avg_ms * 1e-3) / 1e12                       # algorithmic TFLOP/s of that launch
            ach, peak = (3.0 * alg, PEAK_F16_MFMA_TFLOPS) if split else (alg, PEAK_FP32_MFMA_TFLOPS)
            traffic, traffic_profile = traffic_for(dom, args.config, split)
            roofline = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                        "traffic_profile": traffic_profile,
                        "avg_launch_ms": round(avg_ms, 4), "flops_per_launch": f_gemm * (3 if split else 1),
                        "gemm_kernels_avg_ms": {k: round(v[0] / v[1], 4) for k, v in sorted(gemm.items())},
                        "gemm_kernels_frac": {k: round(f_gemm * (3 if split else 1) / (v[0] / v[1] * 1e-3) / 1e12 / peak, 4)
                                              for k, v in sorted(gemm.items())},
                        "kernels_ms_per_step": breakdown}
            if split:
                roofline["note"] = ("executed f16 FLOPs (3 per algorithmic FLOP: hi*hi + hi*lo + lo*hi) against the f16 "
                                    "dense peak; algorithmic rate %.1f TFLOP/s = %.2f of the 157.3 TF fp32-MFMA peak"
                                    % (alg, alg / PEAK_FP32_MFMA_TFLOPS))
        out = {"metric": METRIC,
               "value": round(global_B * args.steps / elapsed, 1), "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
               "scaling": "strong" if strong else "weak", "vs_baseline": None,
               "dtype": "f32 (fp16x3 split-operand MFMA, fp32 accumulate)" if split else "f32", "data": "synthetic",
               "config": {"workload": cfg["name"], "baseline_config": args.config, "global_batch": global_B,
                          "per_gpu_batch": local_B, "pixels": N, "parallelism": "dp%d" % world,
                          "decoder_step_gflop_per_gpu": round(f_step / 1e9, 1),
                          "decoder_mfma_frac_of_step": round(f_step / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
               "roofline": roofline}
        if dp.collectives_on():     # N > 1, or the one-rank RCCL rehearsal (SVAE_DP_SOLO=1)
            segs = step.allreduce_segments()
            nbytes = 4 * sum(hi - lo for lo, hi in segs)
            out["allreduce"] = {**(joined or {"world": 1, "ranks_devices": [[0, dev.index]]}),
                                "bytes_per_step": nbytes, "buckets_bytes": [4 * (hi - lo) for lo, hi in segs],
                                "gradient_bytes": step.grads.buffer.numel() * 4,
                                "backend": dist.get_backend(),
                                "compute_stream_wait_ms_per_step": round(sum(comm_ms) / max(len(comm_ms), 1), 4)}
        if dp.collectives_on() and step._lowrank is not None:
            # large encoder layers: the factors (x, dy) of dW are all-gathered and dW formed locally (dp.LowRankExchange)
            out["allgather"] = {"bytes_per_step": step._lowrank.bytes_last, "layers": sorted(step._lowrank.keys),
                                "replaces_allreduce_bytes": out["allreduce"]["gradient_bytes"] - out["allreduce"]["bytes_per_step"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds, ctf_filters.cpu() if ctf_filters is not None else None)
        else:
            out["cpu_baseline"] = None
        if sustained is not None:
            if roofline and roofline["kernel"] in sustained.get("gemm_kernels_avg_ms", {}):
                ms_k = sustained["gemm_kernels_avg_ms"][roofline["kernel"]]
                sustained["roofline_frac"] = round(f_gemm * (3 if split else 1) / (ms_k * 1e-3) / 1e12 / roofline["peak"], 4)
            out["sustained"] = sustained
        if world == 1 and not split and not args.no_secondary and args.config == 2:
            out["fp16x3_mode"] = secondary_mode(args)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
