"""Data-parallel training step with the real HIP decoder: two ranks SHARING cuda:0 (SVAE_SHARE_GPU=1, gloo for the
collectives -- a 1-GPU box has no second device for RCCL) against a one-rank run of the same global minibatches.
Covers everything of SURVEY.md section 8e except the RCCL transport itself: parameter broadcast from rank 0, the shared
seed, contiguous ragged shards (5+3), a shard with NO rows (a 1-row global batch), the -local/global backward seed,
the decoder bucket launched from inside backward on a side stream, and the metrics riding in the second bucket."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import contextlib, io, math, os, sys
sys.path.insert(0, os.environ["SVAE_ROOT"])
import numpy as np, torch, torch.nn as nn, torch.distributed as dist
import spatial_vae.models as models
from spatial_vae_amd import dp, elbo as E, cli, ops

out_path = os.environ["SVAE_DP_REF"]
lowrank = os.environ.get("SVAE_TEST_LOWRANK") == "1"
if lowrank:      # treat every encoder layer above 1000 weights as "large": 32 x 144 and 32 x 32 then take ops.sink_linear
    ops.ENC_LINEAR_MAX_WEIGHT = 1000
rank, world, local = dp.init_process_group(device_is_gpu=True)
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
n = m = 12
torch.manual_seed(100 + rank)                       # every rank initialises differently; rank 0's weights must win
with contextlib.redirect_stdout(io.StringIO()):
    p_net = models.SpatialGenerator(2, 64, num_layers=2, activation=nn.Tanh).to(dev)
    q_net = models.InferenceNetwork(n * m, 5, 32, num_layers=2, activation=nn.Tanh).to(dev)
bucketed = os.environ.get("SVAE_TEST_BUCKETED") == "1"
step = dp.TrainStep(p_net, q_net, E.eval_minibatch_mnist, lr=1e-2, bucketed=bucketed if world > 1 else None, rotate=True,
                    translate=True, dx_scale=0.1, theta_prior=math.pi / 4)
assert step._bucketed == (bucketed and world > 1) and step.aliased()      # small encoder: one bucket unless asked otherwise
finished = [0]
if lowrank and world > 1:
    # the two "large" layers exchange the factors of their weight gradients; their ranges are not all-reduced
    assert step._lowrank is not None and step._lowrank.keys == {"layers.0", "layers.2"}
    skipped = step.grads.buffer.numel() - sum(hi - lo for lo, hi in step.allreduce_segments())
    assert skipped >= 32 * 144 + 32 + 32 * 32 + 32, skipped
    orig_finish = step._lowrank.finish
    def counting_finish():
        finished[0] += 1
        return orig_finish()
    step._lowrank.finish = counting_finish
else:
    assert step._lowrank is None
seed = dp.shared_seed(dev)
x = cli.coord_grid(n, m).to(dev)
rs = np.random.RandomState(7)
sizes = [8, 8, 1, 6]                                # 4+4, then 5+3 via an uneven split below, 1+0 (EMPTY shard), 3+3
if lowrank:                                         # the factor exchange pads shards to ceil(global / world) rows: it is built
    sizes = [8, 7, 1, 6]                            # for dp.shard_bounds' near-equal slices -- 4+4, 4+3 (ragged), 1+0, 3+3
ys = [torch.from_numpy(rs.uniform(size=(b, n * m)).astype(np.float32)).to(dev) for b in sizes]
rs_ = [torch.from_numpy(rs.normal(size=(b, 5)).astype(np.float32)).to(dev) for b in sizes]

def bounds(i, b):
    if world == 2 and i == 1 and not lowrank:       # a deliberately ragged 5 + 3 split of the second batch
        return (0, 5) if rank == 0 else (5, 8)
    return dp.shard_bounds(b, rank, world)

metrics = []
for i, (y, r) in enumerate(zip(ys, rs_)):
    lo, hi = bounds(i, y.size(0))
    step(x, y[lo:hi], weight=(hi - lo) / y.size(0), global_batch=y.size(0), noise=r[lo:hi])
    metrics.append(step.metrics.clone())
torch.cuda.synchronize()
assert finished[0] == (len(sizes) if (lowrank and world > 1) else 0)       # the low-rank path really ran, every step
flat = step.grads.flat_param.detach().cpu()
met = torch.stack(metrics).cpu()
assert step.aliased()
if world == 1:
    torch.save({"flat": flat, "metrics": met}, out_path)
    print("reference written", float(met[0, 0]))
else:
    ref = torch.load(out_path, weights_only=True)
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "replicas diverged"
    perr = (flat - ref["flat"]).abs().max().item() / ref["flat"].abs().max().item()
    merr = ((met - ref["metrics"]).abs().max(1).values / ref["metrics"].abs().max(1).values).max().item()
    print("rank", rank, "seed", seed, "param err %.3e metric err %.3e" % (perr, merr))
    assert perr < 2e-6 and merr < 2e-6, (perr, merr)
    dist.destroy_process_group()
'''


@pytest.mark.parametrize("bucketed,lowrank", [(False, False), (True, False), (True, True), (False, True)],
                         ids=["one_bucket", "two_buckets", "two_buckets_lowrank", "one_bucket_lowrank"])
def test_two_ranks_on_one_gpu_match_the_single_rank_run(tmp_path, bucketed, lowrank):
    """Both collective schemes: ONE all-reduce after backward() (the default for small encoders) and the two-bucket form whose
    first all-reduce is launched from inside backward() on a side stream (the default for the galaxy encoder) -- and each of
    them with the large-layer path of the galaxy configuration: the encoder layers above the size limit all-gather the
    factors (x, dy) of their weight gradients and form the global dW locally instead of all-reducing it (dp.LowRankExchange;
    the limit is lowered in the worker so that a small encoder takes it), asserted taken on every step including the one
    where rank 1 has no rows."""
    script = tmp_path / "dp_gpu_worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, SVAE_ROOT=ROOT, SVAE_DP_REF=str(tmp_path / "ref.pt"), PYTHONPATH=ROOT,
               SVAE_TEST_BUCKETED="1" if bucketed else "0", SVAE_TEST_LOWRANK="1" if lowrank else "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SVAE_SHARE_GPU", "SVAE_DP_BUCKETS", "SVAE_DP_LOWRANK"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout[-1500:] + one.stderr[-3000:]
    env["SVAE_SHARE_GPU"] = "1"
    code = ("import sys; sys.path.insert(0, %r); from spatial_vae_amd import dp; "
            "sys.exit(dp.launch_ranks(2, [%r]))" % (ROOT, str(script)))
    two = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stdout[-1500:] + two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("rank")]
    assert len(lines) == 2 and len({l.split()[3] for l in lines}) == 1, lines


def test_checkpointing_keeps_the_modules_inside_the_flat_buffers(tmp_path):
    """cli.save_models must not move the live modules (their parameters are views of TrainStep's flat buffers; a
    .cpu()/.to() round trip re-allocates them and training silently freezes)."""
    import contextlib
    import io
    import math
    import numpy as np
    import torch
    import torch.nn as nn
    import spatial_vae.models as models
    from spatial_vae_amd import cli, dp, elbo as E
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        p_net = models.SpatialGenerator(2, 32, num_layers=2, activation=nn.Tanh).to(dev)
        q_net = models.InferenceNetwork(100, 5, 16, num_layers=1, activation=nn.Tanh).to(dev)
    step = dp.TrainStep(p_net, q_net, E.eval_minibatch_mnist, lr=1e-2, rotate=True, translate=True, dx_scale=0.1,
                        theta_prior=math.pi / 4)
    x = cli.coord_grid(10, 10).to(dev)
    y = torch.from_numpy(np.random.RandomState(0).uniform(size=(16, 100)).astype(np.float32)).to(dev)
    step(x, y)
    cli.save_models(str(tmp_path / "ck"), "1", p_net, q_net, dev)
    assert step.aliased() and p_net.training is not None
    assert "_grad_sinks" in p_net.__dict__ and "_grad_sinks" in q_net.__dict__
    saved = torch.load(tmp_path / "ck_generator_epoch1.sav", weights_only=False)           # a file this test just wrote
    assert not hasattr(saved, "_grad_sinks") and not saved.training
    assert all(not p.is_cuda for p in saved.parameters())
    before = {k: v.detach().cpu().clone() for k, v in p_net.state_dict().items()}
    assert all(torch.equal(before[k], v) for k, v in saved.state_dict().items())
    step(x, y)
    after = p_net.state_dict()
    assert any(not torch.equal(before[k], after[k].cpu()) for k in before), "training froze after the checkpoint"
    e0 = float(step.metrics[0])
    for _ in range(20):
        step(x, y)
    assert float(step.metrics[0]) > e0, "the ELBO no longer improves after a checkpoint"


def test_train_particles_saving_every_epoch_keeps_training(tmp_path):
    """The reference defaults (--save-interval 10, --num-epochs 100) checkpoint in mid-run; with --save-interval 1 every
    epoch's checkpoint must differ from the previous one."""
    import torch
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train_particles.py"), "x", "y", "--synthetic", "64", "--num-epochs", "3",
                          "--minibatch-size", "32", "--p-hidden-dim", "32", "--q-hidden-dim", "32", "--save-prefix", "p",
                          "--save-interval", "1", "--progress-every", "0", "-l", "1e-3"], cwd=str(tmp_path), env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    sd = [torch.load(tmp_path / ("p_generator_epoch%d.sav" % e), weights_only=False).state_dict() for e in (1, 2, 3)]
    for a, b in ((sd[0], sd[1]), (sd[1], sd[2])):
        assert any(not torch.equal(a[k], b[k]) for k in a)
    rows = [l.split("\t") for l in out.stdout.splitlines() if "\ttrain\t" in l]
    assert float(rows[2][2]) > float(rows[0][2])                      # the training ELBO keeps improving


def test_command_line_under_two_ranks_prints_the_single_rank_table(tmp_path):
    """train_mnist.py end to end, one rank against two ranks sharing cuda:0, same --seed: the loop shards every GLOBAL
    minibatch (ragged last one: 200 images in batches of 64 -> 64, 64, 64, 8), draws one noise tensor per global batch, and
    all-reduces the metrics inside the gradient buckets -- so both runs print the same table (up to fp32 summation order)."""
    import numpy as np
    args = ["--synthetic", "200", "--num_epochs", "2", "--minibatch_size", "64", "--p_hidden_dim", "64", "--q_hidden_dim", "32",
            "--progress_every", "0", "--save_interval", "100", "--seed", "11", "-l", "1e-3"]
    script = os.path.join(ROOT, "train_mnist.py")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SVAE_SHARE_GPU"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, script] + args + ["--save_prefix", "one"], cwd=str(tmp_path), env=env, capture_output=True,
                         text=True, timeout=600)
    assert one.returncode == 0, one.stdout[-1500:] + one.stderr[-3000:]
    env["SVAE_SHARE_GPU"] = "1"
    code = ("import sys; sys.path.insert(0, %r); from spatial_vae_amd import dp; "
            "sys.exit(dp.launch_ranks(2, [%r] + %r))" % (ROOT, script, args + ["--save_prefix", "two"]))
    two = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stdout[-1500:] + two.stderr[-3000:]

    def table(out):
        rows = [l.split("\t") for l in out.splitlines() if "\t" in l]
        assert rows[0] == ["Epoch", "ELBO", "BCE loss", "KL"]
        return np.array([[float(v) for v in r] for r in rows[1:]])

    a, b = table(one.stdout), table(two.stdout)
    assert a.shape == b.shape == (4, 4)                              # rank 0 alone prints, once per line
    assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max(), (a, b)
    assert a[2, 1] > a[0, 1]                                         # and the model trained
    import torch
    p1 = torch.load(tmp_path / "outputs_one" / "trained" / "one_generator_epoch2.sav", weights_only=False).state_dict()
    p2 = torch.load(tmp_path / "outputs_two" / "trained" / "two_generator_epoch2.sav", weights_only=False).state_dict()
    for k in p1:
        assert (p1[k] - p2[k]).abs().max().item() <= 1e-5 * max(p1[k].abs().max().item(), 1e-3), k


def test_unseeded_galaxy_run_under_two_ranks_holds_one_dataset(tmp_path):
    """train_galaxy.py shuffles the training images with np.random inside build() (train_galaxy.py:372) and draws augmentation
    angles from it; without --seed every process would shuffle differently and the ranks would slice DIFFERENT images out
    of 'the same' global minibatch.  The loop seeds np.random from rank 0's seed before build() and checks the resident
    dataset across ranks; the run must pass that check, train, and print one table."""
    import numpy as np
    args = ["x", "y", "--synthetic", "96", "--num_epochs", "2", "--minibatch_size", "32", "--p_hidden_dim", "32", "--q_hidden_dim",
            "32", "--augment_rotation", "--save_prefix", "g2", "--progress_every", "0", "--save_interval", "100", "-l", "1e-3"]
    script = os.path.join(ROOT, "train_galaxy.py")
    env = dict(os.environ, PYTHONPATH=ROOT, SVAE_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    code = ("import sys; sys.path.insert(0, %r); from spatial_vae_amd import dp; "
            "sys.exit(dp.launch_ranks(2, [%r] + %r))" % (ROOT, script, args))
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    rows = [l.split("\t") for l in out.stdout.splitlines() if "\t" in l]
    assert rows[0] == ["Epoch", "ELBO", "BCE loss", "KL"] and len(rows) == 5
    vals = np.array([[float(v) for v in r] for r in rows[1:]])
    assert np.isfinite(vals).all() and vals[2, 1] > vals[0, 1]


_SOLO_WORKER = r'''
import contextlib, io, math, os, sys
sys.path.insert(0, os.environ["SVAE_ROOT"])
import numpy as np, torch, torch.nn as nn, torch.distributed as dist
import spatial_vae.models as models
from spatial_vae_amd import dp, elbo as E, cli
rank, world, local = dp.init_process_group(device_is_gpu=True)
dev = torch.device("cuda", local)
torch.manual_seed(5)
with contextlib.redirect_stdout(io.StringIO()):
    p_net = models.SpatialGenerator(2, 64, num_layers=2, activation=nn.Tanh).to(dev)
    q_net = models.InferenceNetwork(144, 5, 32, num_layers=2, activation=nn.Tanh).to(dev)
step = dp.TrainStep(p_net, q_net, E.eval_minibatch_mnist, lr=1e-2, rotate=True, translate=True, dx_scale=0.1,
                    theta_prior=math.pi / 4)
x = cli.coord_grid(12, 12).to(dev)
rs = np.random.RandomState(3)
mets = []
for i in range(4):
    y = torch.from_numpy(rs.uniform(size=(8, 144)).astype(np.float32)).to(dev)
    r = torch.from_numpy(rs.normal(size=(8, 5)).astype(np.float32)).to(dev)
    step(x, y, noise=r)
    mets.append(step.metrics.clone())
torch.cuda.synchronize()
print("backend", dist.get_backend() if dist.is_initialized() else "none", "collectives", dp.collectives_on())
torch.save({"flat": step.grads.flat_param.detach().cpu(), "metrics": torch.stack(mets).cpu()}, os.environ["SVAE_OUT"])
if dist.is_initialized():
    dist.destroy_process_group()
'''


def test_one_rank_rccl_group_executes_the_collectives_and_changes_nothing(tmp_path):
    """SVAE_DP_SOLO=1 runs the data-parallel step over a ONE-rank process group with backend "nccl" (= RCCL): the parameter
    broadcast, the decoder bucket's all-reduce launched from inside backward() on the side stream, the second bucket with the
    metric tail on the compute stream and the stream-level waits all execute on the real transport -- which a 1-GPU box
    cannot otherwise reach -- and the trained parameters and metrics must be bit-identical to the plain run."""
    import torch
    script = tmp_path / "solo_worker.py"
    script.write_text(_SOLO_WORKER)
    outs = {}
    for mode in ("plain", "solo", "solo2"):
        env = dict(os.environ, SVAE_ROOT=ROOT, PYTHONPATH=ROOT, SVAE_OUT=str(tmp_path / (mode + ".pt")))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SVAE_SHARE_GPU", "SVAE_DP_SOLO", "SVAE_DP_BUCKETS"):
            env.pop(k, None)
        if mode != "plain":
            env["SVAE_DP_SOLO"] = "1"
            env["SVAE_DP_BUCKETS"] = "2" if mode == "solo2" else "1"
        res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
        outs[mode] = (res.stdout, torch.load(tmp_path / (mode + ".pt"), weights_only=True))
    assert "backend none collectives False" in outs["plain"][0]
    for mode in ("solo", "solo2"):      # one all-reduce after backward / two buckets with the side-stream launch
        assert "backend nccl collectives True" in outs[mode][0]
        a, b = outs["plain"][1], outs[mode][1]
        assert torch.equal(a["flat"], b["flat"]) and torch.equal(a["metrics"], b["metrics"]), mode


def test_epoch_log_of_the_one_rank_rccl_run_equals_the_plain_run(tmp_path):
    """The per-epoch TRAINING means are collected lazily from step.metrics; in every data-parallel mode -- including the one-rank
    RCCL rehearsal (world == 1!) -- that is the shared metric tail of the gradient buffer, overwritten by each step, so the
    loop must keep a copy per step (r02 kept references whenever world == 1: the logged means were the last step's values
    repeated).  Same --seed, SVAE_DP_SOLO=1 against the plain run: the printed tables must agree line for line."""
    import numpy as np
    args = ["--synthetic", "200", "--num_epochs", "2", "--minibatch_size", "64", "--p_hidden_dim", "64", "--q_hidden_dim", "32",
            "--progress_every", "0", "--save_interval", "100", "--seed", "21", "-l", "1e-3"]
    script = os.path.join(ROOT, "train_mnist.py")
    tables = {}
    for mode in ("plain", "solo"):
        env = dict(os.environ, PYTHONPATH=ROOT)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SVAE_SHARE_GPU", "SVAE_DP_SOLO"):
            env.pop(k, None)
        if mode == "solo":
            env["SVAE_DP_SOLO"] = "1"
        out = subprocess.run([sys.executable, script] + args + ["--save_prefix", mode], cwd=str(tmp_path), env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        rows = [l.split("\t") for l in out.stdout.splitlines() if "\t" in l]
        tables[mode] = np.array([[float(v) for v in r] for r in rows[1:]])
    a, b = tables["plain"], tables["solo"]
    assert a.shape == b.shape == (4, 4)
    assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max(), (a, b)
    # and the four minibatches of an epoch do differ, so a repeated last value could not have passed
    assert abs(a[0, 1] - a[2, 1]) > 1e-3 * abs(a[0, 1])
