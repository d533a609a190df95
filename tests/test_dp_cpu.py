"""Data-parallel TrainStep on CPU: two gloo ranks against a single-process run.

The decoder has no CPU form, so these tests drive dp.TrainStep with a small pure-torch ELBO (TrainStep takes any
eval_minibatch callable and any modules): what is under test is the DP machinery -- the parameter broadcast at
construction, the -local/global backward seed, the two gradient buckets, the three metrics riding in the second
bucket, ragged and EMPTY shards -- not the decoder.  The same contract with the real HIP decoder is
tests/test_gpu_dp.py (two ranks sharing cuda:0)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["SVAE_ROOT"])
import torch, torch.nn as nn, torch.distributed as dist
from spatial_vae_amd import dp

def toy_elbo(x, y, p_net, q_net, noise=None):
    q = q_net(y)
    mu, logstd = q[:, :2], q[:, 2:]
    z = mu + logstd.exp() * noise
    y_hat = p_net(z)
    log_p = -((y_hat - y) ** 2).sum(1).mean()
    kl = (-logstd + 0.5 * logstd.exp() ** 2 + 0.5 * mu ** 2 - 0.5).sum(1).mean()
    return log_p - kl, log_p, kl

def nets(seed):
    torch.manual_seed(seed)
    return (nn.Sequential(nn.Linear(2, 8), nn.Tanh(), nn.Linear(8, 5)),
            nn.Sequential(nn.Linear(5, 8), nn.Tanh(), nn.Linear(8, 4)))

rank, world, _ = dp.init_process_group(device_is_gpu=False)
bucketed = os.environ.get("BUCKETED") == "1"
p_net, q_net = nets(100 + rank)                     # every rank starts from its OWN weights: the broadcast must fix that
step = dp.TrainStep(p_net, q_net, toy_elbo, lr=1e-2, bucketed=bucketed)
assert step.aliased()
seed = dp.shared_seed(torch.device("cpu"))
gen = torch.Generator().manual_seed(1234)
sizes = [8, 5, 1, 6]                                # 4+4, 3+2 (ragged), 1+0 (rank 1 has NO rows), 3+3
batches = [torch.randn(b, 5, generator=gen) for b in sizes]
noises = [torch.randn(b, 2, generator=gen) for b in sizes]

# single-process reference from rank 0's initial weights: plain modules, plain torch.optim.Adam, whole batches
rp, rq = nets(100)
opt = torch.optim.Adam(list(rp.parameters()) + list(rq.parameters()), lr=1e-2)
for y, r in zip(batches, noises):
    lo, hi = dp.shard_bounds(y.size(0), rank, world)
    step(None, y[lo:hi], weight=(hi - lo) / y.size(0), noise=r[lo:hi])
    got = step.metrics.clone()
    e, lp, kl = toy_elbo(None, y, rp, rq, noise=r)
    opt.zero_grad(); (-e).backward(); opt.step()
    want = torch.stack([e, lp, kl]).detach()
    assert (got - want).abs().max().item() <= 1e-6 * want.abs().max().item(), (got, want)

ref = torch.cat([p.detach().reshape(-1) for p in list(rp.parameters()) + list(rq.parameters())])
mine = step.grads.flat_param
packed = torch.cat([mine[o:o + p.numel()] for p, o in zip(step.grads.params, step.grads.offsets)])   # without the alignment gaps
err = (packed - ref).abs().max().item() / ref.abs().max().item()
both = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
assert all(torch.equal(both[0], b) for b in both[1:]), "replicas diverged"
assert step.aliased()
assert all(torch.equal(p.detach().reshape(-1), step.grads.flat_param[o:o + p.numel()]) for p, o in
           zip(step.grads.params, step.grads.offsets))
assert all(o % 64 == 0 for o in step.grads.offsets)
print("rank", rank, "seed", seed, "param err", err)
assert err < 1e-6, err
dist.destroy_process_group()
'''


def _run(tmp_path, bucketed, world=2):
    sys.path.insert(0, ROOT)
    from spatial_vae_amd import dp
    script = tmp_path / "dp_step_worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, SVAE_ROOT=ROOT, BUCKETED="1" if bucketed else "0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    code = ("import sys; sys.path.insert(0, %r); from spatial_vae_amd import dp; "
            "sys.exit(dp.launch_ranks(%d, [%r]))" % (ROOT, world, str(script)))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("rank")]
    assert len(lines) == world
    assert len({l.split()[3] for l in lines}) == 1, "ranks disagree on the shared seed: %r" % lines


def test_trainstep_world2_matches_single_process_one_bucket(tmp_path):
    _run(tmp_path, bucketed=False)


def test_trainstep_world2_matches_single_process_two_buckets(tmp_path):
    _run(tmp_path, bucketed=True)


def test_trainstep_world4_matches_single_process(tmp_path):
    """Four ranks (the driver's scaling run goes to eight): the global batches of 8, 5, 1 and 6 rows shard as 2+2+2+2, 2+1+1+1,
    1+0+0+0 (three ranks with NO rows join the collectives with zeros) and 2+2+1+1."""
    _run(tmp_path, bucketed=True, world=4)


def test_launch_ranks_reports_a_failing_rank(tmp_path):
    """A rank that dies takes the job down with its exit code instead of leaving its peers blocked."""
    script = tmp_path / "boom.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(7)\n"
                      "time.sleep(60)\n")
    code = ("import sys; sys.path.insert(0, %r); from spatial_vae_amd import dp; "
            "sys.exit(dp.launch_ranks(2, [%r]))" % (ROOT, str(script)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 7


def test_launch_ranks_has_a_wall_clock_limit(tmp_path):
    """A job whose ranks never finish (one stuck in a collective) is terminated at the limit and reported as 124."""
    import time
    script = tmp_path / "hang.py"
    script.write_text("import time\ntime.sleep(300)\n")
    code = ("import sys; sys.path.insert(0, %r); from spatial_vae_amd import dp; "
            "sys.exit(dp.launch_ranks(2, [%r], timeout=3))" % (ROOT, str(script)))
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 124 and time.time() - t0 < 60
    assert "terminating the job" in out.stderr


def test_a_rank_without_a_gpu_of_its_own_fails_with_a_clear_message(tmp_path):
    """init_process_group(device_is_gpu=True) checks torch.cuda.device_count() before joining the group: here there is no GPU at
    all, so rank 1 of 2 must exit with the message, not hang in a rendezvous."""
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.update(RANK='1', LOCAL_RANK='1', WORLD_SIZE='2', "
            "MASTER_ADDR='127.0.0.1', MASTER_PORT='29999'); os.environ.pop('SVAE_SHARE_GPU', None); "
            "from spatial_vae_amd import dp; dp.init_process_group(device_is_gpu=True)" % ROOT)
    import torch
    if torch.cuda.device_count() >= 2:
        return
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "needs GPU 1" in out.stderr, out.stderr[-2000:]


_LOWRANK_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["SVAE_ROOT"])
import torch, torch.distributed as dist
from spatial_vae_amd import dp
rank, world, _ = dp.init_process_group(device_is_gpu=False)
g = torch.Generator().manual_seed(7)
rows_of = [[3, 2], [1, 0], [4, 4]]                    # per step: rows on rank 0, rank 1 (ragged; rank 1 EMPTY in step 2)
w0, b0 = torch.full((6, 11), 9.0), torch.full((6,), 9.0)      # the sinks: views of the flat gradient buffer in TrainStep
w1, b1 = torch.full((4, 6), 9.0), torch.full((4,), 9.0)
ex = dp.LowRankExchange([("layers.0", w0, b0), ("layers.2", w1, b1)], torch.device("cpu"))
assert ex.has("layers.0") and not ex.has("layers.4")
for rows in rows_of:
    xs = [(torch.randn(r, 11, generator=g), torch.randn(r, 6, generator=g), torch.randn(r, 6, generator=g),
           torch.randn(r, 4, generator=g)) for r in rows]              # every rank draws every rank's factors: same stream
    mine = xs[rank]
    if rows[rank] > 0:
        ex.add("layers.0", mine[0], mine[1])
        ex.add("layers.2", mine[2], mine[3])
    cap = (sum(rows) + world - 1) // world
    cap = max(cap, max(rows))                          # the caller's shards are contiguous near-equal slices; here ragged on purpose
    ex.start(cap).wait()
    ex.finish()
    want_w0 = sum(dy.t() @ x for x, dy, _, _ in xs)
    want_w1 = sum(dy.t() @ x for _, _, x, dy in xs)
    assert torch.allclose(w0, want_w0, atol=1e-5), (w0 - want_w0).abs().max()
    assert torch.allclose(w1, want_w1, atol=1e-5)
    assert torch.allclose(b0, sum(dy.sum(0) for _, dy, _, _ in xs), atol=1e-5)
    assert torch.allclose(b1, sum(dy.sum(0) for _, _, _, dy in xs), atol=1e-5)
    both = [torch.empty_like(w0) for _ in range(world)]
    dist.all_gather(both, w0)
    assert torch.equal(both[0], both[1]), "replicas must be bit-equal: every rank multiplies the same gathered factors"
    assert ex.bytes_last == world * cap * (11 + 6 + 6 + 4) * 4
print("rank", rank, "lowrank ok")
dist.destroy_process_group()
'''


def test_low_rank_exchange_forms_the_global_weight_gradient(tmp_path):
    """dp.LowRankExchange (the galaxy encoder's 983 MB first-layer gradient never crosses the wire): two gloo ranks with ragged
    and EMPTY shards all-gather the factors of two layers' weight gradients and must each end up with the exact global
    dW = sum_r dy_r^T x_r and db, bit-equal between the ranks."""
    script = tmp_path / "lowrank_worker.py"
    script.write_text(_LOWRANK_WORKER)
    env = dict(os.environ, SVAE_ROOT=ROOT, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    code = ("import sys; sys.path.insert(0, %r); from spatial_vae_amd import dp; "
            "sys.exit(dp.launch_ranks(2, [%r]))" % (ROOT, str(script)))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("lowrank ok") == 2
