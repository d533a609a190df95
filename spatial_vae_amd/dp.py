"""Data-parallel training step: one process per GPU, gradients summed with one RCCL all-reduce.

The reference is single-device; this is the build's multi-GPU path
(section 8e).  Images are independent given the parameters and every loss term is a batch mean,
so rank g of G takes a contiguous slice of each global minibatch, computes the gradient of its
local mean weighted by local_count / global_count, and ONE all-reduce (sum) of the flat fp32
gradient over xGMI yields the gradient of the global mean.  All ranks hold full replicas of
p_net, q_net and the Adam state; TrainStep broadcasts rank 0's parameters once at construction
(every process initialises its modules from its own RNG) and from then on all ranks apply the
identical Adam update to identical gradients, so no further broadcast is needed.

Every parameter's .grad is a view into one flat buffer, so the collective runs on the buffer the
backward pass wrote -- no pack/unpack copies.  The buffer is laid out [p_net | q_net | 3 metrics]
(the three logged scalars elbo, log_p, kl, pre-weighted like the gradients: no collective for
logging).  By default ONE all-reduce of the whole buffer follows backward().  With a heavy encoder
(>= 32 M parameters: the galaxy configuration) the buffer goes out in two buckets: the decoder's
gradients are complete as soon as svae_decoder_backward has been enqueued, so their bucket is
all-reduced on a side stream while autograd is still running the encoder's backward, and the
second bucket (q_net + metrics) follows when backward() returns.
Backend: "nccl" (= RCCL on ROCm) when the parameters live on a GPU, "gloo" on CPU (tests) and
for the shared-GPU rehearsal (SVAE_SHARE_GPU=1: every rank on cuda:0).
"""
import datetime
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_process_group(device_is_gpu):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun sets them)."""
    rank, world, local = env_world()
    # rehearsal on a 1-GPU box: SVAE_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo for the collective
    share = os.environ.get("SVAE_SHARE_GPU") == "1"
    if share:
        local = 0
    # a rank that never arrives at a collective (a dead peer, a wedged GPU) must end the job, not hang it: every collective
    # of the group carries this limit (SVAE_DP_TIMEOUT seconds, default 300; launch_ranks has its own wall-clock limit on top)
    limit = datetime.timedelta(seconds=int(os.environ.get("SVAE_DP_TIMEOUT", "300")))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if device_is_gpu and not share:
            have = torch.cuda.device_count()        # counts devices without initialising the GPU
            if have <= local:
                raise SystemExit("spatial_vae_amd.dp: rank %d of %d needs GPU %d but this node shows %d device(s) "
                                 "(one process per GPU: WORLD_SIZE must not exceed the GPUs of the node)" % (rank, world, local, have))
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local), timeout=limit)
        else:
            dist.init_process_group(backend="gloo", timeout=limit)
    elif world == 1 and solo_collectives() and device_is_gpu and not dist.is_initialized():
        # SVAE_DP_SOLO=1: a ONE-rank RCCL group whose collectives are really issued (see collectives_on)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    return rank, world, local


def solo_collectives():
    """SVAE_DP_SOLO=1: run the data-parallel code path -- parameter broadcast, the bucketed all-reduces on the side and compute
    streams, the metric tail -- with a process group of ONE rank.  On a 1-GPU box this is the only way to execute the RCCL
    transport calls themselves (backend "nccl"); the results must equal the plain single-process run."""
    return os.environ.get("SVAE_DP_SOLO") == "1"


def collectives_on():
    """True when the step must issue its collectives: more than one rank, or the one-rank rehearsal."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or solo_collectives())


def launch_ranks(nproc, argv, env=None, poll=0.2, timeout=None):
    """Start `nproc` fresh Python processes running `argv` (script + arguments), one per GPU of this node, with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way torch.distributed.run sets them, and wait.  Returns 0
    when every rank exited 0, otherwise the first non-zero exit code (the remaining ranks are terminated: a dead peer
    would leave them blocked in a collective).  `timeout` (seconds; default SVAE_LAUNCH_TIMEOUT or 1500) is a wall-clock
    limit for the whole job: when it passes, every rank is terminated (killed if it ignores that) and 124 is returned -- one
    rank stuck in a collective cannot hold the caller forever.  The children inherit stdout/stderr, so rank 0's output is the
    job's.  The caller must not have touched the GPU: the children are new processes (fork + exec of the interpreter), never a
    re-exec of this one."""
    if timeout is None:
        timeout = float(os.environ.get("SVAE_LAUNCH_TIMEOUT", "1500"))
    deadline = time.monotonic() + timeout
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(nproc):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL needs it)
        e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // nproc)))
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(poll)
        if time.monotonic() > deadline:
            sys.stderr.write("spatial_vae_amd.dp.launch_ranks: %d rank(s) still running after %.0f s -- terminating the job\n"
                             % (len(live), timeout))
            for pr in live:
                pr.terminate()
            rc = rc or 124
            break
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in live:
                    other.terminate()
    for pr in procs:
        try:
            pr.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pr.kill()
    return rc


def shard_bounds(n, rank, world):
    """Contiguous, near-equal slices; the first n % world ranks get one extra row."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class FlatGrads(object):
    """One flat fp32 gradient buffer (and optionally one flat parameter buffer) for a set of parameters.

    * every parameter's storage is re-pointed into `flat_param` (p.data becomes a view), so a single
      fused Adam update over one tensor steps all of them;
    * gradients live in `flat`: parameters in `sinks` (the decoder's, whose backward kernels take an
      output pointer) get their view handed to the kernels and are written in place -- their .grad stays
      None until autograd assigns that very view, no accumulate kernel runs; every other parameter's
      .grad IS its view and autograd accumulates into it in place.
    """

    ALIGN = 64   # elements: every parameter starts on a 256-byte boundary of the flat buffers (the kernels read weights with
                 # 16-byte loads); the padding holds zeros in both buffers and stays zero under Adam

    def __init__(self, params, sink_params=None, flatten_params=False, tail=0):
        self.params = [p for p in params if p.requires_grad]
        self.offsets = []
        n = 0
        for p in self.params:
            n = (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            self.offsets.append(n)
            n += p.numel()
        n = (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        ref = self.params[0]
        self.n = n
        # `tail` extra floats after the gradients ride along in the same collective (TrainStep: the logged scalars)
        self.buffer = torch.zeros(n + tail, dtype=torch.float32, device=ref.device)
        self.flat = self.buffer[:n]
        self.tail = self.buffer[n:] if tail else None
        self.flat_param = None
        sink_ids = {id(p): name for name, p in (sink_params or {}).items()}
        self.sinks = {}
        self._sink_list = []
        if flatten_params:
            self.flat_param = torch.zeros(n, dtype=torch.float32, device=ref.device)
        for p, off in zip(self.params, self.offsets):
            view = self.flat[off:off + p.numel()].view_as(p)
            if flatten_params:
                pv = self.flat_param[off:off + p.numel()].view_as(p)
                pv.copy_(p.data)
                p.data = pv
            if id(p) in sink_ids:
                self.sinks[sink_ids[id(p)]] = view
                self._sink_list.append(p)
                p.grad = None
            else:
                p.grad = view

    def zero(self, already_cleared=False):
        """optim.zero_grad(): clear the flat buffer (unless the optimiser kernel already did) and detach the sink
        parameters' .grad (the decoder kernels write their views directly)."""
        if not already_cleared:
            self.flat.zero_()
        for p in self._sink_list:
            p.grad = None

    def all_reduce(self, weight=1.0):
        """buffer <- sum over ranks of weight * buffer (weight = local_count / global_count)."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            if weight != 1.0:
                self.buffer.mul_(weight)
            dist.all_reduce(self.buffer, op=dist.ReduceOp.SUM)


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def shared_seed(device):
    """Rank 0's torch.initial_seed(), the same integer on every rank: seeds the shuffle permutation, the global noise
    draw and (under DP) the augmentation angles, so that every rank slices the SAME global minibatch."""
    seed = torch.initial_seed() % (2 ** 62)
    if collectives_on():
        t = torch.tensor([seed], dtype=torch.int64, device=device)
        dist.broadcast(t, src=0)
        seed = int(t.item())
    return seed


def assert_same_on_all_ranks(t, what):
    """Raise on every rank if the values of `t` (a small tensor) differ between ranks."""
    if not collectives_on():
        return
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError("data-parallel ranks disagree on %s (min %s, max %s): they must be identical replicas"
                           % (what, lo.tolist(), hi.tolist()))


class LowRankExchange(object):
    """Data-parallel weight gradients of LARGE Linear layers without moving the gradient.

    dW = dy^T x has rank <= the number of rows (images) behind it.  For the galaxy encoder (BASELINE configs[3]:
    InferenceNetwork(49152, 23, 5000, 2), train_galaxy.py:306, :461-470) the flat gradient is 1.09 GB, 983 MB of it the
    49 152 x 5 000 first layer whose gradient is the LAST to complete -- an all-reduce nothing can hide.  But with a global
    minibatch of 128 that gradient is a sum of 128 outer products: each rank contributes its rows of the two factors
    (x: B_local x n_in, dy: B_local x n_out, already weighted local/global through the backward seed), ONE all-gather moves
    B_global x (n_in + n_out) floats per layer (27.7 MB + 5.1 MB for the two large layers) and every rank forms the GLOBAL
    dW = dy_all^T x_all and db = colsum(dy_all) locally (63 GF + 6 GF).  Every rank multiplies the same gathered matrices in
    the same order, so the replicas stay bit-equal; against the one-rank run the sum over images is merely associated
    differently (fp32 rounding).  Ragged and empty shards are zero-padded to cap = ceil(B_global / world) rows.
    Used for the Linear layers that already take ops.sink_linear (more than ops.ENC_LINEAR_MAX_WEIGHT weights); their
    parameter ranges are left out of the step's all-reduce (TrainStep._segments)."""

    def __init__(self, layers, device):
        # layers: [(key, sink_w, sink_b)] with sink_w (n_out, n_in)
        self.layers = list(layers)
        self.keys = {k for k, _, _ in self.layers}
        self.device = device
        self.width = sum(w.size(0) + w.size(1) for _, w, _ in self.layers)
        self.pending = {}
        self._send = self._recv = None
        self.bytes_last = 0

    def has(self, key):
        return key in self.keys

    def add(self, key, x, dy):
        self.pending[key] = (x.detach(), dy.detach())

    def start(self, cap):
        """Pack this rank's factors (zero rows beyond its own, all zeros for an empty shard) and start the all-gather."""
        world = dist.get_world_size()
        if self._send is None or self._send.size(0) != cap:
            self._send = torch.zeros(cap, self.width, dtype=torch.float32, device=self.device)
            self._recv = torch.empty(world * cap, self.width, dtype=torch.float32, device=self.device)
        else:
            self._send.zero_()
        off = 0
        for key, w, _ in self.layers:
            n_out, n_in = w.shape
            if key in self.pending:
                x, dy = self.pending[key]
                rows = x.size(0)
                if rows > cap:
                    raise RuntimeError("LowRankExchange: %d local rows but the global minibatch allows %d per rank" % (rows, cap))
                self._send[:rows, off:off + n_in].copy_(x)
                self._send[:rows, off + n_in:off + n_in + n_out].copy_(dy)
            off += n_in + n_out
        self.pending = {}
        self.bytes_last = self._recv.numel() * 4
        return dist.all_gather_into_tensor(self._recv, self._send, async_op=True)

    def finish(self):
        """dW and db of every large layer from the gathered factors, written into the flat gradient buffer."""
        off = 0
        for _, w, b in self.layers:
            n_out, n_in = w.shape
            xa = self._recv[:, off:off + n_in]
            dya = self._recv[:, off + n_in:off + n_in + n_out]
            torch.mm(dya.t(), xa, out=w)
            torch.sum(dya, 0, out=b)
            off += n_in + n_out


class TrainStep(object):
    """forward + backward + (all-reduce) + Adam step for one (local) minibatch.

    Mirrors the loop body of train_epoch (/root/reference/train_mnist.py:143-150):
    loss = -elbo; backward; optim.step; optim.zero_grad -- with the zeroing done on the flat
    buffer and the metrics left on the device (the caller decides when to pay for .item()).
    torch.optim.Adam is unchanged; it simply sees ONE parameter (the flat buffer every module
    parameter is a view of), which is the same element-wise update in one kernel.

    Data parallel (see the module docstring): call with this rank's slice of the global minibatch,
    weight = local_rows / global_rows and global_batch = global_rows (needed by a rank whose slice is empty when large encoder
    layers exchange gradient factors: LowRankExchange).  The weight enters as the seed of backward() (-weight instead of -1: no
    scaling pass over the gradient buffer).  A rank whose slice is EMPTY (ragged last batch smaller than the world)
    skips forward and backward but still joins both collectives with zeros.  After the call `metrics` holds
    (elbo, log_p, kl) of the GLOBAL minibatch on every rank, valid until the next call.
    """

    def __init__(self, p_net, q_net, eval_minibatch, lr=1e-4, fused_adam=None, bucketed=None, **eval_kwargs):
        self.p_net, self.q_net = p_net, q_net
        self.eval_minibatch = eval_minibatch
        self.eval_kwargs = eval_kwargs
        p_params = [p for p in p_net.parameters() if p.requires_grad]
        params = p_params + list(q_net.parameters())
        on_gpu = params[0].is_cuda
        sink_params = p_net.decoder_parameters() if (on_gpu and hasattr(p_net, "decoder_parameters")) else None
        q_sinks = {}
        if on_gpu and sink_params is not None and hasattr(q_net, "layers"):
            # the encoder's plain Linear layers: elbo._encode runs them through ops.sink_linear, whose backward writes the
            # weight / bias gradients into the flat buffer directly (no AccumulateGrad add per parameter)
            for idx, m in enumerate(q_net.layers):
                if isinstance(m, torch.nn.Linear) and m.bias is not None:
                    q_sinks["layers.%d.weight" % idx] = m.weight
                    q_sinks["layers.%d.bias" % idx] = m.bias
            sink_params = dict(sink_params, **q_sinks)
        self.grads = FlatGrads(params, sink_params=sink_params, flatten_params=True, tail=3)
        # [0, n_p) = the decoder's bucket: up to where the first encoder parameter starts
        self.n_p = self.grads.offsets[len(p_params)] if len(p_params) < len(self.grads.params) else self.grads.n
        # static (never data dependent), so that every rank issues the same collectives in the same order
        if bucketed is None:
            # Two buckets pay a cross-stream fork/join (measured on a one-rank RCCL group at BASELINE cfg 2: +0.080 ms per step
            # against +0.045 ms for ONE all-reduce after backward()), which only an encoder backward long enough to hide the
            # decoder bucket's transfer can repay: the galaxy encoder (271 M parameters, ~1 ms of GEMMs) yes, the 0.04 ms
            # encoders of the other configs no.  SVAE_DP_BUCKETS=1|2 overrides the rule.
            env = os.environ.get("SVAE_DP_BUCKETS")
            n_q = sum(p.numel() for p in q_net.parameters())
            bucketed = bool(sink_params) and (env == "2" or (env != "1" and n_q >= 32 * 1024 * 1024))
        self._bucketed = bool(bucketed)
        if sink_params:
            p_net._grad_sinks = {k: v for k, v in self.grads.sinks.items() if k not in q_sinks}
            p_net._grad_sinks["__ready__"] = self._decoder_grads_ready
            if q_sinks:
                q_net._grad_sinks = {k: v for k, v in self.grads.sinks.items() if k in q_sinks}
        self.device = params[0].device
        # Large encoder layers (those elbo._encode runs through ops.sink_linear) exchange the factors of their weight gradient
        # instead of the gradient (LowRankExchange); SVAE_DP_LOWRANK=0 keeps them in the all-reduce.
        self._lowrank = None
        skip = set()
        if on_gpu and q_sinks and collectives_on() and os.environ.get("SVAE_DP_LOWRANK") != "0":
            from .ops import ENC_LINEAR_MAX_WEIGHT
            big = []
            for idx, m in enumerate(q_net.layers):
                if isinstance(m, torch.nn.Linear) and m.bias is not None and m.weight.numel() > ENC_LINEAR_MAX_WEIGHT:
                    big.append(("layers.%d" % idx, self.grads.sinks["layers.%d.weight" % idx], self.grads.sinks["layers.%d.bias" % idx]))
                    skip.update((id(m.weight), id(m.bias)))
            if big:
                self._lowrank = LowRankExchange(big, self.device)
                q_net._grad_sinks["__lowrank__"] = self._lowrank
        # element ranges of the gradient buffer that go through the all-reduce: everything except the low-rank layers; the
        # metric tail rides at the end of the last range
        self._segments = []
        lo = 0
        for p, off in zip(self.grads.params, self.grads.offsets):
            if id(p) in skip:
                if off > lo:
                    self._segments.append((lo, off))
                lo = (off + p.numel() + FlatGrads.ALIGN - 1) // FlatGrads.ALIGN * FlatGrads.ALIGN
        self._segments.append((lo, self.grads.buffer.numel()))
        self._seeds = {}
        self.master = torch.nn.Parameter(self.grads.flat_param)
        self.master.grad = self.grads.flat
        if on_gpu and fused_adam is None:
            from .ops import FlatAdam    # torch.optim.Adam's arithmetic in one small kernel over the flat buffer
            self.optim = FlatAdam([self.master], lr=lr, zero_grad=True)   # the update clears the flat gradient behind itself
        else:
            kw = {"fused": True, "capturable": True} if (fused_adam and on_gpu) else {}
            self.optim = torch.optim.Adam([self.master], lr=lr, **kw)
        self._graph = None
        self._work_p = None
        self._side = torch.cuda.Stream(self.device) if on_gpu else None
        self.metrics = self.grads.tail
        self.comm_events = None          # bench.py: (start, end) HIP events around the step's collectives
        self.sync_replicas()

    def sync_replicas(self):
        """Make every replica rank 0's: each process initialised its modules from its own RNG."""
        if collectives_on():
            dist.broadcast(self.grads.flat_param, src=0)

    def aliased(self):
        """True while every module parameter still lives inside the flat parameter buffer (a Module._apply round trip --
        net.cpu(), net.to() -- silently breaks that: the optimiser would then update orphaned memory)."""
        base = self.grads.flat_param
        lo, hi = base.data_ptr(), base.data_ptr() + base.numel() * 4
        return all(lo <= p.data_ptr() < hi for p in self.grads.params)

    # ---- collectives ------------------------------------------------------------------------------
    def _decoder_grads_ready(self):
        """Called by ops._Decoder.backward once svae_decoder_backward is enqueued: every p_net gradient is final, so its
        bucket goes out now, on a side stream, under the encoder's backward."""
        if collectives_on() and self._bucketed and self._work_p is None:
            self._work_p = self._launch(self.grads.buffer[:self.n_p], side=True)

    def _launch(self, tensor, side):
        if self._side is not None and side:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=True)
        return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=True)

    def allreduce_segments(self):
        """(lo, hi) element ranges of the gradient buffer, in launch order, that this step all-reduces: the decoder's bucket
        first when it goes out early, then every other range outside the low-rank layers (the metric tail is in the last)."""
        segs = list(self._segments)
        if self._bucketed:
            out = [(0, self.n_p)]
            for lo, hi in segs:
                lo = max(lo, self.n_p)
                if hi > lo:
                    out.append((lo, hi))
            return out
        return segs

    def _reduce(self, global_batch=None):
        if not collectives_on():
            return
        ev = self.comm_events
        segs = self.allreduce_segments()
        work = []
        if self._bucketed:
            if self._work_p is None:                      # empty shard (no backward ran): join the first bucket too
                self._work_p = self._launch(self.grads.buffer[:self.n_p], side=True)
            work.append(self._work_p)
            segs = segs[1:]
        if ev:
            ev[0].record()
        gather = None
        if self._lowrank is not None:
            # the factors first: the local GEMMs that follow then run under the remaining all-reduce
            world = dist.get_world_size()
            if global_batch is None:
                raise RuntimeError("TrainStep: the low-rank gradient exchange needs global_batch= (rows of the GLOBAL minibatch)")
            gather = self._lowrank.start((int(global_batch) + world - 1) // world)
        for lo, hi in segs:
            work.append(self._launch(self.grads.buffer[lo:hi], side=False))
        if gather is not None:
            gather.wait()
            self._lowrank.finish()
        for w in work:
            w.wait()                                      # the compute stream waits; the host does not
        if ev:
            ev[1].record()
        self._work_p = None

    def _seed(self, weight):
        t = self._seeds.get(weight)
        if t is None:
            t = self._seeds[weight] = torch.tensor(-float(weight), dtype=torch.float32, device=self.device)
        return t

    def _step(self, x, batch, weight, kw):
        out = None
        kw = dict(kw)
        global_batch = kw.pop("global_batch", None)
        rows = batch[0].size(0) if torch.is_tensor(batch[0]) else 1
        if global_batch is None and weight > 0:
            global_batch = int(round(rows / weight))
        if rows > 0:
            args = dict(self.eval_kwargs)
            args.update(kw)
            out = self.eval_minibatch(x, *batch, self.p_net, self.q_net, **args)
            elbo = out[0]
            # loss = -elbo (train_mnist.py:147-148) without a negation kernel; under DP the seed is -local/global
            elbo.backward(self._seed(weight))
            base = getattr(elbo, "_base", None)           # ops.elbo_head returns views of one (elbo, log_p, kl) vector
            vec = base if (base is not None and base.numel() == 3) else torch.stack([out[0], out[1], out[2]])
            if not collectives_on() and weight == 1.0:
                self.metrics = vec.detach()               # nothing to reduce: the minibatch's own metrics, no copy
            else:
                torch.mul(vec.detach(), float(weight), out=self.grads.tail)
                self.metrics = self.grads.tail
        else:
            self.grads.tail.zero_()
            self.metrics = self.grads.tail
        self._reduce(global_batch)
        self.optim.step()
        self.grads.zero(already_cleared=getattr(self.optim, "zero_grad_in_step", False))
        return out

    def __call__(self, x, *batch, weight=1.0, **kw):
        if self._graph is not None:
            return self._replay(x, batch, weight, kw)
        return self._step(x, batch, weight, kw)

    # ---- optional: the whole step as one HIP graph (single-GPU, fixed shapes) ------------------------
    def capture(self, x, *batch, warmup=3, **kw):
        """(Needs TrainStep(..., fused_adam=True): torch's capturable Adam keeps its step counter on the device.)
        Record forward + backward + Adam for inputs of these shapes into a HIP graph (torch.cuda.CUDAGraph:
        the C-ABI launches go to the capture stream like any torch op).  Afterwards __call__ copies the batch
        into the static input buffers and replays: ~60 kernel launches become one.  Noise is drawn inside the
        graph from torch's graph-safe Philox generator, as the reference draws it on the device."""
        if world_size() > 1:
            raise RuntimeError("graph capture is wired for the single-GPU step only")
        self._static_x = x
        self._static_batch = [b.clone() if torch.is_tensor(b) else b for b in batch]
        self._static_kw = dict(kw)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step(x, self._static_batch, 1.0, self._static_kw)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._static_out = self._step(x, self._static_batch, 1.0, self._static_kw)
        self._graph = g
        return self

    def _replay(self, x, batch, weight, kw):
        kw = {k: v for k, v in kw.items() if k != "global_batch"}
        if weight != 1.0 or kw:
            raise RuntimeError("a captured step replays fixed arguments")
        for dst, src in zip(self._static_batch, batch):
            if torch.is_tensor(dst) and src is not dst:
                dst.copy_(src, non_blocking=True)
        self._graph.replay()
        return self._static_out
