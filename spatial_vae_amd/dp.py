"""Data-parallel training step: one process per GPU, gradients summed with one RCCL all-reduce.

The reference is single-device; this is the build's multi-GPU path
(section 8e).  Images are independent given the parameters and every loss term is a batch mean,
so rank g of G takes a contiguous slice of each global minibatch, computes the gradient of its
local mean weighted by local_count / global_count, and ONE all-reduce (sum) of the flat fp32
gradient over xGMI yields the gradient of the global mean.  All ranks hold full replicas of
p_net, q_net and the Adam state and apply the identical update, so no broadcast is needed.

Every parameter's .grad is a view into one flat buffer, so the collective runs on the buffer the
backward pass wrote -- no pack/unpack copies.  Backend: "nccl" (= RCCL on ROCm) when the
parameters live on a GPU, "gloo" on CPU (tests).
"""
import os

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
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if device_is_gpu and not share:
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo")
    return rank, world, local


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

    def __init__(self, params, sink_params=None, flatten_params=False):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=torch.float32, device=ref.device)
        self.flat_param = None
        sink_ids = {id(p): name for name, p in (sink_params or {}).items()}
        self.sinks = {}
        self._sink_list = []
        if flatten_params:
            self.flat_param = torch.empty(n, dtype=torch.float32, device=ref.device)
        off = 0
        for p in self.params:
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
            off += p.numel()

    def zero(self):
        self.flat.zero_()
        for p in self._sink_list:
            p.grad = None

    def all_reduce(self, weight=1.0):
        """flat <- sum over ranks of weight * flat (weight = local_count / global_count)."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            if weight != 1.0:
                self.flat.mul_(weight)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)


class TrainStep(object):
    """forward + backward + (all-reduce) + Adam step for one (local) minibatch.

    Mirrors the loop body of train_epoch (/root/reference/train_mnist.py:143-150):
    loss = -elbo; backward; optim.step; optim.zero_grad -- with the zeroing done on the flat
    buffer and the metrics left on the device (the caller decides when to pay for .item()).
    torch.optim.Adam is unchanged; it simply sees ONE parameter (the flat buffer every module
    parameter is a view of), which is the same element-wise update in one kernel.
    """

    def __init__(self, p_net, q_net, eval_minibatch, lr=1e-4, fused_adam=None, **eval_kwargs):
        self.p_net, self.q_net = p_net, q_net
        self.eval_minibatch = eval_minibatch
        self.eval_kwargs = eval_kwargs
        params = list(p_net.parameters()) + list(q_net.parameters())
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
        self.grads = FlatGrads(params, sink_params=sink_params, flatten_params=True)
        if sink_params:
            p_net._grad_sinks = {k: v for k, v in self.grads.sinks.items() if k not in q_sinks}
            if q_sinks:
                q_net._grad_sinks = {k: v for k, v in self.grads.sinks.items() if k in q_sinks}
        self._minus_one = torch.tensor(-1.0, device=params[0].device)
        self.master = torch.nn.Parameter(self.grads.flat_param)
        self.master.grad = self.grads.flat
        if on_gpu and fused_adam is None:
            from .ops import FlatAdam    # torch.optim.Adam's arithmetic in one small kernel over the flat buffer
            self.optim = FlatAdam([self.master], lr=lr)
        else:
            kw = {"fused": True, "capturable": True} if (fused_adam and on_gpu) else {}
            self.optim = torch.optim.Adam([self.master], lr=lr, **kw)
        self._graph = None

    def _step(self, x, batch, weight, kw):
        args = dict(self.eval_kwargs)
        args.update(kw)
        out = self.eval_minibatch(x, *batch, self.p_net, self.q_net, **args)
        elbo = out[0]
        elbo.backward(self._minus_one)                  # loss = -elbo (train_mnist.py:147-148) without a negation kernel
        self.grads.all_reduce(weight)
        self.optim.step()
        self.grads.zero()
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
        if dist.is_initialized() and dist.get_world_size() > 1:
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
        if weight != 1.0 or kw:
            raise RuntimeError("a captured step replays fixed arguments")
        for dst, src in zip(self._static_batch, batch):
            if torch.is_tensor(dst) and src is not dst:
                dst.copy_(src, non_blocking=True)
        self._graph.replay()
        return self._static_out
