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
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if device_is_gpu:
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
    """Owns one flat fp32 buffer; every parameter's .grad is a view into it."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=torch.float32, device=ref.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

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
    """

    def __init__(self, p_net, q_net, eval_minibatch, lr=1e-4, fused_adam=None, **eval_kwargs):
        self.p_net, self.q_net = p_net, q_net
        self.eval_minibatch = eval_minibatch
        self.eval_kwargs = eval_kwargs
        params = list(p_net.parameters()) + list(q_net.parameters())
        self.grads = FlatGrads(params)
        on_gpu = params[0].is_cuda
        if fused_adam is None:
            fused_adam = on_gpu
        kw = {"fused": True} if fused_adam else {}
        self.optim = torch.optim.Adam(params, lr=lr, **kw)

    def __call__(self, x, *batch, weight=1.0, **kw):
        args = dict(self.eval_kwargs)
        args.update(kw)
        out = self.eval_minibatch(x, *batch, self.p_net, self.q_net, **args)
        elbo = out[0]
        (-elbo).backward()
        self.grads.all_reduce(weight)
        self.optim.step()
        self.grads.zero()
        return out
