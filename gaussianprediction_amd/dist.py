"""View-parallel data parallelism: one process per GPU, Gaussians + MLP
replicated, each rank renders a different camera, gradients are SUMMED across ranks -- exactly the
reference's single-GPU `--batch` accumulation (`loss_ = stack(batch_loss).sum()`, [REF train.py:113-119])
run in parallel.  One flat fp32 bucket holds every parameter's .grad (views into it), so the exchange
is a single RCCL all-reduce over xGMI with no packing copies; `radii` are combined with MAX and the
visibility filter follows from it [REF train.py:121-122].
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGradBucket:
    """Makes every parameter's .grad a view into one contiguous buffer."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.offsets = []
        n = 0
        for p in self.params:                       # 64-element (256 B) aligned segments
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero(self):
        self.flat.zero_()

    def rebind(self):
        """Re-attach views (needed if something replaced .grad, e.g. zero_grad(set_to_none=True))."""
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat[off:off + 1].data_ptr():
                p.grad = self.flat[off:off + p.numel()].view_as(p)

    def all_reduce_sum(self, group=None, async_op=False):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return None


def reduce_view_stats(radii: torch.Tensor, group=None):
    """radii = elementwise max over the views of the batch; visibility = radii > 0
    [REF train.py:121-122]."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=group)
    return radii, radii > 0


def init_from_env(backend: str | None = None):
    """torchrun-style init (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
