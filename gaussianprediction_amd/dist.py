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

    def segment(self, p):
        """The bucket slice that backs p.grad (padded to the 64-element granule)."""
        i = next(k for k, q in enumerate(self.params) if q is p)
        off = self.offsets[i]
        end = self.offsets[i + 1] if i + 1 < len(self.offsets) else self.flat.numel()
        return self.flat[off:end]


class OverlappedGradReducer:
    """SUM-all-reduce of the gradient bucket, overlapped with the rest of backward.

    The rasterizer backward finishes the largest gradients first (SH coefficients: 192 of the 364
    bytes per Gaussian) while the deformation backward (blend, MLP) and the Adam launches are still
    ahead.  A post-accumulate-grad hook on every large leaf starts an asynchronous all-reduce of that
    leaf's bucket segment the moment its gradient is final; RCCL runs it on its own stream over xGMI
    while the compute stream continues.  Small tensors (MLP weights, keypoints: < `small_numel`) are
    reduced together in one trailing collective.  `finish()` makes the compute stream wait for all of
    them; the optimizer then sees exactly sum-over-ranks, i.e. the reference's `--batch` semantics
    [REF train.py:113-119].  Every rank issues the same collectives in the same order (the order is
    fixed by the autograd graph, which is identical on all ranks)."""

    def __init__(self, bucket: FlatGradBucket, group=None, small_numel=1 << 20):
        self.bucket, self.group = bucket, group
        self.handles = []
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.large = [p for p in bucket.params if p.numel() >= small_numel]
        self.small = [p for p in bucket.params if p.numel() < small_numel]
        self._fired = set()
        self._late = set()
        self._sink_cb = None
        if self.enabled:
            from . import grad_sink
            hooks = {id(p): self._make_hook(p) for p in self.large}
            for p in self.large:
                p.register_post_accumulate_grad_hook(hooks[id(p)])
            # leaves whose gradient is written directly by a kernel (grad_sink) never run AccumulateGrad:
            # they announce completion through the sink registry instead
            self._sink_cb = grad_sink.register_callback(lambda p: hooks[id(p)](p) if id(p) in hooks else None)

    def close(self):
        """Detach from the gradient-sink registry (call before the bucket is rebuilt)."""
        if getattr(self, "_sink_cb", None) is not None:
            from . import grad_sink
            grad_sink.unregister_callback(self._sink_cb)
            self._sink_cb = None
        self.enabled = False

    def set_late(self, params):
        """Leaves whose gradient has more than one producer in the coming backward (several views per step; the lifecycle
        opacity's second MLP pass into `_xyz`): a producer's completion notice is not "final" for them, so they are left to
        finish(), which runs after backward."""
        self._late = {id(p) for p in params}

    def _make_hook(self, p):
        def hook(param):
            if id(p) in self._fired or id(p) in self._late:
                return
            self._fired.add(id(p))
            self.handles.append(dist.all_reduce(self.bucket.segment(p), op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return hook

    def finish(self):
        """Call after loss.backward(): reduces what the hooks did not cover, waits for everything."""
        if not self.enabled:
            return
        for p in self.large:                       # leaves that received no gradient this step (still zero)
            if id(p) not in self._fired:
                self.handles.append(dist.all_reduce(self.bucket.segment(p), op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if self.small:
            # the small tensors are contiguous at the end of the bucket when they were registered last;
            # otherwise reduce them one by one (they are tiny)
            idx = [next(k for k, q in enumerate(self.bucket.params) if q is p) for p in self.small]
            if idx == list(range(idx[0], idx[0] + len(idx))):
                off = self.bucket.offsets[idx[0]]
                end = self.bucket.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.bucket.offsets) else self.bucket.flat.numel()
                self.handles.append(dist.all_reduce(self.bucket.flat[off:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                for p in self.small:
                    self.handles.append(dist.all_reduce(self.bucket.segment(p), op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in self.handles:
            h.wait()
        self.handles.clear()
        self._fired.clear()


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
    if os.environ.get("GP_FORCE_LOCAL_RANK") is not None:     # test hook: several ranks on one GPU (gloo only)
        local = int(os.environ["GP_FORCE_LOCAL_RANK"])
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("GP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" IS RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
