"""Direct gradient sinks.

When a leaf parameter already owns an allocated, contiguous fp32 `.grad` (as the flat gradient bucket of
`dist.FlatGradBucket` arranges), the backward kernels accumulate straight into it and the autograd
Function returns None for that input: no temporary gradient tensor, no separate AccumulateGrad pass
(for the SH coefficients that pass alone moves 3 x 180 MB per step at 1M Gaussians).  Standard PyTorch
flows (grads None before backward / zero_grad(set_to_none=True)) never take this path.

Because autograd's AccumulateGrad node does not run for such a leaf, post-accumulate hooks do not
fire; consumers that need the "gradient of this leaf is final" event (the overlapped all-reduce)
register a callback here instead."""
from __future__ import annotations

import torch

_callbacks = []


def sink_of(t):
    """The tensor to accumulate into, or None if `t` is not an eligible leaf."""
    if t is None or not isinstance(t, torch.Tensor) or not t.is_leaf or not t.requires_grad:
        return None
    g = t.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.device != t.device or g.shape != t.shape:
        return None
    if g.data_ptr() % 16 != 0:
        return None
    return g


def register_callback(fn):
    _callbacks.append(fn)
    return fn


def unregister_callback(fn):
    if fn in _callbacks:
        _callbacks.remove(fn)


def notify(param):
    for fn in list(_callbacks):
        fn(param)
