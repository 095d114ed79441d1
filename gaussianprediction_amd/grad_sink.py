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


# Gradient buffers whose CONTENTS are stale: the optimizer consumed them and skipped the zeroing pass because the next
# producer is known to overwrite the whole tensor (the rasterizer's SH gradients: 192 B per Gaussian that would
# otherwise be zero-written by Adam and read back by the next backward).  Keyed by data_ptr.
_stale = set()


def mark_stale(g):
    _stale.add(g.data_ptr())


def take_stale(g):
    """True (and forget it) if `g` must be overwritten rather than accumulated into."""
    k = g.data_ptr()
    if k in _stale:
        _stale.discard(k)
        return True
    return False


def forget_all():
    """Drop every mark (the gradient buffers were re-allocated)."""
    _stale.clear()
    _fresh.clear()


# Gradient buffers known to hold ZEROS that nobody has written since (the optimizer marks them right after its zeroing pass):
# the first producer may WRITE its gradient into such a buffer instead of handing autograd a temporary to be added -- writing
# over zeros is accumulating.  The mark is consumed by the first taker; every later contribution of the same backward (or of
# another view of the same step) goes through autograd's accumulate as usual, so the protocol holds for any number of producers.
_fresh = set()


def mark_fresh(g):
    _fresh.add(g.data_ptr())


def unmark_fresh(g):
    if g is not None:
        _fresh.discard(g.data_ptr())


def take_fresh(g):
    """True (and forget it) if `g` is all zeros and unwritten since: the caller may overwrite it with a first contribution."""
    k = g.data_ptr()
    if k in _fresh:
        _fresh.discard(k)
        return True
    return False


def is_stale(g):
    return g.data_ptr() in _stale


def register_callback(fn):
    _callbacks.append(fn)
    return fn


def unregister_callback(fn):
    if fn in _callbacks:
        _callbacks.remove(fn)


def notify(param):
    for fn in list(_callbacks):
        fn(param)


# ---- optimizer step fused into a producer kernel ---------------------------------------------------------------------------
# A training harness that knows the next backward is the ONLY producer of some leaves' gradients (one view, one rank) can ask the
# producer to apply the optimizer update itself: arm_fused_update(leaves, payload) before backward; the producer calls
# take_fused_update(leaves) -- which disarms -- and, if it gets a payload, updates the parameters in place and writes no gradient.
_fused = {}


def arm_fused_update(leaves, payload):
    _fused[tuple(id(t) for t in leaves)] = payload


def take_fused_update(leaves):
    return _fused.pop(tuple(id(t) for t in leaves), None)


def disarm_fused_update(leaves):
    """True if the update is still armed (no producer took it): the caller runs the ordinary optimizer step for these leaves."""
    return _fused.pop(tuple(id(t) for t in leaves), None) is not None
