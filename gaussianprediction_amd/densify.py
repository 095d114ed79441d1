"""Densify / prune with optimizer-state surgery (second half), mirroring
[REF scene/gaussian_model.py:532-690, 739-760] on top of this package's flat gradient bucket + fused Adam.

The reference edits `torch.optim.Adam`'s per-parameter state in place (`cat_tensors_to_optimizer`, `_prune_optimizer`).
Here the per-Gaussian parameters, their Adam moments and the gradient bucket are rebuilt together by
`TrainStep.rebuild_optimizer`: moments of surviving rows are carried over, new rows start at zero (as
`torch.zeros_like(extension_tensor)` does in the reference), the step count is preserved.

This is bookkeeping that runs every few hundred iterations; it is written with torch tensor ops (plumbing), not kernels.
"""
from __future__ import annotations

import torch
from torch import nn

PER_GAUSSIAN = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
                ("scaling", "_scaling"), ("rotation", "_rotation"), ("motion_feature", "motion_feature"))


class DensificationStats:
    """xyz_gradient_accum / denom / max_radii2D  [REF scene/gaussian_model.py:434-437, 755-760; train.py:164-170]."""

    def __init__(self, n, device):
        self.reset(n, device)

    def reset(self, n, device):
        self.xyz_gradient_accum = torch.zeros(n, 1, device=device)
        self.xyz_gradient_accum_max = torch.zeros(n, 1, device=device)
        self.denom = torch.zeros(n, 1, device=device)
        self.max_radii2D = torch.zeros(n, device=device)

    def add(self, viewspace_point_tensor, update_filter, radii=None):
        """add_densification_stats [REF :755-760] + the max_radii2D update of train.py:166."""
        grad = torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
        self.xyz_gradient_accum[update_filter] += grad
        self.denom[update_filter] += 1
        cur = self.xyz_gradient_accum_max[update_filter]
        self.xyz_gradient_accum_max[update_filter] = torch.where(grad > cur, grad, cur)
        if radii is not None:
            self.max_radii2D[update_filter] = torch.max(self.max_radii2D[update_filter], radii[update_filter].to(torch.float32))


def build_rotation(r):
    """[REF utils/general_utils.py:78-99] rotation matrices of (w, x, y, z) quaternions, normalised inside."""
    q = r / torch.sqrt((r * r).sum(-1, keepdim=True))
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


def _per_gaussian(model):
    return {name: getattr(model, attr) for name, attr in PER_GAUSSIAN if hasattr(model, attr)}


def _apply(model, step, new_params, keep, n_new):
    """Install `new_params` (name -> tensor) and rebuild bucket + optimizer.  `keep` = bool mask over the OLD rows that
    survive (in order); `n_new` rows were appended after them with zero moments."""
    old = _per_gaussian(model)
    moments = step.adam_moments() if step is not None else {}
    carried = {}
    for name, p in old.items():
        if id(p) in moments:
            m, v = moments[id(p)]
            tail = [n_new] + list(p.shape[1:])
            carried[name] = (torch.cat([m[keep], torch.zeros(tail, device=m.device)]), torch.cat([v[keep], torch.zeros(tail, device=v.device)]))
    fresh = {}
    for name, attr in PER_GAUSSIAN:
        if name in new_params:
            q = nn.Parameter(new_params[name].contiguous().requires_grad_(old[name].requires_grad))
            setattr(model, attr, q)
            fresh[name] = q
    if step is not None:
        step.rebuild_optimizer({id(fresh[name]): mv for name, mv in carried.items()})


def densify_and_prune(model, step, stats, max_grad, min_opacity, extent, max_screen_size, percent_dense=0.01, n_split=2,
                      generator=None):
    """`densify` (clone + split) followed by `prune` [REF scene/gaussian_model.py:690-694, 739-747; train.py:171-175].
    Returns (n_cloned, n_split_sources, n_pruned)."""
    dev = model._xyz.device
    if hasattr(step, "wait_side"):
        step.wait_side()            # a side-stream parameter update (TrainStep: the SH Adam) must have landed
    P = {k: v.detach() for k, v in _per_gaussian(model).items()}
    N0 = P["xyz"].shape[0]
    grads = stats.xyz_gradient_accum / stats.denom
    grads[grads.isnan()] = 0.0
    scaling = torch.exp(P["scaling"])                              # get_scaling [REF :138-140]
    big = scaling.max(dim=1).values > percent_dense * extent
    # ---- clone: small Gaussians with a large view-space gradient [REF :668-688]
    clone = (torch.norm(grads, dim=-1) >= max_grad) & ~big
    n_clone = int(clone.sum())
    # ---- split: large ones; gradients padded with zeros for the rows the clone step appended [REF :645-666]
    split = (grads.squeeze(-1) >= max_grad) & big
    n_src = int(split.sum())
    stds = scaling[split].repeat(n_split, 1)
    samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
    rots = build_rotation(P["rotation"][split]).repeat(n_split, 1, 1)
    split_new = {k: v[split].repeat(n_split, *([1] * (v.dim() - 1))) for k, v in P.items()}
    split_new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + P["xyz"][split].repeat(n_split, 1)
    split_new["scaling"] = torch.log(scaling[split].repeat(n_split, 1) / (0.8 * n_split))
    # rows after densify: [old N0 | clones | split children]; the split sources are then pruned [REF :664-666]
    appended = {k: torch.cat([P[k][clone], split_new[k]]) for k in P}
    n_app = n_clone + n_split * n_src
    keep_old = ~split
    # ---- prune: transparent / oversized [REF :739-747], evaluated on the densified set
    opacity_all = torch.sigmoid(torch.cat([P["opacity"][keep_old], appended["opacity"]])).squeeze(-1)
    prune = opacity_all < min_opacity
    if max_screen_size:
        radii_all = torch.cat([stats.max_radii2D[keep_old], torch.zeros(n_app, device=dev)])    # reset to 0 by densification_postfix
        scale_all = torch.exp(torch.cat([P["scaling"][keep_old], appended["scaling"]])).max(dim=1).values
        prune = prune | (radii_all > max_screen_size) | (scale_all > 0.1 * extent)
    n_kept_old = int(keep_old.sum())
    keep_final_old = keep_old.clone()
    keep_final_old[keep_old] = ~prune[:n_kept_old]
    keep_app = ~prune[n_kept_old:]
    new_params = {k: torch.cat([P[k][keep_final_old], appended[k][keep_app]]) for k in P}
    _apply(model, step, new_params, keep_final_old, int(keep_app.sum()))
    stats.reset(new_params["xyz"].shape[0], dev)
    return n_clone, n_src, int(prune.sum())


def reset_opacity(model, step):
    """[REF scene/gaussian_model.py:526-530]: opacity <- inverse_sigmoid(min(opacity, 0.01)), Adam moments of it zeroed."""
    o = torch.sigmoid(model._opacity.detach())
    new = torch.minimum(o, torch.full_like(o, 0.01))
    new = torch.log(new / (1 - new))
    with torch.no_grad():
        model._opacity.copy_(new)
    if step is not None:
        mv = step.adam_moments().get(id(model._opacity))
        if mv is not None:
            mv[0].zero_(); mv[1].zero_()
