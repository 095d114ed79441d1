"""Seed-fixed synthetic scenes (datasets are not available offline).

Follows SURVEY.md section 8(d): positions uniform in a cube (mirrors the reference's own random init,
[REF scene/dataset_readers.py:259]), log-uniform scales sized for a target mean screen footprint,
un-normalised N(0,1) quaternions (as the model stores `_rotation`), opacity logit of U(0.05,0.95),
SH f_dc ~ N(0,1), f_rest ~ N(0,0.1), motion_feature ~ 1e-3 U(-1,1) [REF scene/gaussian_model.py:362].
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class SceneSpec:
    n_gaussians: int
    extent: tuple = (1.3, 1.3, 1.3)       # half-extent of the position box
    scale_lo: float = 0.004               # world-space sigma range (log-uniform)
    scale_hi: float = 0.02
    anisotropy: float = 3.0               # per-axis sigma jitter factor (log-uniform in [1/a, a]^(1/2))
    sh_degree: int = 3
    feature_dim: int = 32
    seed: int = 2024


def make_gaussians(spec: SceneSpec, device="cpu", dtype=torch.float32):
    """Returns a dict of *raw* (pre-activation) tensors laid out exactly as GaussianModel's
    parameters [REF scene/gaussian_model.py:347-362]."""
    g = torch.Generator().manual_seed(spec.seed)
    N = spec.n_gaussians
    ext = torch.tensor(spec.extent, dtype=torch.float64)
    xyz = (torch.rand(N, 3, generator=g, dtype=torch.float64) * 2 - 1) * ext
    base = torch.exp(torch.rand(N, 1, generator=g, dtype=torch.float64) *
                     (math.log(spec.scale_hi) - math.log(spec.scale_lo)) + math.log(spec.scale_lo))
    jit = torch.exp((torch.rand(N, 3, generator=g, dtype=torch.float64) - 0.5) * math.log(spec.anisotropy))
    scaling = torch.log(base * jit)
    rotation = torch.randn(N, 4, generator=g, dtype=torch.float64)
    op = torch.rand(N, 1, generator=g, dtype=torch.float64) * 0.9 + 0.05
    opacity = torch.log(op / (1 - op))
    M = (spec.sh_degree + 1) ** 2
    f_dc = torch.randn(N, 1, 3, generator=g, dtype=torch.float64)
    f_rest = torch.randn(N, M - 1, 3, generator=g, dtype=torch.float64) * 0.1
    mf = 1e-3 * (2 * torch.rand(N, spec.feature_dim, generator=g, dtype=torch.float64) - 1)
    out = dict(xyz=xyz, scaling=scaling, rotation=rotation, opacity=opacity, features_dc=f_dc,
               features_rest=f_rest, motion_feature=mf)
    return {k: v.to(dtype).to(device).contiguous() for k, v in out.items()}


def make_keypoints(xyz: torch.Tensor, motion_feature: torch.Tensor, K: int, nearest_num: int,
                   feature_amplify: float = 5.0, seed: int = 2024, chunk: int = 65536):
    """K keypoints by strided sampling of the positions, exact brute-force kNN indices in the
    reference's "hybird" 3+feature_dim space [REF scene/gaussian_model.py:110-125], and random raw
    blend weights [N, 2*nearest_num] standing in for the hash-grid weights model
    [REF scene/gaussian_model.py:257] (both are *inputs* to the hot path)."""
    N = xyz.shape[0]
    stride = max(N // K, 1)
    idx = torch.arange(K, device=xyz.device) * stride
    kp_xyz = xyz[idx].clone()
    kp_feat = motion_feature[idx].clone()
    a = torch.cat([xyz, motion_feature * feature_amplify], dim=-1)
    b = torch.cat([kp_xyz, kp_feat * feature_amplify], dim=-1)
    nn_idx = torch.empty(N, nearest_num, dtype=torch.int64, device=xyz.device)
    for s in range(0, N, chunk):
        d = torch.cdist(a[s:s + chunk].double(), b.double())
        nn_idx[s:s + chunk] = d.topk(nearest_num, dim=-1, largest=False).indices
    g = torch.Generator().manual_seed(seed + 1)
    raw_w = torch.randn(N, 2 * nearest_num, generator=g).to(xyz.device).to(xyz.dtype)
    return kp_xyz, kp_feat, nn_idx, raw_w
