"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the two per-frame stage-2/3 steps that feed
the hot path its keypoint weights.

**parity unpinned**: both live in un-vendored, un-pinned CUDA dependencies of the reference (`tinycudann` float32 fork
and `frnn`, installed by `env.sh:7-14`; neither is in /root/reference), and the reference has no tests or golden
vectors for them.  What is restated here is the *published* algorithm each call site relies on:

* `weights_model = tcnn.NetworkWithInputEncoding(3 -> 2*nearest_num, Grid/Hash L=16 F=4 T=2^19 N_min=16
  b=exp(ln(2048/16)/15), Linear interpolation; FullyFusedMLP 64 neurons, 2 hidden layers, ReLU, no output activation)`
  [REF scene/gaussian_model.py:370-392], called as `weights_model(self.get_xyz.detach())` [REF :257].
  Multiresolution hash encoding as published (Mueller et al. 2022, and tiny-cuda-nn's grid.h conventions):
  per level l: scale_l = 2^(l log2 b) N_min - 1, resolution_l = ceil(scale_l) + 1, pos = x scale_l + 0.5,
  cell = floor(pos), w = pos - cell; the 8 corners are indexed densely (x + y res + z res^2) while res^3 fits the
  level's table, else by the spatial hash (x*1) ^ (y*2654435761) ^ (z*805459861) (uint32), both modulo the level's
  table size; table size = min(res^3 rounded up to 8, 2^19); features are trilinearly blended.  Inputs are used as they
  are (the reference does not normalise xyz to [0,1]; negative cells wrap through uint32 exactly as in the CUDA code).
  The MLP has no biases; its output is padded to 16 and the first 2*nearest_num columns are returned.
* `frnn.frnn_grid_points(points, keypoints, K=nearest_num, r=1e8)` [REF scene/gaussian_model.py:110-125]: the K nearest
  keypoints of every Gaussian by squared Euclidean distance, ascending, in 3-D ("3D") or in the 35-D space
  [xyz | feature_amplify * motion_feature] ("hybird"); ties go to the lower keypoint index here.
"""
from __future__ import annotations

import math

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)


def grid_meta(n_levels=16, n_features=4, log2_hashmap_size=19, base_resolution=16, per_level_scale=None):
    """Per-level (scale, resolution, table entries, offset in entries)."""
    if per_level_scale is None:
        per_level_scale = math.exp(math.log(2048 / base_resolution) / (n_levels - 1))
    log2_b = math.log2(float(np.float32(per_level_scale)))      # double precision, then one rounding to float32 per level
    scales, ress, sizes, offs = [], [], [], []
    off = 0
    for lvl in range(n_levels):
        scale = np.float32(2.0 ** (lvl * log2_b) * base_resolution - 1.0)
        res = int(math.ceil(float(scale))) + 1
        full = res ** 3
        size = min((full + 7) // 8 * 8, 1 << log2_hashmap_size)
        scales.append(float(scale)); ress.append(res); sizes.append(size); offs.append(off)
        off += size
    return dict(scales=scales, resolutions=ress, sizes=sizes, offsets=offs, total=off, n_levels=n_levels,
                n_features=n_features)


def hash_encode(xyz: torch.Tensor, table: torch.Tensor, meta) -> torch.Tensor:
    """xyz [N,3] float32, table [total, F] float32 -> [N, L*F] (level-major, as tcnn lays its output out)."""
    N = xyz.shape[0]
    F = meta["n_features"]
    outs = []
    for lvl in range(meta["n_levels"]):
        scale, res, size, off = meta["scales"][lvl], meta["resolutions"][lvl], meta["sizes"][lvl], meta["offsets"][lvl]
        # fmaf(scale, x, 0.5): the double product of two floats is exact, so rounding the double sum once is the fused result
        pos = (xyz.to(torch.float64) * float(np.float32(scale)) + 0.5).to(torch.float32)
        cell_f = torch.floor(pos)
        w = (pos - cell_f)
        cell = cell_f.to(torch.int64) & 0xFFFFFFFF          # (uint32)(int) wrap
        dense = res ** 3 <= size
        acc = torch.zeros(N, F, dtype=table.dtype)
        for corner in range(8):
            cw = torch.ones(N, dtype=table.dtype)
            cpos = []
            for d in range(3):
                bit = (corner >> d) & 1
                cw = cw * (w[:, d] if bit else (1 - w[:, d])).to(table.dtype)
                cpos.append((cell[:, d] + bit) & 0xFFFFFFFF)
            if dense:
                idx = (cpos[0] + cpos[1] * res + cpos[2] * res * res) & 0xFFFFFFFF
            else:
                idx = ((cpos[0] * PRIMES[0]) & 0xFFFFFFFF) ^ ((cpos[1] * PRIMES[1]) & 0xFFFFFFFF) ^ ((cpos[2] * PRIMES[2]) & 0xFFFFFFFF)
            idx = idx % size
            acc = acc + cw[:, None] * table[off + idx]
        outs.append(acc)
    return torch.cat(outs, dim=1)


def mlp_forward(feat: torch.Tensor, w1, w2, w3) -> torch.Tensor:
    h = torch.relu(feat @ w1.t())
    h = torch.relu(h @ w2.t())
    return h @ w3.t()


def weights_model(xyz, params, meta, n_out):
    """params: flat tensor [64*64 + 64*64 + 16*64 | table]; returns [N, n_out]."""
    w1 = params[0:4096].view(64, 64)
    w2 = params[4096:8192].view(64, 64)
    w3 = params[8192:9216].view(16, 64)
    table = params[9216:].view(-1, meta["n_features"])
    return mlp_forward(hash_encode(xyz, table, meta), w1, w2, w3)[:, :n_out]


def knn(points: np.ndarray, keypoints: np.ndarray, k: int):
    """indices [N,k] (ascending squared distance, ties to the lower index) and the squared distances, float32 arithmetic
    with the per-dimension differences summed in index order."""
    p = points.astype(np.float32)
    q = keypoints.astype(np.float32)
    d2 = np.zeros((p.shape[0], q.shape[0]), dtype=np.float32)
    for d in range(p.shape[1]):
        diff = p[:, d:d + 1] - q[None, :, d]
        d2 = d2 + diff * diff
    idx = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return idx.astype(np.int64), np.take_along_axis(d2, idx, axis=1)


def knn3_mean_dist2(points):
    """simple_knn's distCUDA2 [REF scene/gaussian_model.py:340]: mean of the squared distances to the three nearest OTHER points
    (self excluded by index).  parity unpinned (simple_knn is absent): the published kernel's result, restated with a k-d tree."""
    import numpy as np
    from scipy.spatial import cKDTree
    p = np.asarray(points, np.float64)
    d, idx = cKDTree(p).query(p, k=min(4, len(p)))
    out = np.empty(len(p))
    for i in range(len(p)):                  # drop ONE occurrence of the point's own index (coincident points stay)
        row = [dd for dd, j in zip(np.atleast_1d(d[i]), np.atleast_1d(idx[i])) if j != i]
        if len(row) == len(np.atleast_1d(d[i])):
            row = row[:-1] if len(row) > 3 else row
        out[i] = np.sum(np.square(row[:3])) / 3.0 if len(row) >= 3 else np.inf
    return out
