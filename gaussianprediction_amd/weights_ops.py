"""The stage-2/3 "keypoint weights" producers, mirroring the reference's two call sites:

* `weights_model = tcnn.NetworkWithInputEncoding(...)` [REF scene/gaussian_model.py:370-392], called once per frame as
  `weights_model(self.get_xyz.detach())` [REF :257]  ->  `WeightsModel`
* `get_nearest_mask` = `frnn.frnn_grid_points(...)` [REF scene/gaussian_model.py:110-125]  ->  `knn_keypoints`

Hash-grid encoding + the 64-wide bias-free MLP behind it are ONE fused HIP kernel each way (gp_weights_forward / _backward,
csrc/weights_kernels.hip: encode per level, MLP on v_mfma_f32_32x32x2_f32); the kNN is a HIP kernel too.  The stand-alone
encoding kernels (gp_hashgrid_forward / _backward, `_HashGridEncode`) stay exported for callers that want the features only.
tinycudann / frnn are absent from the reference tree: parity unpinned (oracle/weights_oracle.py states the algorithm).
"""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch import nn

from . import _lib

MLP_FLOATS = 64 * 64 + 64 * 64 + 16 * 64


class HashGridConfigC(C.Structure):   # == gp_hashgrid_config
    _fields_ = [("n_levels", C.c_int32), ("n_features_per_level", C.c_int32), ("log2_hashmap_size", C.c_int32),
                ("base_resolution", C.c_int32), ("per_level_scale", C.c_float)]


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: HIP kernels only (no CPU fallback)")


def morton_order(xyz):
    """int32 permutation of the points along a 30-bit Morton curve (plumbing: torch ops; refreshed rarely)."""
    x = xyz.detach().to(torch.float32)
    lo, hi = x.min(0).values, x.max(0).values
    q = ((x - lo) / (hi - lo).clamp_min(1e-12) * 1023.0).clamp(0, 1023).to(torch.int64)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(code).to(torch.int32)


class _HashGridEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, table, cfg, perm=None):
        _need_cuda(xyz, "hash grid")
        x = xyz.detach().to(torch.float32).contiguous()
        t = table.detach().contiguous()
        n = x.shape[0]
        out = torch.empty(n, cfg.n_levels * 4, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().gp_hashgrid_forward(C.byref(cfg), C.c_int64(n), _lib.ptr(x), _lib.ptr(perm), _lib.ptr(t),
                                                      _lib.ptr(out), _lib.stream_ptr(x.device)), "gp_hashgrid_forward")
        ctx.save_for_backward(x)
        ctx.perm = perm
        ctx.cfg, ctx.table_shape = cfg, table.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        dtable = torch.zeros(ctx.table_shape, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().gp_hashgrid_backward(C.byref(ctx.cfg), C.c_int64(x.shape[0]), _lib.ptr(x), _lib.ptr(ctx.perm),
                                                       _lib.ptr(g), _lib.ptr(dtable), _lib.stream_ptr(x.device)),
                       "gp_hashgrid_backward")
        return None, dtable, None, None


class _WeightsModelFused(torch.autograd.Function):
    """encode + MLP in one HIP kernel each way (gp_weights_forward / gp_weights_backward)."""

    @staticmethod
    def forward(ctx, xyz, params, cfg, n_out, perm):
        _need_cuda(xyz, "weights model")
        x = xyz.detach().to(torch.float32).contiguous()
        p = params.detach().contiguous()
        n = x.shape[0]
        need = params.requires_grad
        out = torch.empty(n, n_out, device=x.device)
        feat = torch.empty(n, 64, device=x.device) if need else None
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().gp_weights_forward(C.byref(cfg), C.c_int64(n), _lib.ptr(x), _lib.ptr(perm), _lib.ptr(p),
                                                     C.c_int32(n_out), _lib.ptr(out), _lib.ptr(feat), _lib.stream_ptr(x.device)),
                       "gp_weights_forward")
        if need:
            ctx.save_for_backward(x, p, feat)
            ctx.cfg, ctx.n_out, ctx.perm, ctx.leaf = cfg, n_out, perm, params
        return out

    @staticmethod
    def backward(ctx, g):
        from . import grad_sink
        x, p, feat = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        sink = grad_sink.sink_of(ctx.leaf)
        if sink is not None and grad_sink.take_stale(sink):
            sink.zero_()
        dparams = sink if sink is not None else torch.zeros_like(p)
        alloc = _lib.TorchAllocator(x.device)
        with _lib.on_device(x.device):
            rc = _lib.lib().gp_weights_backward(C.byref(ctx.cfg), C.c_int64(x.shape[0]), _lib.ptr(x), _lib.ptr(ctx.perm), _lib.ptr(p),
                                                C.c_int32(ctx.n_out), _lib.ptr(feat), _lib.ptr(g), _lib.ptr(dparams), alloc.cb, None,
                                                _lib.stream_ptr(x.device))
        err = alloc.error
        alloc.release()
        if err is not None:
            raise err
        _lib.check(rc, "gp_weights_backward")
        if sink is not None:
            grad_sink.notify(ctx.leaf)
            return None, None, None, None, None
        return None, dparams, None, None, None


class WeightsModel(nn.Module):
    """Drop-in for the reference's `tcnn.NetworkWithInputEncoding(n_input_dims=3, n_output_dims=2*nearest_num, Grid/Hash
    encoding, FullyFusedMLP 64 x 2 hidden, ReLU)`.  One flat parameter `params` (as tcnn exposes it): the three
    row-major weight matrices [64,64], [64,64], [16,64] (output padded to 16, no biases) followed by the hash table
    [entries, 4]."""

    def __init__(self, n_output_dims, n_levels=16, n_features_per_level=4, log2_hashmap_size=19, base_resolution=16,
                 per_level_scale=None, seed=1337, device="cuda"):
        super().__init__()
        if n_output_dims > 16:
            raise RuntimeError("WeightsModel: n_output_dims must be <= 16")
        if n_levels != 16 or n_features_per_level != 4:
            # the reference's only configuration [REF scene/gaussian_model.py:370-392]; the fused encode + MLP kernel is built for it
            raise RuntimeError("WeightsModel: built for n_levels == 16, n_features_per_level == 4 (the reference's hash-grid "
                               "configuration); other grids are not implemented")
        if per_level_scale is None:
            per_level_scale = math.exp(math.log(2048 / base_resolution) / (n_levels - 1))   # [REF :372]
        self.n_output_dims = n_output_dims
        self.cfg = HashGridConfigC(n_levels, n_features_per_level, log2_hashmap_size, base_resolution, per_level_scale)
        entries = int(_lib.lib().gp_hashgrid_table_entries(C.byref(self.cfg)))
        if entries < 0:
            raise RuntimeError(_lib.lib().gp_last_error().decode(errors='replace'))
        self.table_entries = entries
        gen = torch.Generator().manual_seed(seed)

        def xavier(o, i):
            a = math.sqrt(6.0 / (i + o))
            return (torch.rand(o, i, generator=gen) * 2 - 1) * a

        mlp = torch.cat([xavier(64, 64).reshape(-1), xavier(64, 64).reshape(-1), xavier(16, 64).reshape(-1)])
        grid = (torch.rand(entries * 4, generator=gen) * 2 - 1) * 1e-4     # tcnn's grid initialisation range
        self.params = nn.Parameter(torch.cat([mlp, grid]).to(device))
        self._perm, self._perm_age = None, 0
        self.perm_refresh = 200          # frames between refreshes of the spatial order (Gaussians move slowly)

    def spatial_order(self, xyz, age=True):
        if self._perm is None or self._perm.shape[0] != xyz.shape[0] or self._perm_age >= self.perm_refresh:
            self._perm, self._perm_age = morton_order(xyz), 0
        self._perm_age += 1 if age else 0
        return self._perm

    def forward(self, xyz):
        perm = self.spatial_order(xyz) if xyz.shape[0] > 4096 else None
        return _WeightsModelFused.apply(xyz, self.params, self.cfg, self.n_output_dims, perm)


def knn_keypoints(xyz, kp_xyz, nearest_num, feat=None, kp_feat=None, feature_amplify=5.0, knn_type="hybird",
                  return_dist=False, order=None):
    """[N, nearest_num] int64 indices of the nearest keypoints, ascending distance [REF scene/gaussian_model.py:110-125].
    `order` (int32 permutation of the points, `morton_order`) makes the wavefronts spatially coherent: same result, ~2.5x faster."""
    _need_cuda(xyz, "knn")
    if knn_type not in ("3D", "hybird"):
        raise RuntimeError('Type error! Should be "3D" or "hybird"')       # [REF :119-121]
    hybrid = knn_type == "hybird"
    x = xyz.detach().to(torch.float32).contiguous()
    k = kp_xyz.detach().to(torch.float32).contiguous()
    f = feat.detach().to(torch.float32).contiguous() if hybrid else None
    kf = kp_feat.detach().to(torch.float32).contiguous() if hybrid else None
    n, K = x.shape[0], k.shape[0]
    idx = torch.empty(n, nearest_num, dtype=torch.int64, device=x.device)
    idx16 = torch.empty(n, nearest_num, dtype=torch.int16, device=x.device) if K < 65536 else None   # the blend kernels' packed copy
    d2 = torch.empty(n, nearest_num, device=x.device) if return_dist else None
    if order is not None and (order.dtype != torch.int32 or order.shape[0] != n or not order.is_contiguous() or order.device != x.device):
        raise RuntimeError("knn: order must be a contiguous int32 permutation of the points, on their device")
    with _lib.on_device(x.device):
        rc = _lib.lib().gp_knn_keypoints(C.c_int64(n), _lib.ptr(x), _lib.ptr(f), C.c_int32(f.shape[1] if hybrid else 0),
                                         C.c_float(feature_amplify), C.c_int64(K), _lib.ptr(k), _lib.ptr(kf),
                                         C.c_int32(nearest_num), _lib.ptr(order), _lib.ptr(idx), _lib.ptr(d2), _lib.ptr(idx16),
                                         _lib.stream_ptr(x.device))
        _lib.check(rc, "gp_knn_keypoints")
    if idx16 is not None:
        idx._gp_idx16 = ((idx._version, idx.data_ptr(), tuple(idx.shape)), idx16)      # (deform_ops.packed_idx16 finds it here)
    return (idx, d2) if return_dist else idx


def dist_cuda2(points):
    """Mean squared distance of every point to its three nearest other points: `simple_knn._C.distCUDA2`
    [REF scene/gaussian_model.py:23, 340] (un-vendored CUDA dependency; HIP kernel gp_knn3_mean_dist2, exact brute force)."""
    _need_cuda(points, "distCUDA2")
    x = points.detach().to(torch.float32).contiguous()
    out = torch.empty(x.shape[0], device=x.device)
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().gp_knn3_mean_dist2(C.c_int64(x.shape[0]), _lib.ptr(x), _lib.ptr(out), _lib.stream_ptr(x.device)),
                   "gp_knn3_mean_dist2")
    return out
