"""The per-frame part of the reference's GaussianModel [REF scene/gaussian_model.py:32-321]:
parameters, activations and `forward(t, iteration, return_weights=False) -> (xyz_t, q_t, scale, opacity_t[, weights_xyz,
weights_r])` with its three stages, on the HIP kernels of this package.  The training bookkeeping the reference keeps on
the same class (optimizer groups per stage, learning-rate schedule, restore, densify / prune, keypoint growth) is in
training.py (TrainingMixin); PLY / checkpoint I/O in io_formats.py.

Two stage-2/3 quantities come from un-vendored CUDA dependencies of the reference:
  * `knn_idx` [N, nearest_num]  -- frnn kNN of Gaussians vs keypoints  [REF :110-125]
  * `raw_weights` [N, 2*nearest_num] -- output of the tcnn hash-grid weights model [REF :257]
They are the hot path's *inputs* (BASELINE.json north_star): `set_keypoint_weights` supplies them directly (what the
bench does).  When they are not supplied, `forward` computes them per frame as the reference does, with this
package's own `WeightsModel` / `knn_keypoints` (weights_ops.py; SURVEY.md section 8f rank 1, parity unpinned).
"""
from __future__ import annotations

import torch
from torch import nn

from .deform_ops import Activations, KeypointBlend
from .deformable_field import Deformable_Field
from .training import TrainingMixin
from .weights_ops import WeightsModel, dist_cuda2, knn_keypoints


class GaussianModel(TrainingMixin, nn.Module):
    def __init__(self, sh_degree: int, args):
        super().__init__()
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.args = args
        self.beta = args.beta
        self.d, self.w = args.d, args.w
        self.motion_feature_dim = args.feature_dim
        self.second_stage_iter = args.second_stage_iteration
        self.third_stage_iter = args.third_stage_iteration
        self.time_input_dim = None
        self.xyz_input_dim = None
        self.knn_idx = None
        self.raw_weights = None
        self.N_pcd_init = None                     # [REF scene/gaussian_model.py:70-71]: set by train.py / eval.py before create_from_pcd
        self.final_kpts_num = None                 #   to size a model that a checkpoint is then loaded into
        self.lifecycle_opacity = None
        self._last_delta = None
        self._last_blend = None
        self._training_init()

    def set_inputDim(self, time_input_dim, xyz_input_dim):   # [REF scene/gaussian_model.py:106-108]
        self.time_input_dim = time_input_dim
        self.xyz_input_dim = xyz_input_dim

    def create_from_pcd(self, pcd, spatial_lr_scale: float, device="cuda"):
        """[REF scene/gaussian_model.py:327-392]: Gaussians from a point cloud (`pcd.points` [N,3], `pcd.colors` [N,3] in 0..1):
        positions = the points, DC colour = RGB2SH(colour), isotropic scale = sqrt of the mean squared distance to the three
        nearest points (`distCUDA2`), identity rotation, opacity 0.1, motion features 1e-3 U(-1,1), keypoints all ones
        (`final_kpts_num` or `max_points` of them: placeholders until set_superKeypoints / a checkpoint), the hash-grid weights
        model.  `N_pcd_init` (set by train.py / eval.py when a checkpoint will be loaded) replaces the cloud by that many
        copies of its first point, exactly as the reference does."""
        import numpy as np
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(pcd.points)).float().to(device)
        col = (torch.tensor(np.asarray(pcd.colors)).float().to(device) - 0.5) / 0.28209479177387814     # RGB2SH [REF utils/sh_utils.py:114-115]
        if self.N_pcd_init is not None:
            pts, col = pts[0:1].repeat(self.N_pcd_init, 1), col[0:1].repeat(self.N_pcd_init, 1)
        self.N_pcd_init = n = pts.shape[0]
        features = torch.zeros((n, 3, (self.max_sh_degree + 1) ** 2), device=device)
        features[:, :3, 0] = col
        dist2 = torch.clamp_min(dist_cuda2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=device)
        rots[:, 0] = 1
        opac = torch.full((n, 1), 0.1, device=device)
        opac = torch.log(opac / (1 - opac))                                   # inverse_sigmoid [REF utils/general_utils.py:18-19]
        motion = 1e-3 * (2 * torch.rand((n, self.motion_feature_dim), device=device) - 1)
        K = self.final_kpts_num if self.final_kpts_num is not None else self.args.max_points
        degree = self.active_sh_degree            # untouched by the reference's create_from_pcd: 0 from __init__ when training
        self.create_from_tensors(pts, features[:, :, 0:1].transpose(1, 2).contiguous(), features[:, :, 1:].transpose(1, 2).contiguous(),
                                 scales, rots, opac, motion, torch.ones(K, 3, device=device),
                                 torch.ones(K, self.motion_feature_dim, device=device), with_weights_model=True)
        self.active_sh_degree = degree            # (oneupSHdegree raises it, train.py:81-83; eval.py:241 / train.py:52 set the maximum beforehand)
        return self

    # ---- construction from raw tensors ---------------------------------------------------------
    def create_from_tensors(self, xyz, features_dc, features_rest, scaling, rotation, opacity, motion_feature,
                            keypoints=None, keypoint_features=None, with_weights_model=False):
        self._xyz = nn.Parameter(xyz.clone().requires_grad_(True))
        self._features_dc = nn.Parameter(features_dc.clone().requires_grad_(True))
        self._features_rest = nn.Parameter(features_rest.clone().requires_grad_(True))
        self._scaling = nn.Parameter(scaling.clone().requires_grad_(True))
        self._rotation = nn.Parameter(rotation.clone().requires_grad_(True))
        self._opacity = nn.Parameter(opacity.clone().requires_grad_(True))
        self.motion_feature = nn.Parameter(motion_feature.clone().requires_grad_(True))
        self.max_radii2D = torch.zeros(xyz.shape[0], device=xyz.device)
        if self.args.step_opacity:               # [REF scene/gaussian_model.py:357-360]; read by opacity_type "explicit" only
            self.opacity_thres = nn.Parameter((-2 * torch.ones_like(opacity)).requires_grad_(True))
        delta_dim = 8 if self.args.step_opacity else 7
        in_dim = self.time_input_dim + self.xyz_input_dim + self.motion_feature_dim
        # per-Gaussian passes (> 2048 rows) run the split-fp16 kernels by default: fp32-grade results (same test bars as the exact-fp32
        # kernels, tests/test_gpu_deform.py) at twice their speed; args.mlp_precision = "fp32" selects the exact-fp32 matrix-core kernels
        self.df_model = Deformable_Field(in_dim, d=self.d, w=self.w, output_dim=delta_dim, split_xyz=False,
                                         precision=getattr(self.args, "mlp_precision", "fp32s")).to(xyz.device)
        if keypoints is not None:
            self.super_gaussians = nn.Parameter(keypoints.clone().requires_grad_(True))
            self.super_gaussians_feature = nn.Parameter(keypoint_features.clone().requires_grad_(True))
        self.weights_model = None
        if with_weights_model:                       # [REF scene/gaussian_model.py:370-392]
            self.weights_model = WeightsModel(2 * self.args.nearest_num, device=xyz.device)
        self.active_sh_degree = self.max_sh_degree
        return self

    def set_keypoint_weights(self, raw_weights, knn_idx):
        self.raw_weights, self.knn_idx = raw_weights, knn_idx

    @staticmethod
    def _state_key(*tensors):
        """Identity + version of every tensor a cached result depends on (optimizer kernels bump the versions of what they write)."""
        return tuple((id(t), t._version, t.data_ptr(), tuple(t.shape)) for t in tensors)

    @torch.no_grad()
    def get_nearest_mask(self, keepshape=False, defer=False):     # [REF scene/gaussian_model.py:110-125]
        """The reference searches the neighbours on every forward; their inputs do not depend on the frame time, so the result
        is kept until one of them changes (every training step -- but not between the frames of an evaluation, nor between the
        views of a `--batch`).  Same kernel, same result.
        `defer=True` (forward's own call): a large search is enqueued on a side stream so that it runs beside the weights model's
        encode (the search is vector-ALU work, the encode waits on random table lines); `_knn_join()` orders the calling stream
        behind it before the indices are consumed."""
        a = self.args
        key = self._state_key(self._xyz, self.motion_feature, self.super_gaussians, self.super_gaussians_feature)
        c = getattr(self, "_knn_cache", None)
        if c is None or c[0] != key:
            self._knn_join()
            wm = getattr(self, "weights_model", None)          # the spatial order its encode already keeps (a performance hint only)
            n = self._xyz.shape[0]
            order = wm.spatial_order(self._xyz.detach(), age=False) if wm is not None and n > 4096 else None
            run = lambda: knn_keypoints(self._xyz, self.super_gaussians, a.nearest_num, self.motion_feature,
                                        self.super_gaussians_feature, getattr(a, "feature_amplify", 5.0),
                                        getattr(a, "knn_type", "hybird"), order=order)
            if defer and self._xyz.is_cuda and n >= (1 << 17):
                dev = self._xyz.device
                side = getattr(self, "_knn_stream", None)
                if side is None or side.device != dev:
                    side = self._knn_stream = torch.cuda.Stream(device=dev)
                cur = torch.cuda.current_stream(dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    nearest = run()
                nearest.record_stream(cur)
                if getattr(nearest, "_gp_idx16", None) is not None:       # (the packed copy the search kernel wrote beside it)
                    nearest._gp_idx16[1].record_stream(cur)
                self._knn_pending = side
            else:
                nearest = run()
            c = self._knn_cache = (key, nearest)
        if not defer:
            self._knn_join()
        nearest = c[1]
        self.nearest_mask = nearest if keepshape else nearest.view([-1])
        return self.nearest_mask

    def _knn_join(self):
        side = getattr(self, "_knn_pending", None)
        if side is not None:
            torch.cuda.current_stream(side.device).wait_stream(side)
            self._knn_pending = None

    def _keypoint_raw_weights(self):
        """`weights_model(self.get_xyz.detach())` [REF scene/gaussian_model.py:257], evaluated once per parameter state: positions
        and the model's parameters do not depend on the frame time.  Without autograd (evaluation) the result is kept until
        a version changes.  With autograd it can only be shared inside ONE backward's graph: a harness that sums several views
        into one backward opens such a scope (`keypoint_weights_scope`; the weights model then runs once, and once backward
        with the summed upstream gradient, instead of once per view)."""
        wm = self.weights_model
        key = self._state_key(self._xyz, wm.params)
        grad = torch.is_grad_enabled() and wm.params.requires_grad
        c = getattr(self, "_rw_cache", None)
        scope = getattr(self, "_rw_scope", None)
        if c is not None and c[0] == key and c[1] == (scope if grad else None) and (not grad or scope is not None):
            return c[2]
        out = wm(self.get_xyz.detach())
        self._rw_cache = (key, scope if grad else None, out) if (not grad or scope is not None) else None
        return out

    def keypoint_weights_scope(self, token):
        """Open (token not None) / close (None) the scope inside which autograd-tracked keypoint weights may be shared."""
        self._rw_scope = token
        if token is None and getattr(self, "_rw_cache", None) is not None and self._rw_cache[1] is not None:
            self._rw_cache = None

    # ---- accessors [REF scene/gaussian_model.py:138-172] -----------------------------------------
    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_scaling(self):
        return Activations.apply(self._scaling, self._opacity, None, 0, 1.0)[0]

    @property
    def get_opacity(self):
        return Activations.apply(self._scaling, self._opacity, None, 0, 1.0)[1]

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    rotation_activation = staticmethod(torch.nn.functional.normalize)      # [REF scene/gaussian_model.py:49]

    def get_rotation_(self, delta):
        """normalize(delta (x) _rotation), Hamilton product in (w, x, y, z) with the RAW stored rotation
        [REF scene/gaussian_model.py:314-315; eval.py:141 calls it on blended keypoint rotations].  Runs the per-Gaussian form
        of the blend kernel (gp_blend_forward, nn = 0), the same arithmetic as forward()'s composition."""
        n = self._rotation.shape[0]
        d = torch.zeros(n, 7, dtype=torch.float32, device=self._rotation.device)
        d[:, 3:7] = delta
        return KeypointBlend.apply(d, None, None, self._xyz.detach(), self._rotation, False)[1]

    @property
    def all_xyz_motion(self):
        """Per-Gaussian displacement of the LAST forward, xyz_t - xyz.  [REF eval.py:60] reads this attribute in
        project_trajectory; no code in the reference assigns it, so the definition is this package's."""
        last = getattr(self, "_last_xyz_t", None)
        return None if last is None else last - self._xyz.detach()

    def get_covariance(self, scaling_modifier=1):
        """Packed world-space covariance [N,6] (xx,xy,xz,yy,yz,zz) = (R S)(R S)^T, with the *raw*
        `_rotation` normalised inside, as the reference's python fallback does
        [REF scene/gaussian_model.py:35-39,320-321; utils/general_utils.py:64-110]."""
        q = torch.nn.functional.normalize(self._rotation)
        r, x, y, z = q.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
        L = R * (scaling_modifier * self.get_scaling)[:, None, :]
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=-1)

    @property
    def get_superGaussians(self):
        return self.super_gaussians

    def get_loss(self, iteration):                 # [REF scene/gaussian_model.py:174-178]
        if iteration < self.args.jointly_iteration:
            return 0.0
        feat = self.super_gaussians_feature if iteration > self.second_stage_iter else self.motion_feature
        return 1.0e-5 * torch.mean(torch.abs(feat))

    # ---- side outputs of forward (lazy: none of them is on the train / eval hot path) -----------------------------
    @property
    def kpts_xyz_motion(self):                     # [REF :269] the MLP's translation output, before the blend
        d = self._last_delta
        return None if d is None else d[:, 0:3]

    @property
    def kpts_rotation_motion(self):                # [REF :266-270] the MLP's rotation output (normalised if norm_rotation)
        d = self._last_delta
        if d is None:
            return None
        q = d[:, 3:7]
        return torch.nn.functional.normalize(q) if self.args.norm_rotation else q

    def dense_weights(self):
        """The reference's dense [N,K] blend weights (`fill_nearest`: two softmaxes scattered at the kNN indices,
        [REF scene/gaussian_model.py:214-229]) from the sparse form the kernels use.  1 GB each at N = 1M, K = 250: built
        only on request (`return_weights=True`, `weights_sum`)."""
        raw_w, idx, K = self._last_blend
        nn_ = idx.shape[1]
        wx = torch.softmax(raw_w[:, :nn_], dim=-1)
        wr = torch.softmax(raw_w[:, nn_:2 * nn_], dim=-1)
        fill = torch.zeros(raw_w.shape[0], K, dtype=raw_w.dtype, device=raw_w.device)
        return torch.scatter(fill, -1, idx, wx), torch.scatter(fill, -1, idx, wr)

    @property
    def weights_sum(self):                         # [REF :263] |weights_xyz| + |weights_r|, dense
        if self._last_blend is None:
            return None
        wx, wr = self.dense_weights()
        return torch.abs(wx) + torch.abs(wr)

    @torch.no_grad()
    def get_teach_motion(self, t, delta_xyz):
        """Stage-1 "teacher" displacement of every Gaussian vs the blended one [REF scene/gaussian_model.py:306-312]."""
        xyz_freq, time_freq = int(self.xyz_input_dim / 6), self.time_input_dim // 2
        teach = self.df_model.forward_fused(self.motion_feature.detach(), self._xyz.detach(), t, xyz_freq, time_freq)
        self.add_desification_stats_motion(delta_xyz - teach[:, 0:3])

    # ---- the hot path [REF scene/gaussian_model.py:231-304] ---------------------------------------
    def stage_transitions(self, iteration):
        """The stage hooks the reference runs inside forward [REF scene/gaussian_model.py:246-250]: at second_stage_iter + 1 the
        keypoints are initialised (k-means of the motion features) and the stage-2 optimizer is installed, at third_stage_iter + 1
        the stage-3 optimizer.  They need a training set-up; a harness may call this ahead of the forward (TrainStep does, so
        that the optimizer / gradient bucket it works with is the one the step's backward fills)."""
        if self.training_args is None:
            return
        if iteration == self.third_stage_iter + 1 and not self.third_stage:
            self.training3stage_setup()
        if iteration == self.second_stage_iter + 1 and not self.second_stage:
            self.set_superKeypoints()
            self.training2stage_setup()

    raw_activations_ok = True      # forward(raw_activations=True) exists (renderer.render asks before it passes the keyword)

    def forward(self, t, iteration, return_weights=False, reference_rng=None, raw_activations=False):
        """`reference_rng` (default: the model's `reference_rng` attribute, False): draw `torch.randn_like` on EVERY stage-1 / stage-2/3
        pass, also once the noise's scale has decayed to zero, as the reference does [REF scene/gaussian_model.py:241,254] -- a seeded
        training run then consumes the random stream exactly like the reference's (matters for bit-reproducible training
        comparisons, not for render parity).  Off: the draw is skipped once its factor is zero (one launch less per step); the fused
        train step is not taken with it on."""
        # `raw_activations` (extension, passes without autograd: renderer.render): where no lifecycle term applies, return the RAW
        # _scaling / _opacity in the places of the activated tensors and set `_forward_raw` -- the rasterizer applies exp / sigmoid
        # itself (GaussianRasterizationSettings.raw_activations: same values, one launch less per frame)
        self._forward_raw = False
        if torch.is_tensor(iteration):
            iteration = iteration.item()
        a = self.args
        if reference_rng is None:
            reference_rng = bool(getattr(self, "reference_rng", False))
        xyz_freq, time_freq = int(self.xyz_input_dim / 6), self.time_input_dim // 2
        if iteration < a.jointly_iteration:          # warm-up: static Gaussians
            s, o = Activations.apply(self._scaling, self._opacity, None, 0, 1.0)
            self._last_xyz_t = self._xyz.detach()       # (what the view was rendered at: the factorised SH exchange's view directions)
            return self._xyz, self.get_rotation, s, o
        t_dev = t.to(self._xyz.device, torch.float32).reshape(-1)[:1]
        self.stage_transitions(iteration)
        self._last_blend = None
        if iteration <= self.second_stage_iter:      # stage 1: MLP over all N Gaussians, xyz detached
            noise = getattr(a, "xyz_noise_iteration", 0)
            xyz_in = self._xyz.detach()
            if noise and (iteration < noise or reference_rng):
                xyz_in = xyz_in + torch.randn_like(xyz_in) * 0.1 * (1 - min(1, iteration / noise))
            delta = self.df_model.forward_fused(self.motion_feature, xyz_in, t_dev, xyz_freq, time_freq)
            self._last_delta = delta.detach()        # (side outputs only: holding the graph alive would pin its AccumulateGrad nodes)
            xyz_t, q_t = KeypointBlend.apply(delta, None, None, self._xyz, self._rotation, a.norm_rotation)
        else:                                        # stage 2/3: MLP over K keypoints + sparse blend
            noise = getattr(a, "xyz_noise_iteration", 0)
            kp = self.super_gaussians
            if noise and ((iteration - self.second_stage_iter) < noise or reference_rng):
                kp = kp + torch.randn_like(kp) * 0.1 * (1 - min(1, (iteration - self.second_stage_iter) / noise))
            raw_weights, knn_idx = self.raw_weights, self.knn_idx
            if raw_weights is None or knn_idx is None:
                if getattr(self, "weights_model", None) is None:
                    raise RuntimeError("stage 2/3 needs set_keypoint_weights(raw_weights, knn_idx) or a weights_model "
                                       "(create_from_tensors(..., with_weights_model=True))")
                knn_idx = self.get_nearest_mask(keepshape=True, defer=True)      # [REF :260] (enqueued first: runs beside the encode)
                raw_weights = self._keypoint_raw_weights()                       # [REF :257]
                self._knn_join()
            delta = self.df_model.forward_fused(self.super_gaussians_feature, kp, t_dev, xyz_freq, time_freq)
            self._last_delta = delta.detach()
            self._last_blend = (raw_weights.detach(), knn_idx, self.super_gaussians.shape[0])
            xyz_t, q_t = KeypointBlend.apply(delta, raw_weights, knn_idx, self._xyz, self._rotation, a.norm_rotation)
            if getattr(a, "densify_from_teaching", False) and self.second_stage:                # [REF :274-283]
                off = self.second_stage_iter
                if a.adaptive_from_iter + off <= iteration < a.adaptive_end_iter + off and self._kpts_room() > 0:
                    self.get_teach_motion(t_dev, (xyz_t - self._xyz).detach())
                    if iteration % a.adaptive_interval == 0:
                        self.sync_teacher_stats()          # (view-parallel: the maximum over every rank's views)
                        self.get_new_kpts(self.xyz_motion_accum_max.squeeze(-1) >= a.teaching_threshold)
        self.lifecycle_opacity = None
        self._last_xyz_t = xyz_t.detach()
        weights = ()
        if return_weights and iteration > self.second_stage_iter:
            weights = self.dense_weights()
        if a.step_opacity and iteration > a.step_opacity_iteration:
            if a.opacity_type == "explicit":
                # per-Gaussian birth time: sigmoid((t - opacity_thres) / beta) [REF :50, :294-295] (the reference also runs the MLP
                # here and drops its output; nothing reads it, so it is not run)
                gate = t_dev.reshape(1, 1) - self.opacity_thres
                s, o = Activations.apply(self._scaling, self._opacity, gate, 0, self.beta)
            else:
                # second MLP pass over all N Gaussians with the per-Gaussian motion feature [REF :291-298]
                delta2 = self.df_model.forward_fused(self.motion_feature, self._xyz, t_dev, xyz_freq, time_freq)
                s, o = Activations.apply(self._scaling, self._opacity, delta2, 7, self.beta)
            self.lifecycle_opacity = o
            if weights:                              # the reference returns the PLAIN opacity beside the weights [REF :299-300]
                return (xyz_t, q_t, s, self.get_opacity) + tuple(weights)
            return xyz_t, q_t, s, o
        if raw_activations and not torch.is_grad_enabled():
            self._forward_raw = True
            return (xyz_t, q_t, self._scaling, self._opacity) + tuple(weights)
        s, o = Activations.apply(self._scaling, self._opacity, None, 0, 1.0)
        return (xyz_t, q_t, s, o) + tuple(weights)
