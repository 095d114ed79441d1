"""Training bookkeeping of the reference's GaussianModel, on this package's flat gradient bucket + fused Adam:

    training_setup / training2stage_setup / training3stage_setup      [REF scene/gaussian_model.py:394-472]
    update_learning_rate                                              [REF :474-491]
    restore                                                           [REF :96-104]
    add_densification_stats, densify, prune, reset_opacity            [REF :526-530, 645-718, 745-760]
    densify_kpts, get_new_kpts, densification_motion_postfix,
    get_teach_motion, add_desification_stats_motion                   [REF :196-212, 306-312, 612-630, 720-744, 762-773]
    set_superKeypoints (k-means keypoint initialisation)              [REF :127-136]

so that `train.py`'s calls on `gaussians` (`update_learning_rate`, `optimizer.step()`, `optimizer.zero_grad(set_to_none=True)`,
`densify`, `reset_opacity`, `prune`, `densify_kpts`, `optimizer.state_dict()`, `restore`) resolve on this model.

The reference edits `torch.optim.Adam`'s per-parameter state in place (`cat_tensors_to_optimizer`, `_prune_optimizer`,
`replace_tensor_to_optimizer`).  Here every optimized parameter's `.grad` is a view into ONE flat buffer (the RCCL
all-reduce operand, dist.FlatGradBucket) and Adam is one multi-tensor HIP launch over it (loss_ops.FusedAdam), so a change
of the per-Gaussian row count rebuilds bucket + optimizer together: moments of surviving rows are carried over, appended
rows start at zero (`torch.zeros_like(extension_tensor)` in the reference), step count and learning rates are preserved.
This is bookkeeping that runs every few hundred iterations: torch tensor ops (plumbing), except furthest-point sampling,
which is a kernel (gp_furthest_point_sampling; the reference's is pointops' CUDA kernel, utils/fps.py:71-88).
Nothing here has a CPU implementation of a kernel: the -m "not gpu" tests of this bookkeeping run it on CPU tensors and install
their own checkers for the two kernels it reaches (the Adam step, furthest-point sampling; tests/host_checkers.py).
"""
from __future__ import annotations

import ctypes as C
import math
from types import SimpleNamespace

import torch
from torch import nn

from . import _lib
from .dist import FlatGradBucket

# (optimizer group name, model attribute) of the per-Gaussian parameters [REF scene/gaussian_model.py:434-451, 632-655]
PER_GAUSSIAN = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
                ("scaling", "_scaling"), ("rotation", "_rotation"), ("motion_feature", "motion_feature"),
                ("opacity_thres", "opacity_thres"))


def default_training_args(**over):
    """The optimisation defaults of the reference [REF arguments/__init__.py:72-96]."""
    a = dict(iterations=30_000, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
             position_lr_max_steps=30_000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
             mfeature_lr=0.0008, mfeature_lr_final=0.00008, kpts_lr=0.0008, kpts_lr_final=0.00008, hash_lr=0.005,
             hash_lr_final=0.00005, mlp_lr=0.0008, percent_dense=0.01, lambda_dssim=0.2, densification_interval=100,
             opacity_reset_interval=3000, densify_from_iter=500, densify_until_iter=15_000, densify_grad_threshold=0.0002)
    a.update(over)
    return SimpleNamespace(**a)


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation from lr_init (step 0) to lr_final (step max_steps), optionally eased in over the first
    lr_delay_steps by lr_delay_mult + (1 - lr_delay_mult) sin(pi/2 * step / lr_delay_steps)
    [REF utils/general_utils.py:29-62]."""
    def rate(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        ease = 1.0
        if lr_delay_steps > 0:
            ease = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        t = min(max(step / max_steps, 0.0), 1.0)
        return ease * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return rate


def build_rotation(r):
    """[REF utils/general_utils.py:78-99] rotation matrices of (w, x, y, z) quaternions, normalised inside."""
    q = r / torch.sqrt((r * r).sum(-1, keepdim=True))
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


def furthest_point_sampling(xyz: torch.Tensor, m: int) -> torch.Tensor:
    """Indices (int64, [m]) of an iterative furthest-point sample of `xyz` [n,3] that starts at point 0 -- the contract of
    pointops' `furthestsampling` for one batch [REF utils/fps.py:71-88].  HIP kernel only; `host_fps` is the seam through
    which the CPU tests of the keypoint-growth bookkeeping supply their own restatement (tests/host_checkers.py)."""
    n = xyz.shape[0]
    m = int(min(m, n))
    if m <= 0:
        return torch.empty(0, dtype=torch.int64, device=xyz.device)
    x = xyz.detach().to(torch.float32).contiguous()
    if x.is_cuda:
        idx = torch.empty(m, dtype=torch.int32, device=x.device)
        tmp = torch.empty(n, dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().gp_furthest_point_sampling(C.c_int64(n), _lib.ptr(x), C.c_int64(m), _lib.ptr(idx), _lib.ptr(tmp),
                                                             _lib.stream_ptr(x.device)), "gp_furthest_point_sampling")
        return idx.to(torch.int64)
    if host_fps is None:
        raise RuntimeError("furthest_point_sampling: HIP kernel only (no CPU fallback)")
    return host_fps(x, m)


host_fps = None


def nearest_index(query: torch.Tensor, base: torch.Tensor, chunk: int = 256) -> torch.Tensor:
    """Index of the nearest `base` point for every `query` point (pytorch3d knn_points K=1 in the reference, :207)."""
    out = torch.empty(query.shape[0], dtype=torch.int64, device=query.device)
    for s in range(0, query.shape[0], chunk):
        out[s:s + chunk] = torch.cdist(query[s:s + chunk], base).argmin(dim=1)
    return out


def kmeans(features: torch.Tensor, K: int, iters: int = 20, seed: int = 0, chunk: int = 65536):
    """Lloyd's k-means (cluster ids, centres), centres initialised from a seeded random subset.  Stands in for
    kmeans_pytorch.kmeans (absent, un-vendored: parity unpinned) in `feature_kmeans` [REF utils/visualizer_utils.py:84-93]."""
    n = features.shape[0]
    g = torch.Generator().manual_seed(seed)
    centres = features[torch.randperm(n, generator=g)[:K].to(features.device)].clone()
    ids = torch.zeros(n, dtype=torch.int64, device=features.device)
    for _ in range(iters):
        for s in range(0, n, chunk):
            ids[s:s + chunk] = torch.cdist(features[s:s + chunk], centres).argmin(dim=1)
        sums = torch.zeros_like(centres).index_add_(0, ids, features)
        cnt = torch.zeros(K, device=features.device).index_add_(0, ids, torch.ones(n, device=features.device))
        new = torch.where(cnt[:, None] > 0, sums / cnt[:, None].clamp_min(1), centres)
        if torch.allclose(new, centres, rtol=0, atol=1e-7):
            centres = new
            break
        centres = new
    return ids, centres


class TrainingMixin:
    """See the module docstring.  Mixed into GaussianModel."""

    No_prune_and_densify = ("s_xyz", "s_motion_feature", "s_weights", "weight_feature", "weight_mlp")   # [REF :88]

    def _training_init(self):
        self.optimizer = None
        self.bucket = None
        self.optimizer_epoch = 0           # bumped whenever bucket + optimizer are rebuilt (harnesses re-attach their hooks)
        self.spatial_lr_scale = 1.0
        self.percent_dense = 0.01
        self.second_stage = False
        self.third_stage = False
        self.training_args = None
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.xyz_gradient_accum_max = torch.empty(0)
        self.denom = torch.empty(0)
        self._vp = None                    # (group, rank, world) of a view-parallel harness: set_view_parallel
        self._surgery_no = 0               # densify calls so far: the seed of the split's random draws under view parallelism
        self.surgery_seed = 2024           # [REF train.py:306 seeds everything with it]
        self.new_kpts_init()

    # ---- view-parallel consistency (every rank must perform the SAME surgery) ---------------------------
    def set_view_parallel(self, group, rank, world):
        """Called by a view-parallel harness (TrainStep under a process group).  From then on the model keeps its replicas
        identical through every operation that changes the parameter SET, not only their values:
          * the k-means keypoints (float atomics in index_add_: not bit-reproducible) are rank 0's, broadcast;
          * the teacher statistic that proposes keypoints is the MAX over the ranks' views (as the views of one `--batch`
            accumulate into it in the reference);
          * densify_and_split draws its samples from a generator seeded identically on every rank;
          * after every surgery N, K and an exact checksum of the parameters are compared across the ranks (RuntimeError on
            disagreement, before the next collective could silently mix rows of different Gaussians)."""
        self._vp = (group, int(rank), int(world))

    def _vp_group(self):
        vp = getattr(self, "_vp", None)
        if vp is None:
            return None
        import torch.distributed as tdist
        return vp if (tdist.is_available() and tdist.is_initialized()) else None

    def _vp_src(self, which=0):
        """Global rank of the group's first (0) / last (-1) member: the `src` of a broadcast."""
        import torch.distributed as tdist
        group, _, world = self._vp
        local = 0 if which == 0 else world - 1
        return tdist.get_global_rank(group, local) if group is not None else local

    def _vp_broadcast(self, tensors, which=0):
        vp = self._vp_group()
        if vp is None:
            return
        import torch.distributed as tdist
        for t in tensors:
            tdist.broadcast(t, src=self._vp_src(which), group=vp[0])

    def _surgery_generator(self):
        """The random stream of the next densify_and_split.  View-parallel: a generator every rank seeds with the same number
        (surgery_seed, how many densify calls there have been); single process: None = torch's global stream, as in the reference."""
        if self._vp_group() is None and not getattr(self, "deterministic_surgery", False):
            return None
        g = torch.Generator(device=self.get_xyz.device)
        g.manual_seed(int(self.surgery_seed) * 1_000_003 + int(self._surgery_no))
        return g

    @torch.no_grad()
    def assert_ranks_agree(self, what=""):
        """View-parallel only: every rank holds the same parameter set -- N, K and an order-independent EXACT checksum (the
        parameters' bit patterns summed as integers) of every optimized tensor agree.  One small all-reduce and one host read,
        after surgery only."""
        vp = self._vp_group()
        if vp is None:
            return
        import torch.distributed as tdist
        dev = self.get_xyz.device
        vals = [self.get_xyz.shape[0], self.super_gaussians.shape[0] if hasattr(self, "super_gaussians") else 0]
        sums = []
        for p in (self.bucket.params if self.bucket is not None else list(self.parameters())):
            bits = p.detach().contiguous().view(torch.int32).to(torch.int64)
            sums.append(bits.sum())
        v = torch.cat([torch.tensor(vals, dtype=torch.int64, device=dev), torch.stack(sums) if sums else torch.zeros(0, dtype=torch.int64, device=dev)])
        both = torch.cat([v, -v])
        tdist.all_reduce(both, op=tdist.ReduceOp.MAX, group=vp[0])
        n = v.numel()
        spread = (both[:n] + both[n:]).cpu()              # max - min per entry
        if bool((spread != 0).any()):
            bad = [i for i in range(n) if int(spread[i]) != 0]
            raise RuntimeError(f"view-parallel replicas diverged ({what}): entries {bad} of [N, K, checksum per optimized tensor] differ "
                               f"across ranks (N spread {int(spread[0])}, K spread {int(spread[1])})")

    # ---- optimizer construction ----------------------------------------------------------------------------------
    def _per_gaussian(self):
        return {name: getattr(self, attr) for name, attr in PER_GAUSSIAN if isinstance(getattr(self, attr, None), nn.Parameter)}

    def _install_optimizer(self, groups):
        optimized = {id(p) for g in groups for p in g["params"]}
        for p in self.parameters():             # the reference computes (and discards) the other gradients; skipping them
            p.requires_grad_(id(p) in optimized)   # changes no result
        params = [p for g in groups for p in g["params"]]
        shard = getattr(self, "optimizer_shard", None)       # (rank, world): set by a view-parallel harness (dist.ShardedExchange)
        from .loss_ops import FusedAdam                      # Adam(lr=0.0, eps=1e-15) [REF :472] as one multi-tensor launch
        if shard is not None:
            # (bucket_small_numel: tensors below it share one region / one collective; tests lower it to reach the per-region paths.)
            # Two regions for the per-Gaussian tensors -- the SH coefficients, which the rasterizer backward finishes first, and
            # the geometry (xyz, scaling, rotation, opacity), final after the blend / activation backward -- and the tail of small
            # tensors: three reduce-scatters + three all-gathers per step instead of one pair per tensor.
            by_attr = lambda *names: [getattr(self, a) for a in names if isinstance(getattr(self, a, None), nn.Parameter)]   # noqa: E731
            self.bucket = FlatGradBucket(params, shards=shard[1], flat_params=True, small_numel=getattr(self, "bucket_small_numel", 1 << 20),
                                         groups=[by_attr("_features_dc", "_features_rest"),
                                                 by_attr("_xyz", "_scaling", "_rotation", "_opacity", "opacity_thres")])
            self.optimizer = FusedAdam(groups, self.bucket, eps=1e-15, shard=shard)
        else:
            self.bucket = FlatGradBucket(params)
            self.optimizer = FusedAdam(groups, self.bucket, eps=1e-15)
        self.optimizer_epoch += 1

    def _gaussian_groups(self):
        a, s = self.training_args, self.spatial_lr_scale
        return [{"params": [self._xyz], "lr": a.position_lr_init * s, "name": "xyz"},
                {"params": [self._features_dc], "lr": a.feature_lr, "name": "f_dc"},
                {"params": [self._features_rest], "lr": a.feature_lr / 20.0, "name": "f_rest"},
                {"params": [self._opacity], "lr": a.opacity_lr, "name": "opacity"},
                {"params": [self._scaling], "lr": a.scaling_lr, "name": "scaling"},
                {"params": [self._rotation], "lr": a.rotation_lr, "name": "rotation"}]

    def _keypoint_groups(self):
        a = self.training_args
        g = [{"params": [self.super_gaussians], "lr": a.kpts_lr, "name": "s_xyz"},
             {"params": [self.super_gaussians_feature], "lr": a.kpts_lr, "name": "s_motion_feature"}]
        if getattr(self, "weights_model", None) is not None:
            g.append({"params": list(self.weights_model.parameters()), "lr": a.hash_lr, "name": "weight_mlp"})
        return g

    def _thres_group(self):
        if self.args.step_opacity and isinstance(getattr(self, "opacity_thres", None), nn.Parameter):
            return [{"params": [self.opacity_thres], "lr": self.training_args.opacity_lr, "name": "opacity_thres"}]
        return []

    def _stage_groups(self, stage):
        a = self.training_args
        mlp = [{"params": list(self.df_model.parameters()), "lr": a.mlp_lr, "name": "df_mlp"}]
        if stage == 1:        # [REF :432-451]
            return self._gaussian_groups() + mlp + \
                [{"params": [self.motion_feature], "lr": a.mfeature_lr, "name": "motion_feature"}] + self._thres_group()
        if stage == 2:        # [REF :413-430]
            return self._keypoint_groups() + mlp
        return self._gaussian_groups() + self._keypoint_groups() + mlp + self._thres_group()     # [REF :394-411]

    def _reset_gaussian_stats(self):
        n, dev = self.get_xyz.shape[0], self.get_xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.xyz_gradient_accum_max = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros(n, device=dev)

    def _reset_kpts_stats(self, k=None):
        dev = self.get_xyz.device
        k = self.super_gaussians.shape[0] if k is None else k
        self.kpts_gradient_accum = torch.zeros((k, 1), device=dev)
        self.kpts_gradient_accum_max = torch.zeros((k, 1), device=dev)
        self.kpts_denom = torch.zeros((k, 1), device=dev)

    def training_setup(self, training_args):
        """Stage 1 [REF scene/gaussian_model.py:432-472]."""
        self.training_args = a = training_args
        self.percent_dense = a.percent_dense
        keep_radii = self.max_radii2D if self.max_radii2D.numel() == self.get_xyz.shape[0] else None
        self._reset_gaussian_stats()
        if keep_radii is not None:               # the reference allocates max_radii2D in create_from_pcd, not here
            self.max_radii2D = keep_radii
        self._reset_kpts_stats(self.args.max_points if hasattr(self.args, "max_points") else None)
        self._stage = 1
        self._install_optimizer(self._stage_groups(1))
        s = self.spatial_lr_scale
        self.xyz_scheduler_args = get_expon_lr_func(a.position_lr_init * s, a.position_lr_final * s,
                                                    lr_delay_mult=a.position_lr_delay_mult, max_steps=a.position_lr_max_steps)
        self.mlp_scheduler_args = get_expon_lr_func(a.mlp_lr, a.position_lr_final, lr_delay_mult=a.position_lr_delay_mult,
                                                    max_steps=a.position_lr_max_steps)
        self.weight_mlp_scheduler_args = get_expon_lr_func(a.hash_lr, a.hash_lr_final, lr_delay_steps=a.position_lr_max_steps,
                                                           max_steps=a.iterations)
        self.motion_feature_scheduler_args = get_expon_lr_func(a.mfeature_lr, a.mfeature_lr_final,
                                                               lr_delay_steps=a.position_lr_max_steps,
                                                               max_steps=a.position_lr_max_steps)
        self.super_xyz_scheduler_args = get_expon_lr_func(a.kpts_lr, a.kpts_lr_final, lr_delay_steps=a.position_lr_max_steps,
                                                          max_steps=a.iterations)

    def training2stage_setup(self):
        """Stage 2: keypoints + MLP only [REF :413-430]."""
        self.second_stage = True
        n, dev = self.get_xyz.shape[0], self.get_xyz.device
        self.xyz_motion_accum = torch.zeros((n, 1), device=dev)
        self.xyz_motion_accum_max = torch.zeros((n, 1), device=dev)
        self.motion_denom = torch.zeros((n, 1), device=dev)
        self._reset_kpts_stats()
        self._stage = 2
        self._install_optimizer(self._stage_groups(2))

    def training3stage_setup(self):
        """Stage 3: everything except the per-Gaussian motion feature [REF :394-411]."""
        self.third_stage = True
        self._stage = 3
        self._install_optimizer(self._stage_groups(3))

    def setup_for_iteration(self, training_args, iteration):
        """The setup call the reference would have made by `iteration` [REF :96-104 restore, :244-249 forward]."""
        if self.training_args is None or iteration <= self.second_stage_iter:
            self.training_setup(training_args)
        if iteration > self.third_stage_iter:
            self.second_stage = True
            self.training3stage_setup()
        elif iteration > self.second_stage_iter:
            self.training2stage_setup()

    def restore(self, opt_dict, training_args, iteration=-1):
        """[REF scene/gaussian_model.py:96-104]"""
        self.training_args = training_args
        if iteration <= self.second_stage_iter:
            self.training_setup(training_args)
        else:
            if not hasattr(self, "xyz_scheduler_args"):
                self.training_setup(training_args)       # schedulers + statistics (the reference keeps them from __init__ time)
            if iteration > self.third_stage_iter:
                self.second_stage = True
                self.training3stage_setup()
            else:
                self.training2stage_setup()
        self.optimizer.load_state_dict(opt_dict)

    def update_learning_rate(self, iteration):
        """Per-step learning-rate schedule, by group name [REF scene/gaussian_model.py:474-491]."""
        for g in self.optimizer.param_groups:
            name = g["name"]
            if name == "xyz" or "delta" in name or "fourier_weights" in name:
                g["lr"] = self.xyz_scheduler_args(iteration)
            elif "mlp" in name and "weight" not in name:
                g["lr"] = self.mlp_scheduler_args(iteration)
            elif name == "s_xyz" or name == "weight_feature":
                g["lr"] = self.super_xyz_scheduler_args(iteration)
            elif "weight_mlp" in name:
                g["lr"] = self.weight_mlp_scheduler_args(iteration)
            elif "motion_feature" in name or "motion_weights" in name or name == "delta_xyz":
                g["lr"] = self.motion_feature_scheduler_args(iteration)

    def oneupSHdegree(self):                      # [REF :323-325]
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- Adam-moment access --------------------------------------------------------------------------------------
    def adam_moments(self):
        """{id(param): (exp_avg, exp_avg_sq)} of the current optimizer."""
        opt = self.optimizer
        if opt is None:
            return {}
        return opt.full_moments()                # (sharded: a collective -- every rank performs the same surgery)

    def _unconsumed_gradient(self, params=None):
        """True when the CURRENT bucket holds a gradient that no optimizer step has consumed yet (a backward ran and
        `optimizer.step()` has not): some non-stale gradient buffer is not all zero.  This package's own harness updates first
        and operates afterwards (everything is zero -- Adam zeroes in the same launch -- or marked stale); the reference's loop
        runs densify / prune / reset_opacity BETWEEN backward and optimizer.step() [REF train.py:164-197].  One host read."""
        if self.bucket is None:
            return False
        from . import grad_sink
        live = [self.bucket.segment(p) for p in (self.bucket.params if params is None else params)
                if p.grad is not None and not grad_sink.is_stale(p.grad)]
        live = [seg for seg in live if seg.numel() > 0]
        if not live:
            return False
        return bool(torch.stack([seg.abs().max() for seg in live]).max() > 0)

    def _rebuild_optimizer(self, carried):
        """New bucket + optimizer over the model's CURRENT Parameters.  `carried` maps id(new per-Gaussian Parameter) to its
        (exp_avg, exp_avg_sq); every other parameter keeps its moments; step count and learning rates are preserved.

        Loops in the reference's order [REF train.py:164-197: backward -> densify / prune -> optimizer.step()] arrive here with an
        unconsumed gradient in the old bucket.  torch.optim.Adam would then update the parameters that SURVIVED the surgery (MLP,
        keypoints, hash grid) from it and pass over the replaced per-Gaussian tensors, whose .grad is None: the surviving
        parameters' gradients are carried into the new bucket and the replaced groups are held back on the next step()
        (`FusedAdam.pending_hold`), which is the same outcome."""
        old = self.optimizer
        if old is None:
            return
        pending = self._unconsumed_gradient()
        old_bucket = self.bucket
        old_off = {id(p): off for p, off in zip(old_bucket.params, old_bucket.offsets)}
        old_mom = self.adam_moments()
        lrs = {g["name"]: g["lr"] for g in old.param_groups}
        old_step = old.step_count
        from . import grad_sink
        grad_sink.forget_all()
        self._install_optimizer(self._stage_groups(self._stage))
        for g in self.optimizer.param_groups:
            if g["name"] in lrs:
                g["lr"] = lrs[g["name"]]
        self.optimizer.step_count = old_step     # the reference keeps the stored state (and its step) [REF :595-598]
        self.optimizer.lag = dict(old.lag)       # ... per group: the steps it skipped
        self.optimizer.pending_hold = set(getattr(old, "pending_hold", ()))
        for p in self.bucket.params:
            src = carried.get(id(p)) or old_mom.get(id(p))
            if src is not None and src[0].shape == p.shape:
                self.optimizer.load_full_moments(p, src[0], src[1])
        if pending:
            for p, off in zip(self.bucket.params, self.bucket.offsets):
                if id(p) in old_off:             # the same Parameter object: its gradient of this iteration is still due
                    self.bucket.flat[off:off + p.numel()].copy_(old_bucket.flat[old_off[id(p)]:old_off[id(p)] + p.numel()])
            self.optimizer.pending_hold |= {g["name"] for g in self.optimizer.param_groups
                                            if any(id(p) not in old_off for p in g["params"] if p.requires_grad)}
        self.assert_ranks_agree("optimizer rebuild")

    def _resize_per_gaussian(self, new_tensors, keep, n_new):
        """Install new per-Gaussian tensors (name -> tensor).  `keep` = bool mask over the OLD rows that survive, in order;
        `n_new` rows follow them with zero moments [REF :551-630 _prune_optimizer / cat_tensors_to_optimizer]."""
        old = self._per_gaussian()
        mom = self.adam_moments()
        carried = {}
        for name, p in old.items():
            if id(p) in mom:
                m, v = mom[id(p)]
                tail = [n_new] + list(p.shape[1:])
                carried[name] = (torch.cat([m[keep], torch.zeros(tail, device=m.device)]),
                                 torch.cat([v[keep], torch.zeros(tail, device=v.device)]))
        fresh = {}
        for name, attr in PER_GAUSSIAN:
            if name in new_tensors and name in old:
                q = nn.Parameter(new_tensors[name].contiguous().requires_grad_(True))
                setattr(self, attr, q)
                fresh[name] = q
        self._rebuild_optimizer({id(fresh[name]): mv for name, mv in carried.items() if name in fresh})

    def _sync_side_stream(self):
        """Before anything outside forward() / render() reads or rewrites parameters: a harness may still be updating them on
        a second stream (`_param_ready_event`) or gathering other ranks' slices into them (`_param_ready_wait`, the
        asynchronous all-gather of dist.ShardedExchange) -- surgery on stale rows, or an in-place reset that a late
        all-gather overwrites, would otherwise depend on every driver remembering TrainStep.sync_params()."""
        ev = getattr(self, "_param_ready_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        wait = getattr(self, "_param_ready_wait", None)
        if wait is not None:
            wait()

    # ---- densification statistics ----------------------------------------------------------------------------------
    @staticmethod
    def _running_max(current, candidate):
        """Elementwise: the candidate where it is larger, else what was there (a NaN candidate never replaces a number)."""
        return torch.where(candidate > current, candidate, current)

    @staticmethod
    def _mean_grad(accum, denom):
        """accum / denom with 0 where nothing was accumulated (the reference divides and then overwrites the NaNs)."""
        return torch.where(denom > 0, accum / denom.clamp_min(1), torch.zeros_like(accum))

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """Per visible Gaussian: sum, count and maximum of the screen-space positional gradient's length
        [REF scene/gaussian_model.py:755-760]."""
        size = viewspace_point_tensor.grad[update_filter, :2].norm(dim=-1, keepdim=True)
        self.xyz_gradient_accum[update_filter] += size
        self.denom[update_filter] += 1
        self.xyz_gradient_accum_max[update_filter] = self._running_max(self.xyz_gradient_accum_max[update_filter], size)

    def add_desification_stats_motion(self, viewspace_point_tensor, update_filter=None):
        """(the reference's spelling)  Two statistics share this entry [REF scene/gaussian_model.py:762-773]: without a filter
        the argument is a per-Gaussian displacement error (blend vs. teacher) and its length feeds the running maximum that
        proposes new keypoints; with one it is a tensor whose .grad is a per-keypoint gradient."""
        per_gaussian = update_filter is None
        size = (viewspace_point_tensor if per_gaussian else viewspace_point_tensor.grad).norm(dim=-1, keepdim=True)
        if per_gaussian:
            self.xyz_motion_accum_max = self._running_max(self.xyz_motion_accum_max, size)
            self.motion_denom += 1
            return
        self.kpts_gradient_accum += size
        self.kpts_gradient_accum_max = self._running_max(self.kpts_gradient_accum_max, size)
        self.kpts_denom += 1

    def sync_teacher_stats(self):
        """View-parallel: the teacher statistic (running maximum over every view rendered so far) becomes the maximum over the
        ranks' views -- what the views of one `--batch` leave behind in a single process [REF scene/gaussian_model.py:274-283]."""
        vp = self._vp_group()
        if vp is not None:
            import torch.distributed as tdist
            tdist.all_reduce(self.xyz_motion_accum_max, op=tdist.ReduceOp.MAX, group=vp[0])

    # ---- densify / prune --------------------------------------------------------------------------------------------
    def prune_points(self, mask):
        """Remove the rows where `mask` is set [REF :566-589]."""
        self._sync_side_stream()
        keep = ~mask
        P = {k: v.detach() for k, v in self._per_gaussian().items()}
        self._resize_per_gaussian({k: v[keep] for k, v in P.items()}, keep, 0)
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.xyz_gradient_accum_max = self.xyz_gradient_accum_max[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]

    def densification_postfix(self, new):
        """Append rows (name -> tensor) and reset EVERY per-Gaussian statistic, max_radii2D included [REF :632-661]."""
        self._sync_side_stream()
        P = {k: v.detach() for k, v in self._per_gaussian().items()}
        n_old = P["xyz"].shape[0]
        n_new = new["xyz"].shape[0]
        keep = torch.ones(n_old, dtype=torch.bool, device=P["xyz"].device)
        self._resize_per_gaussian({k: torch.cat([P[k], new[k]]) for k in P}, keep, n_new)
        self._reset_gaussian_stats()

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        """[REF :696-711]"""
        self._sync_side_stream()
        P = {k: v.detach() for k, v in self._per_gaussian().items()}
        sel = torch.norm(grads, dim=-1) >= grad_threshold
        sel = sel & (torch.exp(P["scaling"]).max(dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix({k: v[sel] for k, v in P.items()})
        return int(sel.sum())

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, generator=None):
        """[REF :663-694]: the gradients are zero-padded for the rows the clone step appended."""
        self._sync_side_stream()
        P = {k: v.detach() for k, v in self._per_gaussian().items()}
        n = P["xyz"].shape[0]
        padded = torch.zeros(n, device=P["xyz"].device)
        padded[:grads.shape[0]] = grads.squeeze()
        scaling = torch.exp(P["scaling"])
        sel = (padded >= grad_threshold) & (scaling.max(dim=1).values > self.percent_dense * scene_extent)
        stds = scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
        rots = build_rotation(P["rotation"][sel]).repeat(N, 1, 1)
        new = {k: v[sel].repeat(N, *([1] * (v.dim() - 1))) for k, v in P.items()}
        new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + P["xyz"][sel].repeat(N, 1)
        new["scaling"] = torch.log(scaling[sel].repeat(N, 1) / (0.8 * N))
        n_src = int(sel.sum())
        self.densification_postfix(new)
        self.prune_points(torch.cat([sel, torch.zeros(N * n_src, dtype=torch.bool, device=sel.device)]))
        return n_src

    def densify(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """Clone + split; NO pruning of transparent / oversized points here [REF :713-718]."""
        self._sync_side_stream()                 # (the selection below READS the parameters: every slice must have landed)
        grads = self._mean_grad(self.xyz_gradient_accum, self.denom)
        if generator is None:                    # (view-parallel: the same seeded stream on every rank)
            generator = self._surgery_generator()
        self._surgery_no += 1
        n_clone = self.densify_and_clone(grads, max_grad, extent)
        n_src = self.densify_and_split(grads, max_grad, extent, generator=generator)
        return n_clone, n_src

    def prune(self, max_grad, min_opacity, extent, max_screen_size):
        """[REF :745-753]; runs with the LIVE max_radii2D, which a preceding densify has zeroed [REF :661]."""
        self._sync_side_stream()
        o = torch.sigmoid(self._opacity.detach())
        mask = (o < min_opacity).squeeze(-1)
        if max_screen_size:
            big_vs = self.max_radii2D > max_screen_size
            big_ws = torch.exp(self._scaling.detach()).max(dim=1).values > 0.1 * extent
            mask = mask | big_vs | big_ws
        self.prune_points(mask)
        return int(mask.sum())

    def reset_opacity(self):
        """opacity <- inverse_sigmoid(min(opacity, 0.01)); its Adam moments are zeroed [REF :526-545]."""
        self._sync_side_stream()
        o = torch.sigmoid(self._opacity.detach())
        new = torch.minimum(o, torch.full_like(o, 0.01))
        new = torch.log(new / (1 - new))
        if self.optimizer is not None and self._opacity.grad is not None and self._unconsumed_gradient([self._opacity]):
            # reference order (backward -> reset_opacity -> optimizer.step()): the reference swaps in a new tensor whose .grad is None,
            # Adam passes over it on this iteration [REF :526-545, train.py:173-197] -- drop the gradient, hold the group once
            self.bucket.segment(self._opacity).zero_()
            self.optimizer.pending_hold.add("opacity")
        with torch.no_grad():
            self._opacity.copy_(new)
        if self.optimizer is not None:           # in place, on every rank its own slice (no collective, no temporaries)
            self.optimizer.zero_moments(self._opacity)
        self.assert_ranks_agree("reset_opacity")

    # ---- keypoint growth -----------------------------------------------------------------------------------------------
    def new_kpts_init(self):                      # [REF :170-172]
        self.new_xyz = None
        self.new_motion_feature = None

    def _kpts_room(self):
        return int(self.args.max_points + self.args.adaptive_points_num - self.super_gaussians.shape[0])

    @torch.no_grad()
    def get_new_kpts(self, mask, ratio=100):
        """Furthest-point sample of the masked Gaussians -> candidate keypoints, each with the motion feature of its
        nearest Gaussian [REF scene/gaussian_model.py:196-212]."""
        self._sync_side_stream()
        sampling = self.get_xyz.detach()[mask].contiguous()
        if sampling.shape[0] >= 1:
            select = sampling.shape[0] // ratio if sampling.shape[0] > ratio else 1
            clip = self._kpts_room()
            select = select if select < clip else clip
            if select <= 0:
                self.new_xyz, self.new_motion_feature = sampling[:0], self.motion_feature.detach()[:0]
                return
            idx = furthest_point_sampling(sampling, select)
            new_xyz = sampling[idx]
            nn_idx = nearest_index(new_xyz, self._xyz.detach())
            self.new_xyz, self.new_motion_feature = new_xyz[:clip], self.motion_feature.detach()[nn_idx][:clip]
        else:
            self.new_xyz, self.new_motion_feature = None, None

    def densification_motion_postfix(self, new_xyz, new_motion_feature):
        """Append keypoints (zero Adam moments) and reset every statistic [REF :612-630]."""
        self._sync_side_stream()
        mom = self.adam_moments()
        old_kp, old_kf = self.super_gaussians, self.super_gaussians_feature
        carried = {}
        new_kp = nn.Parameter(torch.cat([old_kp.detach(), new_xyz]).contiguous().requires_grad_(True))
        new_kf = nn.Parameter(torch.cat([old_kf.detach(), new_motion_feature]).contiguous().requires_grad_(True))
        for old, new, ext in ((old_kp, new_kp, new_xyz), (old_kf, new_kf, new_motion_feature)):
            if id(old) in mom:
                m, v = mom[id(old)]
                carried[id(new)] = (torch.cat([m, torch.zeros_like(ext)]), torch.cat([v, torch.zeros_like(ext)]))
        self.super_gaussians, self.super_gaussians_feature = new_kp, new_kf
        self._rebuild_optimizer(carried)
        n, dev = self.get_xyz.shape[0], self.get_xyz.device
        self.xyz_motion_accum = torch.zeros((n, 1), device=dev)
        self.xyz_motion_accum_max = torch.zeros((n, 1), device=dev)
        self.motion_denom = torch.zeros((n, 1), device=dev)
        self._reset_kpts_stats()
        self._reset_gaussian_stats()
        self.knn_idx = None if getattr(self, "_knn_from_model", False) else self.knn_idx    # stale once K changes

    # Three ways of PROPOSING keypoints (each leaves its proposal in new_xyz / new_motion_feature, None = nothing to add) ...
    def _propose_kpts_from_hot_gaussians(self, max_grad, ratio):
        """"down_sampling": a furthest-point sample of the Gaussians whose mean screen-space gradient exceeds the threshold."""
        hot = self._mean_grad(self.xyz_gradient_accum, self.denom) > max_grad
        self.get_new_kpts(hot.squeeze(-1), ratio=ratio)

    def _take_existing_kpts(self, which):
        """Copies of existing keypoints (`which`: indices or a mask), as many as there is room for."""
        room = self._kpts_room()
        self.new_xyz = self.super_gaussians.detach()[which][:room]
        self.new_motion_feature = self.super_gaussians_feature.detach()[which][:room]

    def _propose_kpts_dominant(self, max_grad, ratio):
        """"gaussian_mean": for every hot Gaussian the keypoint with the largest blend weight; each such keypoint once."""
        hot = (self._mean_grad(self.xyz_gradient_accum, self.denom) > max_grad).squeeze(-1)
        k = self.super_gaussians.shape[0]
        self._take_existing_kpts(self.weights_sum[hot, :k].argmax(dim=-1).unique())

    def _propose_kpts_own_gradient(self, max_grad, ratio):
        """any other mode: the keypoints whose OWN mean gradient reaches the threshold."""
        self._take_existing_kpts((self._mean_grad(self.kpts_gradient_accum, self.kpts_denom) >= max_grad).view(-1))

    @torch.no_grad()
    def densify_kpts(self, max_grad, mode="gaussian_mean", ratio=100):
        """... and one adoption step: append the proposal (zero Adam moments, statistics reset), forget it
        [REF scene/gaussian_model.py:720-744]."""
        self._sync_side_stream()
        propose = {"down_sampling": self._propose_kpts_from_hot_gaussians,
                   "gaussian_mean": self._propose_kpts_dominant}.get(mode, self._propose_kpts_own_gradient)
        propose(max_grad, ratio)
        if self.new_xyz is None:
            return
        self.densification_motion_postfix(self.new_xyz, self.new_motion_feature)
        self.new_kpts_init()

    @torch.no_grad()
    def set_superKeypoints(self, seed=0):
        """k-means of [xyz | motion_feature] -> K = max_points keypoints: positions = per-cluster mean of the Gaussian
        positions, features = the motion-feature part of the cluster centres [REF scene/gaussian_model.py:127-136]."""
        self._sync_side_stream()
        xyz = self.get_xyz.detach()
        feature = torch.cat([xyz, self.motion_feature.detach()], dim=-1)
        ids, centres = kmeans(feature, int(self.args.max_points), seed=seed)
        K = centres.shape[0]
        sums = torch.zeros(K, 3, device=xyz.device).index_add_(0, ids, xyz)
        cnt = torch.zeros(K, device=xyz.device).index_add_(0, ids, torch.ones(xyz.shape[0], device=xyz.device))
        means = torch.where(cnt[:, None] > 0, sums / cnt[:, None].clamp_min(1), centres[:, :3])
        kf, kp = centres[:, 3:].contiguous(), means.contiguous()
        self._vp_broadcast([kp, kf])             # (index_add_ sums with float atomics on the device: rank 0's result is everyone's)
        self.super_gaussians_feature = nn.Parameter(kf.requires_grad_(True))
        self.super_gaussians = nn.Parameter(kp.requires_grad_(True))
