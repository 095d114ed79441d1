"""The stage-3 train step through ONE call of the library (gp_train_step_run, include/gp_hip.h) instead of a dozen calls out of
an autograd graph: what TrainStep uses when a step has the plain shape the bench, the long tail of a training run and every rank
of a view-parallel job spend their time in --

    stage 3 (iteration > third_stage_iter), one view per step, keypoint weights + neighbour indices supplied
    (GaussianModel.set_keypoint_weights), no lifecycle opacity, no position noise, the split SH layout, capacity-mode binning,
    nothing held back, every optimized tensor fp32 and contiguous

-- and the drop-in surfaces (render(), GaussianRasterizer, loss_ops, FusedAdam.step) for everything else.  Same kernels in the same
order as the autograd path (the C entry calls the same entry points); the host's work per step is what changes: ~0.6 ms of Python
down to the launches.  [REF train.py:101-133, 196-197]"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib, grad_sink
from .deform_ops import packed_idx16


class FusedStage3:
    """Plan + per-step call.  Built for ONE (model, optimizer epoch, image size); TrainStep drops it when the bucket is rebuilt."""

    def __init__(self, ts):
        pc, dev = ts.pc, ts.pc.get_xyz.device
        self.ts, self.pc, self.dev = ts, pc, dev
        self.epoch = pc.optimizer_epoch
        a = pc.args
        N, K = pc._xyz.shape[0], pc.super_gaussians.shape[0]
        cam0 = ts.cameras[0]
        H, W = int(cam0.image_height), int(cam0.image_width)
        fd = pc.super_gaussians_feature.shape[1]
        xyz_freq, time_freq = int(pc.xyz_input_dim / 6), pc.time_input_dim // 2
        wb = pc.df_model._wb()
        in_dim, out_dim = wb[0].shape[1], wb[8].shape[0]
        nn_ = pc.knn_idx.shape[1]
        f32 = dict(dtype=torch.float32, device=dev)
        # The intermediates are carved out of ONE block (their addresses never change: the plan is filled once).  (An experiment on the
        # way: the first version of the fused step ran its HBM-bound kernels 4 - 7 % slower than the graph path; staggering these
        # buffers' start addresses against each other changed nothing -- the cause was the step's working set, see GP_BUF_TEMP_DONE.)
        sizes = []

        def e(*shape, dtype=torch.float32):
            n = 1
            for d_ in shape:
                n *= int(d_)
            sizes.append((shape, dtype, n * torch.empty(0, dtype=dtype).element_size()))
            return len(sizes) - 1
        in_pad = (in_dim + 7) // 8 * 8
        self.buf = dict(
            delta=e(K, out_dim), acts=e((K * in_pad + 63) // 64 * 64 + 4 * K * 256 + 4 * K * 8), xyz_t=e(N, 3), q_t=e(N, 4), scale=e(N, 3),
            opacity_t=e(N, 1), color=e(3, H, W), radii=e(N, dtype=torch.int32), depth=e(1, H, W),
            tidx=e(H, W, dtype=torch.int32), visible=e(N, dtype=torch.uint8),
            loss_sums=e(2 * _lib.GP_LOSS_SUM_SLOTS(H, W), dtype=torch.float64), loss=e(1),
            dL_dimage=e(3, H, W), g_xyz_t=e(N, 3), g_q_t=e(N, 4), g_scale=e(N, 3), g_opacity_t=e(N, 1), g_means2D=e(N, 3),
            g_delta=e(K, out_dim), g_feature_tmp=e(K, fd))
        offs, off = [], 0
        for k, (_, _, nbytes) in enumerate(sizes):
            off = (off + 255) // 256 * 256
            offs.append(off)
            off += nbytes
        self._block = torch.empty(off + 256, dtype=torch.uint8, device=dev)
        base = (256 - self._block.data_ptr() % 256) % 256
        for name, k in list(self.buf.items()):
            shape, dtype, nbytes = sizes[k]
            self.buf[name] = self._block[base + offs[k]: base + offs[k] + nbytes].view(dtype).view(*shape)
        # what render() hands back as `viewspace_points`: a tensor whose .grad is the screen-space gradient [REF scene/gaussian_model.py:757]
        self.viewspace = torch.zeros(N, 3, **f32).requires_grad_(True)
        self.viewspace.grad = self.buf["g_means2D"]
        self.raw_w = pc.raw_weights.detach().to(torch.float32).contiguous()
        self.knn = pc.knn_idx.detach().to(torch.int64).contiguous()
        self.idx16 = packed_idx16(pc.knn_idx, K)
        P = _lib.StepPlanC()
        P.num_gaussians, P.num_keypoints, P.nearest_num, P.norm_rotation = N, K, nn_, int(bool(a.norm_rotation))
        P.image_height, P.image_width = H, W
        P.lambda_dssim = float(ts.lambda_dssim)
        for name, t in (("xyz", pc._xyz), ("scaling", pc._scaling), ("rotation", pc._rotation), ("opacity", pc._opacity),
                        ("features_dc", pc._features_dc), ("features_rest", pc._features_rest), ("keypoints", pc.super_gaussians),
                        ("keypoint_features", pc.super_gaussians_feature)):
            setattr(P, name, t.data_ptr())
            setattr(P, "g_" + name, t.grad.data_ptr())
        P.mlp.in_dim, P.mlp.width, P.mlp.depth, P.mlp.out_dim = in_dim, 256, 4, out_dim
        for l in range(5):
            P.mlp.w[l], P.mlp.b[l] = wb[2 * l].data_ptr(), wb[2 * l + 1].data_ptr()
            P.g_mlp.dw[l], P.g_mlp.db[l] = wb[2 * l].grad.data_ptr(), wb[2 * l + 1].grad.data_ptr()
        # (K <= 512: the feature-split forward of the keypoint MLP; in the fused step the saved activations are its exchange buffer, the
        # scratch holds the counters.  The plan runs on one stream: the scratch of that stream, deform_ops.mlp_scratch)
        from .deform_ops import mlp_scratch
        self.mlp_scratch = mlp_scratch(dev, K)
        P.mlp.scratch = self.mlp_scratch.data_ptr() if self.mlp_scratch is not None else None
        P.feature_dim, P.xyz_freq, P.time_freq = fd, xyz_freq, time_freq
        P.raw_w, P.knn_idx = self.raw_w.data_ptr(), self.knn.data_ptr()
        P.knn_idx16 = self.idx16.data_ptr() if self.idx16 is not None else None
        P.dmaps = None      # (a TEMP buffer of the call: it shares memory with the sort's and the backward's scratch)
        for k in ("delta", "acts", "xyz_t", "q_t", "scale", "opacity_t", "loss_sums", "loss", "dL_dimage", "g_xyz_t", "g_q_t",
                  "g_scale", "g_opacity_t", "g_means2D", "g_delta", "g_feature_tmp"):
            setattr(P, k, self.buf[k].data_ptr())
        P.out = _lib.RasterOutputsC(self.buf["color"].data_ptr(), self.buf["radii"].data_ptr(), self.buf["depth"].data_ptr(),
                                    self.buf["tidx"].data_ptr(), self.buf["visible"].data_ptr())
        self.plan = P
        self.params = [pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features_dc, pc._features_rest, pc.super_gaussians,
                       pc.super_gaussians_feature] + list(wb)
        self._addr = [(p.data_ptr(), p.grad.data_ptr()) for p in self.params]          # (raw_w / knn: tracked by _kw_key below)
        self._kw_key = (id(pc.raw_weights), pc.raw_weights._version, id(pc.knn_idx), pc.knn_idx._version)
        # one view struct per camera (matrices, target, time all resident on the device)
        self.views, self._keep = [], []
        for cam, gt, t in zip(ts.cameras, ts.gt, ts.times):
            mats = [x.detach().to(torch.float32).contiguous() for x in (ts.bg, cam.world_view_transform, cam.full_proj_transform, cam.camera_center)]
            g = gt.detach().to(torch.float32).contiguous()
            if tuple(g.shape) != (3, H, W) or int(cam.image_height) != H or int(cam.image_width) != W:
                raise RuntimeError("fused step: every camera must have the plan's image size")
            self._keep += mats + [g]
            self.views.append(_lib.StepViewC(math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), mats[0].data_ptr(), mats[1].data_ptr(),
                                             mats[2].data_ptr(), mats[3].data_ptr(), g.data_ptr(), t.data_ptr()))
        # the optimizer's launch table, SH pair excluded when its update rides in the rasterizer backward
        self.upd = _lib.StepUpdateC()
        self._adam_key = None
        self._hook_cb = _lib.STEP_HOOK_FN(self._on_point)      # (kept alive with the plan)
        self._hook_error = None

    # ---- eligibility ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def eligible(ts, binning, hold):
        pc, a = ts.pc, ts.pc.args
        if not getattr(ts, "fused", True) or binning is None or not binning[0] or hold or ts.batch != 1:
            return False
        if pc.get_xyz.device.type != "cuda" or ts.iteration <= pc.third_stage_iter or ts.overlap_sh_adam:
            return False
        if getattr(pc.df_model, "generic", False):       # (--d / --w off the operating point: the layer-by-layer network is an autograd graph)
            return False
        if ts.reducer.enabled and not (ts.sharded and hasattr(ts.reducer, "gather_params")):
            return False                     # (view-parallel: the sharded exchange only; all-reduce + replicated Adam stays on the graph path)
        if pc.raw_weights is None or pc.knn_idx is None:      # (the per-frame weights model + kNN have autograd functions of their own)
            return False
        if a.step_opacity and ts.iteration > a.step_opacity_iteration:
            return False
        noise = getattr(a, "xyz_noise_iteration", 0)
        if noise and ((ts.iteration - pc.second_stage_iter) < noise or getattr(pc, "reference_rng", False)):
            return False                     # (the keypoint noise -- or, with reference_rng, its zero-scaled draw -- is a torch op of the graph path)
        if getattr(a, "densify_from_teaching", False) or ts.pipe.convert_SHs_python or ts.pipe.compute_cov3D_python:
            return False
        if pc._features_rest.dim() != 3 or pc._features_rest.shape[1] != 15 or pc.super_gaussians.shape[0] * pc.super_gaussians_feature.shape[1] > 65536:
            return False
        if pc.knn_idx.shape[1] < 1 or pc.raw_weights.shape != (pc._xyz.shape[0], 2 * pc.knn_idx.shape[1]):
            return False
        if ts.iteration < pc.args.jointly_iteration:
            return False
        opt = pc.optimizer
        if opt is None or opt.pending_hold or (opt.shard is not None) != bool(ts.reducer.enabled):
            return False
        if len(opt.items) > 32:              # (gp_step_update's launch table and its 32-bit masks)
            return False
        need = [pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features_dc, pc._features_rest, pc.super_gaussians,
                pc.super_gaussians_feature] + list(pc.df_model.parameters())
        have = {id(p) for p in pc.bucket.params}
        for p in need:
            if id(p) not in have or p.dtype != torch.float32 or not p.is_contiguous() or grad_sink.sink_of(p) is None:
                return False
        return True

    def stale(self):
        """True when something the plan holds an ADDRESS of has moved (then TrainStep builds a new plan)."""
        pc = self.pc
        if self.epoch != pc.optimizer_epoch:
            return True
        if self._kw_key != (id(pc.raw_weights), pc.raw_weights._version, id(pc.knn_idx), pc.knn_idx._version):
            return True
        return any(p.data_ptr() != a or p.grad is None or p.grad.data_ptr() != g for p, (a, g) in zip(self.params, self._addr))

    # ---- one step -----------------------------------------------------------------------------------------------------------------
    def _adam_tables(self, opt, fuse_sh):
        """The launch table of FusedAdam as the C entry takes it (sharded: this rank's slices); rebuilt only when the set of tensors
        changes.  `fuse_sh`: the SH pair is updated elsewhere (inside the rasterizer backward, or -- view-parallel -- on the exchange's
        side stream: TrainStep._chain_sh)."""
        pc = self.pc
        skip = {id(pc._features_dc), id(pc._features_rest)} if fuse_sh else set()
        key = (id(opt), bool(fuse_sh), len(opt.items))
        if self._adam_key != key:
            n = len(opt.items)
            self._tab = dict(
                P=(C.c_void_p * n)(*[p.data_ptr() for _, p, _, _, _ in opt.items]),
                G=(C.c_void_p * n)(*[opt.bucket.flat.data_ptr() + 4 * off for _, _, off, _, _ in opt.items]),
                M=(C.c_void_p * n)(*[m.data_ptr() for _, _, _, m, _ in opt.items]),
                V=(C.c_void_p * n)(*[v.data_ptr() for _, _, _, _, v in opt.items]),
                NUM=(C.c_int64 * n)(*[0 if id(own) in skip else p.numel() for (_, p, _, _, _), own in zip(opt.items, opt.owner)]),
                LR=(C.c_float * n)(), STEPS=(C.c_int64 * n)())
            u = self.upd
            u.adam_count = n
            for f, k in (("adam_params", "P"), ("adam_grads", "G"), ("adam_exp_avgs", "M"), ("adam_exp_avg_sqs", "V"), ("adam_numels", "NUM"),
                         ("adam_lrs", "LR")):
                setattr(u, f, C.cast(self._tab[k], C.c_void_p))
            self._adam_key = key
        return self._tab

    # ---- view-parallel: the exchange is issued from the host BETWEEN the library's enqueues (gp_step_update.hook) ----------------
    def _on_point(self, _ctx, point):
        """GP_STEP_AFTER_FORWARD: the overflow word is final -> its MAX over the ranks.  GP_STEP_AFTER_RASTER_BACKWARD: the SH
        gradients (3/4 of the bytes) and the screen-space gradient are final -> their reduce-scatter (chained to its Adam slice and
        all-gather on the side stream), the densification input's broadcast.  GP_STEP_AFTER_BACKWARD: everything else, then the
        compute stream waits for the reduced slices in front of the optimizer launch.  Never raises across the C boundary."""
        try:
            ts, ex = self.ts, self.ts.reducer
            import torch.distributed as dist
            if point == 2:
                if self._skip_flag is not None:
                    ts._flag_handle = dist.all_reduce(self._skip_flag, op=dist.ReduceOp.MAX, group=ts.group, async_op=True)
            elif point == 0:
                if self._stats_due:
                    self._h_vs = dist.broadcast(self.viewspace.grad, src=self.pc._vp_src(-1), group=ts.group, async_op=True)
                for region in self._sh_regions:
                    h = ex._reduce_scatter(region)
                    if ex.chain is not None and ex.chain(region, h):
                        ex._chained.add(region[0])
                    else:
                        ex.handles.append(h)
            elif point == 1:
                sh = {r[0] for r in self._sh_regions}
                for region in ex.bucket.regions:
                    if region[0] not in sh:
                        ex.handles.append(ex._reduce_scatter(region))
                ex._timed_wait("reduce_scatter", ex.handles)
                ex.handles.clear()
                if ts._flag_handle is not None:
                    ex._timed_wait("overflow_flag", [ts._flag_handle])
                n = ex.bucket.flat.numel()
                ex.bytes_sent_per_step = 2 * 4 * n * (ex.world - 1) // ex.world
        except BaseException as e:       # noqa: BLE001  (re-raised by run() once the library call has returned)
            self._hook_error = self._hook_error or e

    def run(self, view, time_tensor, capacity, status, skip_flag, keep, depth_key=None):
        """Enqueue the whole step for camera `view`; returns (loss [] tensor, result dict shaped like render()'s).
        `depth_key`: None or (bits, base), gp_raster_settings.depth_key_bits (status then has the library's scratch word)."""
        ts, pc, opt = self.ts, self.pc, self.pc.optimizer
        sh_pair = (pc._features_dc, pc._features_rest)
        ex = ts.reducer if ts.reducer.enabled else None
        u = self.upd
        fuse, fuse_c, chained = None, None, False
        if ex is None:
            fuse = opt.fused_payload(sh_pair[0], sh_pair[1], skip_flag) if ts.fuse_sh_adam else None
            if fuse is not None:
                fuse_c = _lib.AdamFuseC(fuse["m_dc"].data_ptr(), fuse["v_dc"].data_ptr(), fuse["m_rest"].data_ptr(), fuse["v_rest"].data_ptr(),
                                        fuse["lr_dc"], fuse["lr_rest"], fuse["beta1"], fuse["beta2"], fuse["eps"], fuse["step"],
                                        skip_flag.data_ptr() if skip_flag is not None else None)
            u.hook, u.sh_ready_event = _lib.STEP_HOOK_FN(), None
        else:
            # what TrainStep._step sets up for its gradient hooks, here for the three hook points of the library call
            sh_ids = {id(sh_pair[0]), id(sh_pair[1])}
            self._sh_regions = [r for r in ex.bucket.regions if r[0] != ex.bucket.tail_start and all(id(ex.bucket.params[k]) in sh_ids for k in r[2])]
            chained = bool(self._sh_regions) and bool(getattr(ts, "chain_sh", True)) and ex.side is not None
            ts._chain_on, ts._keep, ts._skip_flag, ts._chained_params, ts._flag_handle = chained, keep, skip_flag, [], None
            self._skip_flag, self._h_vs, self._stats_due, self._hook_error = skip_flag, None, ts._view_stats_due(), None
            late = getattr(pc, "_param_late_event", None)
            ev = late() if late is not None else None            # (waits for the early gathers; an event for the SH coefficients')
            u.sh_ready_event = int(ev.cuda_event) if ev is not None else None
            self._keep_ev = ev
            u.hook = self._hook_cb
        tab = self._adam_tables(opt, fuse is not None or chained)
        step_no = opt.step_count + 1
        for k, (g, _, _, _, _) in enumerate(opt.items):
            tab["LR"][k] = float(g["lr"])
        if opt.lag:
            for k, st in enumerate(opt.item_steps(step_no)):
                tab["STEPS"][k] = max(1, st)
            u.adam_steps = C.cast(tab["STEPS"], C.c_void_p)
        else:
            u.adam_steps = None
        keep_ids = {id(p) for p in keep}
        mask = 0
        for k, own in enumerate(opt.owner):
            if id(own) in keep_ids and tab["NUM"][k] != 0:
                mask |= 1 << k
        # the per-Gaussian tensors' gradients are final after the blend backward and the keypoint MLP's backward touches none of them: their
        # update rides in the launch of that backward's data kernel (gp_step_update.adam_early_mask; single-rank only -- a view-parallel
        # step reduces the gradients first).  Bit-identical to one launch; -0.015 ms per step on the bench workload
        # (profiles/r06_adam_rider_ab.txt; round 5's two-stream form of the same idea measured slower and is gone).
        early = 0
        # (not beside the feature-split MLP kernels, K <= 512: their chain of dependent trips to memory slows down under the optimizer's
        # stream by more than the overlap brings -- 0.092 ms fused against 0.019 + 0.054)
        if not ts.reducer.enabled and getattr(ts, "early_adam", False) and self.mlp_scratch is None:
            early_ids = {id(t) for t in (pc._xyz, pc._rotation, pc._scaling, pc._opacity, pc._features_dc, pc._features_rest)}
            for k, own in enumerate(opt.owner):
                if id(own) in early_ids and tab["NUM"][k] != 0:
                    early |= 1 << k
        u.adam_early_mask = early
        u.binning_capacity, u.binning_status = int(capacity), status.data_ptr()
        u.depth_key_bits, u.depth_key_base = (int(depth_key[0]), int(depth_key[1]) & 0xFFFFFFFF) if depth_key else (0, 0)
        u.adam_shs = C.cast(C.pointer(fuse_c), C.c_void_p) if fuse_c is not None else None
        u.beta1, u.beta2, u.eps, u.step, u.keep_grad_mask = float(opt.betas[0]), float(opt.betas[1]), float(opt.eps), int(step_no), mask
        u.skip_flag = skip_flag.data_ptr() if skip_flag is not None else None
        v = self.views[view]
        if time_tensor is not None:                 # (a jittered time: the view struct is per camera, the time pointer per step)
            v = _lib.StepViewC(v.tanfovx, v.tanfovy, v.bg, v.viewmatrix, v.projmatrix, v.campos, v.gt_image, time_tensor.data_ptr())
        P = self.plan
        P.sh_degree = int(pc.active_sh_degree)
        P.reg_scale = 1.0e-5 if ts.iteration >= pc.args.jointly_iteration else 0.0       # [REF scene/gaussian_model.py:174-178]
        alloc = _lib.TorchAllocator(self.dev)
        with _lib.on_device(self.dev):
            try:
                rc = _lib.lib().gp_train_step_run(C.byref(P), C.byref(v), C.byref(u), alloc.cb, None, _lib.stream_ptr(self.dev))
                if self._hook_error is not None:
                    err, self._hook_error = self._hook_error, None
                    raise err
                if alloc.error is not None:
                    raise alloc.error
                _lib.check(rc, "gp_train_step_run")
            finally:
                alloc.release()
        # ---- what FusedAdam.step does around its launch
        excluded = sh_pair if fuse is not None else (tuple(ts._chained_params) if chained else None)
        opt.external_step(keep_grad=keep, exclude=excluded)
        if fuse is not None:                         # updated inside the rasterizer backward; their (unwritten) gradient buffers are stale
            bump = torch.autograd.graph.increment_version
            for p in sh_pair:
                bump(p)
                if p.grad is not None:
                    grad_sink.mark_stale(p.grad)
        b = self.buf
        radii, vis = b["radii"], b["visible"].view(torch.bool)
        if ex is not None:
            ex.gather_params()                       # asynchronous; awaited by the next step / render
            from .dist import reduce_view_stats      # radii = max over the ranks' views, visibility = any [REF train.py:121-122]
            radii, vis = reduce_view_stats(radii, ts.group)
            if self._h_vs is not None:
                self._h_vs.wait()
        # the forward's side outputs GaussianModel.forward leaves behind (all_xyz_motion, kpts_*_motion, weights_sum, dense_weights():
        # densify_kpts(mode="gaussian_mean") reads them): views of the plan's buffers, overwritten by the next step -- without this a
        # fused step left the LAST GRAPH-PATH step's values there (round-5 advisor)
        pc._last_delta, pc._last_xyz_t = b["delta"], b["xyz_t"]
        pc._last_blend = (self.raw_w, self.knn, int(pc.super_gaussians.shape[0]))
        pkg = {"render": b["color"], "viewspace_points": self.viewspace, "visibility_filter": vis, "radii": radii,
               "depth": b["depth"], "tidx": b["tidx"]}
        if ex is not None:
            pkg["viewspace_point_tensor_grad"] = self.viewspace.grad        # (the graph path's multi-rank key)
        return b["loss"].reshape(()).clone(), pkg
