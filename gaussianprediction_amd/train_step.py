"""Train-step harness: what the reference's loop does around the hot
path for one iteration [REF train.py:101-133, 196-197]:
    render -> 0.8*L1 + 0.2*(1-SSIM_11x11) + 1e-5*mean|motion feature| -> backward -> Adam(eps=1e-15).
Render, deformation, loss (fused L1+SSIM) and optimizer (fused Adam + gradient zeroing) all run on
this package's HIP kernels; there is no other implementation in the package (the tests build their own
plain-torch restatement of the step as the checker, tests/host_checkers.py).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

from .dist import OverlappedGradReducer, ShardedExchange
from . import grad_sink
from .loss_ops import l1_ssim_loss
from .renderer import render
from .training import default_training_args


class TrainStep:
    """One optimisation step: `batch` views per rank rendered, their losses SUMMED, one backward, one Adam step
    [REF train.py:101-133, 196-197]; view-parallel over the ranks when world_size > 1 (rank r takes the views
    (step * world + r) * batch + b).  Optimizer, gradient bucket, learning-rate schedule and densification live on the
    model, as in the reference (training.py)."""

    # capacity mode of the rasterizer's binning stage (include/gp_hip.h, gp_raster_settings.binning_capacity): the host
    # never waits for R.  Protocol: the first `len(cameras)` steps run in exact mode and report R through the status word;
    # afterwards the capacity is SPEC_MARGIN x the largest R seen, every step writes {R, overflow} into one of SPEC_SLOTS
    # status slots, Adam takes the overflow word as its skip flag, and the slot is read back (pinned, asynchronous) when it
    # comes round again SPEC_SLOTS steps later: the high-water mark follows the scene, and a frame that overflowed (its
    # update was skipped on the device) is repeated in exact mode.
    SPEC_SLOTS = 4
    SPEC_MARGIN = 1.1
    SPEC_PAD = 4096
    # The same protocol carries the DEPTH-KEY speculation (gp_raster_settings.depth_key_bits): the exact-mode steps also ask for the
    # {min, max} depth key of their visible Gaussians (status words 4, 5); once every view has reported, the capacity-mode steps
    # promise that the keys lie in a window of 2^KEY_BITS keys (two octaves: a factor of 4 in depth) centred on the range seen
    # so far -- if that leaves at least KEY_MIN_MARGIN keys (1/32 octave: 2 % in depth) on either side; the depth sort then runs three
    # passes instead of four.  A Gaussian outside the promised range raises the overflow word like a binning overflow: the frame is
    # repeated in exact mode, which also refreshes the range.
    KEY_BITS, KEY_MIN_MARGIN = 24, 1 << 18
    STATUS_WORDS = 8                # {R, overflow, library scratch, -, key min, key max, -, -}

    def __init__(self, pc, cameras, gt_images, iteration, lambda_dssim=0.2, lrs=None, group=None, speculative=False,
                 overlap_sh_adam=False, batch=1, schedule=False, training_args=None, sharded=None, fuse_sh_adam=True, chain_sh=True,
                 view_stats=None, fused=True, factorised_sh=False):
        """`fused`: take plain stage-3 steps through ONE call of the library (fused_step.FusedStage3 / gp_train_step_run) instead of
        the autograd graph -- same kernels, same results, a fraction of the host work; steps of any other shape use the graph.
        Buffers of the returned dict (`render`, `radii`, ...) are then the plan's own and are overwritten by the next step.
        `view_stats` (view-parallel only): True / False = always / never exchange the densification inputs' screen-space gradient
        (see _step); None = exactly while the reference's loop reads it (below densify_until_iter, and during keypoint growth)."""
        self.pc, self.cameras, self.gt, self.iteration = pc, cameras, gt_images, iteration
        self.view_stats = view_stats
        # view-parallel, replicated optimizer only (sharded=False), one view per step: the SH gradients travel as (dL/dRGB, view direction)
        # factors -- one all-gather of 24 B per Gaussian and rank -- instead of an all-reduce of 192 B (dist.OverlappedGradReducer.set_factorised)
        self.factorised_sh = bool(factorised_sh)
        self._view_center = None
        self.fused = bool(fused)
        self.early_adam = True                  # fused step: gp_step_update.adam_early_mask (the Adam rider of the MLP backward's launch)
        self._fused_plan = None
        self.fused_steps = 0
        self.lambda_dssim = lambda_dssim
        self.group = group
        self.chain_sh = bool(chain_sh)      # sharded exchange: SH regions' Adam + all-gather on a side stream (_chain_sh)
        self.batch = int(batch)
        self.schedule = bool(schedule)      # True: update_learning_rate(iteration) every step, as train.py:79 does
        dev = pc.get_xyz.device
        self.speculative = bool(speculative) and dev.type == "cuda"
        if self.speculative:
            K = self.SPEC_SLOTS
            self._status = torch.zeros(K, self.STATUS_WORDS, dtype=torch.int32, device=dev)
            self._status_host = torch.zeros(K, self.STATUS_WORDS, dtype=torch.int32).pin_memory()
            self._key_lo = self._key_hi = None      # smallest / largest visible depth key reported so far (over all views)
            self._key_none = torch.tensor([-1, 0], dtype=self._status.dtype, device=self._status.device)    # {0xFFFFFFFF, 0}: nothing reported
            self.depth_key_speculation = os.environ.get("GP_DEPTH_KEY_SPEC", "1") != "0"
            self._events = [None] * K
            self._slot_view = [None] * K            # view rendered by the step that last used the slot
            self._slot_spec = [False] * K           # ... and whether it ran in capacity mode
            self._slot_toff = [None] * K            # ... and its time offset (a repeated frame is repeated with it)
            self._slot_hold = [()] * K              # ... and the optimizer groups it held back
            self._slot_epoch = [None] * K           # ... and which optimizer (pc.optimizer_epoch) took the step
            self._r_max, self._n_steps, self.redone = 0, 0, 0
            self.last_depth_key_promise = None
            if os.environ.get("GP_SPEC_MARGIN"):     # test hook: a margin < 1 forces overflows (and the redo protocol)
                self.SPEC_MARGIN, self.SPEC_PAD = float(os.environ["GP_SPEC_MARGIN"]), 0
        # Optional (off: measured 1.83 -> 1.90 ms on the bench): Adam for the SH coefficients (3/4 of all parameter bytes) on
        # a second stream.  Their gradients are final as soon as the rasterizer backward has run and their values are not
        # read again before the next rasterizer forward, so the update can overlap the deformation backward of this step and
        # the deformation forward of the next -- but those kernels share the HBM with it and queue behind its 2800
        # workgroups, which costs more than the overlap hides.  Single-process only.
        self.overlap_sh_adam = bool(overlap_sh_adam) and dev.type == "cuda"
        # single view, single rank: the rasterizer backward applies Adam to the SH coefficients itself (gp_adam_fuse): their gradient
        # (192 of the 236 B per Gaussian) is never written or read back.  Bit-identical arithmetic to the optimizer kernel.
        self.fuse_sh_adam = bool(fuse_sh_adam) and dev.type == "cuda" and not self.overlap_sh_adam
        self._side = torch.cuda.Stream(device=dev) if self.overlap_sh_adam else None
        self._sh_early = False
        self._sink_cb = None
        self.bg = torch.zeros(3, device=dev)          # black background [REF train.py:59]
        self.pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
        # camera times live on the device: a per-step H2D copy from pageable memory would be a host sync
        self.times = [torch.from_numpy(c.time).to(torch.float32).to(dev) for c in cameras]
        # learning rates: the reference's defaults [REF arguments/__init__.py:74-90], overridable per group through `lrs`
        # (xyz, f_dc, opacity, scaling, rotation, mfeature, kpts, mlp, hash) or wholesale through `training_args`
        if training_args is None:
            m = dict(xyz="position_lr_init", f_dc="feature_lr", opacity="opacity_lr", scaling="scaling_lr", rotation="rotation_lr",
                     mfeature="mfeature_lr", kpts="kpts_lr", mlp="mlp_lr", hash="hash_lr")
            over = {m[k]: v for k, v in (lrs or {}).items() if k in m}
            if lrs and "f_rest" in lrs and "f_dc" not in lrs:
                over["feature_lr"] = 20.0 * lrs["f_rest"]
            training_args = default_training_args(**over)
        self.training_args = training_args
        # view-parallel: reduce-scatter -> sharded Adam -> all-gather by default (dist.ShardedExchange); sharded=False keeps
        # the all-reduce + replicated Adam of round 1
        import torch.distributed as tdist
        from .dist import active as dist_active
        world = tdist.get_world_size(group) if (tdist.is_available() and tdist.is_initialized()) else 1
        self.sharded = bool(dist_active(group) and (sharded is None or sharded))
        shard = (tdist.get_rank(group), world) if self.sharded else None
        if dist_active(group):                       # the replicas must stay identical through densify / prune / keypoint growth
            pc.set_view_parallel(group, tdist.get_rank(group), world)
        if pc.optimizer is None or getattr(pc, "_stage", None) != self._stage_of(iteration) or getattr(pc, "optimizer_shard", None) != shard:
            pc.optimizer_shard = shard
            pc.setup_for_iteration(training_args, iteration)
        self._epoch = None
        self._attach()

    def _view_stats_due(self):
        """Does the loop read `viewspace_points.grad` after this iteration?  [REF train.py:164-167] below densify_until_iter;
        [REF train.py:179-183] in the second stage while keypoints may still be added."""
        if self.view_stats is not None:
            return bool(self.view_stats)
        pc, a, it = self.pc, self.pc.args, self.iteration
        if it < getattr(self.training_args, "densify_until_iter", 0):
            return True
        if not getattr(pc, "second_stage", False) or not hasattr(a, "adaptive_end_iter") or not hasattr(pc, "super_gaussians"):
            return False
        return it < a.adaptive_end_iter + pc.second_stage_iter and pc.super_gaussians.shape[0] < a.max_points + a.adaptive_points_num

    def _stage_of(self, iteration):
        return 1 if iteration <= self.pc.second_stage_iter else (2 if iteration <= self.pc.third_stage_iter else 3)

    @property
    def bucket(self):
        return self.pc.bucket

    @property
    def optimizer(self):
        return self.pc.optimizer

    def _attach(self):
        """(Re-)attach the gradient reducer and the side-stream callback to the model's CURRENT bucket."""
        if self._epoch == self.pc.optimizer_epoch:
            return
        if getattr(self, "reducer", None) is not None:
            self.reducer.close()
        if self.sharded:
            self.reducer = ShardedExchange(self.pc.bucket, self.group)
            sh = {id(self.pc._features_dc), id(self.pc._features_rest)}
            self._early_params = {id(p) for p in self.pc.bucket.params} - sh
            self.pc._param_ready_wait = self.reducer.wait_params          # surgery / checkpoints / plain renders: wait for everything
            self.pc._param_late_event = self.reducer.late_event           # render(): an event for the SH gather instead of a wait
            self.reducer.chain = self._chain_sh
            self._chain_on, self._flag_handle = False, None
        else:
            self.reducer = OverlappedGradReducer(self.pc.bucket, self.group)
            self.pc._param_late_event = None
            if self.factorised_sh and self.reducer.enabled:
                if self.batch != 1:
                    raise RuntimeError("factorised_sh needs one view per step and rank (batch=1): a rank's SH gradient must be ONE view's")
                self.reducer.set_factorised(self.pc._features_dc, self.pc._features_rest, self._view_dirs, lambda: int(self.pc.active_sh_degree))
        if getattr(self, "overlap_sh_adam", False) and self._sink_cb is None:
            self._armed = False
            self._ev_bwd, self._ev_sh = torch.cuda.Event(), torch.cuda.Event()
            self._sink_cb = grad_sink.register_callback(self._on_sink)
        self._epoch = self.pc.optimizer_epoch

    # ---- optimizer-state surgery lives on the model (training.py); kept here for callers of the round-1 interface ------
    def adam_moments(self):
        self.wait_side()
        return self.pc.adam_moments()

    def rebuild_optimizer(self, carried):
        self.pc._rebuild_optimizer(carried)
        self._attach()

    def wait_side(self):
        """Make the current stream wait for the side-stream SH update (before anything but render() touches the SH
        parameters, their gradients or their Adam moments)."""
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def _chain_sh(self, region, h_rs):
        """ShardedExchange.chain: an SH region's reduce-scatter has just been issued (from the gradient hook, inside
        loss.backward(), the moment the rasterizer backward finished writing that gradient).  Its remaining life runs on the
        exchange's side stream, off the compute stream: wait for the reduced slice -> Adam on this rank's 1/world of the region ->
        all-gather of the updated slices, all under the blend / MLP backward, the main Adam launch and the next step's
        deformation, projection and sorts; the next rasterizer forward waits for the gather only in front of its SH -> RGB kernel
        (late_event -> gp_raster_settings.sh_ready_event).  Returns False (the ordinary path) unless this is a single-producer
        step on device tensors."""
        if not self._chain_on or self.reducer.side is None:
            return False
        params = [self.bucket.params[k] for k in region[2]]
        if not all(p is self.pc._features_dc or p is self.pc._features_rest for p in params):
            return False
        side = self.reducer.side
        with torch.cuda.stream(side):
            h_rs.wait()                                   # the SIDE stream waits for the reduced slice
            if self._flag_handle is not None:
                self._flag_handle.wait()                  # (capacity mode: the overflow flag, MAX over the ranks)
            self.optimizer.step(zero_grad=True, keep_grad=self._keep, skip_flag=self._skip_flag, only=params, stream=side, advance=False)
            self.reducer.gather_region(region, late=True)
        self._chained_params.extend(params)
        return True

    def _on_sink(self, p):
        """grad_sink callback (runs on the autograd thread, inside loss.backward()): the rasterizer backward has written the
        SH gradients -> launch their Adam update on the side stream, behind everything enqueued so far."""
        if not self._armed or p is not self.pc._features_rest:
            return
        self._armed = False
        self._ev_bwd.record()                                   # the autograd thread's current stream = the forward's stream
        self._side.wait_event(self._ev_bwd)
        self.optimizer.step(zero_grad=True, keep_grad=self._keep, skip_flag=self._skip_flag,
                            only=(self.pc._features_dc, self.pc._features_rest), stream=self._side, advance=False)
        self._ev_sh.record(self._side)
        self.pc._param_ready_event = self._ev_sh                # render() waits for it right before the rasterizer call
        self._sh_early = True

    def loss_of(self, image, gt):
        if self.iteration >= self.pc.args.jointly_iteration:            # [REF scene/gaussian_model.py:174-178]
            feat = self.pc.super_gaussians_feature if self.iteration > self.pc.second_stage_iter else self.pc.motion_feature
            return l1_ssim_loss(image, gt, self.lambda_dssim, feat, 1.0e-5)
        return l1_ssim_loss(image, gt, self.lambda_dssim)

    def step(self, view_index: int, time_offset=None, hold=()):
        """`time_offset`: a [1] device tensor added to every rendered view's time (the decaying time noise of the training
        loop [REF train.py:92-99]); None = the cameras' own times.
        `hold`: optimizer group names that skip this step's Adam update (densify.held_groups: the tensors the reference's loop
        replaces before its optimizer.step() on this iteration) -- FusedAdam.step(hold=...)."""
        self._time_offset = time_offset
        self._hold = tuple(hold)
        if not self.speculative:
            return self._step(view_index, None, None)
        K = self.SPEC_SLOTS
        slot = self._n_steps % K
        if self._events[slot] is not None:           # the step that used this slot K steps ago: long finished
            self._events[slot].synchronize()
            r, overflow = int(self._status_host[slot, 0]), int(self._status_host[slot, 1])
            self._r_max = max(self._r_max, r)
            if not self._slot_spec[slot]:            # an exact-mode step: it also reported the visible Gaussians' depth-key range
                lo, hi = int(self._status_host[slot, 4]) & 0xFFFFFFFF, int(self._status_host[slot, 5]) & 0xFFFFFFFF
                if lo <= hi and hi != 0:            # ({0xFFFFFFFF, 0} = nothing reported: see _run_slot)
                    self._key_lo = lo if self._key_lo is None else min(self._key_lo, lo)
                    self._key_hi = hi if self._key_hi is None else max(self._key_hi, hi)
            if overflow and self._slot_spec[slot]:   # its Adam update was skipped on the device: repeat the frame, exactly
                self.redone += 1
                if self._slot_epoch[slot] == self.pc.optimizer_epoch:
                    self.optimizer.step_count -= 1
                    for name in self._slot_hold[slot]:   # (the repeated step holds the same groups again)
                        left = self.optimizer.lag.get(name, 0) - 1
                        if left > 0:
                            self.optimizer.lag[name] = left
                        else:
                            self.optimizer.lag.pop(name, None)
                # (else: bucket + optimizer were rebuilt since -- a stage change or surgery: the skipped update's count belongs to an
                # optimizer that no longer exists, and the frame is repeated on the current model as an ordinary step)
                self._events[slot] = None
                self._time_offset, self._hold = self._slot_toff[slot], self._slot_hold[slot]
                self._run_slot(slot, self._slot_view[slot], exact=True)
                self._n_steps += 1
                return self.step(view_index, time_offset, hold)
        exact = self._n_steps < len(self.cameras) + K or self._r_max == 0   # until every view's R has been read back
        out = self._run_slot(slot, view_index, exact)
        self._n_steps += 1
        return out

    def _depth_key_promise(self):
        """(bits, base) for gp_raster_settings.depth_key_bits / depth_key_base, or None.  The depth keys are the bit patterns of positive
        floats (monotonic in the depth; an octave is 2^23 consecutive keys): the 2^24-key window is centred on the range seen so far,
        so the room left over is the margin on both sides."""
        if not getattr(self, "depth_key_speculation", False) or self._key_lo is None or self._key_hi < self._key_lo:
            return None
        bits = self.KEY_BITS
        slack = (1 << bits) - 1 - (self._key_hi - self._key_lo)
        if slack < 2 * self.KEY_MIN_MARGIN:
            return None
        return bits, max(1, self._key_lo - slack // 2)

    def _run_slot(self, slot, view_index, exact):
        status = self._status[slot]
        capacity = 0 if exact else (int(self._r_max * self.SPEC_MARGIN) + self.SPEC_PAD)
        # (capacity, {R, overflow, scratch}, the depth-key promise of a capacity-mode step, the key-range words of an exact one)
        binning = (capacity, status[0:3], None if exact else self._depth_key_promise(), status[4:6] if exact else None)
        self.last_depth_key_promise = binning[2]
        if exact:
            # "no report": the library fills the two range words only on the sorting path of a render that sorts something -- a step that
            # returns early (no Gaussians, nothing visible) must not leave {0, 0}, which would read as a range and pin the window's base
            # at 0 (round-5 advisor).  (min, max) start values: a batch's views then accumulate through the kernel's atomics.
            status[4:6].copy_(self._key_none, non_blocking=True)
        out = self._step(view_index, binning, None if exact else status[1:2])
        self._status_host[slot].copy_(status, non_blocking=True)
        ev = self._events[slot] or torch.cuda.Event()
        ev.record()
        self._events[slot], self._slot_view[slot], self._slot_spec[slot] = ev, view_index, not exact
        self._slot_toff[slot] = self._time_offset
        self._slot_hold[slot] = tuple(getattr(self, "_hold", ()))
        self._slot_epoch[slot] = self.pc.optimizer_epoch
        return out

    def _step(self, view_index: int, binning, skip_flag):
        pc = self.pc
        if self.iteration >= pc.args.jointly_iteration:
            pc.stage_transitions(self.iteration)     # (the hooks forward would run: the optimizer must be this step's from the start)
        self._attach()                               # (densify / prune / stage changes rebuild bucket + optimizer)
        if self.sharded:                             # parameters the deformation reads: their all-gather must have landed
            self.reducer.wait_params(only=self._early_params)
        if self.schedule:
            pc.update_learning_rate(self.iteration)  # [REF train.py:79]
        a = pc.args
        lifecycle = bool(a.step_opacity and self.iteration > a.step_opacity_iteration)
        if self.fused and self.batch == 1:
            from .fused_step import FusedStage3
            if FusedStage3.eligible(self, binning, getattr(self, "_hold", ())):
                plan = self._fused_plan
                if plan is None or plan.stale():
                    plan = self._fused_plan = FusedStage3(self)
                v = view_index % len(self.cameras)
                t_view = None
                if getattr(self, "_time_offset", None) is not None:
                    t_view = self.times[v] + self._time_offset
                # per-Gaussian "=" gradients: their producers overwrite them, the optimizer launch need not zero them (and the graph path
                # finds them marked stale, as after its own steps); the small tensors are zeroed as the graph path leaves them
                keep_f = (pc._features_dc, pc._features_rest, pc._rotation, pc._scaling, pc._opacity, pc._xyz)
                self.fused_steps += 1
                return plan.run(v, t_view, binning[0], binning[1], skip_flag, keep_f, depth_key=binning[2] if len(binning) > 2 else None)
        keep = ()
        if self.batch == 1 and not self.pipe.convert_SHs_python and self.iteration > pc.third_stage_iter:
            # the per-Gaussian gradients each have exactly one producer kernel that writes the whole tensor (SH: rasterizer
            # backward; xyz / rotation: blend backward; scaling / opacity: activation backward): their zeroing pass is
            # skipped and the next backward overwrites instead of accumulating.  With the lifecycle opacity the second MLP
            # pass is a second producer of the xyz gradient, so xyz keeps the zero-and-accumulate protocol.
            keep = (pc._features_dc, pc._features_rest, pc._rotation, pc._scaling, pc._opacity) + (() if lifecycle else (pc._xyz,))

        # A gradient may leave for its exchange from a kernel's completion notice only if that kernel is its ONLY producer in this
        # backward: the per-Gaussian tensors of a single-view step (SH: rasterizer backward; rotation / xyz: blend backward;
        # scaling / opacity: activation backward).  Everything else -- several views
        # per step; under the lifecycle opacity `_xyz`, the motion feature AND the MLP weights, which the second MLP pass also
        # differentiates; keypoints -- is exchanged after backward (finish()).  (Found by the two-rank test once it gave every
        # tensor a region of its own: the MLP weights left after the first of their two producers.)
        # (The stage-1 motion feature is NOT among them: the regulariser 1e-5 mean|motion_feature| [REF scene/gaussian_model.py:174-178]
        # is a second producer of its gradient beside the MLP backward -- found by the two-rank schedule test of round 5.)
        single = [] if self.batch > 1 else [pc._features_dc, pc._features_rest, pc._rotation, pc._scaling, pc._opacity] + \
            ([] if lifecycle else [pc._xyz])
        sid = {id(p_) for p_ in single}
        self.reducer.set_late([p_ for p_ in self.bucket.params if id(p_) not in sid])
        # groups that skip this step's update (the caller's, and the ones surgery left pending: FusedAdam.step merges `pending_hold` in,
        # but the early / fused / chained forms of the SH update are decided HERE -- round-5 advisor): with any, all of those forms are
        # off and one ordinary optimizer launch runs at the end
        hold = tuple(dict.fromkeys(tuple(getattr(self, "_hold", ())) + tuple(getattr(self.optimizer, "pending_hold", ()) or ())))
        plain = bool(hold)
        if self.overlap_sh_adam and keep and not self.reducer.enabled and not plain:
            self._armed, self._keep, self._skip_flag = True, keep, skip_flag
        sh_pair, fuse = (pc._features_dc, pc._features_rest), None
        if self.fuse_sh_adam and keep and not self.reducer.enabled and not self.sharded and not plain:
            from . import grad_sink
            fuse = self.optimizer.fused_payload(sh_pair[0], sh_pair[1], skip_flag)
            if fuse is not None:
                grad_sink.arm_fused_update(sh_pair, fuse)
        world = torch.distributed.get_world_size(self.group) if self.reducer.enabled else 1
        if self.sharded:
            # SH regions may leave the compute stream right after their reduce-scatter (_chain_sh): single-producer steps only
            self._chain_on = bool(keep) and self.reducer.enabled and getattr(self, "chain_sh", True) and not plain
            self._keep, self._skip_flag, self._chained_params, self._flag_handle = keep, skip_flag, [], None
        losses, pkgs = [], []
        if self.batch > 1 and hasattr(pc, "keypoint_weights_scope"):
            self._scope_no = getattr(self, "_scope_no", 0) + 1
            pc.keypoint_weights_scope(self._scope_no)        # the views of this step share one evaluation of the weights model
        try:
            for b in range(self.batch):              # [REF train.py:101-119]
                v = view_index * self.batch + b
                cam = self.cameras[v % len(self.cameras)]
                t_view = self.times[v % len(self.cameras)]
                if getattr(self, "_time_offset", None) is not None:
                    t_view = t_view + self._time_offset
                self._view_center = cam.camera_center
                pkg = render(cam, pc, self.pipe, self.bg, time=t_view, it=self.iteration, binning=binning)
                losses.append(self.loss_of(pkg["render"], self.gt[v % len(self.gt)]))
                pkgs.append(pkg)
            if skip_flag is not None and self.reducer.enabled:
                # one rank's overflow invalidates the summed gradient: every rank must skip (and later repeat) this step.  The
                # flag is final once the forward is enqueued, so its MAX over the ranks is started here, ahead of the backward
                self._flag_handle = torch.distributed.all_reduce(skip_flag, op=torch.distributed.ReduceOp.MAX, group=self.group, async_op=True)
            loss = losses[0] if self.batch == 1 else torch.stack(losses, dim=0).sum()
            seed = getattr(self, "_seed_one", None)  # (autograd would fill a ones_like(loss) every step: one more launch)
            if seed is None or seed.device != loss.device or seed.dtype != loss.dtype or seed.shape != loss.shape:
                seed = self._seed_one = torch.ones_like(loss)
            loss.backward(gradient=seed)             # hooks start the all-reduce of each large gradient as it completes
        except BaseException:
            if fuse is not None:                     # an armed update must not outlive the step it was meant for
                from . import grad_sink
                grad_sink.disarm_fused_update(sh_pair)
            raise
        finally:
            if self.batch > 1 and hasattr(pc, "keypoint_weights_scope"):
                pc.keypoint_weights_scope(None)
        self._armed = False
        h_vs = None
        if self.reducer.enabled and self._view_stats_due():
            # The densification input of a batch is the LAST view's screen-space gradient [REF train.py:123-127 compute the sum over
            # the batch, :167 hands add_densification_stats the loop variable = the last view's tensor, whose .grad holds that view
            # alone].  The ranks of a view-parallel step ARE that batch, rank order = view order: every rank receives the last rank's
            # last view's gradient (one [N,3] broadcast, asynchronous: it lands under the exchange and the optimizer launch), so that
            # xyz_gradient_accum -- and with it every densify / prune / keypoint-growth decision -- is the same on every rank.
            vs = pkgs[-1]["viewspace_points"]
            if vs.grad is None:                      # (nothing visible on this rank: the collective still needs an operand)
                vs.grad = torch.zeros_like(vs)
            h_vs = torch.distributed.broadcast(vs.grad, src=self.pc._vp_src(-1), group=self.group, async_op=True)
        self.reducer.finish()                        # SUM over views == the reference's --batch semantics
        if getattr(self, "_flag_handle", None) is not None:
            if self.sharded:
                self.reducer._timed_wait("overflow_flag", [self._flag_handle])
            else:
                self._flag_handle.wait()
        elif skip_flag is not None and self.reducer.enabled:
            torch.distributed.all_reduce(skip_flag, op=torch.distributed.ReduceOp.MAX, group=self.group)
        pkg = pkgs[-1]
        if self.batch > 1 or world > 1:
            # densification inputs of a batch [REF train.py:121-127]: radii = max over the views, visibility = any; the summed
            # screen-space gradient is computed -- and then the reference feeds add_densification_stats the LAST view's
            # tensor, whose .grad holds that view alone (train.py:167); both are returned.  View-parallel: `viewspace_points.grad`
            # is the last RANK's last view's on every rank (broadcast above); `viewspace_point_tensor_grad`, which the reference
            # computes and never reads, stays this rank's own sum.
            from .dist import reduce_view_stats
            radii = torch.stack([p["radii"] for p in pkgs]).max(dim=0).values
            radii, vis = reduce_view_stats(radii, self.group)
            grads = [p["viewspace_points"].grad for p in pkgs if p["viewspace_points"].grad is not None]
            pkg = dict(pkg, radii=radii, visibility_filter=vis,
                       viewspace_point_tensor_grad=torch.stack(grads).sum(0) if grads else None)
        if fuse is not None:
            from . import grad_sink
            if grad_sink.disarm_fused_update(sh_pair):      # no producer took it (e.g. python SH conversion): ordinary step below
                fuse = None
            else:                                           # done inside the backward: the (unwritten) gradient buffers are stale
                for p_ in sh_pair:
                    if p_.grad is not None:
                        grad_sink.mark_stale(p_.grad)
        # the small leaf inputs of the keypoint MLP / the regulariser: their zeroed gradient buffers are marked fresh, so that the first
        # producer of the next backward writes into them (deform_ops._input_sink) instead of launching an accumulate per tensor
        fresh = (pc.super_gaussians, pc.super_gaussians_feature) if self.iteration > pc.second_stage_iter else ()
        chained = getattr(self, "_chained_params", None) if self.sharded else None
        if chained:                  # their Adam slice and all-gather are already running on the exchange's side stream
            self.optimizer.step(zero_grad=True, keep_grad=keep, skip_flag=skip_flag, exclude=tuple(chained), fresh_grad=fresh)
        elif fuse is not None:
            self.optimizer.step(zero_grad=True, keep_grad=keep, skip_flag=skip_flag, exclude=sh_pair, fresh_grad=fresh)
        elif self._sh_early:         # the SH tensors were updated on the side stream during the backward
            self._sh_early = False
            self.optimizer.step(zero_grad=True, keep_grad=keep, skip_flag=skip_flag, exclude=(pc._features_dc, pc._features_rest), fresh_grad=fresh)
        else:
            self.optimizer.step(zero_grad=True, keep_grad=keep, skip_flag=skip_flag, hold=hold, fresh_grad=fresh)
        if self.sharded:
            self.reducer.gather_params()             # asynchronous; awaited by the next step / render
        if h_vs is not None:
            h_vs.wait()                              # (the current stream waits; the host does not)
        return loss.detach(), pkg

    def _view_dirs(self):
        """Unit directions camera -> deformed Gaussian of the view this rank has just rendered: where the rasterizer evaluated the SH basis
        (raster_kernels.hip sh_color: (p - campos) / |p - campos|)."""
        d = self.pc._last_xyz_t - self._view_center.to(self.pc._last_xyz_t.dtype).reshape(1, 3)
        return d / d.norm(dim=1, keepdim=True)

    def sync_params(self):
        """Wait for every outstanding parameter exchange (before anything outside step() / render() reads the parameters)."""
        if self.sharded:
            self.reducer.wait_params()
