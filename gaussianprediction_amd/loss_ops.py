"""Fused L1 + SSIM loss and Adam step on libgp_hip.so.

  loss = (1 - lambda) * L1(image, gt) + lambda * (1 - SSIM(image, gt))        [REF train.py:105-108]
with the reference's 11x11 Gaussian-window SSIM [REF utils/loss_utils.py:54-100], and
torch.optim.Adam(eps=1e-15) semantics per parameter group [REF scene/gaussian_model.py:394-472].
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class L1SSIMLoss(torch.autograd.Function):
    """(1 - lambda) L1 + lambda (1 - SSIM) [+ reg_scale * mean|reg_x|: the motion-feature regulariser of
    [REF scene/gaussian_model.py:174-178] folded into the same launches when reg_x is small (<= 65536 elements)]."""

    @staticmethod
    def forward(ctx, image, gt, lambda_dssim, reg_x=None, reg_scale=0.0):
        if not image.is_cuda:
            raise RuntimeError("L1SSIMLoss: HIP kernels only (no CPU fallback)")
        dev = image.device
        img = image.detach().to(torch.float32).contiguous()
        g = gt.detach().to(torch.float32).contiguous()
        if img.dim() != 3 or img.shape[0] != 3 or g.shape != img.shape:
            raise RuntimeError("L1SSIMLoss expects two [3,H,W] images")
        _, H, W = img.shape
        sums = torch.empty(2 * _lib.GP_LOSS_SUM_SLOTS(H, W), dtype=torch.float64, device=dev)
        need = image.requires_grad or (reg_x is not None and reg_x.requires_grad)
        dmaps = torch.empty(3, 3, H, W, device=dev) if need else None
        with _lib.on_device(dev):
            rc = _lib.lib().gp_loss_l1_ssim_forward(_lib.ptr(img), _lib.ptr(g), C.c_int32(3), C.c_int32(H), C.c_int32(W),
                                                    _lib.ptr(sums), _lib.ptr(dmaps), _lib.stream_ptr(dev))
            _lib.check(rc, "gp_loss_l1_ssim_forward")
        lam = float(lambda_dssim)
        loss_t = torch.empty(1, dtype=torch.float32, device=dev)
        xc = reg_x.detach().to(torch.float32).contiguous() if reg_x is not None else None
        with _lib.on_device(dev):
            if xc is None:
                rc = _lib.lib().gp_loss_l1_ssim_finalize(_lib.ptr(sums), C.c_int32(3), C.c_int32(H), C.c_int32(W), C.c_float(lam),
                                                         _lib.ptr(loss_t), _lib.stream_ptr(dev))
            else:
                rc = _lib.lib().gp_loss_l1_ssim_finalize_reg(_lib.ptr(sums), C.c_int32(3), C.c_int32(H), C.c_int32(W), C.c_float(lam),
                                                             _lib.ptr(xc), C.c_int64(xc.numel()), C.c_float(float(reg_scale)),
                                                             _lib.ptr(loss_t), _lib.stream_ptr(dev))
            _lib.check(rc, "gp_loss_l1_ssim_finalize")
        loss = loss_t.reshape(())
        if need:
            ctx.save_for_backward(img, g, dmaps, xc if xc is not None else torch.empty(0, device=dev))
            ctx.lam, ctx.reg = lam, (float(reg_scale), reg_x.shape) if xc is not None else None
            ctx.reg_leaf = reg_x
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        img, g, dmaps, xc = ctx.saved_tensors
        dev = img.device
        _, H, W = img.shape
        up = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        dimg = torch.empty_like(img)
        gx = None
        with _lib.on_device(dev):
            if ctx.reg is None:
                rc = _lib.lib().gp_loss_l1_ssim_backward(_lib.ptr(img), _lib.ptr(g), _lib.ptr(dmaps), C.c_int32(3), C.c_int32(H),
                                                         C.c_int32(W), C.c_float(ctx.lam), _lib.ptr(up), _lib.ptr(dimg),
                                                         _lib.stream_ptr(dev))
            else:
                # (a leaf whose .grad buffer is marked fresh takes the regulariser's gradient directly: deform_ops._input_sink)
                from .deform_ops import _input_sink
                sink = _input_sink(getattr(ctx, "reg_leaf", None), ctx.reg[1]) if xc.is_contiguous() else None
                gx = sink if sink is not None else torch.empty_like(xc)
                rc = _lib.lib().gp_loss_l1_ssim_backward_reg(_lib.ptr(img), _lib.ptr(g), _lib.ptr(dmaps), C.c_int32(3), C.c_int32(H),
                                                             C.c_int32(W), C.c_float(ctx.lam), _lib.ptr(up), _lib.ptr(dimg), _lib.ptr(xc),
                                                             C.c_int64(xc.numel()), C.c_float(ctx.reg[0]), _lib.ptr(gx),
                                                             _lib.stream_ptr(dev))
                gx = gx.reshape(ctx.reg[1])
                if sink is not None:
                    from . import grad_sink
                    grad_sink.notify(ctx.reg_leaf)
                    gx = None
            _lib.check(rc, "gp_loss_l1_ssim_backward")
        return dimg, None, None, gx, None


class _AddL1Mean(torch.autograd.Function):
    """loss + scale * mean(|x|) in one kernel each way (the reference composes abs / mean / mul / add: ten launches on a
    tensor of a few thousand elements)."""

    @staticmethod
    def forward(ctx, loss, x, scale):
        if not x.is_cuda:
            raise RuntimeError("add_l1_mean: HIP kernels only (no CPU fallback)")
        xc = x.detach().to(torch.float32).contiguous()
        base = loss.detach().to(torch.float32).reshape(1).contiguous()
        out = torch.empty(1, device=xc.device)
        with _lib.on_device(xc.device):
            _lib.check(_lib.lib().gp_l1_mean_forward(_lib.ptr(xc), C.c_int64(xc.numel()), C.c_float(scale), _lib.ptr(base),
                                                     _lib.ptr(out), _lib.stream_ptr(xc.device)), "gp_l1_mean_forward")
        ctx.save_for_backward(xc)
        ctx.scale, ctx.shape = float(scale), x.shape
        return out.reshape(loss.shape)

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        up = g.detach().to(torch.float32).reshape(1).contiguous()
        gx = torch.empty_like(xc)
        with _lib.on_device(xc.device):
            _lib.check(_lib.lib().gp_l1_mean_backward(_lib.ptr(xc), C.c_int64(xc.numel()), C.c_float(ctx.scale), _lib.ptr(up),
                                                      _lib.ptr(gx), _lib.stream_ptr(xc.device)), "gp_l1_mean_backward")
        return g, gx.reshape(ctx.shape), None


def add_l1_mean(loss, x, scale):
    return _AddL1Mean.apply(loss, x, scale)


GP_LOSS_REG_MAX = 65536


def l1_ssim_loss(image, gt, lambda_dssim=0.2, reg_x=None, reg_scale=0.0):
    """reg_x (optional): + reg_scale * mean|reg_x| -- in the same launches when it is small, through add_l1_mean otherwise."""
    if reg_x is None:
        return L1SSIMLoss.apply(image, gt, lambda_dssim)
    if reg_x.numel() <= GP_LOSS_REG_MAX:
        return L1SSIMLoss.apply(image, gt, lambda_dssim, reg_x, reg_scale)
    return add_l1_mean(L1SSIMLoss.apply(image, gt, lambda_dssim), reg_x, reg_scale)


class FusedAdam:
    """Adam over the segments of a FlatGradBucket: ONE multi-tensor HIP launch, gradient zeroed in the same pass.  State
    layout mirrors torch.optim.Adam (exp_avg, exp_avg_sq, step).

    `shard=(rank, world)`: the sharded form (dist.ShardedExchange).  This rank holds moments for, and updates, only its
    1/world slice of every region of the bucket (parameters live in `bucket.pflat`); the launch table is the intersection of
    every tensor with that slice, so a slice of the small-tensor region that spans several tensors still gets each one's own
    learning rate."""

    host_step = None      # test seam: see step()

    def __init__(self, param_groups, bucket, betas=(0.9, 0.999), eps=1e-15, shard=None):
        self.param_groups = param_groups
        # the per-group step counts (`lag`), `hold` and `pending_hold` are keyed by group name: unnamed groups get one, duplicates are refused
        for k, g in enumerate(param_groups):
            if g.get("name") is None:
                g["name"] = f"group{k}"
        names = [g["name"] for g in param_groups]
        if len(set(names)) != len(names):
            raise ValueError(f"FusedAdam needs a unique 'name' on every parameter group (got {names})")
        self.bucket = bucket
        self.betas, self.eps = betas, eps
        self.step_count = 0
        # torch.optim.Adam counts steps per parameter; here: one counter + how far each GROUP is behind it (name -> skipped steps).
        # A group falls behind when a step holds it back (step(hold=...): the reference's densify / prune / reset_opacity replace the
        # per-Gaussian tensors BEFORE optimizer.step() of the same iteration, their .grad is None and Adam skips them
        # [REF train.py:164-197]) or when a loaded state dict says so.  Bias corrections use step_count - lag per tensor.
        self.lag = {}
        # groups the NEXT full step holds back once, whatever its `hold` argument says: set by optimizer-state surgery that found an
        # unconsumed gradient (a loop in the reference's order: backward -> densify / prune / reset_opacity -> optimizer.step(); the
        # replaced tensors' .grad is None there and torch.optim.Adam passes over them) -- training.TrainingMixin._rebuild_optimizer
        self.pending_hold = set()
        self.shard = shard
        off_of = {id(p): off for p, off in zip(bucket.params, bucket.offsets)}
        self.items = []        # (group, tensor the launch updates, its offset in the flat buffers, exp_avg, exp_avg_sq)
        self.owner = []        # the Parameter each item belongs to (itself when not sharded)
        for g in param_groups:
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous parameters")
                if shard is None:
                    self.items.append((g, p, off_of[id(p)], torch.zeros_like(p), torch.zeros_like(p)))
                    self.owner.append(p)
                    continue
                if bucket.pflat is None or bucket.shards != shard[1]:
                    raise RuntimeError("sharded FusedAdam needs FlatGradBucket(shards=world, flat_params=True)")
                lo, hi = bucket.shard_slice(bucket.region_of(p), shard[0])
                a, b = max(lo, off_of[id(p)]), min(hi, off_of[id(p)] + p.numel())
                if a < b:          # this rank owns elements [a, b) of the flat layout, all inside tensor p
                    view = bucket.pflat[a:b]
                    self.items.append((g, view, a, torch.zeros_like(view), torch.zeros_like(view)))
                    self.owner.append(p)

    def fused_payload(self, p_dc, p_rest, skip_flag=None):
        """What a producer kernel needs to apply THIS optimizer's next step to the SH pair itself (gp_adam_fuse; the caller then
        runs step(exclude=(p_dc, p_rest))).  None when the pair is not (both) optimized here, sharded, or not in the [N,15,3] layout."""
        if self.shard is not None or not self.bucket.flat.is_cuda:
            return None
        it = {id(p): (g, m, v) for g, p, _, m, v in self.items}
        if id(p_dc) not in it or id(p_rest) not in it or p_rest.dim() != 3 or p_rest.shape[1] != 15:
            return None
        (g0, m0, v0), (g1, m1, v1) = it[id(p_dc)], it[id(p_rest)]
        if self.lag_of(g0) != self.lag_of(g1):      # (one step number per fused launch)
            return None
        return dict(m_dc=m0, v_dc=v0, m_rest=m1, v_rest=v1, lr_dc=float(g0["lr"]), lr_rest=float(g1["lr"]), beta1=float(self.betas[0]),
                    beta2=float(self.betas[1]), eps=float(self.eps), step=int(self.step_count + 1 - self.lag_of(g0)), skip_flag=skip_flag)

    def lag_of(self, group):
        return int(self.lag.get(group.get("name"), 0))

    def item_steps(self, step_no):
        """The step number of every launch-table entry for the optimisation step `step_no` (its group's own count)."""
        return [step_no - self.lag_of(g) for g, _, _, _, _ in self.items]

    def full_moments(self):
        """{id(param): (exp_avg, exp_avg_sq)} as whole tensors.  Sharded: a COLLECTIVE (every rank must call it)."""
        if self.shard is None:
            return {id(p): (m, v) for _, p, _, m, v in self.items}
        import torch.distributed as dist
        n = self.bucket.flat.numel()
        full = [torch.zeros(n, device=self.bucket.flat.device) for _ in range(2)]
        for (_, view, a, m, v) in self.items:
            full[0][a:a + view.numel()] = m
            full[1][a:a + view.numel()] = v
        for f in full:             # every element is owned by exactly one rank, the others hold zero there
            dist.all_reduce(f, op=dist.ReduceOp.SUM)
        return {id(p): (full[0][off:off + p.numel()].view_as(p).clone(), full[1][off:off + p.numel()].view_as(p).clone())
                for p, off in zip(self.bucket.params, self.bucket.offsets)}

    def zero_moments(self, p):
        """Zero the Adam moments of parameter `p` in place -- on every rank its own slice when sharded; no collective
        [REF scene/gaussian_model.py:547-559 replace_tensor_to_optimizer]."""
        for (_, _, _, m, v), owner in zip(self.items, self.owner):
            if owner is p:
                m.zero_(); v.zero_()

    def load_full_moments(self, p, m_full, v_full):
        """Install whole-tensor moments for parameter `p` (each rank keeps its own slice when sharded)."""
        off = next(o for q, o in zip(self.bucket.params, self.bucket.offsets) if q is p)
        for (_, view, a, m, v), owner in zip(self.items, self.owner):
            if owner is p:
                if self.shard is None:
                    m.copy_(m_full.to(m.device, m.dtype)); v.copy_(v_full.to(v.device, v.dtype))
                else:
                    m.copy_(m_full.reshape(-1)[a - off:a - off + view.numel()].to(m.device, m.dtype))
                    v.copy_(v_full.reshape(-1)[a - off:a - off + view.numel()].to(v.device, v.dtype))

    # ---- the torch.optim.Adam surface train.py and the checkpoint format use ---------------------------------------
    @property
    def state(self):
        """{param: {"step", "exp_avg", "exp_avg_sq"}} (views of the live moments), empty before the first step as in torch."""
        if self.step_count == 0:
            return {}
        mom = self.full_moments()
        step_of = {id(p): self.step_count - self.lag_of(g) for g in self.param_groups for p in g["params"]}
        return {p: {"step": torch.tensor(float(step_of.get(id(p), self.step_count))), "exp_avg": mom[id(p)][0], "exp_avg_sq": mom[id(p)][1]}
                for p in self.bucket.params if id(p) in mom}

    def zero_grad(self, set_to_none=True):
        """train.py calls `optimizer.zero_grad(set_to_none=True)` [REF train.py:197]; the gradients here are views into the
        flat bucket (the all-reduce operand), so they are zeroed in place, never detached."""
        self.bucket.zero()

    def _group_template(self):
        probe = torch.optim.Adam([torch.zeros(1, requires_grad=True)], lr=0.0, betas=self.betas, eps=self.eps)
        return {k: v for k, v in probe.state_dict()["param_groups"][0].items() if k not in ("params", "lr")}

    def state_dict(self):
        """Same layout as `torch.optim.Adam.state_dict()` for the same groups: `(state_dict(), optimizer.state_dict(),
        iteration)` checkpoints interchange with the reference's [REF train.py:199-201, scene/gaussian_model.py:96-104]."""
        tmpl = self._group_template()
        mom = self.full_moments()
        groups, state, k = [], {}, 0
        for g in self.param_groups:
            entry = dict(tmpl)
            entry.update({kk: vv for kk, vv in g.items() if kk != "params"})
            entry["params"] = list(range(k, k + len(g["params"])))
            for p in g["params"]:
                if self.step_count > 0 and id(p) in mom:
                    m, v = mom[id(p)]
                    state[k] = {"step": torch.tensor(float(self.step_count - self.lag_of(g))), "exp_avg": m.detach().clone(),
                                "exp_avg_sq": v.detach().clone()}
                k += 1
            groups.append(entry)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts a `torch.optim.Adam.state_dict()` (or this class's) for the same groups."""
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError(f"loaded state dict has {len(groups)} parameter groups, the optimizer has {len(self.param_groups)}")
        known = {id(p) for p in self.bucket.params}
        steps = {}                                   # group name -> smallest stored step of its tensors
        for g, saved in zip(self.param_groups, groups):
            if len(saved["params"]) != len(g["params"]):
                raise ValueError(f"group {g.get('name')}: {len(saved['params'])} parameters saved, {len(g['params'])} expected")
            if "name" in saved and saved["name"] != g.get("name"):
                raise ValueError(f"group order differs: saved {saved['name']!r}, optimizer {g.get('name')!r}")
            g["lr"] = float(saved["lr"])
            for p, idx in zip(g["params"], saved["params"]):
                st = sd["state"].get(idx)
                if st is None or id(p) not in known:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"group {g.get('name')}: moment shape {tuple(st['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
                self.load_full_moments(p, st["exp_avg"], st["exp_avg_sq"])
                steps[g.get("name")] = min(steps.get(g.get("name"), 1 << 62), int(float(st["step"])))
        # torch.optim.Adam counts per parameter, and in the reference the counts differ: densify / prune / reset_opacity run BEFORE
        # optimizer.step() of the same iteration [REF train.py:164-197] and replace the per-Gaussian Parameters, whose .grad is then
        # None, so Adam skips them on that iteration and their `step` falls behind the MLP's by one per event.  The stored counts
        # are kept as they are: step_count = the largest, every group's distance to it in `lag` (bias corrections per tensor).
        self.step_count = max(steps.values()) if steps else 0
        self.lag = {name: self.step_count - st for name, st in steps.items() if st != self.step_count}

    def _launch(self, n, step_no, zero_grad, keep_ids, skip_flag, only, exclude, stream):
        """gp_adam_step_multi over the launch table (this rank's slices), restricted by `only` / `exclude`."""
        if not hasattr(self, "_tab"):
            P = (C.c_void_p * n)(*[p.data_ptr() for _, p, _, _, _ in self.items])
            G = (C.c_void_p * n)(*[self.bucket.flat.data_ptr() + 4 * off for _, _, off, _, _ in self.items])
            M = (C.c_void_p * n)(*[m.data_ptr() for _, _, _, m, _ in self.items])
            V = (C.c_void_p * n)(*[v.data_ptr() for _, _, _, _, v in self.items])
            NUM = (C.c_int64 * n)(*[p.numel() for _, p, _, _, _ in self.items])
            self._tab = (P, G, M, V, NUM)
            self._num_subsets = {}
        P, G, M, V, NUM = self._tab
        active = None
        if only is not None or exclude is not None:
            inc = {id(p) for p in only} if only is not None else None
            exc = {id(p) for p in exclude} if exclude is not None else set()
            key = (frozenset(inc) if inc is not None else None, frozenset(exc))
            sub = self._num_subsets.get(key)
            if sub is None:     # a tensor with numel 0 is skipped by the library
                flags = [(inc is None or id(p) in inc) and id(p) not in exc for p in self.owner]
                sub = ((C.c_int64 * n)(*[p.numel() if f else 0 for f, (_, p, _, _, _) in zip(flags, self.items)]), flags)
                self._num_subsets[key] = sub
            NUM, active = sub
        LR = (C.c_float * n)(*[float(g["lr"]) for g, _, _, _, _ in self.items])
        STEPS = (C.c_int64 * n)(*[max(1, st) for st in self.item_steps(step_no)]) if self.lag else None   # (no group behind: one count)
        mask = 0
        if zero_grad:
            for k, p in enumerate(self.owner):
                if id(p) in keep_ids and (active is None or active[k]):
                    mask |= 1 << k
        b1, b2 = self.betas
        dev = self.bucket.flat.device
        sp = _lib.stream_ptr(dev) if stream is None else C.c_void_p(stream.cuda_stream)
        with _lib.on_device(dev):
            if STEPS is None:
                rc = _lib.lib().gp_adam_step_multi(C.c_int32(n), P, G, M, V, NUM, LR, C.c_float(b1), C.c_float(b2), C.c_float(self.eps),
                                                   C.c_int64(step_no), C.c_int32(1 if zero_grad else 0), C.c_uint32(mask),
                                                   _lib.ptr(skip_flag), sp)
            else:
                rc = _lib.lib().gp_adam_step_multi_steps(C.c_int32(n), P, G, M, V, NUM, LR, STEPS, C.c_float(b1), C.c_float(b2),
                                                         C.c_float(self.eps), C.c_int32(1 if zero_grad else 0), C.c_uint32(mask),
                                                         _lib.ptr(skip_flag), sp)
            _lib.check(rc, "gp_adam_step_multi")

    def step(self, zero_grad=True, keep_grad=(), skip_flag=None, only=None, exclude=None, stream=None, advance=True, hold=(), fresh_grad=()):
        """One launch for all parameter tensors (gp_adam_step_multi_steps).  Parameters listed in `keep_grad` are not
        zeroed: their gradient buffers are marked stale (grad_sink.mark_stale) and the next backward overwrites them.
        `skip_flag`: optional int32 device word; non-zero = leave parameters and moments untouched (invalid frame).
        `only` / `exclude`: restrict the launch to a subset of the parameter tensors; `stream`: a torch.cuda.Stream other
        than the current one; `advance=False` uses step number step_count + 1 without committing it (the first of two
        partial launches of one optimisation step).
        `hold`: group names that SKIP this optimisation step as torch.optim.Adam skips a parameter whose .grad is None: no update,
        moments untouched, the group's own step count does not advance (`lag`), its gradient is discarded.
        `fresh_grad`: parameters whose gradient buffer -- zeroed by this very call -- is marked fresh (grad_sink.mark_fresh): the
        first producer of the next backward may write into it instead of going through autograd's accumulate."""
        step_no = self.step_count + 1
        held_params = ()
        if advance:
            self.step_count = step_no
            if self.pending_hold:
                hold = tuple(set(hold) | self.pending_hold)
                self.pending_hold = set()
        if hold:
            names = set(hold)
            held = [g for g in self.param_groups if g.get("name") in names]
            if advance:
                for g in held:
                    self.lag[g["name"]] = self.lag_of(g) + 1
            held_params = [p for g in held for p in g["params"] if p.requires_grad]
            if held_params:
                exclude = tuple(exclude or ()) + tuple(held_params)
        n = len(self.items)
        keep_ids = {id(p) for p in keep_grad}
        if not self.bucket.flat.is_cuda:
            # No CPU implementation ships.  The -m "not gpu" tests of the host logic around this class (sharding, checkpoint
            # layout, optimizer-state surgery) install their own checker here (tests/host_checkers.py); the bookkeeping below
            # (version counters, stale marks, zeroing of foreign slices) is the product's either way.
            if FusedAdam.host_step is None:
                raise RuntimeError("FusedAdam.step: HIP kernels only (no CPU fallback)")
            inc_h = {id(p) for p in only} if only is not None else None
            exc_h = {id(p) for p in exclude} if exclude is not None else set()
            FusedAdam.host_step(self, step_no, zero_grad, keep_ids,
                                [(inc_h is None or id(p) in inc_h) and id(p) not in exc_h for p in self.owner])
        else:
            self._launch(n, step_no, zero_grad, keep_ids, skip_flag, only, exclude, stream)
        self._after_launch(only, exclude, zero_grad, keep_ids, held_params, fresh_grad, stream)

    def external_step(self, keep_grad=(), exclude=None):
        """The bookkeeping of step() for a launch somebody else enqueued with THIS optimizer's tables (gp_train_step_run: the optimizer
        launch is the last thing the fused train step enqueues): the step count, the parameters' version counters, the stale marks,
        and -- sharded -- the zeroing of the slices the other ranks own."""
        self.step_count += 1
        self._after_launch(None, exclude, True, {id(p) for p in keep_grad}, (), (), None)

    def _after_launch(self, only, exclude, zero_grad, keep_ids, held_params, fresh_grad, stream):
        # which PARAMETERS this launch covered -- not only those this rank holds a slice of: under the sharded layout a tensor can
        # lie entirely inside another rank's slice of its region (regions shared by several tensors), and its gradient buffer
        # here still has to be marked stale / zeroed like everyone else's
        inc = {id(p) for p in only} if only is not None else None
        exc = {id(p) for p in exclude} if exclude is not None else set()
        covered = [p for p in self.bucket.params if (inc is None or id(p) in inc) and id(p) not in exc]
        # The parameters were written through raw pointers (or, on ranks that own no slice of a tensor, will be by the all-gather
        # into the flat parameter buffer, which does not touch p._version either: p.data is a view with a counter of its own).
        # Whatever caches on `_version` (GaussianModel's neighbour-search and keypoint-weights caches) or saved a parameter for a
        # backward must see the change ON EVERY RANK, so every covered parameter is bumped, not only this rank's slices --
        # otherwise ranks without a slice of _xyz / the keypoints keep blending with stale neighbour indices (rank-divergent
        # forwards, no error).
        bump = getattr(torch.autograd.graph, "increment_version", None)
        if bump is None:
            raise RuntimeError("torch.autograd.graph.increment_version is missing: cached keypoint weights could not be invalidated")
        for p in covered:
            bump(p)
        if zero_grad and held_params:
            off_of = {id(q): o for q, o in zip(self.bucket.params, self.bucket.offsets)}
            for p in held_params:                # a held group's gradient is dropped, as the replaced tensor's is in the reference
                self.bucket.flat[off_of[id(p)]:off_of[id(p)] + p.numel()].zero_()
        if zero_grad and keep_ids:
            from . import grad_sink
            for p in covered:
                if id(p) in keep_ids and p.grad is not None:
                    grad_sink.mark_stale(p.grad)
        if zero_grad and fresh_grad:
            from . import grad_sink
            cov = {id(p) for p in covered} | {id(p) for p in held_params}          # (zeroed by the launch, or explicitly above)
            for p in fresh_grad:
                if id(p) in cov and id(p) not in keep_ids and p.grad is not None:
                    if not getattr(p, "_gp_fresh_hook", False):      # (once per Parameter, whatever optimizer objects come and go)
                        # a contribution that arrives through autograd instead ends the buffer's freshness
                        p.register_post_accumulate_grad_hook(lambda q: grad_sink.unmark_fresh(q.grad))
                        p._gp_fresh_hook = True
                    grad_sink.mark_fresh(p.grad)
        if self.shard is not None and zero_grad:
            # the launch zeroed this rank's slices only; the slices the other ranks own hold this rank's partial sums still.
            # Only regions this launch covered (`only` / `exclude`), on the stream of the launch.
            act_ids = {id(p) for p in covered}
            for region in self.bucket.regions:
                ps = [self.bucket.params[i] for i in region[2]]
                if not any(id(p) in act_ids for p in ps):
                    continue
                if all(id(p) in keep_ids for p in ps):
                    continue                     # every tensor of the region is overwritten by its producer next step
                if stream is not None:
                    with torch.cuda.stream(stream):
                        self.bucket.flat[region[0]:region[1]].zero_()
                else:
                    self.bucket.flat[region[0]:region[1]].zero_()
