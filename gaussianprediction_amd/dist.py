"""View-parallel data parallelism: one process per GPU, Gaussians + MLP
replicated, each rank renders a different camera, gradients are SUMMED across ranks -- exactly the
reference's single-GPU `--batch` accumulation (`loss_ = stack(batch_loss).sum()`, [REF train.py:113-119])
run in parallel.  One flat fp32 bucket holds every parameter's .grad (views into it), so the exchange
is a single RCCL all-reduce over xGMI with no packing copies; `radii` are combined with MAX and the
visibility filter follows from it [REF train.py:121-122].
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def active(group=None) -> bool:
    """True when gradients / parameters have to be exchanged: a process group with more than one rank.
    GP_DIST_FORCE_SINGLE=1 (test hook) also runs every collective on a ONE-rank group, so that the RCCL code path
    (in-place reduce_scatter_tensor / all_gather_into_tensor, hooks, stream waits) is exercised on a single-GPU box."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("GP_DIST_FORCE_SINGLE") == "1"


class FlatGradBucket:
    """Makes every parameter's .grad a view into one contiguous buffer.

    `shards` > 1 lays the buffer out for a sharded optimizer (reduce-scatter -> 1/shards of Adam per rank -> all-gather):
    the buffer is a sequence of REGIONS, each a multiple of 64 * shards elements so that rank r owns the r-th equal slice of
    every region -- one region per large tensor (its gradient can be exchanged the moment it is final), one region for all
    the small tensors together (MLP weights, keypoints: one collective instead of twenty).  `flat_params` additionally moves
    the parameters' storage into a second buffer of the same layout (p.data becomes a view), the all-gather operand."""

    def __init__(self, params, shards=1, flat_params=False, small_numel=1 << 20, groups=None):
        """`groups` (sharded layout only): lists of parameters that share ONE region each, in the given order -- one reduce-scatter
        and one all-gather per group instead of one per tensor (xGMI collectives have a fixed cost of tens of microseconds:
        fewer, larger ones).  Tensors in no group: a region of their own when large, the common tail region when small."""
        params = [p for p in params if p.requires_grad]
        self.shards = int(shards)
        self.small_numel = int(small_numel)
        regioned = self.shards > 1 or flat_params   # (a one-rank group under GP_DIST_FORCE_SINGLE keeps the sharded layout)
        plan = []                                   # regioned: [[params of region 0], ...], the tail region last
        if regioned:
            known = {id(p) for p in params}
            grouped = set()
            for g in (groups or []):
                members = [p for p in g if id(p) in known and id(p) not in grouped]
                if members and sum(p.numel() for p in members) >= small_numel:
                    plan.append(members)
                    grouped.update(id(p) for p in members)
            rest = [p for p in params if id(p) not in grouped]
            plan += [[p] for p in rest if p.numel() >= small_numel]
            tail = [p for p in rest if p.numel() < small_numel]
            if tail:
                plan.append(tail)
            params = [p for region in plan for p in region]
        self.tail_start = None                      # start of the region of small tensors (exchanged after backward, never early)
        self.params = params
        self.offsets, self.regions = [], []         # regions: (start, end, [param indices])
        unit = 64 * self.shards
        n = 0
        if regioned:
            k = 0
            for members in plan:
                start, idx = n, []
                for p in members:                   # 64-element (256 B) aligned segments inside the region
                    self.offsets.append(n)
                    idx.append(k)
                    k += 1
                    n += (p.numel() + 63) // 64 * 64
                end = start + (n - start + unit - 1) // unit * unit
                self.regions.append((start, end, idx))
                if tail and members is plan[-1]:
                    self.tail_start = start
                n = end
        else:
            for p in params:                        # 64-element (256 B) aligned segments
                self.offsets.append(n)
                n += (p.numel() + 63) // 64 * 64
            self.regions.append((0, n, list(range(len(params)))))
        dev = params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.pflat = None
        if flat_params:
            self.pflat = torch.zeros(n, dtype=torch.float32, device=dev)
            for p, off in zip(params, self.offsets):
                view = self.pflat[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
        for p, off in zip(params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero(self):
        self.flat.zero_()

    def rebind(self):
        """Re-attach views (needed if something replaced .grad, e.g. zero_grad(set_to_none=True))."""
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat[off:off + 1].data_ptr():
                p.grad = self.flat[off:off + p.numel()].view_as(p)

    def all_reduce_sum(self, group=None, async_op=False):
        if active(group):
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return None

    def segment(self, p):
        """The bucket slice that backs p.grad (padded to the 64-element granule)."""
        i = next(k for k, q in enumerate(self.params) if q is p)
        off = self.offsets[i]
        end = self.offsets[i + 1] if i + 1 < len(self.offsets) else self.flat.numel()
        return self.flat[off:end]

    def region_of(self, p):
        i = next(k for k, q in enumerate(self.params) if q is p)
        return next(r for r in self.regions if i in r[2])

    def shard_slice(self, region, rank):
        start, end, _ = region
        n = (end - start) // self.shards
        return start + rank * n, start + (rank + 1) * n


class ShardedExchange:
    """The gradient / parameter exchange of the sharded optimizer (ZeRO-1 shaped, sized for xGMI):

        backward  ->  reduce-scatter(SUM) of every region, each large tensor's the moment its gradient is final
                  ->  Adam on this rank's 1/world slice of every region (loss_ops.FusedAdam(shard=...))
                  ->  all-gather of the updated parameter slices, asynchronous: the small tensors and the per-Gaussian
                      geometry are awaited before the next deformation, the SH coefficients (3/4 of the bytes) only in front
                      of the next rasterizer call (renderer._wait_params), i.e. behind the next step's deformation.

    Against all-reduce + replicated Adam the bytes on the links are the same (an all-reduce IS a reduce-scatter + all-gather),
    but Adam's 0.32 ms at configs[2] shrinks by the world size and the all-gather half overlaps the next forward.
    SUM semantics = the reference's --batch accumulation [REF train.py:113-119].  Backends: RCCL ("nccl") uses in-place
    reduce_scatter_tensor / all_gather_into_tensor; gloo (CPU tests, two ranks on one GPU) has neither, so there the same
    result is produced with all_reduce / all_gather on the same views."""

    def __init__(self, bucket: FlatGradBucket, group=None):
        self.bucket, self.group = bucket, group
        self.enabled = active(group)
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        assert bucket.shards == self.world, "bucket layout and process group disagree on the number of shards"
        self.nccl = self.enabled and dist.get_backend(group) == "nccl"
        self.handles, self._fired, self._late = [], set(), set()
        self._gather = []                       # (param ids of the region, handle, late)
        self._sink_cb = None
        self.bytes_sent_per_step = 0
        # `chain(region, handle) -> bool`: set by the harness.  Called from the gradient hook right after a large region's
        # reduce-scatter has been issued; returning True takes the handle over (the harness runs that region's Adam and
        # all-gather on `side`, off the compute stream: train_step.TrainStep._chain_sh) and finish() / gather_params() skip it.
        self.chain = None
        self._chained = set()                   # region starts taken over this step
        self.side = torch.cuda.Stream(device=bucket.flat.device) if (self.enabled and bucket.flat.is_cuda) else None
        self._late_event, self._late_ids = None, set()
        # optional: how long the COMPUTE stream sits in each kind of wait (a pair of events around the wait: with nothing else queued
        # on the stream between them, their distance is the exposed part of the collective).  bench.py --time-waits.
        self.time_waits = False
        self._wait_events = []                  # (kind, before, after)
        self._issued = set()                    # region starts whose reduce-scatter the hooks have issued this step
        if self.enabled:
            from . import grad_sink
            # every region but the tail of small tensors may leave early: when ALL of its tensors' gradients are final
            self.large = [bucket.params[k] for r in bucket.regions if r[0] != bucket.tail_start for k in r[2]]
            hooks = {id(p): self._make_hook(p) for p in self.large}
            self._hook_handles = [p.register_post_accumulate_grad_hook(hooks[id(p)]) for p in self.large]
            self._sink_cb = grad_sink.register_callback(lambda p: hooks[id(p)](p) if id(p) in hooks else None)

    def _timed_wait(self, kind, handles):
        """h.wait() for every handle (the current stream waits for the collective), bracketed by events when time_waits is on."""
        handles = [h for h in handles if h is not None]
        if not handles:
            return
        if self.time_waits and self.bucket.flat.is_cuda:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for h in handles:
                h.wait()
            b.record()
            self._wait_events.append((kind, a, b))
        else:
            for h in handles:
                h.wait()

    def exposed_wait_ms(self, steps):
        """{kind: ms per step the compute stream spent waiting} over the waits recorded since the last call (synchronises)."""
        if not self._wait_events:
            return {}
        torch.cuda.synchronize(self.bucket.flat.device)
        out = {}
        for kind, a, b in self._wait_events:
            out[kind] = out.get(kind, 0.0) + a.elapsed_time(b)
        self._wait_events = []
        return {k: round(v / max(steps, 1), 4) for k, v in out.items()}

    def close(self):
        """Detach from the parameters and the gradient-sink registry and drain what is in flight (call before the bucket is
        rebuilt: a late all-gather would otherwise still be writing the old parameter buffer, and a hook left on a Parameter
        that survives the rebuild would fire a stray reduce-scatter on the old bucket every step)."""
        if self._sink_cb is not None:
            from . import grad_sink
            grad_sink.unregister_callback(self._sink_cb)
            self._sink_cb = None
        for h in getattr(self, "_hook_handles", ()):
            h.remove()
        self._hook_handles = []
        for h in self.handles:
            h.wait()
        self.handles.clear()
        self.wait_params()
        if self.side is not None:
            torch.cuda.current_stream(self.bucket.flat.device).wait_stream(self.side)
        self.chain = None
        self.enabled = False

    def set_late(self, params):
        self._late = {id(p) for p in params}

    def _reduce_scatter(self, region):
        start, end, _ = region
        seg = self.bucket.flat[start:end]
        if self.nccl:
            lo, hi = self.bucket.shard_slice(region, self.rank)
            return dist.reduce_scatter_tensor(self.bucket.flat[lo:hi], seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _make_hook(self, p):
        region = self.bucket.region_of(p)
        members = [id(self.bucket.params[k]) for k in region[2]]

        def hook(param):
            if not self.enabled or id(p) in self._fired or id(p) in self._late:
                return
            self._fired.add(id(p))
            if any(m in self._late for m in members) or not all(m in self._fired for m in members):
                return                              # the region leaves with its last tensor -- or, with a late member, in finish()
            self._issued.add(region[0])
            h = self._reduce_scatter(region)
            if self.chain is not None and self.chain(region, h):
                self._chained.add(region[0])
            else:
                self.handles.append(h)
        return hook

    def finish(self):
        """After loss.backward(): reduce-scatter what the hooks did not cover; every rank then holds the SUM over ranks in its
        own slice of every region."""
        if not self.enabled:
            return
        for region in self.bucket.regions:
            if region[0] not in self._issued:
                self.handles.append(self._reduce_scatter(region))
        self._timed_wait("reduce_scatter", self.handles)
        self.handles.clear()
        self._fired.clear()
        self._issued.clear()
        n = self.bucket.flat.numel()
        self.bytes_sent_per_step = 2 * 4 * n * (self.world - 1) // self.world     # reduce-scatter + all-gather, per rank

    def gather_region(self, region, late=False):
        """Start the all-gather of ONE region's updated parameter slices, ordered behind the work already on the CURRENT stream
        (a collective waits for the stream it is issued under).  `late`: render() does not wait for it on the compute stream but
        hands the rasterizer an event instead (late_event)."""
        pf = self.bucket.pflat
        start, end, idx = region
        lo, hi = self.bucket.shard_slice(region, self.rank)
        if self.nccl:
            h = dist.all_gather_into_tensor(pf[start:end], pf[lo:hi], group=self.group, async_op=True)
        else:
            n = (end - start) // self.world
            outs = [pf[start + k * n:start + (k + 1) * n] for k in range(self.world)]
            h = dist.all_gather(outs, pf[lo:hi].clone(), group=self.group, async_op=True)
        self._gather.append(({id(self.bucket.params[k]) for k in idx}, h, bool(late)))

    def gather_params(self):
        """Start the all-gather of the updated parameter slices (after the optimizer step), small / geometry regions first;
        regions the harness chained this step (their gather is already under way) are skipped."""
        if not self.enabled:
            return
        order = sorted(self.bucket.regions, key=lambda r: r[1] - r[0])          # smallest first: the SH regions come last
        for region in order:
            if region[0] not in self._chained:
                self.gather_region(region)
        self._chained.clear()

    def wait_params(self, only=None, exclude=None):
        """Make the current stream wait for the gathered parameters (`only` / `exclude`: sets of id(param))."""
        rest, now = [], []
        for ids, h, late in self._gather:
            if (only is not None and not (ids & only)) or (exclude is not None and ids <= exclude):
                rest.append((ids, h, late))
            else:
                now.append(h)
        self._timed_wait("all_gather", now)
        self._gather = rest
        # late gathers handed to a rasterizer call as an event (late_event) are awaited on the SIDE stream only; if that call
        # never consumed the event (no Gaussians, an exception before the kernel) nothing else orders the compute stream behind
        # them -- so any wait that covers their tensors (checkpoint, save_ply, surgery: only=None) also waits for the event
        ev = self._late_event
        if ev is not None and (only is None or (self._late_ids & only)) and not (exclude is not None and self._late_ids <= exclude):
            torch.cuda.current_stream(self.bucket.flat.device).wait_event(ev)
            self._late_event = None

    def late_event(self):
        """For render(): wait -- on the current stream -- for every outstanding gather EXCEPT the late ones (the SH coefficients),
        and return a torch.cuda.Event that fires when those have landed (None if there are none): the rasterizer forward waits
        for it in front of its SH -> RGB kernel only (gp_raster_settings.sh_ready_event)."""
        late = [g for g in self._gather if g[2]]
        self._timed_wait("all_gather", [g[1] for g in self._gather if not g[2]])
        self._gather = []
        if not late:
            return None
        if self.side is None:                   # (CPU tensors: nothing to overlap with)
            for g in late:
                g[1].wait()
            return None
        with torch.cuda.stream(self.side):
            for g in late:
                g[1].wait()                     # blocks the SIDE stream until the gather is complete
            ev = torch.cuda.Event()
            ev.record(self.side)
        self._late_event, self._late_ids = ev, set().union(*[g[0] for g in late])
        return ev


SH_C0 = 0.28209479177387814


def sh_basis(dirs: torch.Tensor, degree: int) -> torch.Tensor:
    """Y_k(dir), k < (degree + 1)^2, as [n, (degree + 1)^2]: the real SH basis of [REF utils/sh_utils.py:57-112] (the statement of what
    gp_sh_factor_gradient evaluates; used by the tests and by the CPU seam of the factorised exchange)."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    xx, yy, zz = x * x, y * y, z * z
    c1, c2, c3 = 0.4886025119029199, (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396), \
        (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277, -0.5900435899266435)
    Y = [torch.full_like(x, SH_C0), -c1 * y, c1 * z, -c1 * x,
         c2[0] * (x * y), c2[1] * (y * z), c2[2] * (2 * zz - xx - yy), c2[3] * (x * z), c2[4] * (xx - yy),
         c3[0] * y * (3 * xx - yy), c3[1] * (x * y) * z, c3[2] * y * (4 * zz - xx - yy), c3[3] * z * (2 * zz - 3 * xx - 3 * yy),
         c3[4] * x * (4 * zz - xx - yy), c3[5] * z * (xx - yy), c3[6] * x * (xx - 3 * yy)]
    return torch.stack(Y[:(degree + 1) ** 2], dim=1)


# CPU tensors (the gloo tests) have no kernel: the tests install a torch restatement here (tests/host_checkers.py); the product itself
# never falls back -- without the seam a CPU call raises
host_sh_factor_gradient = None


def sh_factor_gradient(factors: torch.Tensor, degree: int, g_dc: torch.Tensor, g_rest: torch.Tensor):
    """g_dc [n,1,3] / g_rest [n,15,3] <- the sum over the views (dim 0 of `factors` [world, n, 6] = (dL/dRGB | unit view direction), in
    order) of the rank-one SH gradients (gp_sh_factor_gradient)."""
    world, n, _ = factors.shape
    if factors.is_cuda:
        import ctypes as C
        from . import _lib
        assert factors.is_contiguous() and g_dc.is_contiguous() and g_rest.is_contiguous() and factors.dtype == torch.float32
        with _lib.on_device(factors.device):
            _lib.check(_lib.lib().gp_sh_factor_gradient(C.c_int64(n), C.c_int32(world), _lib.ptr(factors), C.c_int32(int(degree)), _lib.ptr(g_dc),
                                                        _lib.ptr(g_rest), _lib.stream_ptr(factors.device)), "gp_sh_factor_gradient")
        return
    if host_sh_factor_gradient is None:
        raise RuntimeError("sh_factor_gradient: CPU tensors have no implementation (the HIP library is the only one)")
    host_sh_factor_gradient(factors, degree, g_dc, g_rest)


class OverlappedGradReducer:
    """SUM-all-reduce of the gradient bucket, overlapped with the rest of backward.

    The rasterizer backward finishes the largest gradients first (SH coefficients: 192 of the 364
    bytes per Gaussian) while the deformation backward (blend, MLP) and the Adam launches are still
    ahead.  A post-accumulate-grad hook on every large leaf starts an asynchronous all-reduce of that
    leaf's bucket segment the moment its gradient is final; RCCL runs it on its own stream over xGMI
    while the compute stream continues.  Small tensors (MLP weights, keypoints: < `small_numel`) are
    reduced together in one trailing collective.  `finish()` makes the compute stream wait for all of
    them; the optimizer then sees exactly sum-over-ranks, i.e. the reference's `--batch` semantics
    [REF train.py:113-119].  Every rank issues the same collectives in the same order (the order is
    fixed by the autograd graph, which is identical on all ranks)."""

    def __init__(self, bucket: FlatGradBucket, group=None, small_numel=1 << 20):
        self.bucket, self.group = bucket, group
        self.handles = []
        self.enabled = active(group)
        # FACTORISED SH exchange (set_factorised): the two SH tensors' gradients are not all-reduced; the ranks all-gather
        # (dL/dRGB, view direction) per Gaussian instead and every rank forms the sum over the views itself
        self._factor = None                     # (features_dc, features_rest, dirs_fn, degree_fn)
        self._factor_pending = None             # (handle, gathered factors) of this step
        self.factor_bytes_received_per_step = 0
        self.large = [p for p in bucket.params if p.numel() >= small_numel]
        self.small = [p for p in bucket.params if p.numel() < small_numel]
        self._fired = set()
        self._late = set()
        self._sink_cb = None
        self._hooks = {}
        if self.enabled:
            from . import grad_sink
            hooks = self._hooks
            hooks.update({id(p): self._make_hook(p) for p in self.large})
            self._hook_handles = [p.register_post_accumulate_grad_hook(hooks[id(p)]) for p in self.large]
            # leaves whose gradient is written directly by a kernel (grad_sink) never run AccumulateGrad:
            # they announce completion through the sink registry instead
            self._sink_cb = grad_sink.register_callback(lambda p: hooks[id(p)](p) if id(p) in hooks else None)

    def close(self):
        """Detach from the parameters and the gradient-sink registry (call before the bucket is rebuilt)."""
        if getattr(self, "_sink_cb", None) is not None:
            from . import grad_sink
            grad_sink.unregister_callback(self._sink_cb)
            self._sink_cb = None
        for h in getattr(self, "_hook_handles", ()):
            h.remove()
        self._hook_handles = []
        for h in self.handles:
            h.wait()
        self.handles.clear()
        self.enabled = False

    def set_late(self, params):
        """Leaves whose gradient has more than one producer in the coming backward (several views per step; the lifecycle
        opacity's second MLP pass into `_xyz`): a producer's completion notice is not "final" for them, so they are left to
        finish(), which runs after backward."""
        self._late = {id(p) for p in params}

    def set_factorised(self, features_dc, features_rest, dirs_fn, degree_fn):
        """Exchange the SH gradients in factorised form from now on (single-view steps only: a rank's gradient must be ONE view's).
        `dirs_fn()` -> [n, 3] unit view directions of this rank's view (the ones the rasterizer evaluated the SH basis at),
        `degree_fn()` -> the active SH degree.  Per Gaussian and step a rank receives 24 (world - 1) bytes instead of sending and
        receiving 2 x 192 (world - 1) / world: 168 against 336 B at world 8 -- the price is that every rank keeps (as in this
        replicated-optimizer path anyway) the full SH moments and streams them every step."""
        self._factor = (features_dc, features_rest, dirs_fn, degree_fn)
        for p in (features_dc, features_rest):          # (small test scenes: the two tensors must be hooked leaves, not part of the tail)
            if any(q is p for q in self.small):
                self.small = [q for q in self.small if q is not p]
                self.large.append(p)
                if self.enabled:
                    self._hooks[id(p)] = self._make_hook(p)
                    self._hook_handles.append(p.register_post_accumulate_grad_hook(self._hooks[id(p)]))

    def _factor_ids(self):
        return () if self._factor is None else (id(self._factor[0]), id(self._factor[1]))

    def _start_factor_gather(self):
        dc, rest, dirs_fn, _ = self._factor
        world = dist.get_world_size(self.group)
        n = dc.shape[0]
        mine = torch.cat([dc.grad.reshape(n, 3) * (1.0 / SH_C0), dirs_fn().reshape(n, 3).to(dc.grad.dtype)], dim=1).contiguous()
        out = torch.empty(world, n, 6, dtype=mine.dtype, device=mine.device)
        if dist.get_backend(self.group) == "nccl":
            h = dist.all_gather_into_tensor(out.view(-1), mine.view(-1), group=self.group, async_op=True)
        else:
            h = dist.all_gather([out[k] for k in range(world)], mine, group=self.group, async_op=True)
        self._factor_pending = (h, out)
        self.factor_bytes_received_per_step = 24 * n * (world - 1)

    def _make_hook(self, p):
        def hook(param):
            if not self.enabled or id(p) in self._fired or id(p) in self._late:
                return
            self._fired.add(id(p))
            fid = self._factor_ids()
            if id(p) in fid:
                if all(i in self._fired for i in fid) and self._factor_pending is None:
                    self._start_factor_gather()          # both SH tensors' gradients are final (one kernel writes them)
                return
            self.handles.append(dist.all_reduce(self.bucket.segment(p), op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return hook

    def finish(self):
        """Call after loss.backward(): reduces what the hooks did not cover, waits for everything."""
        if not self.enabled:
            return
        fid = self._factor_ids()
        if fid and self._factor_pending is None:
            if any(i in self._late for i in fid):
                raise RuntimeError("factorised SH exchange: the SH gradients of this step have several producers (more than one view per step?)")
            for i in fid:
                self._fired.add(i)
            self._start_factor_gather()            # (the hooks did not fire: e.g. nothing visible -- the collective still needs every rank)
        for p in self.large:                       # leaves that received no gradient this step (still zero)
            if id(p) not in self._fired:
                self.handles.append(dist.all_reduce(self.bucket.segment(p), op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if self.small:
            # the small tensors are contiguous at the end of the bucket when they were registered last;
            # otherwise reduce them one by one (they are tiny)
            idx = [next(k for k, q in enumerate(self.bucket.params) if q is p) for p in self.small]
            if idx == list(range(idx[0], idx[0] + len(idx))):
                off = self.bucket.offsets[idx[0]]
                end = self.bucket.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.bucket.offsets) else self.bucket.flat.numel()
                self.handles.append(dist.all_reduce(self.bucket.flat[off:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                for p in self.small:
                    self.handles.append(dist.all_reduce(self.bucket.segment(p), op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in self.handles:
            h.wait()
        self.handles.clear()
        if self._factor_pending is not None:       # the sum over the views, in rank order: identical on every rank
            h, factors = self._factor_pending
            h.wait()
            dc, rest, _, degree_fn = self._factor
            sh_factor_gradient(factors, int(degree_fn()), dc.grad, rest.grad)
            self._factor_pending = None
        self._fired.clear()


def reduce_view_stats(radii: torch.Tensor, group=None):
    """radii = elementwise max over the views of the batch; visibility = radii > 0
    [REF train.py:121-122]."""
    if active(group):
        dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=group)
    return radii, radii > 0


def init_from_env(backend: str | None = None):
    """torchrun-style init (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GP_FORCE_LOCAL_RANK") is not None:     # test hook: several ranks on one GPU (gloo only)
        local = int(os.environ["GP_FORCE_LOCAL_RANK"])
    if (world > 1 or os.environ.get("GP_DIST_FORCE_SINGLE") == "1") and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("GP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" IS RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
