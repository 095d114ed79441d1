"""torch.autograd.Function wrappers over the deformation entry points of libgp_hip.so.

These are the HIP replacements of the PyTorch-composed ops on the reference's per-frame path
(SURVEY.md section 2.2(c)): positional encoding + concat + MLP [REF scene/gaussian_model.py:180-189,
scene/deformable_field.py:63-72,102-127], keypoint blend + quaternion compose
[REF scene/gaussian_model.py:214-229,266-273,285-286,314-315], activations
[REF scene/gaussian_model.py:41-51,291-298].  No CPU fallback: inputs must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, grad_sink


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor on {t.device}; the deformation path only exists as HIP kernels (no CPU fallback)")


def _c(t):
    return t.detach().to(torch.float32).contiguous()


_PACK_CACHE = {}           # id(layer-0 weight) -> (key, packed tensor)
SMALL_ROWS = 2048          # csrc/gp_capi_deform.hip: GP_MLP_SMALL_ROWS
FORCE_PACKED = False       # (tests: use the packed copy under autograd too)
FORCE_ROW_TILES = False    # (tests, A/B: the 16-row kernels where the feature-split forward would run)


def packed_weights(wb, ws):
    """The fragment-ordered copy of the four hidden-layer weight matrices for the small-row kernels (gp_mlp_pack), rebuilt when
    a weight's version counter has moved (the optimizer kernels bump them; load_state_dict copies in place)."""
    leaves = wb[0:8:2]
    key = tuple((id(w), w._version, w.data_ptr()) for w in leaves)
    c = _PACK_CACHE.get(id(leaves[0]))
    if c is not None and c[0] == key:
        return c[1]
    dev = ws[0].device
    in_dim = ws[0].shape[1]
    n = int(_lib.lib().gp_mlp_packed_floats(C.c_int32(in_dim)))
    if n <= 0:
        return None
    pk = c[1] if (c is not None and c[1].numel() == n and c[1].device == dev) else torch.empty(n, device=dev)
    params = _lib.MlpParamsC(in_dim, 256, 4, 7)
    for l in range(4):
        params.w[l] = ws[l].data_ptr()
    with _lib.on_device(dev):
        _lib.check(_lib.lib().gp_mlp_pack(C.byref(params), _lib.ptr(pk), _lib.stream_ptr(dev)), "gp_mlp_pack")
    if len(_PACK_CACHE) > 64:
        _PACK_CACHE.clear()
    _PACK_CACHE[id(leaves[0])] = (key, pk)
    return pk


SPLIT_ROWS = 512           # csrc/gp_capi_deform.hip: GP_MLP_SPLIT_ROWS
_SCRATCH = {}              # (device index, stream) -> zero-initialised scratch of the feature-split small-row forward


def mlp_scratch(dev, rows):
    """gp_mlp_params.scratch for a pass over `rows` rows on the CURRENT stream of `dev` (None where the library would not use one): one
    buffer per (device, stream) -- calls on one stream run one after the other, and the library returns the counters to zero at the end
    of every call -- sized for the largest row count the feature-split kernel serves, zeroed once."""
    if not (0 < rows <= SPLIT_ROWS) or FORCE_PACKED or FORCE_ROW_TILES:
        return None
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(torch.cuda.current_stream(dev).cuda_stream))
    t = _SCRATCH.get(key)
    if t is None:
        n = int(_lib.lib().gp_mlp_scratch_bytes(C.c_int64(SPLIT_ROWS)))
        if n <= 0:
            return None
        t = torch.zeros((n + 3) // 4, dtype=torch.int32, device=dev)
        if len(_SCRATCH) > 64:
            _SCRATCH.clear()
        _SCRATCH[key] = t
    return t


def _input_sink(leaf, shape):
    """The .grad buffer of `leaf` if a kernel may WRITE this backward's gradient into it: an eligible leaf whose buffer is marked
    fresh (zeros, unwritten since the optimizer's zeroing pass: grad_sink.mark_fresh; the mark is consumed), else None."""
    g = grad_sink.sink_of(leaf)
    if g is None or tuple(g.shape) != tuple(shape) or not grad_sink.take_fresh(g):
        return None
    return g


class FusedMlp(torch.autograd.Function):
    """out = Deformable_Field(cat[feature, PE(xyz, xyz_freq), PE(t, time_freq)])  (d=4, w=256)."""

    @staticmethod
    def forward(ctx, feature, xyz, t, xyz_freq, time_freq, *wb):
        _need_cuda(feature, "FusedMlp")
        dev = feature.device
        ws = [_c(w) for w in wb[0::2]]
        bs = [_c(b) for b in wb[1::2]]
        feature_c = _c(feature)
        xyz_c = _c(xyz) if xyz is not None else None
        t_c = _c(t).reshape(-1)[:1] if t is not None else None
        rows, fd = feature_c.shape
        in_dim, out_dim = ws[0].shape[1], ws[4].shape[0]
        in_pad = (in_dim + 7) // 8 * 8
        need_grad = any(x is not None and torch.is_tensor(x) and x.requires_grad for x in (feature, xyz) + tuple(wb))
        out = torch.empty(rows, out_dim, device=dev)
        x_floats = (rows * in_pad + 63) // 64 * 64
        acts = torch.empty(x_floats + 4 * rows * 256 + 4 * rows * 8, device=dev) if need_grad else None   # x | h | ReLU bit masks
        params = _lib.MlpParamsC(in_dim, 256, 4, out_dim)
        for l in range(5):
            params.w[l] = ws[l].data_ptr()
            params.b[l] = bs[l].data_ptr()
        # Used where the copy is free: passes without autograd (evaluation: the weights stand still, one pack serves every frame;
        # forward 0.040 -> 0.035 ms at 250 rows).  In training the weights change every step and the 0.9 MB repack (6 us, a launch
        # of its own) costs what the faster forward + backward save (measured: 5 + 2 us) -- there the kernels read w[] directly.
        # (rows <= 512: the feature-split kernel reads w[] -- 16 KB per layer and workgroup -- and needs no copy, only its scratch)
        scratch = mlp_scratch(dev, rows)
        pk = packed_weights(wb, ws) if (0 < rows <= SMALL_ROWS and scratch is None and (not need_grad or FORCE_PACKED)) else None
        params.packed = pk.data_ptr() if pk is not None else None
        params.scratch = scratch.data_ptr() if scratch is not None else None
        inp = _lib.MlpInputC(rows, fd, int(xyz_freq), int(time_freq), feature_c.data_ptr(),
                             xyz_c.data_ptr() if xyz_c is not None else None, t_c.data_ptr() if t_c is not None else None)
        with _lib.on_device(dev):
            rc = _lib.lib().gp_mlp_forward(C.byref(params), C.byref(inp), _lib.ptr(out), _lib.ptr(acts), _lib.stream_ptr(dev))
            _lib.check(rc, "gp_mlp_forward")
        if need_grad:
            ctx.save_for_backward(feature_c, xyz_c if xyz_c is not None else torch.empty(0, device=dev),
                                  t_c if t_c is not None else torch.empty(0, device=dev), acts, *ws, *bs)
            ctx.meta = (int(xyz_freq), int(time_freq), xyz_c is not None, t_c is not None)
            ctx.needs = (feature.requires_grad, xyz is not None and xyz.requires_grad)
            ctx.wb_leaves = tuple(wb)
            ctx.in_leaves = (feature, xyz)      # (their .grad may take the input gradients directly: _input_sink)
            ctx.pk = pk              # (the weights do not change between a forward and its backward)
            ctx.split = scratch is not None
        return out

    @staticmethod
    def backward(ctx, g_out):
        saved = ctx.saved_tensors
        feature_c, xyz_c, t_c, acts = saved[:4]
        ws, bs = saved[4:9], saved[9:14]
        xyz_freq, time_freq, has_xyz, has_t = ctx.meta
        dev = feature_c.device
        rows, fd = feature_c.shape
        in_dim, out_dim = ws[0].shape[1], ws[4].shape[0]
        g = g_out.to(torch.float32).contiguous()
        params = _lib.MlpParamsC(in_dim, 256, 4, out_dim)
        grads = _lib.MlpGradsC()
        # the kernels "+=" into dW/db: when every weight owns an allocated .grad, use it directly
        leaves = ctx.wb_leaves
        sinks = [grad_sink.sink_of(t) for t in leaves]
        use_sink = all(s_ is not None for s_ in sinks)
        if use_sink:
            dws, dbs = list(sinks[0::2]), list(sinks[1::2])
        else:
            dws = [torch.zeros_like(w) for w in ws]
            dbs = [torch.zeros_like(b) for b in bs]
        for l in range(5):
            params.w[l] = ws[l].data_ptr()
            params.b[l] = bs[l].data_ptr()
            grads.dw[l] = dws[l].data_ptr()
            grads.db[l] = dbs[l].data_ptr()
        params.packed = ctx.pk.data_ptr() if getattr(ctx, "pk", None) is not None else None
        scratch = mlp_scratch(dev, rows) if getattr(ctx, "split", False) else None     # (the feature-split data backward, as the forward)
        params.scratch = scratch.data_ptr() if scratch is not None else None
        inp = _lib.MlpInputC(rows, fd, xyz_freq, time_freq, feature_c.data_ptr(), xyz_c.data_ptr() if has_xyz else None,
                             t_c.data_ptr() if has_t else None)
        need_f, need_x = ctx.needs
        # The input gradients are WRITTEN (not accumulated) by the kernels.  Where the input is a leaf whose .grad buffer is marked
        # fresh (zeros since the optimizer's pass, nobody has written: TrainStep asks FusedAdam to mark the keypoint tensors), the
        # kernel writes straight into it and autograd gets None: no temporary, no AccumulateGrad add launch (three 4 us kernels per
        # step on K x 32 / K x 3 tensors, round 4).  A later contribution to the same leaf finds no mark and is accumulated by autograd.
        f_leaf, x_leaf = getattr(ctx, "in_leaves", (None, None))
        f_sink = _input_sink(f_leaf, (rows, fd)) if need_f else None
        x_sink = _input_sink(x_leaf, (rows, 3)) if (need_x and has_xyz and xyz_freq > 0) else None
        g_feat = f_sink if f_sink is not None else (torch.empty(rows, fd, device=dev) if need_f else None)
        g_xyz = x_sink if x_sink is not None else (torch.empty(rows, 3, device=dev) if (need_x and has_xyz and xyz_freq > 0) else None)
        alloc = _lib.TorchAllocator(dev)
        with _lib.on_device(dev):
            rc = _lib.lib().gp_mlp_backward(C.byref(params), C.byref(inp), _lib.ptr(acts), _lib.ptr(g), C.byref(grads),
                                            _lib.ptr(g_feat), _lib.ptr(g_xyz), alloc.cb, None, _lib.stream_ptr(dev))
            if alloc.error is not None:
                err = alloc.error
                alloc.release()
                raise err
            alloc.release()
            _lib.check(rc, "gp_mlp_backward")
        wb_grads = []
        for l in range(5):
            wb_grads += [None, None] if use_sink else [dws[l], dbs[l]]
        if use_sink:
            for t in leaves:
                grad_sink.notify(t)
        if f_sink is not None:
            grad_sink.notify(f_leaf)
            g_feat = None
        if x_sink is not None:
            grad_sink.notify(x_leaf)
            g_xyz = None
        return (g_feat, g_xyz, None, None, None, *wb_grads)


def _to16(w, tdt, split):
    """16-bit copy of a weight matrix; split mode: [hi copy, lo' copy] with hi = fp16(w) (zero below the fp16 normal
    range) and lo' = fp16((w - hi) * 2^11) (see GP_DTYPE_F16_SPLIT in include/gp_hip.h).  The torch statement of what
    gp_mlp16_pack does in one launch (kept as the checker of tests/test_gpu_deform.py::test_mlp16_pack_matches_the_torch_form)."""
    def pack(m):            # [F, K] (F % 32 == 0, K % 16 == 0) -> fragment order [k-step][feature tile][half][j][8] (include/gp_hip.h)
        F_, K_ = m.shape
        return m.reshape(F_ // 32, 32, K_ // 16, 2, 8).permute(2, 0, 3, 1, 4).contiguous()
    w = w.contiguous()
    if w.shape[0] % 32:
        w = torch.cat([w, w.new_zeros(32 - w.shape[0] % 32, w.shape[1])])
    if not split:
        return pack(w.to(tdt))
    c = w.clamp(-65504.0, 65504.0)
    hi = torch.where(c.abs() < 6.103515625e-05, torch.zeros_like(c), c).to(torch.float16)
    lo = ((c - hi.float()) * 2048.0).to(torch.float16)
    return torch.stack([pack(hi), pack(lo)]).contiguous()


def _torch_w16(ws, tdt, split, transposed):
    """The five 16-bit weight copies built with torch ops (the form of rounds 1-5; the checker of gp_mlp16_pack)."""
    dev = ws[0].device
    in_dim, out_dim = ws[0].shape[1], ws[4].shape[0]
    in_pad = (in_dim + 15) // 16 * 16
    if not transposed:
        w0p = torch.zeros(256, in_pad, device=dev)
        w0p[:, :in_dim] = ws[0]
        w4p = torch.zeros(32, 256, device=dev)
        w4p[:out_dim] = ws[4]
        return [_to16(w, tdt, split) for w in (w0p, ws[1], ws[2], ws[3], w4p)]
    wt0 = torch.zeros(in_pad, 256, device=dev)
    wt0[:in_dim] = ws[0].t()
    wt4 = torch.zeros(256, 16, device=dev)
    wt4[:, :out_dim] = ws[4].t()
    return [_to16(w, tdt, split) for w in (wt0, ws[1].t(), ws[2].t(), ws[3].t(), wt4)]


def packed_w16(ws, tdt, cdt, transposed):
    """The five zero-padded, fragment-packed 16-bit weight copies of gp_mlp16_forward (transposed=False) / gp_mlp16_backward (True):
    one buffer, one launch (gp_mlp16_pack).  Returns (buffer, [data pointers])."""
    dev = ws[0].device
    in_dim, out_dim = ws[0].shape[1], ws[4].shape[0]
    ns = 2 if cdt == _lib.GP_DTYPE_F16_SPLIT else 1
    L = _lib.lib()
    sizes = [ns * int(L.gp_mlp16_packed_elems(C.c_int32(l), C.c_int32(in_dim), C.c_int32(1 if transposed else 0))) for l in range(5)]
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + (n + 127) // 128 * 128)          # 256-byte aligned pieces
    buf = torch.empty(offs[-1], device=dev, dtype=tdt)
    params = _lib.MlpParamsC(in_dim, 256, 4, out_dim)
    for l in range(5):
        params.w[l] = ws[l].data_ptr()
    ptrs = (C.c_void_p * 5)(*[buf.data_ptr() + 2 * offs[l] for l in range(5)])
    with _lib.on_device(dev):
        _lib.check(L.gp_mlp16_pack(C.byref(params), C.c_int32(cdt), C.c_int32(1 if transposed else 0), ptrs, _lib.stream_ptr(dev)), "gp_mlp16_pack")
    return buf, [int(ptrs[l]) for l in range(5)]


class FusedMlp16(torch.autograd.Function):
    """FusedMlp with 16-bit operands on the matrix cores (fp16 or bf16, fp32 accumulate; "fp32s" = split fp16 pairs with
    fp32-grade results).  Opt-in."""

    @staticmethod
    def forward(ctx, feature, xyz, t, xyz_freq, time_freq, precision, range_flag, *wb):
        """`range_flag`: None, or an int32 [1] device tensor the split-mode ("fp32s") kernel ORs 1 into when a hidden activation
        reached 2^15 (gp_mlp16_params.range_flag; Deformable_Field's range guard)."""
        _need_cuda(feature, "FusedMlp16")
        dev = feature.device
        tdt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32s": torch.float16}[precision]
        cdt = {"fp16": _lib.GP_DTYPE_F16, "bf16": _lib.GP_DTYPE_BF16, "fp32s": _lib.GP_DTYPE_F16_SPLIT}[precision]
        split = precision == "fp32s"
        ns = 2 if split else 1
        ws = [_c(w) for w in wb[0::2]]
        bs = [_c(b) for b in wb[1::2]]
        feature_c = _c(feature)
        xyz_c = _c(xyz) if xyz is not None else None
        t_c = _c(t).reshape(-1)[:1] if t is not None else None
        rows, fd = feature_c.shape
        in_dim, out_dim = ws[0].shape[1], ws[4].shape[0]
        in_pad = (in_dim + 15) // 16 * 16
        w16_buf, w16 = packed_w16(ws, tdt, cdt, False)
        need_grad = any(x is not None and torch.is_tensor(x) and x.requires_grad for x in (feature, xyz) + tuple(wb))
        out = torch.empty(rows, out_dim, device=dev)
        # blocked layout [row block of 16][feature][16]: rows zero-padded to 64, every tensor's extent padded to 128 rows (include/gp_hip.h)
        rows128 = (rows + 127) // 128 * 128
        xT = torch.empty(ns * in_pad * rows128, device=dev, dtype=tdt) if need_grad else None
        hT = torch.empty(ns * 4 * 256 * rows128, device=dev, dtype=tdt) if need_grad else None
        masks = torch.empty(4, rows, 8, device=dev, dtype=torch.int32) if need_grad else None
        params = _lib.Mlp16ParamsC(cdt, in_dim, 256, 4, out_dim)
        for l in range(5):
            params.w16[l] = w16[l]
            params.b[l] = bs[l].data_ptr()
        if split and range_flag is not None:
            params.range_flag = range_flag.data_ptr()
        inp = _lib.MlpInputC(rows, fd, int(xyz_freq), int(time_freq), feature_c.data_ptr(),
                             xyz_c.data_ptr() if xyz_c is not None else None, t_c.data_ptr() if t_c is not None else None)
        with _lib.on_device(dev):
            rc = _lib.lib().gp_mlp16_forward(C.byref(params), C.byref(inp), _lib.ptr(out), _lib.ptr(xT), _lib.ptr(hT), _lib.ptr(masks),
                                             _lib.stream_ptr(dev))
            _lib.check(rc, "gp_mlp16_forward")
        if need_grad:
            e = torch.empty(0, device=dev)
            ctx.save_for_backward(feature_c, xyz_c if xyz_c is not None else e, t_c if t_c is not None else e, xT, hT, masks, *ws, *bs)
            ctx.meta = (int(xyz_freq), int(time_freq), xyz_c is not None, t_c is not None, tdt, cdt, in_pad)
            ctx.needs = (feature.requires_grad, xyz is not None and xyz.requires_grad)
            ctx.wb_leaves = tuple(wb)
        return out

    @staticmethod
    def backward(ctx, g_out):
        saved = ctx.saved_tensors
        feature_c, xyz_c, t_c, xT, hT, masks = saved[:6]
        ws, bs = saved[6:11], saved[11:16]
        xyz_freq, time_freq, has_xyz, has_t, tdt, cdt, in_pad = ctx.meta
        dev = feature_c.device
        rows, fd = feature_c.shape
        in_dim, out_dim = ws[0].shape[1], ws[4].shape[0]
        g = g_out.to(torch.float32).contiguous()
        split = cdt == _lib.GP_DTYPE_F16_SPLIT
        wt_buf, wt = packed_w16(ws, tdt, cdt, True)
        params = _lib.Mlp16ParamsC(cdt, in_dim, 256, 4, out_dim)
        grads = _lib.MlpGradsC()
        leaves = ctx.wb_leaves
        sinks = [grad_sink.sink_of(t_) for t_ in leaves]
        use_sink = all(s_ is not None for s_ in sinks)
        if use_sink:
            dws, dbs = list(sinks[0::2]), list(sinks[1::2])
        else:
            dws = [torch.zeros_like(w) for w in ws]
            dbs = [torch.zeros_like(b) for b in bs]
        for l in range(5):
            params.w16[l] = wt[l]
            params.b[l] = bs[l].data_ptr()
            grads.dw[l] = dws[l].data_ptr()
            grads.db[l] = dbs[l].data_ptr()
        inp = _lib.MlpInputC(rows, fd, xyz_freq, time_freq, feature_c.data_ptr(), xyz_c.data_ptr() if has_xyz else None,
                             t_c.data_ptr() if has_t else None)
        need_f, need_x = ctx.needs
        g_feat = torch.empty(rows, fd, device=dev) if need_f else None
        g_xyz = torch.empty(rows, 3, device=dev) if (need_x and has_xyz and xyz_freq > 0) else None
        alloc = _lib.TorchAllocator(dev)
        with _lib.on_device(dev):
            rc = _lib.lib().gp_mlp16_backward(C.byref(params), C.byref(inp), _lib.ptr(xT), _lib.ptr(hT), _lib.ptr(masks), _lib.ptr(g),
                                              C.byref(grads), _lib.ptr(g_feat), _lib.ptr(g_xyz), alloc.cb, None, _lib.stream_ptr(dev))
            if alloc.error is not None:
                err = alloc.error
                alloc.release()
                raise err
            alloc.release()
            _lib.check(rc, "gp_mlp16_backward")
        wb_grads = []
        for l in range(5):
            wb_grads += [None, None] if use_sink else [dws[l], dbs[l]]
        if use_sink:
            for t_ in leaves:
                grad_sink.notify(t_)
        return (g_feat, g_xyz, None, None, None, None, None, *wb_grads)


def _overwrite_sink(leaf):
    """`leaf.grad` if it is a direct-sink buffer marked stale (grad_sink.mark_stale): the kernel then writes the gradient
    straight into it; None otherwise (the usual temporary + autograd accumulation)."""
    sk = grad_sink.sink_of(leaf)
    if sk is not None and grad_sink.take_stale(sk):
        return sk
    return None


def packed_idx16(knn_idx, K):
    """The neighbour indices as 16-bit words (gp_blend_args.knn_idx16): the reference's tensor is int64, 8 bytes per neighbour for
    values below K <= 512 -- 48 of the ~170 bytes per Gaussian each blend kernel moves.  Made once per index tensor state (the
    neighbour-search kernel emits it itself, weights_ops.knn_keypoints) and kept on the tensor."""
    if K > 65535 or knn_idx is None:
        return None
    key = (knn_idx._version, knn_idx.data_ptr(), tuple(knn_idx.shape))
    c = getattr(knn_idx, "_gp_idx16", None)
    if c is not None and c[0] == key:
        return c[1]
    t16 = knn_idx.detach().to(torch.int32).to(torch.int16).contiguous()      # (K < 65536: the low 16 bits are the index)
    try:
        knn_idx._gp_idx16 = (key, t16)
    except Exception:
        pass
    return t16


class KeypointBlend(torch.autograd.Function):
    """(xyz_t, q_t) from per-keypoint (nn>0) or per-Gaussian (raw_w is None) deltas."""

    @staticmethod
    def forward(ctx, delta, raw_w, knn_idx, xyz, rot, norm_rotation):
        _need_cuda(xyz, "KeypointBlend")
        dev = xyz.device
        delta_c, xyz_c, rot_c = _c(delta), _c(xyz), _c(rot)
        N = xyz_c.shape[0]
        if raw_w is not None:
            raw_c = _c(raw_w)
            idx_c = knn_idx.detach().to(torch.int64).contiguous()
            nn_ = idx_c.shape[1]
            K = delta_c.shape[0]
            if raw_c.shape != (N, 2 * nn_):
                raise RuntimeError("raw_w must be [N, 2*nearest_num]")
        else:
            raw_c, idx_c, nn_, K = None, None, 0, 0
        idx16 = packed_idx16(knn_idx, K) if raw_w is not None else None
        args = _lib.BlendArgsC(N, K, nn_, delta_c.shape[1], int(bool(norm_rotation)), delta_c.data_ptr(),
                               raw_c.data_ptr() if raw_c is not None else None,
                               idx_c.data_ptr() if idx_c is not None else None, xyz_c.data_ptr(), rot_c.data_ptr(),
                               idx16.data_ptr() if idx16 is not None else None)
        xyz_t = torch.empty(N, 3, device=dev)
        q_t = torch.empty(N, 4, device=dev)
        with _lib.on_device(dev):
            rc = _lib.lib().gp_blend_forward(C.byref(args), _lib.ptr(xyz_t), _lib.ptr(q_t), _lib.stream_ptr(dev))
            _lib.check(rc, "gp_blend_forward")
        e = torch.empty(0, device=dev)
        ctx.save_for_backward(delta_c, raw_c if raw_c is not None else e,
                              idx_c if idx_c is not None else torch.empty(0, dtype=torch.int64, device=dev), xyz_c, rot_c)
        ctx.meta = (nn_, K, int(bool(norm_rotation)))
        ctx.idx16 = idx16
        ctx.leaves = (xyz, rot)
        return xyz_t, q_t

    @staticmethod
    def backward(ctx, g_xyz_t, g_q_t):
        delta_c, raw_c, idx_c, xyz_c, rot_c = ctx.saved_tensors
        nn_, K, norm_rotation = ctx.meta
        dev = xyz_c.device
        N = xyz_c.shape[0]
        gx = g_xyz_t.to(torch.float32).contiguous() if g_xyz_t is not None else torch.zeros(N, 3, device=dev)
        gq = g_q_t.to(torch.float32).contiguous() if g_q_t is not None else torch.zeros(N, 4, device=dev)
        args = _lib.BlendArgsC(N, K, nn_, delta_c.shape[1], norm_rotation, delta_c.data_ptr(),
                               raw_c.data_ptr() if nn_ else None, idx_c.data_ptr() if nn_ else None, xyz_c.data_ptr(),
                               rot_c.data_ptr(), ctx.idx16.data_ptr() if (nn_ and ctx.idx16 is not None) else None)
        g_delta = torch.empty_like(delta_c)          # every element is written (stage 1: per row; stage 2/3: by the reduction)
        g_raw = torch.empty_like(raw_c) if (nn_ and ctx.needs_input_grad[1]) else None    # frozen weights: not computed
        # a leaf whose gradient buffer the optimizer left stale is OVERWRITTEN in place (no temporary + AccumulateGrad pass)
        sinks = [_overwrite_sink(t_) for t_ in ctx.leaves]
        g_xyz = sinks[0] if sinks[0] is not None else torch.empty(N, 3, device=dev)
        g_rot = sinks[1] if sinks[1] is not None else torch.empty(N, 4, device=dev)
        alloc = _lib.TorchAllocator(dev)
        with _lib.on_device(dev):
            rc = _lib.lib().gp_blend_backward(C.byref(args), _lib.ptr(gx), _lib.ptr(gq), _lib.ptr(g_delta), _lib.ptr(g_raw),
                                              _lib.ptr(g_xyz), _lib.ptr(g_rot), alloc.cb, None, _lib.stream_ptr(dev))
            if alloc.error is not None:
                err = alloc.error
                alloc.release()
                raise err
            alloc.release()
            _lib.check(rc, "gp_blend_backward")
        for t_, sk in zip(ctx.leaves, sinks):
            if sk is not None:
                grad_sink.notify(t_)
        return g_delta, g_raw, None, (None if sinks[0] is not None else g_xyz), (None if sinks[1] is not None else g_rot), None


class Activations(torch.autograd.Function):
    """scale = exp(_scaling); opacity = sigmoid(_opacity) [* sigmoid(delta[:, col] / beta)]."""

    @staticmethod
    def forward(ctx, scaling_raw, opacity_raw, delta, col, beta):
        _need_cuda(scaling_raw, "Activations")
        dev = scaling_raw.device
        s_c, o_c = _c(scaling_raw), _c(opacity_raw)
        N = s_c.shape[0]
        d_c = _c(delta) if delta is not None else None
        stride = d_c.shape[1] if d_c is not None else 0
        scale = torch.empty(N, 3, device=dev)
        opacity = torch.empty(N, 1, device=dev)
        dptr = C.c_void_p(d_c.data_ptr() + 4 * int(col)) if d_c is not None else None
        with _lib.on_device(dev):
            rc = _lib.lib().gp_activations_forward(C.c_int64(N), _lib.ptr(s_c), _lib.ptr(o_c), dptr, C.c_int32(stride),
                                                   C.c_float(float(beta)), _lib.ptr(scale), _lib.ptr(opacity),
                                                   _lib.stream_ptr(dev))
            _lib.check(rc, "gp_activations_forward")
        ctx.save_for_backward(s_c, o_c, d_c if d_c is not None else torch.empty(0, device=dev))
        ctx.meta = (int(col), float(beta), d_c is not None)
        ctx.leaves = (scaling_raw, opacity_raw)
        return scale, opacity

    @staticmethod
    def backward(ctx, g_scale, g_opacity):
        s_c, o_c, d_c = ctx.saved_tensors
        col, beta, has_d = ctx.meta
        dev = s_c.device
        N = s_c.shape[0]
        gs = g_scale.to(torch.float32).contiguous() if g_scale is not None else None
        go = g_opacity.to(torch.float32).contiguous() if g_opacity is not None else None
        sinks = [_overwrite_sink(t_) for t_ in ctx.leaves]
        g_sraw = sinks[0] if sinks[0] is not None else torch.empty(N, 3, device=dev)
        g_oraw = sinks[1] if sinks[1] is not None else torch.empty(N, 1, device=dev)
        g_delta = torch.zeros_like(d_c) if has_d else None
        stride = d_c.shape[1] if has_d else 0
        dptr = C.c_void_p(d_c.data_ptr() + 4 * col) if has_d else None
        gdptr = C.c_void_p(g_delta.data_ptr() + 4 * col) if has_d else None
        with _lib.on_device(dev):
            rc = _lib.lib().gp_activations_backward(C.c_int64(N), _lib.ptr(s_c), _lib.ptr(o_c), dptr, C.c_int32(stride),
                                                    C.c_float(beta), _lib.ptr(gs), _lib.ptr(go), _lib.ptr(g_sraw),
                                                    _lib.ptr(g_oraw), gdptr, _lib.stream_ptr(dev))
            _lib.check(rc, "gp_activations_backward")
        for t_, sk in zip(ctx.leaves, sinks):
            if sk is not None:
                grad_sink.notify(t_)
        return (None if sinks[0] is not None else g_sraw), (None if sinks[1] is not None else g_oraw), g_delta, None, None


class MlpInput(torch.autograd.Function):
    """[feature | PE(xyz, xyz_freq) | PE(t, time_freq)] as a tensor (gp_mlp_input_forward): the input of the GENERIC Deformable_Field
    path -- the fused kernels build it in LDS [REF scene/gaussian_model.py:180-189, scene/deformable_field.py:63-72]."""

    @staticmethod
    def forward(ctx, feature, xyz, t, xyz_freq, time_freq):
        _need_cuda(feature, "MlpInput")
        dev = feature.device
        f_c = _c(feature)
        x_c = _c(xyz) if xyz is not None and xyz_freq > 0 else None
        t_c = _c(t).reshape(-1)[:1] if t is not None and time_freq > 0 else None
        rows, fd = f_c.shape
        xf, tf = (int(xyz_freq) if x_c is not None else 0), (int(time_freq) if t_c is not None else 0)
        out = torch.empty(rows, fd + 6 * xf + 2 * tf, device=dev)
        inp = _lib.MlpInputC(rows, fd, xf, tf, f_c.data_ptr(), x_c.data_ptr() if x_c is not None else None,
                             t_c.data_ptr() if t_c is not None else None)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().gp_mlp_input_forward(C.byref(inp), _lib.ptr(out), _lib.stream_ptr(dev)), "gp_mlp_input_forward")
        ctx.save_for_backward(x_c if x_c is not None else torch.empty(0, device=dev))
        ctx.meta = (rows, fd, xf, tf, feature.requires_grad, xyz is not None and torch.is_tensor(xyz) and xyz.requires_grad and xf > 0)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (x_c,) = ctx.saved_tensors
        rows, fd, xf, tf, need_f, need_x = ctx.meta
        dev = g_out.device
        g = g_out.to(torch.float32).contiguous()
        d_f = torch.empty(rows, fd, device=dev) if need_f else None
        d_x = torch.empty(rows, 3, device=dev) if need_x else None
        if need_f or need_x:
            inp = _lib.MlpInputC(rows, fd, xf, tf, None, x_c.data_ptr() if xf > 0 else None, None)
            with _lib.on_device(dev):
                _lib.check(_lib.lib().gp_mlp_input_backward(C.byref(inp), _lib.ptr(g), _lib.ptr(d_f), _lib.ptr(d_x), _lib.stream_ptr(dev)),
                           "gp_mlp_input_backward")
        return d_f, d_x, None, None, None


class GenericMlp(torch.autograd.Function):
    """nn.Sequential(Linear, ReLU, ..., Linear[, Softmax(dim=-1)]) of ANY depth and width, layer by layer on the library's generic
    dense-layer entries (gp_linear_forward / gp_linear_backward, exact fp32 on the matrix cores) -- Deformable_Field for the shapes the
    fused kernels do not cover [REF scene/deformable_field.py:74-127].  `wb` = w0, b0, w1, b1, ..., w_out, b_out."""

    @staticmethod
    def forward(ctx, x, use_softmax, *wb):
        _need_cuda(x, "GenericMlp")
        dev = x.device
        L = _lib.lib()
        ws = [_c(w) for w in wb[0::2]]
        bs = [_c(b) for b in wb[1::2]]
        h = _c(x)
        rows = h.shape[0]
        acts = [h]
        with _lib.on_device(dev):
            st = _lib.stream_ptr(dev)
            for l, (w, b) in enumerate(zip(ws, bs)):
                if w.shape[1] != acts[-1].shape[1]:
                    raise RuntimeError(f"GenericMlp: layer {l} takes {w.shape[1]} inputs, got {acts[-1].shape[1]}")
                y = torch.empty(rows, w.shape[0], device=dev)
                _lib.check(L.gp_linear_forward(_lib.ptr(acts[-1]), C.c_int64(rows), C.c_int32(w.shape[1]), _lib.ptr(w), _lib.ptr(b),
                                               C.c_int32(w.shape[0]), C.c_int32(1 if l < len(ws) - 1 else 0), _lib.ptr(y), st), "gp_linear_forward")
                acts.append(y)
            out = acts[-1]
            if use_softmax:
                out = torch.empty_like(acts[-1])
                _lib.check(L.gp_softmax_forward(_lib.ptr(acts[-1]), C.c_int64(rows), C.c_int32(out.shape[1]), _lib.ptr(out), st), "gp_softmax_forward")
        need = x.requires_grad or any(t.requires_grad for t in wb)
        if need:
            ctx.save_for_backward(out if use_softmax else torch.empty(0, device=dev), *acts[:-1], *ws)
            ctx.meta = (len(ws), bool(use_softmax), x.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g_out):
        n, use_softmax, need_x = ctx.meta
        saved = ctx.saved_tensors
        sm, acts, ws = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n]
        dev = g_out.device
        L = _lib.lib()
        rows = acts[0].shape[0]
        g = g_out.to(torch.float32).contiguous()
        grads = [None] * (2 * n)
        with _lib.on_device(dev):
            st = _lib.stream_ptr(dev)
            if use_softmax:
                g2 = torch.empty_like(g)
                _lib.check(L.gp_softmax_backward(_lib.ptr(sm), _lib.ptr(g), C.c_int64(rows), C.c_int32(g.shape[1]), _lib.ptr(g2), st), "gp_softmax_backward")
                g = g2
            for l in range(n - 1, -1, -1):
                w = ws[l]
                relu = 1 if l < n - 1 else 0
                # behind a ReLU the layer's own output masks dy: it is the next layer's saved input
                y = acts[l + 1] if relu else None
                dw, db = torch.zeros_like(w), torch.zeros(w.shape[0], device=dev)
                dx = torch.empty(rows, w.shape[1], device=dev) if (l > 0 or need_x) else None
                _lib.check(L.gp_linear_backward(_lib.ptr(acts[l]), _lib.ptr(y), _lib.ptr(g), C.c_int64(rows), C.c_int32(w.shape[1]), _lib.ptr(w),
                                                C.c_int32(w.shape[0]), C.c_int32(relu), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), st), "gp_linear_backward")
                grads[2 * l], grads[2 * l + 1] = dw, db
                g = dx
        return (g if need_x else None, None, *grads)
