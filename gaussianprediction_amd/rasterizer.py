"""Drop-in for the reference's `diff_gaussian_rasterization` package, backed by libgp_hip.so.

Same names, argument meaning and error behaviour as the interface the reference imports and calls
[REF gaussian_renderer/__init__.py:14, 37-52, 98-106]:

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                                  viewmatrix, projmatrix, sh_degree, campos, prefiltered[, debug])
    GaussianRasterizer(raster_settings)(means3D=, means2D=, shs=, colors_precomp=, opacities=,
                                        scales=, rotations=, cov3D_precomp=)
        -> (rendered_image[3,H,W], radii[N] int32, depth[1,H,W], tidx[H,W] int32)

`loss.backward()` populates .grad on means3D, shs/colors_precomp, opacities, scales, rotations
(or cov3D_precomp) and on means2D (the retained `screenspace_points`, [:, :2] = NDC-space gradient
of the projected centre, read at [REF scene/gaussian_model.py:757]).
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib, grad_sink


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool = False  # the reference does not pass it [REF gaussian_renderer/__init__.py:49]
    # extension (include/gp_hip.h, gp_raster_settings): capacity mode of the binning stage.  0 / None = exact mode (one
    # host sync per forward to read R).  capacity > 0 with a 2-word int32 device tensor = no host sync; status = {R, overflow}
    binning_capacity: int = 0
    binning_status: Optional[torch.Tensor] = None
    # extension (gp_raster_settings.sh_ready_event): a torch.cuda.Event after which the SH tensors are valid (the asynchronous
    # all-gather of the updated coefficients in view-parallel training).  The forward then reads them in a separate SH -> RGB
    # kernel right before the composite and waits for the event only there.  None = they are ready now.
    sh_ready_event: Optional[object] = None
    # extension (gp_raster_outputs.visible): a contiguous uint8 [N] tensor on the render device that the projection kernel fills with
    # radii > 0 -- render()'s visibility_filter without a compare launch per frame.  None = not wanted.
    visible_out: Optional[torch.Tensor] = None
    # extension (gp_raster_settings.depth_key_bits / depth_key_base / depth_key_range): the caller promises that the visible Gaussians'
    # depth keys k (the bit patterns of their view-space depths) satisfy 0 <= k - depth_key_base < 2^depth_key_bits -- the depth sort
    # then runs over that many bits only (24: three passes instead of four, same order); a broken promise raises binning_status[1]
    # (status needs 3 words).  `depth_key_range`: int32 [2] device tensor that receives {min, max} visible key.
    depth_key_bits: int = 0
    depth_key_base: int = 0
    depth_key_range: Optional[torch.Tensor] = None
    # extension (gp_raster_settings.raw_activations): `scales` are the model's log-scales and `opacities` its logits -- the projection
    # kernel applies exp / sigmoid (get_scaling / get_opacity) itself and the backward returns the gradients of the RAW tensors.
    raw_activations: bool = False


_bump_version = getattr(torch.autograd.graph, "increment_version", lambda t: None)


def _f32c(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    if t is None or t.numel() == 0 and t.dim() <= 1:
        return None
    if t.device != device:
        raise RuntimeError(f"rasterizer input on {t.device}, expected {device}")
    return t.detach().to(torch.float32).contiguous()


def _settings_c(rs: GaussianRasterizationSettings, device, sh_coeffs: int):
    keep = [_f32c(rs.bg, device), _f32c(rs.viewmatrix, device), _f32c(rs.projmatrix, device), _f32c(rs.campos, device)]
    if keep[0] is None or keep[0].numel() != 3 or keep[1].numel() != 16 or keep[2].numel() != 16 or keep[3].numel() != 3:
        raise RuntimeError("bg[3], viewmatrix[4,4], projmatrix[4,4], campos[3] expected")
    st = _lib.RasterSettingsC(int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
                              float(rs.scale_modifier), int(rs.sh_degree), int(sh_coeffs), int(bool(rs.prefiltered)),
                              int(bool(rs.debug)), keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(),
                              keep[3].data_ptr(), 0, None)
    status = getattr(rs, "binning_status", None)
    if status is not None:
        if status.device != device or status.dtype != torch.int32 or status.numel() < 2 or not status.is_contiguous():
            raise RuntimeError("binning_status: contiguous int32 tensor of >= 2 elements on the render device expected")
        st.binning_capacity = int(getattr(rs, "binning_capacity", 0) or 0)
        st.binning_status = status.data_ptr()
        keep.append(status)
    elif getattr(rs, "binning_capacity", 0):
        raise RuntimeError("binning_capacity needs binning_status")
    kb = int(getattr(rs, "depth_key_bits", 0) or 0)
    if kb not in (0, 32):
        if status is None or status.numel() < 3:
            raise RuntimeError("depth_key_bits needs a binning_status of >= 3 words")
        st.depth_key_bits, st.depth_key_base = kb, int(getattr(rs, "depth_key_base", 0)) & 0xFFFFFFFF
    st.raw_activations = 1 if getattr(rs, "raw_activations", False) else 0
    kr = getattr(rs, "depth_key_range", None)
    if kr is not None:
        if kr.device != device or kr.dtype != torch.int32 or kr.numel() < 2 or not kr.is_contiguous():
            raise RuntimeError("depth_key_range: contiguous int32 tensor of >= 2 elements on the render device expected")
        st.depth_key_range = kr.data_ptr()
        keep.append(kr)
    ev = getattr(rs, "sh_ready_event", None)
    if ev is not None:
        st.sh_ready_event = int(ev.cuda_event)          # (hipEvent_t; an event that was never recorded counts as complete)
        keep.append(ev)
    return st, keep


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                sh_rest=None):
        if not means3D.is_cuda:
            raise RuntimeError("gaussianprediction_amd rasterizer needs tensors on a HIP device (no CPU fallback)")
        device = means3D.device
        L = _lib.lib()
        m3 = _f32c(means3D, device)
        N = means3D.shape[0]
        if m3 is None:
            m3 = torch.empty(0, 3, device=device)
        shs = _f32c(sh, device)
        shs_r = _f32c(sh_rest, device)
        cols = _f32c(colors_precomp, device)
        ops = _f32c(opacities, device)
        scl = _f32c(scales, device)
        rot = _f32c(rotations, device)
        cov = _f32c(cov3Ds_precomp, device)
        sh_coeffs = int(shs.shape[1]) if shs is not None else 0
        if shs_r is not None:
            sh_coeffs += int(shs_r.shape[1])
        rs = raster_settings
        H, W = int(rs.image_height), int(rs.image_width)
        with _lib.on_device(device):
            st, keep = _settings_c(rs, device, sh_coeffs)
            inp = _lib.RasterInputsC(N, _lib.ptr(m3), _lib.ptr(shs), _lib.ptr(shs_r), _lib.ptr(cols), _lib.ptr(ops),
                                     _lib.ptr(scl), _lib.ptr(rot), _lib.ptr(cov))
            color = torch.empty(3, H, W, device=device, dtype=torch.float32)
            radii = torch.empty(N, device=device, dtype=torch.int32)
            depth = torch.empty(1, H, W, device=device, dtype=torch.float32)
            tidx = torch.empty(H, W, device=device, dtype=torch.int32)
            vis = getattr(rs, "visible_out", None)
            if vis is not None and (vis.device != device or vis.dtype != torch.uint8 or vis.numel() != N or not vis.is_contiguous()):
                raise RuntimeError("visible_out: contiguous uint8 tensor of N elements on the render device expected")
            out = _lib.RasterOutputsC(_lib.ptr(color), _lib.ptr(radii), _lib.ptr(depth), _lib.ptr(tidx), _lib.ptr(vis))
            saved = _lib.RasterSavedC()
            alloc = _lib.TorchAllocator(device)
            try:
                rc = L.gp_raster_forward(C.byref(st), C.byref(inp), C.byref(out), C.byref(saved), alloc.cb, None,
                                         _lib.stream_ptr(device))
                if alloc.error is not None:
                    raise alloc.error
                _lib.check(rc, "gp_raster_forward")
            except BaseException:
                alloc.release()          # (breaks the allocator's self-reference: the buffers are freed now, not at the next GC)
                raise
        ctx.set_materialize_grads(False)      # an unused `depth` output must arrive as None, not as zeros
        ctx.sh_leaves = (sh, sh_rest)         # (leaf Parameters: candidates for direct gradient sinks)
        ctx.raster_settings = rs
        ctx.num_rendered = int(saved.num_rendered)
        ctx.sh_coeffs = sh_coeffs
        ctx.flags = (shs is not None, cols is not None, scl is not None, cov is not None, shs_r is not None)
        geom, binning, image = alloc.first(_lib.GP_BUF_GEOM), alloc.first(_lib.GP_BUF_BINNING), alloc.first(_lib.GP_BUF_IMAGE)
        alloc.release()
        empty = torch.empty(0, device=device)
        ctx.save_for_backward(m3, shs if shs is not None else empty, shs_r if shs_r is not None else empty,
                              cols if cols is not None else empty, ops,
                              scl if scl is not None else empty, rot if rot is not None else empty,
                              cov if cov is not None else empty, color, radii, depth, tidx, geom,
                              binning if binning is not None else torch.empty(0, dtype=torch.uint8, device=device), image)
        ctx.mark_non_differentiable(radii, tidx)
        return color, radii, depth, tidx

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_tidx):
        (m3, shs, shs_r, cols, ops, scl, rot, cov, color, radii, depth, tidx, geom, binning, image) = ctx.saved_tensors
        has_sh, has_col, has_sr, has_cov, has_rest = ctx.flags
        device = m3.device
        N = m3.shape[0]
        rs = ctx.raster_settings
        L = _lib.lib()
        gc = grad_color.to(torch.float32).contiguous() if grad_color is not None else torch.zeros_like(color)
        gd = grad_depth.to(torch.float32).contiguous() if grad_depth is not None else None
        with _lib.on_device(device):
            st, keep = _settings_c(rs, device, ctx.sh_coeffs)
            inp = _lib.RasterInputsC(N, _lib.ptr(m3), _lib.ptr(shs) if has_sh else None, _lib.ptr(shs_r) if has_rest else None,
                                     _lib.ptr(cols) if has_col else None,
                                     _lib.ptr(ops), _lib.ptr(scl) if has_sr else None, _lib.ptr(rot) if has_sr else None,
                                     _lib.ptr(cov) if has_cov else None)
            out = _lib.RasterOutputsC(_lib.ptr(color), _lib.ptr(radii), _lib.ptr(depth), _lib.ptr(tidx))
            saved = _lib.RasterSavedC(geom.data_ptr(), geom.numel(), binning.data_ptr() if binning.numel() else None,
                                      binning.numel(), image.data_ptr(), image.numel(), ctx.num_rendered)
            g_m3 = torch.empty(N, 3, device=device)
            g_m2 = torch.empty(N, 3, device=device)
            # direct sinks: accumulate SH gradients straight into the parameters' pre-allocated .grad
            leaf_sh, leaf_rest = ctx.sh_leaves
            sink_sh = grad_sink.sink_of(leaf_sh) if has_sh else None
            sink_rest = grad_sink.sink_of(leaf_rest) if has_rest else None
            m16 = ctx.sh_coeffs == 16 and (shs.data_ptr() % 16 == 0)
            use_sink = has_sh and m16 and sink_sh is not None and (not has_rest or sink_rest is not None) and \
                sink_sh.shape == shs.shape and (not has_rest or sink_rest.shape == shs_r.shape)
            accumulate = 0
            # the harness may have asked for the optimizer step of the SH tensors inside this backward (grad_sink.arm_fused_update):
            # only when the saved inputs ARE the parameters (no cast / copy in between) and the layout is the split one
            fuse, fuse_c = None, None
            if has_sh and has_rest and m16 and isinstance(leaf_sh, torch.Tensor) and isinstance(leaf_rest, torch.Tensor) \
                    and shs.data_ptr() == leaf_sh.data_ptr() and shs_r.data_ptr() == leaf_rest.data_ptr():
                fuse = grad_sink.take_fused_update((leaf_sh, leaf_rest))
            if fuse is not None:
                fuse_c = _lib.AdamFuseC(fuse["m_dc"].data_ptr(), fuse["v_dc"].data_ptr(), fuse["m_rest"].data_ptr(), fuse["v_rest"].data_ptr(),
                                        fuse["lr_dc"], fuse["lr_rest"], fuse["beta1"], fuse["beta2"], fuse["eps"], fuse["step"],
                                        fuse["skip_flag"].data_ptr() if fuse.get("skip_flag") is not None else None)
                use_sink, g_sh, g_shr = False, None, None
                fuse["taken"] = True
            elif use_sink:
                g_sh, g_shr = sink_sh, (sink_rest if has_rest else None)
                # a sink the optimizer left un-zeroed (grad_sink.mark_stale) is overwritten; mixed states are normalised first
                st_sh = grad_sink.take_stale(g_sh)
                st_r = grad_sink.take_stale(g_shr) if has_rest else st_sh
                if st_sh and st_r:
                    accumulate = 0
                else:
                    if st_sh:
                        g_sh.zero_()
                    if st_r and has_rest:
                        g_shr.zero_()
                    accumulate = 1
            elif fuse is None:
                for leaf in (leaf_sh, leaf_rest):   # autograd will ACCUMULATE into these: stale contents must not survive
                    if isinstance(leaf, torch.Tensor) and leaf.grad is not None and grad_sink.take_stale(leaf.grad):
                        leaf.grad.zero_()
                g_sh = torch.empty(N, 1 if has_rest else ctx.sh_coeffs, 3, device=device) if has_sh else None
                g_shr = torch.empty(N, ctx.sh_coeffs - 1, 3, device=device) if has_rest else None
            g_col = torch.empty(N, 3, device=device) if has_col else None
            g_op = torch.empty(N, 1, device=device)
            g_scl = torch.empty(N, 3, device=device) if has_sr else None
            g_rot = torch.empty(N, 4, device=device) if has_sr else None
            g_cov = torch.empty(N, 6, device=device) if has_cov else None
            grads = _lib.RasterGradsC(_lib.ptr(g_m3), _lib.ptr(g_m2), _lib.ptr(g_sh), _lib.ptr(g_shr), _lib.ptr(g_col), _lib.ptr(g_op),
                                      _lib.ptr(g_scl), _lib.ptr(g_rot), _lib.ptr(g_cov), accumulate,
                                      C.cast(C.pointer(fuse_c), C.c_void_p) if fuse_c is not None else None)
            alloc = _lib.TorchAllocator(device)
            try:
                rc = L.gp_raster_backward(C.byref(st), C.byref(inp), C.byref(out), C.byref(saved), _lib.ptr(gc), _lib.ptr(gd),
                                          C.byref(grads), alloc.cb, None, _lib.stream_ptr(device))
                if alloc.error is not None:
                    raise alloc.error
                _lib.check(rc, "gp_raster_backward")
            finally:
                alloc.release()
        if fuse is not None:         # the kernel rewrote the two parameters through raw pointers: keep their version counters honest
            _bump_version(leaf_sh)
            _bump_version(leaf_rest)
        if use_sink:
            grad_sink.notify(leaf_sh)
            if has_rest:
                grad_sink.notify(leaf_rest)
            g_sh, g_shr = None, None
        return g_m3, g_m2, g_sh, g_col, g_op, g_scl, g_rot, g_cov, None, g_shr


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        sh_rest=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, sh_rest)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Boolean mask of the Gaussians that pass the near-plane test of this camera."""
        with torch.no_grad():
            p = positions.detach().to(torch.float32).contiguous()
            vm = self.raster_settings.viewmatrix.detach().to(torch.float32).contiguous()
            present = torch.empty(p.shape[0], dtype=torch.uint8, device=p.device)
            with _lib.on_device(p.device):
                rc = _lib.lib().gp_raster_mark_visible(C.c_int64(p.shape[0]), _lib.ptr(p), _lib.ptr(vm), _lib.ptr(present),
                                                      _lib.stream_ptr(p.device))
                _lib.check(rc, "gp_raster_mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, shs_rest=None):
        """`shs_rest` is an extension to the reference signature: pass shs=features_dc [N,1,3] and
        shs_rest=features_rest [N,15,3] to skip the per-frame torch.cat of get_features."""
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs,
                                   shs_rest)


def raster_forward_debug(raster_settings, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                         cov3D_precomp=None):
    """Test/diagnostic helper: forward pass that also returns the binning result
    (point_list[R] in composite order, ranges[T,2]) via gp_raster_debug_binning."""
    device = means3D.device
    L = _lib.lib()
    rs = raster_settings
    N = means3D.shape[0]
    m3 = _f32c(means3D, device)
    shs_c, cols, ops = _f32c(shs, device), _f32c(colors_precomp, device), _f32c(opacities, device)
    scl, rot, cov = _f32c(scales, device), _f32c(rotations, device), _f32c(cov3D_precomp, device)
    sh_coeffs = int(shs_c.shape[1]) if shs_c is not None else 0
    H, W = int(rs.image_height), int(rs.image_width)
    with _lib.on_device(device):
        st, keep = _settings_c(rs, device, sh_coeffs)
        inp = _lib.RasterInputsC(N, _lib.ptr(m3), _lib.ptr(shs_c), None, _lib.ptr(cols), _lib.ptr(ops), _lib.ptr(scl),
                                 _lib.ptr(rot), _lib.ptr(cov))
        color = torch.empty(3, H, W, device=device)
        radii = torch.empty(N, device=device, dtype=torch.int32)
        depth = torch.empty(1, H, W, device=device)
        tidx = torch.empty(H, W, device=device, dtype=torch.int32)
        out = _lib.RasterOutputsC(_lib.ptr(color), _lib.ptr(radii), _lib.ptr(depth), _lib.ptr(tidx))
        saved = _lib.RasterSavedC()
        alloc = _lib.TorchAllocator(device)
        try:
            _lib.check(L.gp_raster_forward(C.byref(st), C.byref(inp), C.byref(out), C.byref(saved), alloc.cb, None,
                                           _lib.stream_ptr(device)), "gp_raster_forward")
        except BaseException:
            alloc.release()
            raise
        R = int(saved.num_rendered)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        point_list = torch.empty(max(R, 1), dtype=torch.int32, device=device)
        ranges = torch.empty(T, 2, dtype=torch.int32, device=device)
        _lib.check(L.gp_raster_debug_binning(C.byref(st), C.byref(saved), _lib.ptr(point_list), _lib.ptr(ranges),
                                             _lib.stream_ptr(device)), "gp_raster_debug_binning")
        torch.cuda.synchronize(device)
        geom = alloc.first(_lib.GP_BUF_GEOM)
        rec = geom[:48 * N].view(torch.float32).view(N, 12).clone() if (geom is not None and N > 0) else None
        img = alloc.first(_lib.GP_BUF_IMAGE)
        al = lambda v: (v + 255) // 256 * 256                     # ImageLayout of gp_capi_raster.hip: 256-byte aligned arrays
        nc_off = al(al(8 * (T + 8)) + 4 * H * W)               # ranges[T] + the instance counter's 16 slots, final_T[P], n_contrib[P]
        n_contrib = img[nc_off:nc_off + 4 * H * W].view(torch.int32).view(H, W).clone() if img is not None else None
        alloc.release()
    return dict(color=color, radii=radii, depth=depth, tidx=tidx, R=R, point_list=point_list[:R], ranges=ranges, rec=rec,
                n_contrib=n_contrib)
