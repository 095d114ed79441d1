"""ctypes binding of libgp_hip.so (C ABI declared in include/gp_hip.h).

The HIP library is the product; there is NO CPU or eager-PyTorch fallback.  If the shared library is
missing or cannot be loaded this module raises immediately and loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # noqa: F401  (must be imported first: libgp_hip.so binds to the HIP runtime torch loaded)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libgp_hip.so")

def GP_LOSS_SUM_SLOTS(H, W):   # include/gp_hip.h
    return 3 * ((W + 31) // 32) * ((H + 31) // 32)

GP_BUF_GEOM, GP_BUF_BINNING, GP_BUF_IMAGE, GP_BUF_TEMP, GP_BUF_TEMP_DONE = 0, 1, 2, 3, 4

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_size_t)

_f = C.POINTER(C.c_float)


class RasterSettingsC(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float),
                ("tanfovy", C.c_float), ("scale_modifier", C.c_float), ("sh_degree", C.c_int32),
                ("sh_coeffs", C.c_int32), ("prefiltered", C.c_int32), ("debug", C.c_int32),
                ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
                ("binning_capacity", C.c_int64), ("binning_status", C.c_void_p), ("sh_ready_event", C.c_void_p),
                ("depth_key_bits", C.c_int32), ("depth_key_base", C.c_uint32), ("depth_key_range", C.c_void_p),
                ("raw_activations", C.c_int32), ("reserved0", C.c_int32)]


class RasterInputsC(C.Structure):
    _fields_ = [("num_gaussians", C.c_int64), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("shs_rest", C.c_void_p),
                ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
                ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p)]


class RasterOutputsC(C.Structure):
    _fields_ = [("color", C.c_void_p), ("radii", C.c_void_p), ("depth", C.c_void_p), ("tidx", C.c_void_p), ("visible", C.c_void_p)]


class RasterSavedC(C.Structure):
    _fields_ = [("geom", C.c_void_p), ("geom_bytes", C.c_size_t), ("binning", C.c_void_p),
                ("binning_bytes", C.c_size_t), ("image", C.c_void_p), ("image_bytes", C.c_size_t),
                ("num_rendered", C.c_int64)]


class RasterGradsC(C.Structure):
    _fields_ = [("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dshs", C.c_void_p),
                ("dL_dshs_rest", C.c_void_p), ("dL_dcolors_precomp", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dscales", C.c_void_p),
                ("dL_drotations", C.c_void_p), ("dL_dcov3D_precomp", C.c_void_p), ("accumulate_shs", C.c_int32),
                ("adam_shs", C.c_void_p)]


class AdamFuseC(C.Structure):
    _fields_ = [("exp_avg_dc", C.c_void_p), ("exp_avg_sq_dc", C.c_void_p), ("exp_avg_rest", C.c_void_p), ("exp_avg_sq_rest", C.c_void_p),
                ("lr_dc", C.c_float), ("lr_rest", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step", C.c_int64), ("skip_flag", C.c_void_p)]


class MlpParamsC(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("width", C.c_int32), ("depth", C.c_int32), ("out_dim", C.c_int32),
                ("w", C.c_void_p * 5), ("b", C.c_void_p * 5), ("packed", C.c_void_p), ("scratch", C.c_void_p)]


class Mlp16ParamsC(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("in_dim", C.c_int32), ("width", C.c_int32), ("depth", C.c_int32), ("out_dim", C.c_int32),
                ("w16", C.c_void_p * 5), ("b", C.c_void_p * 5), ("range_flag", C.c_void_p)]


GP_DTYPE_F16, GP_DTYPE_BF16, GP_DTYPE_F16_SPLIT = 1, 2, 3


class MlpGradsC(C.Structure):
    _fields_ = [("dw", C.c_void_p * 5), ("db", C.c_void_p * 5)]


class MlpInputC(C.Structure):
    _fields_ = [("rows", C.c_int64), ("feature_dim", C.c_int32), ("xyz_freq", C.c_int32), ("time_freq", C.c_int32),
                ("feature", C.c_void_p), ("xyz", C.c_void_p), ("t", C.c_void_p)]


class BlendArgsC(C.Structure):
    _fields_ = [("num_gaussians", C.c_int64), ("num_keypoints", C.c_int64), ("nearest_num", C.c_int32),
                ("out_dim", C.c_int32), ("norm_rotation", C.c_int32), ("delta", C.c_void_p), ("raw_w", C.c_void_p),
                ("knn_idx", C.c_void_p), ("xyz", C.c_void_p), ("rot", C.c_void_p), ("knn_idx16", C.c_void_p)]


class StepPlanC(C.Structure):
    """gp_step_plan (include/gp_hip.h)."""
    _fields_ = [("num_gaussians", C.c_int64), ("num_keypoints", C.c_int64), ("nearest_num", C.c_int32), ("norm_rotation", C.c_int32),
                ("sh_degree", C.c_int32), ("image_height", C.c_int32), ("image_width", C.c_int32), ("lambda_dssim", C.c_float),
                ("reg_scale", C.c_float),
                ("xyz", C.c_void_p), ("scaling", C.c_void_p), ("rotation", C.c_void_p), ("opacity", C.c_void_p),
                ("features_dc", C.c_void_p), ("features_rest", C.c_void_p), ("keypoints", C.c_void_p), ("keypoint_features", C.c_void_p),
                ("mlp", MlpParamsC), ("feature_dim", C.c_int32), ("xyz_freq", C.c_int32), ("time_freq", C.c_int32), ("reserved0", C.c_int32),
                ("raw_w", C.c_void_p), ("knn_idx", C.c_void_p), ("knn_idx16", C.c_void_p),
                ("g_xyz", C.c_void_p), ("g_scaling", C.c_void_p), ("g_rotation", C.c_void_p), ("g_opacity", C.c_void_p),
                ("g_features_dc", C.c_void_p), ("g_features_rest", C.c_void_p), ("g_keypoints", C.c_void_p), ("g_keypoint_features", C.c_void_p),
                ("g_mlp", MlpGradsC),
                ("delta", C.c_void_p), ("acts", C.c_void_p), ("xyz_t", C.c_void_p), ("q_t", C.c_void_p), ("scale", C.c_void_p),
                ("opacity_t", C.c_void_p), ("out", RasterOutputsC), ("loss_sums", C.c_void_p), ("dmaps", C.c_void_p), ("loss", C.c_void_p),
                ("dL_dimage", C.c_void_p), ("g_xyz_t", C.c_void_p), ("g_q_t", C.c_void_p), ("g_scale", C.c_void_p), ("g_opacity_t", C.c_void_p),
                ("g_means2D", C.c_void_p), ("g_delta", C.c_void_p), ("g_feature_tmp", C.c_void_p)]


class StepViewC(C.Structure):
    """gp_step_view."""
    _fields_ = [("tanfovx", C.c_float), ("tanfovy", C.c_float), ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
                ("campos", C.c_void_p), ("gt_image", C.c_void_p), ("time", C.c_void_p)]


STEP_HOOK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)


class StepUpdateC(C.Structure):
    """gp_step_update."""
    _fields_ = [("binning_capacity", C.c_int64), ("binning_status", C.c_void_p), ("depth_key_bits", C.c_int32), ("depth_key_base", C.c_uint32),
                ("sh_ready_event", C.c_void_p), ("adam_shs", C.c_void_p),
                ("adam_count", C.c_int32), ("adam_early_mask", C.c_uint32), ("adam_params", C.c_void_p), ("adam_grads", C.c_void_p), ("adam_exp_avgs", C.c_void_p),
                ("adam_exp_avg_sqs", C.c_void_p), ("adam_numels", C.c_void_p), ("adam_lrs", C.c_void_p), ("adam_steps", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("step", C.c_int64), ("keep_grad_mask", C.c_uint32),
                ("skip_flag", C.c_void_p), ("hook", STEP_HOOK_FN), ("hook_ctx", C.c_void_p)]


class ProfileEntryC(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int32), ("total_ms", C.c_float)]


EXPORTS = [
    "gp_raster_forward", "gp_raster_backward", "gp_raster_mark_visible", "gp_raster_debug_binning",
    "gp_mlp_forward", "gp_mlp_backward", "gp_mlp_pack", "gp_mlp_packed_floats", "gp_mlp_scratch_bytes", "gp_mlp16_forward", "gp_mlp16_backward", "gp_mlp16_pack", "gp_mlp16_packed_elems", "gp_blend_forward", "gp_blend_backward",
    "gp_activations_forward", "gp_activations_backward", "gp_profile_enable", "gp_profile_collect",
    "gp_loss_l1_ssim_forward", "gp_loss_l1_ssim_finalize", "gp_loss_l1_ssim_backward", "gp_loss_l1_ssim_fused", "gp_adam_step",
    "gp_adam_step_multi", "gp_adam_step_multi_steps",
    "gp_hashgrid_table_entries", "gp_hashgrid_forward", "gp_hashgrid_backward", "gp_knn_keypoints",
    "gp_weights_forward", "gp_weights_backward", "gp_l1_mean_forward", "gp_l1_mean_backward", "gp_loss_l1_ssim_finalize_reg", "gp_loss_l1_ssim_backward_reg", "gp_furthest_point_sampling", "gp_knn3_mean_dist2",
    "gp_microbench_copy", "gp_microbench_read", "gp_microbench_mfma", "gp_microbench_valu", "gp_microbench_gather",
    "gp_debug_option", "gp_debug_counters", "gp_train_step_run", "gp_sh_factor_gradient",
    "gp_mlp_input_forward", "gp_mlp_input_backward", "gp_linear_forward", "gp_linear_backward", "gp_softmax_forward", "gp_softmax_backward",
    "gp_last_error", "gp_version", "gp_abi_version",
]
GP_ABI_VERSION = 7         # include/gp_hip.h: the struct layouts / signatures / buffer-size macros this binding was written against

_lib = None
_lock = threading.Lock()


class GpHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libgp_hip.so once.  Raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise GpHipError(
                f"{LIB_PATH} not found: the HIP extension is the only implementation of this path (no CPU/eager "
                "fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).")
        l = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            if not hasattr(l, name):
                raise GpHipError(f"{LIB_PATH} does not export {name}")
        l.gp_last_error.restype = C.c_char_p
        l.gp_version.restype = C.c_char_p
        for name in EXPORTS:
            if name not in ("gp_last_error", "gp_version"):
                getattr(l, name).restype = C.c_int
        l.gp_hashgrid_table_entries.restype = C.c_int64
        l.gp_mlp_packed_floats.restype = C.c_int64
        l.gp_mlp_scratch_bytes.restype = C.c_int64
        l.gp_mlp16_packed_elems.restype = C.c_int64
        if int(l.gp_abi_version()) != GP_ABI_VERSION:
            raise GpHipError(f"{LIB_PATH} implements ABI {int(l.gp_abi_version())}, this binding is written against ABI "
                             f"{GP_ABI_VERSION} (include/gp_hip.h): rebuild the library (__graft_entry__.build(force=True))")
        _lib = l
        return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise GpHipError(f"{what}: {lib().gp_last_error().decode(errors='replace')}")


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` (the raw getter costs ~0.3 us; constructing a torch.cuda.Stream
    object for it ~7 us, nine times per train step)."""
    if _raw_stream is not None and _raw_device is not None:
        idx = getattr(device, "index", device)
        if idx is None:
            idx = _raw_device()
        return C.c_void_p(_raw_stream(int(idx)))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """`with on_device(dev):` == `with torch.cuda.device(dev):`, but free when `dev` already is the current device (every
    call of a single-GPU process; the torch context manager costs ~10 us each way, two dozen times per train step)."""
    idx = getattr(device, "index", device)
    if idx is None or (_raw_device is not None and _raw_device() == idx):
        return _NO_GUARD
    return torch.cuda.device(idx)


class TorchAllocator:
    """Backs gp_alloc_fn with torch's caching allocator.

    GEOM / BINNING / IMAGE buffers outlive the call (they are saved for backward) and are rounded up to a
    coarse granule so the caching allocator re-uses blocks although R changes from view to view.
    TEMP buffers are only touched by kernels enqueued during the call, on the calling stream, so a
    per-(device, stream, slot) arena that only ever grows is re-used across calls without any
    allocator traffic (stream order makes the re-use safe)."""

    _temp_arena: dict = {}
    GRANULE = 8 << 20

    def __init__(self, device):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.bufs = {GP_BUF_GEOM: [], GP_BUF_BINNING: [], GP_BUF_IMAGE: [], GP_BUF_TEMP: []}
        self.error = None
        self.cb = ALLOC_FN(self._alloc)
        self._temp_slot = 0
        self._stream = stream_ptr(self.device).value if self.device.type == "cuda" else 0

    def _alloc(self, _ctx, which, nbytes):
        try:
            if which == GP_BUF_TEMP_DONE:     # (gp_train_step_run, between its sub-calls: the TEMP slots start over, as they do for
                self._temp_slot = 0           # the separate entry points, each of which gets an allocator of its own)
                return 0
            nbytes = max(int(nbytes), 1)
            if which == GP_BUF_TEMP and self.device.type == "cuda":
                key = (self.device.index, self._stream, self._temp_slot)
                self._temp_slot += 1
                t = TorchAllocator._temp_arena.get(key)
                if t is None or t.numel() < nbytes:
                    t = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
                    TorchAllocator._temp_arena[key] = t
            else:
                g = TorchAllocator.GRANULE
                size = nbytes if nbytes < g // 8 else (nbytes + g - 1) // g * g
                t = torch.empty(size, dtype=torch.uint8, device=self.device)
            self.bufs[which].append(t)
            return t.data_ptr()
        except Exception as e:  # never let an exception cross the C boundary
            self.error = e
            return 0

    def first(self, which):
        b = self.bufs[which]
        return b[0] if b else None

    def release(self):
        """Break the self -> callback -> bound method -> self cycle so the buffers are freed by
        reference counting right away (not at the next cyclic GC, seconds and gigabytes later)."""
        self.cb = None
        self.bufs = None


def profile_enable(level) -> None:
    """0/False = off, 1 = only the roofline kernel (composite forward), 2/True = every kernel."""
    level = 2 if level is True else int(level)
    check(lib().gp_profile_enable(C.c_int(level)), "gp_profile_enable")


def profile_collect() -> dict:
    """{kernel name: (launches, total_ms)} since the previous collect (waits for the recorded events)."""
    arr = (ProfileEntryC * 64)()
    n = C.c_int(0)
    check(lib().gp_profile_collect(arr, C.c_int(64), C.byref(n)), "gp_profile_collect")
    return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(n.value)}
