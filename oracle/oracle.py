"""ctypes front-end of the CPU ORACLE (oracle/gp_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gp_oracle.c.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.  PARITY STATUS of the rasterizer
arithmetic: **parity unpinned** (reference rasterizer source absent; see gp_oracle.c).

`RasterOracle("f32")` mirrors the float32 expression trees of the HIP kernels; `RasterOracle("f64")`
is the float64 shadow used as gradient ground truth.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> None:
    """Compile the oracle shared libraries (gcc, a few seconds)."""
    libs = [os.path.join(_HERE, f"libgp_oracle_{p}.so") for p in ("f32", "f64")]
    src = os.path.join(_HERE, "gp_oracle.c")
    stale = force or any((not os.path.exists(l)) or os.path.getmtime(l) < os.path.getmtime(src) for l in libs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class RasterSettings:
    """Plain-number mirror of GaussianRasterizationSettings
    (reference kwargs: gaussian_renderer/__init__.py:37-50)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: np.ndarray            # [3]
    scale_modifier: float
    viewmatrix: np.ndarray    # [4,4] row-major == W2C^T  (scene/cameras.py:59)
    projmatrix: np.ndarray    # [4,4] full_proj_transform  (scene/cameras.py:61)
    sh_degree: int
    campos: np.ndarray        # [3]


class RasterOracle:
    def __init__(self, precision: str = "f32", threads: int | None = None):
        assert precision in ("f32", "f64")
        build()
        self.lib = C.CDLL(os.path.join(_HERE, f"libgp_oracle_{precision}.so"))
        self.dt = np.float32 if precision == "f32" else np.float64
        self.real = C.c_float if precision == "f32" else C.c_double
        assert self.lib.gpo_real_bytes() == np.dtype(self.dt).itemsize
        self.lib.gpo_bin.restype = C.c_long
        if threads is not None:
            try:
                omp = C.CDLL("libgomp.so.1")
                omp.omp_set_num_threads(int(threads))
            except OSError:
                pass

    def _a(self, x, shape=None):
        if x is None:
            return None
        a = np.ascontiguousarray(np.asarray(x, dtype=self.dt))
        if shape is not None:
            a = a.reshape(shape)
        return a

    # ---------------------------------------------------------------------------------------
    def preprocess(self, st: RasterSettings, means3D, opacities, shs=None, colors_precomp=None, scales=None,
                   rotations=None, cov3D_precomp=None):
        dt = self.dt
        means3D = self._a(means3D)
        N = means3D.shape[0]
        M = 0 if shs is None else np.asarray(shs).shape[1]
        s = dict(
            N=N, M=M, means3D=means3D, opacities=self._a(opacities, (N,)), shs=self._a(shs),
            colors_precomp=self._a(colors_precomp), scales=self._a(scales), rotations=self._a(rotations),
            cov3D_precomp=self._a(cov3D_precomp),
            view=self._a(st.viewmatrix, (16,)), proj=self._a(st.projmatrix, (16,)), campos=self._a(st.campos, (3,)),
            bg=self._a(st.bg, (3,)),
            radii=np.zeros(N, np.int32), xy=np.zeros((N, 2), dt), depths=np.zeros(N, dt), cov3D=np.zeros((N, 6), dt),
            rgb=np.zeros((N, 3), dt), conic_opacity=np.zeros((N, 4), dt), rect=np.zeros((N, 4), np.int32),
            tiles_touched=np.zeros(N, np.uint32), clamped=np.zeros((N, 3), np.uint8), st=st,
        )
        rc = self.lib.gpo_preprocess_fwd(
            C.c_int(N), C.c_int(st.sh_degree), C.c_int(M), _ptr(s["means3D"]), _ptr(s["scales"]),
            self.real(st.scale_modifier), _ptr(s["rotations"]), _ptr(s["opacities"]), _ptr(s["shs"]),
            _ptr(s["colors_precomp"]), _ptr(s["cov3D_precomp"]), _ptr(s["view"]), _ptr(s["proj"]), _ptr(s["campos"]),
            C.c_int(st.image_width), C.c_int(st.image_height), self.real(st.tanfovx), self.real(st.tanfovy),
            _ptr(s["radii"]), _ptr(s["xy"]), _ptr(s["depths"]), _ptr(s["cov3D"]), _ptr(s["rgb"]),
            _ptr(s["conic_opacity"]), _ptr(s["rect"]), _ptr(s["tiles_touched"]), _ptr(s["clamped"]))
        assert rc == 0
        return s

    def bin(self, s):
        st = s["st"]
        W, H = st.image_width, st.image_height
        T = ((W + 15) // 16) * ((H + 15) // 16)
        Rn = self.lib.gpo_bin(C.c_int(s["N"]), C.c_int(W), C.c_int(H), _ptr(s["tiles_touched"]), _ptr(s["rect"]),
                              _ptr(s["depths"]), None, None, None)
        s["R"] = int(Rn)
        s["point_list"] = np.zeros(max(Rn, 1), np.uint32)
        s["point_tile"] = np.zeros(max(Rn, 1), np.uint32)
        s["ranges"] = np.zeros((T, 2), np.int32)
        self.lib.gpo_bin(C.c_int(s["N"]), C.c_int(W), C.c_int(H), _ptr(s["tiles_touched"]), _ptr(s["rect"]),
                         _ptr(s["depths"]), _ptr(s["point_list"]), _ptr(s["point_tile"]), _ptr(s["ranges"]))
        return s

    def composite(self, s):
        st = s["st"]
        W, H, dt = st.image_width, st.image_height, self.dt
        s["out_color"] = np.zeros((3, H, W), dt)
        s["out_depth"] = np.zeros((H, W), dt)
        s["out_tidx"] = np.zeros((H, W), np.int32)
        s["final_T"] = np.zeros((H, W), dt)
        s["n_contrib"] = np.zeros((H, W), np.int32)
        s["ambiguous"] = np.zeros((H, W), np.uint8)
        rc = self.lib.gpo_composite_fwd(
            C.c_int(W), C.c_int(H), _ptr(s["ranges"]), _ptr(s["point_list"]), _ptr(s["xy"]), _ptr(s["rgb"]),
            _ptr(s["depths"]), _ptr(s["conic_opacity"]), _ptr(s["bg"]), _ptr(s["out_color"]), _ptr(s["out_depth"]),
            _ptr(s["out_tidx"]), _ptr(s["final_T"]), _ptr(s["n_contrib"]), _ptr(s["ambiguous"]))
        assert rc == 0
        return s

    def composite_published(self, s, band_scale=1.0):
        """The composite of state `s` (preprocess + bin done) in the PUBLISHED form of the algorithm -- opacity * exp(power),
        T * (1 - alpha); none of the kernels' folded algebra (gp_oracle.c section 3b).  -> dict(out_color, out_tidx, final_T,
        n_contrib, ambiguous); `s` is left untouched.  `band_scale` widens the ambiguity bands (a float64 run compared with a float32 one)."""
        st = s["st"]
        W, H, dt = st.image_width, st.image_height, self.dt
        o = dict(out_color=np.zeros((3, H, W), dt), out_tidx=np.zeros((H, W), np.int32), final_T=np.zeros((H, W), dt),
                 n_contrib=np.zeros((H, W), np.int32), ambiguous=np.zeros((H, W), np.uint8))
        rc = self.lib.gpo_composite_fwd_published(
            C.c_int(W), C.c_int(H), _ptr(s["ranges"]), _ptr(s["point_list"]), _ptr(s["xy"]), _ptr(s["rgb"]), _ptr(s["depths"]),
            _ptr(s["conic_opacity"]), _ptr(s["bg"]), _ptr(o["out_color"]), _ptr(o["out_tidx"]), _ptr(o["final_T"]),
            _ptr(o["n_contrib"]), _ptr(o["ambiguous"]), C.c_double(band_scale))
        assert rc == 0
        return o

    def forward(self, st: RasterSettings, means3D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        """-> state dict with out_color[3,H,W], radii[N], out_depth[H,W], out_tidx[H,W] (+ saved state)."""
        s = self.preprocess(st, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp)
        self.bin(s)
        self.composite(s)
        return s

    def forward_with_binning_of(self, ref, st: RasterSettings, means3D, opacities, shs=None, colors_precomp=None, scales=None,
                                rotations=None, cov3D_precomp=None):
        """The float64 shadow of a float32 forward `ref` (another precision's state): per-Gaussian quantities and the
        composite are evaluated in THIS precision, the discrete binning result (visibility, per-tile depth order) is taken
        from `ref`.  Two Gaussians whose depths coincide in float32 but not in float64 would otherwise be blended in the
        opposite order by the two precisions -- at 1 M Gaussians a few hundred tiles contain such a pair -- and the
        gradient "ground truth" would be that of a (slightly) different image."""
        s = self.preprocess(st, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp)
        s["R"], s["point_list"], s["ranges"] = ref["R"], ref["point_list"].copy(), ref["ranges"].copy()
        s["radii"] = ref["radii"].copy()
        self.composite(s)
        return s

    def backward(self, s, dL_dcolor, dL_ddepth=None):
        """-> dict of gradients wrt means3D, means2D(NDC), shs|colors_precomp, opacities, scales, rotations,
        cov3D_precomp."""
        st = s["st"]
        W, H, N, M, dt = st.image_width, st.image_height, s["N"], s["M"], self.dt
        dL_dpix = self._a(dL_dcolor, (3, H, W))
        dL_dd = self._a(dL_ddepth, (H, W)) if dL_ddepth is not None else None
        g_mean2D = np.zeros((N, 2), np.float64)
        g_conic = np.zeros((N, 3), np.float64)
        g_opac = np.zeros(N, np.float64)
        g_col = np.zeros((N, 3), np.float64)
        g_depth = np.zeros(N, np.float64)
        rc = self.lib.gpo_composite_bwd(
            C.c_int(W), C.c_int(H), C.c_int(N), _ptr(s["ranges"]), _ptr(s["point_list"]), _ptr(s["xy"]),
            _ptr(s["rgb"]), _ptr(s["depths"]), _ptr(s["conic_opacity"]), _ptr(s["bg"]), _ptr(s["final_T"]),
            _ptr(s["n_contrib"]), _ptr(dL_dpix), _ptr(dL_dd), _ptr(g_mean2D), _ptr(g_conic), _ptr(g_opac),
            _ptr(g_col), _ptr(g_depth))
        assert rc == 0
        g_means3D = np.zeros((N, 3), dt)
        g_shs = np.zeros((N, max(M, 1), 3), dt)
        g_colors = np.zeros((N, 3), dt)
        g_scales = np.zeros((N, 3), dt)
        g_rots = np.zeros((N, 4), dt)
        g_cov3D = np.zeros((N, 6), dt)
        rc = self.lib.gpo_preprocess_bwd(
            C.c_int(N), C.c_int(st.sh_degree), C.c_int(M), _ptr(s["means3D"]), _ptr(s["scales"]),
            self.real(st.scale_modifier), _ptr(s["rotations"]), _ptr(s["shs"]),
            C.c_int(0 if s["colors_precomp"] is None else 1), _ptr(s["cov3D_precomp"]), _ptr(s["view"]),
            _ptr(s["proj"]), _ptr(s["campos"]), C.c_int(W), C.c_int(H), self.real(st.tanfovx), self.real(st.tanfovy),
            _ptr(s["radii"]), _ptr(s["cov3D"]), _ptr(s["clamped"]), _ptr(g_mean2D), _ptr(g_conic), _ptr(g_col),
            _ptr(g_depth), _ptr(g_means3D), _ptr(g_shs), _ptr(g_colors), _ptr(g_scales), _ptr(g_rots), _ptr(g_cov3D))
        assert rc == 0
        return dict(means3D=g_means3D, means2D=g_mean2D.astype(dt), shs=g_shs if M else None,
                    colors_precomp=g_colors, opacities=g_opac.astype(dt).reshape(N, 1), scales=g_scales,
                    rotations=g_rots, cov3D_precomp=g_cov3D, conic=g_conic, color=g_col, depth=g_depth)
