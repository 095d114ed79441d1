#!/usr/bin/env python
"""A/B laboratory for the composite kernels on the bench workload (configs[2]): selects kernel variants through
gp_debug_option, checks every variant's outputs against the baseline variant bit for bit, and times them with the
library's hipEvent brackets.  Also prints the instruction-rate microbenchmarks (gp_microbench_valu).
    python tools/composite_lab.py [--fwd 1,0] [--bwd 0] [--bwdk 1,0,2] [--reps 20] [--ubench]
--bwdk: the composite backward's kernel (gp_debug_option(7, v)): 0 = quadrant kernel (shipped), 3 = sub-block kernel (round 5 experiment),
2 = the sub-block kernel's counting variant (prints evaluated / contributing pairs)."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np
import torch
import bench
from gaussianprediction_amd import _lib
from gaussianprediction_amd.rasterizer import GaussianRasterizer, raster_forward_debug
from gaussianprediction_amd.renderer import _settings


def ubench(dev):
    L = _lib.lib()
    sink = torch.zeros(4, device=dev)
    st = _lib.stream_ptr(dev)
    out = {}
    n = C.c_double(0)
    for rep in range(2):
        _lib.profile_enable(1); _lib.profile_collect()
        counts = {}
        for kind in range(10):
            _lib.check(L.gp_microbench_valu(C.c_int(kind), C.c_int(4096), _lib.ptr(sink), C.byref(n), st), "valu")
            counts[kind] = n.value
        torch.cuda.synchronize()
        prof = _lib.profile_collect(); _lib.profile_enable(0)
    names = ["mb_valu_fma", "mb_valu_pk_fma", "mb_valu_exp", "mb_valu_cmp_cndmask", "mb_valu_rcp", "mb_valu_dpp_add",
             "mb_lds_read_b128", "mb_valu_min", "mb_valu_pk_mul", "mb_valu_sqrt"]
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    for kind, nm in enumerate(names):
        ms = prof[nm][1] / prof[nm][0]
        per_simd = counts[kind] / (cus * 4)
        # cycles per wave-instruction per SIMD at a nominal 2.4 GHz
        out[nm] = {"ms": round(ms, 4), "cyc_per_instr_per_simd_at_2.4GHz": round(ms * 1e-3 * 2.4e9 / per_simd, 3)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fwd", default="2,0")      # 2 = compiler-scheduled visit loop (the readable reference), 0 = shipped
    ap.add_argument("--bwd", default="0")
    ap.add_argument("--bwdk", default="0")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--ubench", action="store_true")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    res = {}
    if a.ubench:
        res["ubench"] = ubench(dev)
        print(json.dumps(res["ubench"]), flush=True)
    args = SimpleNamespace(gaussians=a.gaussians, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                           scale_lo=0.003, scale_hi=0.012)
    pc, cams, gts, margs = bench.build_workload(args, dev)
    cam = cams[3]
    with torch.no_grad():
        t = torch.from_numpy(cam.time).float().to(dev)
        xyz, q, s, o = pc(t, 50000)
        shs = pc.get_features.contiguous()
    rs = _settings(cam, pc, torch.zeros(3, device=dev), 1.0)
    base = None
    fwd_variants = [int(v) for v in a.fwd.split(",")]
    bwd_variants = [(int(v), int(kk)) for kk in a.bwdk.split(",") for v in a.bwd.split(",")]
    gw = torch.randn(3, 1014, 1352, device=dev)
    for fv in fwd_variants:
        for bv, bk in bwd_variants:
            _lib.check(L.gp_debug_option(0, fv), "opt"); _lib.check(L.gp_debug_option(1, bv), "opt"); _lib.check(L.gp_debug_option(7, bk), "opt")
            cnt = (C.c_uint64 * 4)()
            _lib.check(L.gp_debug_counters(cnt), "counters")
            with torch.no_grad():
                dbg = raster_forward_debug(rs, xyz, o, shs=shs, scales=s, rotations=q)
            key = f"fwd{fv}_bwd{bv}_k{bk}"
            r = {"R": dbg["R"]}
            leaves = [x.detach().clone().requires_grad_(True) for x in (xyz, o, shs, s, q)]
            m2 = torch.zeros(xyz.shape[0], 3, device=dev, requires_grad=True)
            rast = GaussianRasterizer(raster_settings=rs)
            img, radii, depth, tidx = rast(means3D=leaves[0], means2D=m2, opacities=leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
            (img * gw).sum().backward()
            torch.cuda.synchronize()
            grads = [l.grad.clone() for l in leaves] + [m2.grad.clone()]
            if bk == 2:
                _lib.check(L.gp_debug_counters(cnt), "counters")
                r["bwd_contributing_pairs"], r["bwd_evaluated_pairs"] = int(cnt[2]), int(cnt[3])
                r["bwd_evaluated_per_contributing"] = round(int(cnt[3]) / max(int(cnt[2]), 1), 3)
            cur = dict(color=dbg["color"], depth=dbg["depth"], tidx=dbg["tidx"], n_contrib=dbg["n_contrib"], grads=grads)
            if base is None:
                base = cur
            else:
                for k in ("color", "depth", "tidx", "n_contrib"):
                    r[f"equal_{k}"] = bool(torch.equal(cur[k], base[k]))
                    if not r[f"equal_{k}"]:
                        diff = (cur[k].float() - base[k].float()).abs()
                        r[f"maxdiff_{k}"] = float(diff.max()); r[f"ndiff_{k}"] = int((diff > 0).sum())
                r["grad_rel_l2"] = [float((g - b).norm() / (b.norm() + 1e-30)) for g, b in zip(grads, base["grads"])]
            # timing
            _lib.profile_enable(2); _lib.profile_collect()
            for _ in range(a.reps):
                leaves = [x.detach().requires_grad_(True) for x in (xyz, o, shs, s, q)]
                img, radii, depth, tidx = rast(means3D=leaves[0], means2D=m2, opacities=leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
                (img * gw).sum().backward()
            torch.cuda.synchronize()
            prof = _lib.profile_collect(); _lib.profile_enable(0)
            r["ms"] = {k: round(v[1] / v[0], 4) for k, v in prof.items() if k in ("composite_fwd", "composite_bwd", "bwd_pixprep", "tile_sort", "duplicate", "preprocess_fwd", "preprocess_bwd")}
            res[key] = r
            print(key, json.dumps(r), flush=True)
    _lib.check(L.gp_debug_option(0, 0), "opt"); _lib.check(L.gp_debug_option(1, 0), "opt"); _lib.check(L.gp_debug_option(7, 0), "opt")


if __name__ == "__main__":
    main()
