#!/usr/bin/env python
"""How many tile-splat instances of the bench workload (configs[2]) can touch NO pixel of their tile at alpha >= 1/255 (sub-block mask 0:
the 3-sigma rectangle of the published algorithm reaches the tile, the threshold ellipse does not), and how many sub-blocks the others touch.
    python tools/probe/mask_stats.py [--scale 1.0]"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
import bench
from gaussianprediction_amd import _lib
from gaussianprediction_amd.rasterizer import _settings_c, _f32c
from gaussianprediction_amd.renderer import _settings


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    args = SimpleNamespace(gaussians=a.gaussians, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                           scale_lo=0.003 * a.scale, scale_hi=0.012 * a.scale)
    pc, cams, gts, margs = bench.build_workload(args, dev)
    out = {}
    for ci in (0, 3):
        cam = cams[ci]
        with torch.no_grad():
            t = torch.from_numpy(cam.time).float().to(dev)
            xyz, q, s, o = pc(t, 50000)
            shs = pc.get_features.contiguous()
        rs = _settings(cam, pc, torch.zeros(3, device=dev), 1.0)
        N = xyz.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        with _lib.on_device(dev):
            st, keep = _settings_c(rs, dev, 16)
            m3, ops, scl, rot = _f32c(xyz, dev), _f32c(o, dev), _f32c(s, dev), _f32c(q, dev)
            inp = _lib.RasterInputsC(N, _lib.ptr(m3), _lib.ptr(shs), None, None, _lib.ptr(ops), _lib.ptr(scl), _lib.ptr(rot), None)
            color = torch.empty(3, H, W, device=dev); radii = torch.empty(N, device=dev, dtype=torch.int32)
            depth = torch.empty(1, H, W, device=dev); tidx = torch.empty(H, W, device=dev, dtype=torch.int32)
            o_c = _lib.RasterOutputsC(_lib.ptr(color), _lib.ptr(radii), _lib.ptr(depth), _lib.ptr(tidx))
            saved = _lib.RasterSavedC()
            alloc = _lib.TorchAllocator(dev)
            _lib.check(L.gp_raster_forward(C.byref(st), C.byref(inp), C.byref(o_c), C.byref(saved), alloc.cb, None, _lib.stream_ptr(dev)), "fwd")
            torch.cuda.synchronize()
            R = int(saved.num_rendered)
            b = alloc.first(_lib.GP_BUF_BINNING)
            off = (R * 4 + 255) // 256 * 256
            sm = b[off:off + 2 * R].view(torch.int16).to(torch.int32) & 0xFFFF
            # a tile whose pixels all saturated early leaves the tail of its list unstaged (mask never written): count only staged ones
            pop = torch.zeros_like(sm)
            for k in range(16):
                pop += (sm >> k) & 1
            hist = torch.bincount(pop, minlength=17).tolist()
            alloc.release()
        out[f"cam{ci}"] = {"R": R, "R_per_gaussian": round(R / N, 3), "empty_mask_frac": round(hist[0] / R, 4),
                          "mean_subblocks_of_nonempty": round(float((pop.float().sum() / max(R - hist[0], 1))), 3), "popcount_hist": hist}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
