#!/usr/bin/env python
"""Forward-only renders of the bench scene (configs[2]) for profiling the binning kernels:
    rocprofv3 --kernel-trace --stats -d out -o s --output-format csv -- python tools/probe/bin_probe.py [--radix]
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS ... -- python tools/probe/bin_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
import bench
from gaussianprediction_amd import _lib
from gaussianprediction_amd.rasterizer import raster_forward_debug
from gaussianprediction_amd.renderer import _settings

dev = torch.device("cuda", 0)
args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
if "--scale" in sys.argv:
    k = float(sys.argv[sys.argv.index("--scale") + 1]); args.scale_lo *= k; args.scale_hi *= k
pc, cams, gts, margs = bench.build_workload(args, dev)
if "--radix" in sys.argv:
    _lib.check(_lib.lib().gp_debug_option(5, 1), "opt")
if "--ablate" in sys.argv:
    _lib.check(_lib.lib().gp_debug_option(6, int(sys.argv[sys.argv.index("--ablate") + 1])), "opt")
cam = cams[3]
with torch.no_grad():
    xyz, q, s, o = pc(torch.from_numpy(cam.time).float().to(dev), 50000)
    shs = pc.get_features.contiguous()
    rs = _settings(cam, pc, torch.zeros(3, device=dev), 1.0)
    for _ in range(int(os.environ.get("REPS", "6"))):
        dbg = raster_forward_debug(rs, xyz, o, shs=shs, scales=s, rotations=q)
torch.cuda.synchronize()
print("R", dbg["R"])
