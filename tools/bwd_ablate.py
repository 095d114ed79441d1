#!/usr/bin/env python
"""Diagnostic (temporary): time composite_bwd with parts of the pixel walk disabled (results are wrong; timing only)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from types import SimpleNamespace
import gaussianprediction_amd as gpa
from gaussianprediction_amd import _lib
from gaussianprediction_amd.renderer import render
args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
pc, cams, gts, margs = bench.build_workload(args, dev)
L = _lib.lib()
cam = cams[3]
t = torch.from_numpy(cam.time).float().to(dev)
bgc = torch.zeros(3, device=dev)
for abl in [0, 32]:
    L.gp_set_abl(ctypes.c_int(abl))
    for it in range(4):
        if it == 1:
            _lib.profile_enable(2); _lib.profile_collect()
        out = render(cam, pc, None, bgc, time=t, it=50000)
        out["render"].sum().backward()
    torch.cuda.synchronize()
    prof = _lib.profile_collect(); _lib.profile_enable(0)
    n, ms = prof["composite_bwd"]
    print(f"abl={abl:3d}  composite_bwd {ms / n:.4f} ms   pixprep %.4f ms" % (prof["bwd_pixprep"][1] / prof["bwd_pixprep"][0]))
