#!/usr/bin/env python
"""Diagnostic: per-tile list length / n_contrib distribution of the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from types import SimpleNamespace
from gaussianprediction_amd.rasterizer import raster_forward_debug
from gaussianprediction_amd.renderer import _settings
args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
pc, cams, gts, margs = bench.build_workload(args, dev)
with torch.no_grad():
    for ci in (0, 3, 7):
        cam = cams[ci]
        t = torch.from_numpy(cam.time).float().to(dev)
        xyz, q, s, o = pc(t, 50000)
        dbg = raster_forward_debug(_settings(cam, pc, torch.zeros(3, device=dev), 1.0), xyz, o, shs=pc.get_features, scales=s, rotations=q)
        lens = (dbg["ranges"][:, 1] - dbg["ranges"][:, 0]).float().cpu().numpy()
        print(f"cam {ci}: R={dbg['R']} tiles={len(lens)} mean={lens.mean():.0f} median={np.median(lens):.0f} p90={np.percentile(lens,90):.0f} "
              f"p99={np.percentile(lens,99):.0f} max={lens.max():.0f} empty={(lens==0).sum()}")
