#!/usr/bin/env python
"""Which host-side torch operations launch the small kernels of a train step?  (torch.profiler: op -> kernels, with the Python frame)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from gaussianprediction_amd.train_step import TrainStep

args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
pc, cams, gts, margs = bench.build_workload(args, dev)
ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=8e-6), speculative=True)
for i in range(24):
    ts.step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ts.step(24)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.kernels]
evs.sort(key=lambda e: e.time_range.start)
for e in evs:
    ks = [(k.name[:48], round(k.duration, 1)) for k in e.kernels]
    st = [s for s in (e.stack or []) if "/root/repo" in s or "gaussianprediction_amd" in s][:2]
    print(f"{e.name[:40]:40s} {str(e.input_shapes)[:60]:60s} {ks}  {st}")
