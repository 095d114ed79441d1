#!/usr/bin/env python
"""Diagnostic: per-step wall time series of the bench workload (sync after every step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from types import SimpleNamespace
from gaussianprediction_amd.train_step import TrainStep

args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
pc, cams, gts, margs = bench.build_workload(args, dev)
torch.cuda.synchronize()
print(f"setup {time.perf_counter() - t0:.2f}s", flush=True)
ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=8e-6))
import gc
if os.environ.get('NOGC'): gc.disable()
ser = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    torch.cuda.synchronize()
    a = time.perf_counter()
    ts.step(i)
    torch.cuda.synchronize()
    ser.append((time.perf_counter() - a) * 1e3)
print(" ".join(f"{x:.2f}" for x in ser))
print('outliers', [(i, round(x,1)) for i,x in enumerate(ser) if x > 5], 'gc counts', gc.get_count(), 'mem reserved MB', torch.cuda.memory_reserved()/2**20)
