#!/usr/bin/env python
"""Diagnostic: where does the HOST time of one train step go (cProfile over N steps)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from types import SimpleNamespace
from gaussianprediction_amd.train_step import TrainStep

args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
pc, cams, gts, margs = bench.build_workload(args, dev)
ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=8e-6))
for i in range(10):
    ts.step(i)
torch.cuda.synchronize()
n = 30
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(n):
    ts.step(i)
pr.disable()
t_cpu = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host-side loop {1e3 * t_cpu / n:.3f} ms/step ; with final sync {1e3 * t_all / n:.3f} ms/step")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
