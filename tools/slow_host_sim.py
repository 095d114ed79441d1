#!/usr/bin/env python
"""Diagnostic: emulate a slower host by burning EXTRA_MS of CPU time after the rasterizer forward returns (i.e. in the
post-R part of the step, where the host must stay ahead of the GPU) and compare exact binning (one host sync per step)
with the capacity mode (no sync).   python tools/slow_host_sim.py [extra_ms ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from types import SimpleNamespace
from gaussianprediction_amd.train_step import TrainStep
import gaussianprediction_amd.train_step as tsm

args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
extra = [float(a) for a in sys.argv[1:]] or [0.0, 0.4, 0.8]
real_render = tsm.render
burn = [0.0]
def slow_render(*a, **k):
    out = real_render(*a, **k)
    t_end = time.perf_counter() + burn[0] * 1e-3
    while time.perf_counter() < t_end:
        pass
    return out
tsm.render = slow_render
for spec in (False, True):
    pc, cams, gts, margs = bench.build_workload(args, dev)
    ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=8e-6), speculative=spec)
    for i in range(20):
        ts.step(i)
    for e in extra:
        burn[0] = e
        for i in range(10):
            ts.step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(100):
            ts.step(i)
        torch.cuda.synchronize()
        print(f"{'capacity mode' if spec else 'exact mode   '}  +{e:.1f} ms host work per step: {1e3 * (time.perf_counter() - t0) / 100:.3f} ms/step", flush=True)
