#!/usr/bin/env python
"""Soak of the fused train step with the feature-split keypoint MLP kernels: many steps back to back, then the scratch's error word
(a counter that never filled / a row tile on two XCDs) and the counters must still be zero, the loss finite.
    python tools/probe/soak_split_mlp.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from types import SimpleNamespace
from gaussianprediction_amd import deform_ops
from gaussianprediction_amd.train_step import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
args = SimpleNamespace(gaussians=200_000, width=640, height=480, keypoints=250, nearest_num=6, time_freq=8, iteration=50000, scale_lo=0.004, scale_hi=0.016)
dev = torch.device("cuda", 0)
pc, cams, gts, margs = bench.build_workload(args, dev)
ts = TrainStep(pc, cams, gts, args.iteration, lrs=dict(xyz=1.6e-6 * 5.0), speculative=True)
t0 = time.perf_counter()
loss = None
for i in range(steps):
    loss, _ = ts.step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
words = [(k, int((t[:1024] != 0).sum()), int(t[1024])) for k, t in deform_ops._SCRATCH.items()]
print(f"{steps} steps in {dt:.1f} s ({1e3 * dt / steps:.3f} ms per step), fused steps {ts.fused_steps}, frames repeated {ts.redone}, loss {float(loss):.5f}; "
      f"scratch (key, nonzero counters, error word): {words}")
assert all(nz == 0 and err == 0 for _, nz, err in words) and torch.isfinite(loss)
