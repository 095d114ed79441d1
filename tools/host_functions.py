#!/usr/bin/env python
"""Diagnostic: HOST time spent inside each autograd Function's forward / backward (the backward bodies run on the autograd
device thread, where cProfile does not see them)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from types import SimpleNamespace
from gaussianprediction_amd.train_step import TrainStep
from gaussianprediction_amd import deform_ops, loss_ops, rasterizer

acc = collections.defaultdict(float)
def wrap(cls):
    for name in ("forward", "backward"):
        fn = getattr(cls, name)
        def make(fn, key):
            def w(*a, **k):
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    acc[key] += time.perf_counter() - t0
            return staticmethod(w)
        setattr(cls, name, make(fn, f"{cls.__name__}.{name}"))
for c in (deform_ops.FusedMlp, deform_ops.KeypointBlend, deform_ops.Activations, loss_ops.L1SSIMLoss, loss_ops._AddL1Mean,
          rasterizer._RasterizeGaussians):
    wrap(c)

args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
pc, cams, gts, margs = bench.build_workload(args, dev)
ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=8e-6))
for i in range(20):
    ts.step(i)
torch.cuda.synchronize()
acc.clear()
n = 100
t0 = time.perf_counter()
for i in range(n):
    ts.step(i)
torch.cuda.synchronize()
total = time.perf_counter() - t0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:40s} {1e3 * v / n:7.3f} ms")
print(f"{'sum inside Functions':40s} {1e3 * sum(acc.values()) / n:7.3f} ms   step {1e3 * total / n:.3f} ms")
