#!/usr/bin/env python
"""Diagnostic: HOST time per phase of a train step (no cProfile, no extra syncs): the host blocks once per step (the
4-byte R read inside the rasterizer forward); everything after that read is enqueue work that must stay ahead of the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from types import SimpleNamespace
from gaussianprediction_amd.train_step import TrainStep
from gaussianprediction_amd.renderer import render

args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
pc, cams, gts, margs = bench.build_workload(args, dev)
ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=8e-6))
for i in range(20):
    ts.step(i)
torch.cuda.synchronize()
n = 100
acc = [0.0] * 5
pc_ = time.perf_counter
t_begin = pc_()
for i in range(n):
    cam = ts.cameras[i % 8]; gt = ts.gt[i % 8]; tm = ts.times[i % 8]
    t0 = pc_()
    pkg = render(cam, ts.pc, ts.pipe, ts.bg, time=tm, it=ts.iteration)
    t1 = pc_()
    loss = ts.loss_of(pkg["render"], gt)
    t2 = pc_()
    loss.backward()
    t3 = pc_()
    ts.reducer.finish()
    keep = (ts.pc._features_dc, ts.pc._features_rest, ts.pc._xyz, ts.pc._rotation, ts.pc._scaling, ts.pc._opacity)
    ts.optimizer.step(zero_grad=True, keep_grad=keep)
    t4 = pc_()
    for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        acc[k] += d
torch.cuda.synchronize()
total = (pc_() - t_begin) / n
names = ("render() incl. the wait for R", "loss forward", "loss.backward()", "reducer.finish + Adam")
for k, nm in enumerate(names):
    print(f"{nm:32s} {1e3 * acc[k] / n:7.3f} ms host")
print(f"{'step (synchronised at the end)':32s} {1e3 * total:7.3f} ms")
