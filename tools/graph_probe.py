#!/usr/bin/env python
"""Probe: can one whole train step (render, loss, backward, fused Adam) be captured in a hipGraph and replayed?"""
import os, sys, time, faulthandler
faulthandler.enable()
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
t0 = time.perf_counter()
for i in range(50):
    ts.step(i)
torch.cuda.synchronize()
print(f"eager: {1e3 * (time.perf_counter() - t0) / 50:.3f} ms/step", flush=True)
cap = int(ts._r_max * 1.1) + 4096
status = torch.zeros(2, dtype=torch.int32, device=dev)
# warm the side stream once (allocator arenas etc.)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for v in range(2):
        print("side-stream warm step", v, flush=True)
        ts._step(v, (cap, status), status[1:2])
        torch.cuda.synchronize()
        print("ok", flush=True)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graphs = []
pool = None
for v in range(8):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool, stream=s):
        out = ts._step(v, (cap, status), status[1:2])
    pool = g.pool() if pool is None else pool
    graphs.append((g, out))
    print("captured view", v, flush=True)
torch.cuda.synchronize()
for g, _ in graphs:
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    graphs[i % 8][0].replay()
torch.cuda.synchronize()
print(f"graph replay: {1e3 * (time.perf_counter() - t0) / 100:.3f} ms/step  loss {float(graphs[0][1][0]):.5f} status {status.tolist()}", flush=True)
