#!/usr/bin/env python
"""Host-side cost of one train step: cProfile over N steps of the bench workload (GP_DIST_FORCE_SINGLE=1 for the sharded exchange's path).
    python tools/probe/step_cpu_profile.py [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from gaussianprediction_amd import dist as gdist
from gaussianprediction_amd.train_step import TrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
from types import SimpleNamespace
from gaussianprediction_amd.dist import init_from_env
rank, local, world = init_from_env()            # (GP_DIST_FORCE_SINGLE=1: a one-rank nccl group)
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
wl = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                     scale_lo=0.003, scale_hi=0.012)
pc, cams, gts, margs = bench.build_workload(wl, dev)
ts = TrainStep(pc, cams, gts, wl.iteration, lrs=dict(xyz=8e-6), speculative=True)
for i in range(40):
    ts.step(i % len(cams))
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    ts.step(i % len(cams))
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_host / n * 1e3:.3f} ms/step   with the GPU drained {t_all / n * 1e3:.3f} ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
