#!/usr/bin/env python
"""Where the step WITH the weights model and the kNN inside (bench.py's train_step_with_weights_model_ms: what the reference's stage-3
loop runs every frame) spends its time: host enqueue time against the drained time, cProfile of the host side, and the library's
per-kernel table.    python tools/probe/weights_step_profile.py [steps]"""
import cProfile, os, pstats, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
import bench
from gaussianprediction_amd import _lib
from gaussianprediction_amd.train_step import TrainStep
from gaussianprediction_amd.weights_ops import WeightsModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
wl = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                     scale_lo=0.003, scale_hi=0.012)
pc, cams, gts, margs = bench.build_workload(wl, dev)
margs.knn_type, margs.feature_amplify = "hybird", 5.0
pc.weights_model = WeightsModel(2 * wl.nearest_num, device=dev)
pc.set_keypoint_weights(None, None)
pc.optimizer = None
ts = TrainStep(pc, cams, gts, wl.iteration, lrs=dict(xyz=1.6e-6 * 5.0))
for i in range(10):
    ts.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    ts.step(10 + i)
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_host / n * 1e3:.3f} ms/step   with the GPU drained {t_all / n * 1e3:.3f} ms/step")
_lib.profile_enable(2); _lib.profile_collect()
for i in range(5):
    ts.step(10 + n + i)
torch.cuda.synchronize()
p = _lib.profile_collect(); _lib.profile_enable(0)
k = {name: round(v[1] / 5, 4) for name, v in sorted(p.items(), key=lambda kv: -kv[1][1])}
print("library kernels ms/step:", json.dumps(k), " sum", round(sum(k.values()), 3))
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
