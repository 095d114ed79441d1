#!/usr/bin/env python
"""Time the keypoint-weights producers (hash-grid weights model fwd/bwd, kNN) at the bench scale (N = 1M, K = 250)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianprediction_amd import _lib
from gaussianprediction_amd.weights_ops import WeightsModel, knn_keypoints

N, K, nn = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 250, 6
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
xyz = ((torch.rand(N, 3, generator=g) * 2 - 1) * torch.tensor([1.5, 1.5, 0.5])).to(dev)
feat = (1e-3 * (torch.rand(N, 32, generator=g) * 2 - 1)).to(dev)
kp = xyz[torch.randperm(N, generator=g)[:K].to(dev)].clone()
kpf = feat[:K].clone()
m = WeightsModel(2 * nn, device=dev)
print("table entries", m.table_entries, "params", m.params.numel())

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n

def fwd():
    with torch.no_grad():
        return m(xyz)
def fwdbwd():
    m.params.grad = None
    out = m(xyz)
    out.sum().backward()
print(f"weights model forward          {timeit(fwd):8.3f} ms")
print(f"weights model forward+backward {timeit(fwdbwd):8.3f} ms")
print(f"kNN hybird (35-D)              {timeit(lambda: knn_keypoints(xyz, kp, nn, feat, kpf, 5.0, 'hybird')):8.3f} ms")
from gaussianprediction_amd.weights_ops import morton_order
order = morton_order(xyz)
print(f"kNN hybird, Morton order       {timeit(lambda: knn_keypoints(xyz, kp, nn, feat, kpf, 5.0, 'hybird', order=order)):8.3f} ms")
print(f"kNN 3D, Morton order           {timeit(lambda: knn_keypoints(xyz, kp, nn, None, None, 5.0, '3D', order=order)):8.3f} ms")
print(f"morton_order (torch ops)       {timeit(lambda: morton_order(xyz)):8.3f} ms")
print(f"kNN 3D                         {timeit(lambda: knn_keypoints(xyz, kp, nn, None, None, 5.0, '3D')):8.3f} ms")
_lib.profile_enable(2); _lib.profile_collect()
for _ in range(5): fwdbwd()
torch.cuda.synchronize()
for k, (n_, ms) in sorted(_lib.profile_collect().items()):
    print(f"  {k:16s} {ms / n_:8.3f} ms")

if os.environ.get('GP_WB_SHORT'): sys.exit(0)
# ---- full stage-3 train step with the model computing its own weights + kNN per frame (what the reference's step does)
import bench
from types import SimpleNamespace
from gaussianprediction_amd.train_step import TrainStep
from gaussianprediction_amd.weights_ops import WeightsModel as WM
args = SimpleNamespace(gaussians=N, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
pc, cams, gts, margs = bench.build_workload(args, dev)
margs.knn_type, margs.feature_amplify = "hybird", 5.0
pc.weights_model = WM(12, device=dev)
pc.set_keypoint_weights(None, None)
ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=8e-6))
for i in range(10): ts.step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(50): ts.step(i)
torch.cuda.synchronize()
print(f"stage-3 train step incl. weights model + kNN: {1e3 * (time.perf_counter() - t0) / 50:.3f} ms")
_lib.profile_enable(2); _lib.profile_collect()
for i in range(10): ts.step(i)
torch.cuda.synchronize()
for k, (n_, ms) in sorted(_lib.profile_collect().items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:18s} {ms / n_:8.3f} ms")
