"""The bench scene with its Gaussians stored in RANDOM order (as bench.py builds it) vs along a Morton curve (what densification /
a sorted checkpoint produces): per-kernel times of the train step.  Checks that no kernel depends on the order being random."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
import bench
from gaussianprediction_amd import _lib, scene_synth
from gaussianprediction_amd.train_step import TrainStep
from gaussianprediction_amd.weights_ops import morton_order
dev = torch.device("cuda", 0)
args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000, scale_lo=0.003, scale_hi=0.012)
orig = scene_synth.make_gaussians
for mode in ("random", "morton"):
    def mk(spec, device=None):
        raw = orig(spec, device=device)
        if mode == "morton":
            o = morton_order(raw["xyz"]).long()
            raw = {k: (v[o].contiguous() if torch.is_tensor(v) and v.shape[:1] == raw["xyz"].shape[:1] else v) for k, v in raw.items()}
        return raw
    scene_synth.make_gaussians = mk
    pc, cams, gts, margs = bench.build_workload(args, dev)
    ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=8e-6))
    for i in range(20): ts.step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): ts.step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    _lib.profile_enable(2); _lib.profile_collect()
    for i in range(20): ts.step(i)
    torch.cuda.synchronize()
    pr = _lib.profile_collect(); _lib.profile_enable(0)
    print(mode, "step_ms", round(1e3 * dt, 4), {k: round(v[1] / v[0], 4) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][1])}, flush=True)
    del pc, ts
