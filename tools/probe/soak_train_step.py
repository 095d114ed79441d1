"""Stability soak of the fused train step on bench.py's headline scene: N steps back to back, the loss read every 1 000 steps (finite),
frames repeated after a binning / depth-key overflow counted, the feature-split MLP's error word read at the end, parameters finite.
    python tools/probe/soak_train_step.py [steps=10000]"""
import json, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, ".")
import bench
from gaussianprediction_amd.train_step import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
pc, cams, gts, _ = bench.build_workload(args, dev)
ts = TrainStep(pc, cams, gts, args.iteration, lrs=dict(xyz=1.6e-6 * 5.0), speculative=True)
losses = []
t0 = time.perf_counter()
for i in range(steps):
    out = ts.step(i)
    if (i + 1) % 1000 == 0:
        l = float(out[0])
        losses.append(round(l, 5))
        assert l == l and abs(l) < 1e3, l
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fin = all(bool(torch.isfinite(p).all()) for p in (pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features_dc, pc._features_rest))
scr = getattr(getattr(ts, "_fused_plan", None), "mlp_scratch", None)
err = int(scr.view(torch.int32)[1024]) if scr is not None else None
print(json.dumps({"steps": steps, "ms_per_step_incl_reads": round(1e3 * dt / steps, 4), "loss_every_1000": losses, "frames_repeated": getattr(ts, "redone", 0),
                  "fused_steps": getattr(ts, "fused_steps", 0), "params_finite": fin, "mlp_split_error_word": err}))
