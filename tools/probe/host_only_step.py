"""Host time of a train step: the same TrainStep on a scene so small that the GPU work is negligible (step time = Python +
launch work).  The full-size step is GPU-bound as long as this stays below its kernel time (1.33 ms at configs[2])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch, bench
from gaussianprediction_amd.train_step import TrainStep
dev = torch.device("cuda", 0)
if os.environ.get("GP_DIST_FORCE_SINGLE") == "1":      # the view-parallel exchange on a one-rank RCCL group: its HOST cost
    from gaussianprediction_amd.dist import init_from_env
    init_from_env()
args = SimpleNamespace(gaussians=2000, width=160, height=128, keypoints=250, nearest_num=6, time_freq=8, iteration=50000, scale_lo=0.01, scale_hi=0.03)
pc, cams, gts, margs = bench.build_workload(args, dev)
for spec, fused in ((True, True), (True, False), (False, False)):
    ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=8e-6), speculative=spec, fused=fused)
    for i in range(50): ts.step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(500): ts.step(i)
    torch.cuda.synchronize()
    name = ("capacity mode, ONE library call per step (gp_train_step_run)" if fused else "capacity mode, autograd graph") if spec else "exact (one host sync per step), autograd graph"
    print(name, round(1e3 * (time.perf_counter() - t0) / 500, 4), "ms per step on a 2 000-Gaussian scene;", ts.fused_steps, "fused steps", flush=True)
