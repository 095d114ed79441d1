"""Which autograd nodes of the train step's graph receive more than one gradient (the engine sums those with elementwise add
kernels: small launches on the critical path)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
from collections import Counter
import torch
import bench
from gaussianprediction_amd.train_step import TrainStep
from gaussianprediction_amd.renderer import render
dev = torch.device("cuda", 0)
args = SimpleNamespace(gaussians=20000, width=256, height=192, keypoints=250, nearest_num=6, time_freq=8, iteration=50000, scale_lo=0.01, scale_hi=0.03)
pc, cams, gts, margs = bench.build_workload(args, dev)
ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=8e-6))
ts.step(0)
pkg = render(cams[1], pc, ts.pipe, ts.bg, time=ts.times[1], it=50000)
loss = ts.loss_of(pkg["render"], gts[1])
cnt, seen, stack, names = Counter(), set(), [loss.grad_fn], {}
while stack:
    fn = stack.pop()
    if fn is None or fn in seen: continue
    seen.add(fn)
    for nxt, idx in fn.next_functions:
        if nxt is None: continue
        cnt[(nxt, idx)] += 1
        names.setdefault(nxt, set()).add(type(fn).__name__)
        stack.append(nxt)
for (fn, idx), c in cnt.items():
    if c > 1:
        shape = tuple(fn.variable.shape) if hasattr(fn, "variable") else None
        print(c, type(fn).__name__, idx, shape, "from", names[fn])
print("nodes", len(seen))
