"""Calibration for tests/test_gpu_train_loop.py::test_reference_order_loop_equals_the_harness_order: how far apart do two runs of the
SAME loop order end (atomics in the backward, Adam eps = 1e-15), per tensor, in units of the learning rate -- beside the distance
between the two orders."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from types import SimpleNamespace
import torch
import test_gpu_train_loop as T
from gaussianprediction_amd import densify as dn
from gaussianprediction_amd.train_step import TrainStep

def run(order, last=34):
    args, opt = T._args_fast(), T._opt_fast()
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    bg = torch.zeros(3, device="cuda")
    g, cams, gts = T._fresh_model(args, opt)
    if order == "reference":
        for it in range(1, last + 1):
            v = (7 * it) % len(cams)
            T._reference_order_iteration(g, cams[v], gts[v], it, opt, pipe, bg, 2.0, args.max_gaussian_size)
    else:
        ts = TrainStep(g, cams, gts, 1, lambda_dssim=opt.lambda_dssim, schedule=True, training_args=opt)
        for it in range(1, last + 1):
            ts.iteration = it
            loss, pkg = ts.step((7 * it) % len(cams), hold=dn.held_groups(g, it, opt))
            with torch.no_grad():
                dn.track_view(g, pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
                dn.densification_step(g, it, opt, 2.0, max_gaussian_size=args.max_gaussian_size)
    lr_of = {id(p): float(gr["lr"]) for gr in g.optimizer.param_groups for p in gr["params"]}
    return {n: (p.detach().clone(), max(lr_of.get(id(p), 1e-3), 1e-6)) for n, p in g.named_parameters()}

runs = {k: run(k.split("_")[0]) for k in ("reference_1", "reference_2", "harness_1", "harness_2")}
for a, b in (("reference_1", "reference_2"), ("harness_1", "harness_2"), ("reference_1", "harness_1")):
    print("==", a, "vs", b)
    for n in runs[a]:
        x, lr = runs[a][n]
        y = runs[b][n][0]
        if x.shape != y.shape:
            print("  ", n, "shapes differ", tuple(x.shape), tuple(y.shape)); continue
        d = (x - y).abs().flatten().float() / lr
        if n in ("_xyz", "motion_feature", "_opacity", "_features_dc", "df_model.mlp.0.weight"):
            print(f"   {n:28s} median {float(d.median()):.4f} lr   q99 {float(torch.quantile(d[:1000000], 0.99)):.3f} lr   max {float(d.max()):.2f} lr")
