#!/usr/bin/env python
"""A/B of the feature-split small-row MLP forward (gp_mlp_fwd_split_small_kernel) against the 16-row kernel, on one box: agreement of
the outputs and saved activations, the scratch's counters back at zero and its error word clear, and the time of each form.
    python tools/probe/mlp_split_ab.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaussianprediction_amd as gpa
from gaussianprediction_amd import _lib, deform_ops

F = 6
torch.manual_seed(0)
net = gpa.Deformable_Field(32 + 60 + 2 * F, output_dim=7, d=4, w=256).cuda()
t = torch.tensor([0.3], device="cuda")
for rows in [int(a) for a in sys.argv[1:]] or [250, 16, 1, 17, 500, 512]:
    feat = (torch.rand(rows, 32, device="cuda") - 0.5)
    xyz = (torch.rand(rows, 3, device="cuda") * 2.6 - 1.3)
    res = {}
    for grad in (False, True):
        for name, flag in (("split", False), ("rows16", True)):
            deform_ops.FORCE_ROW_TILES = flag
            f = feat.clone().requires_grad_(grad)
            with torch.set_grad_enabled(grad):
                for _ in range(5):
                    y = net.forward_fused(f, xyz, t, 10, F)
                torch.cuda.synchronize()
                _lib.profile_enable(True); _lib.profile_collect()
                for _ in range(50):
                    y = net.forward_fused(f, xyz, t, 10, F)
                torch.cuda.synchronize()
                prof = _lib.profile_collect(); _lib.profile_enable(False)
            n, ms = prof["mlp_fwd"]
            res[(grad, name)] = (y.detach().clone(), ms / n * 1e3)
            if grad:
                y.sum().backward()
                res[(grad, name, "g")] = f.grad.clone()
    deform_ops.FORCE_ROW_TILES = False
    sc = deform_ops.mlp_scratch(feat.device, rows)
    torch.cuda.synchronize()
    for grad in (False, True):
        a, ta = res[(grad, "split")]; b, tb = res[(grad, "rows16")]
        print(f"rows {rows:4d} {'train' if grad else 'infer'}: split {ta:6.1f} us   16-row {tb:6.1f} us   max |diff| {float((a - b).abs().max()):.2e} (|out| max {float(b.abs().max()):.2e})"
              + (f"   dfeature diff {float((res[(True, 'split', 'g')] - res[(True, 'rows16', 'g')]).abs().max()):.2e}" if grad else ""), flush=True)
    print(f"          scratch counters nonzero: {int((sc[:1024] != 0).sum())}, error word {int(sc[1024])}")

# stress: many launches with fresh inputs, every result against the 16-row kernel (a stale read of the exchange would show here)
rows = 250
bad = 0
worst = 0.0
for it in range(1500):
    feat = (torch.rand(rows, 32, device="cuda") - 0.5)
    xyz = (torch.rand(rows, 3, device="cuda") * 2.6 - 1.3)
    tt = torch.rand(1, device="cuda")
    with torch.no_grad():
        deform_ops.FORCE_ROW_TILES = False
        a = net.forward_fused(feat, xyz, tt, 10, F)
        deform_ops.FORCE_ROW_TILES = True
        b = net.forward_fused(feat, xyz, tt, 10, F)
    d = float((a - b).abs().max())
    worst = max(worst, d)
    bad += d > 1e-6
deform_ops.FORCE_ROW_TILES = False
sc = deform_ops.mlp_scratch(feat.device, rows)
print(f"stress: 1500 launches, {bad} beyond 1e-6 of the 16-row kernel, worst {worst:.2e}; error word {int(sc[1024])}")
