#!/usr/bin/env python
"""The split-fp16 weight-gradient kernel's row-slab size (gp_debug_option(10, n): n 16-row blocks per slab; default 64 = 1024 rows) against
its time at several row counts: every slab ends in 64 k atomic adds per (layer, term), fewer slabs are fewer workgroups.
    python tools/probe/mlp16_wgrad_slabs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaussianprediction_amd as gpa
from gaussianprediction_amd import _lib

F = 6
L = _lib.lib()
for rows in (65536, 200000, 1048576):
    net = gpa.Deformable_Field(32 + 60 + 2 * F, output_dim=7, d=4, w=256, precision="fp32s").cuda()
    feat = (torch.rand(rows, 32, device="cuda") - 0.5).requires_grad_(True)
    xyz = (torch.rand(rows, 3, device="cuda") * 2.6 - 1.3).requires_grad_(True)
    t = torch.tensor([0.3], device="cuda")
    g = torch.randn(rows, 7, device="cuda")
    for slab in (0, 16, 32, 64, 128, 256, 512):
        _lib.check(L.gp_debug_option(10, slab), "opt")
        def step():
            for p in net.parameters():
                p.grad = None
            feat.grad = None; xyz.grad = None
            (net.forward_fused(feat, xyz, t, 10, F) * g).sum().backward()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        _lib.profile_enable(True); _lib.profile_collect()
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        p = _lib.profile_collect(); _lib.profile_enable(False)
        print(f"rows {rows:8d} slab {slab:4d} x16 rows: bwd_weight {p['mlp16_bwd_weight'][1] / p['mlp16_bwd_weight'][0] * 1e3:8.1f} us   fwd {p['mlp16_fwd'][1] / p['mlp16_fwd'][0] * 1e3:8.1f}  bwd_data {p['mlp16_bwd_data'][1] / p['mlp16_bwd_data'][0] * 1e3:8.1f}", flush=True)
_lib.check(L.gp_debug_option(10, 0), "opt")
