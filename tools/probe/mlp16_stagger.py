#!/usr/bin/env python
"""Probe: does a start offset between the two workgroups of a CU (gp_debug_option(9, 128 | n << 8)) make product and epilogue overlap?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaussianprediction_amd as gpa
from gaussianprediction_amd import _lib
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
F = 6
net = gpa.Deformable_Field(32 + 60 + 2 * F, output_dim=7, d=4, w=256, precision="fp32s").cuda()
feat = (torch.rand(rows, 32, device="cuda") - 0.5).requires_grad_(True)
xyz = (torch.rand(rows, 3, device="cuda") * 2.6 - 1.3).requires_grad_(True)
t = torch.tensor([0.3], device="cuda")
L = _lib.lib()
for rnd in range(2):
    for n in (0,):
        for extra, what in ((0, "full"), (3, "no stores"), (256, "epilogue prio 3"), (256 + 3, "epi prio 3, no stores"), (512, "product prio 3"), (512 + 3, "prod prio 3, no stores")):
            bits = extra
            _lib.check(L.gp_debug_option(9, bits), "opt")
            for _ in range(3):
                net.forward_fused(feat, xyz, t, 10, F)
            torch.cuda.synchronize()
            _lib.profile_enable(True); _lib.profile_collect()
            for _ in range(8):
                net.forward_fused(feat, xyz, t, 10, F)
            torch.cuda.synchronize()
            p = _lib.profile_collect(); _lib.profile_enable(False)
            print(f"round {rnd} stagger {n} x 8k cycles [{what:22s}]: fwd {p['mlp16_fwd'][1] / p['mlp16_fwd'][0] * 1e3:8.1f} us", flush=True)
_lib.check(L.gp_debug_option(9, 0), "opt")
