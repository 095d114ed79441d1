#!/usr/bin/env python
"""Where the 16-bit MLP forward's time goes: the kernel timed with pieces switched off (gp_debug_option(9, bits): 1 no saved-tensor
stores, 2 no mask stores, 4 no matrix products, 8 no input staging, 16 no epilogue).  python tools/probe/mlp16_ablate.py [precision] [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaussianprediction_amd as gpa
from gaussianprediction_amd import _lib

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32s"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
F = 6
net = gpa.Deformable_Field(32 + 60 + 2 * F, output_dim=7, d=4, w=256, precision=prec).cuda()
feat = (torch.rand(rows, 32, device="cuda") - 0.5).requires_grad_(True)
xyz = (torch.rand(rows, 3, device="cuda") * 2.6 - 1.3).requires_grad_(True)
t = torch.tensor([0.3], device="cuda")
L = _lib.lib()
for bits, what in [(0, "full (first: clocks still settling)"), (0, "full"), (1, "no saved-tensor stores"), (3, "no stores at all"), (8, "no input staging"), (4, "no matrix products"),
                   (16, "no epilogue"), (4 + 16, "no products, no epilogue"), (1 + 2 + 4 + 8 + 16, "nothing but barriers and the output layer"), (0, "full"),
                   (32, "ONE workgroup per CU"), (32 + 3, "one workgroup per CU, no stores"), (32 + 3 + 8 + 16, "one workgroup per CU: products only"), (3 + 8 + 16, "products only")]:
    _lib.check(L.gp_debug_option(9, bits), "opt")
    for _ in range(4):
        net.forward_fused(feat, xyz, t, 10, F)
    torch.cuda.synchronize()
    _lib.profile_enable(True); _lib.profile_collect()
    for _ in range(8):
        net.forward_fused(feat, xyz, t, 10, F)
    torch.cuda.synchronize()
    p = _lib.profile_collect(); _lib.profile_enable(False)
    print(f"{prec} rows {rows}  ablate {bits:2d} ({what:44s}): fwd {p['mlp16_fwd'][1] / p['mlp16_fwd'][0] * 1e3:8.1f} us", flush=True)
_lib.check(L.gp_debug_option(9, 0), "opt")
