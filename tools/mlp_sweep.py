#!/usr/bin/env python
"""Diagnostic: fused PE+MLP kernel timings vs row count (hipEvent timings from the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gaussianprediction_amd as gpa
from gaussianprediction_amd import _lib

dev = "cuda"
F = 6
d_in = 32 + 60 + 2 * F
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
net = gpa.Deformable_Field(d_in, output_dim=7, d=4, w=256, precision=prec).to(dev)
pre = "mlp16" if prec != "fp32" else "mlp"
print("precision", prec)
flop_row = 2 * (d_in * 256 + 3 * 256 * 256 + 256 * 7)
row_list = [int(a) for a in sys.argv[2:]] or ([250, 1024] if prec == "fp32" else []) + [8192, 65536, 262144, 1048576]
for rows in row_list:   # 16-bit: large inputs only (small ones use the fp32 small-row kernels)
    feat = (torch.rand(rows, 32, device=dev) - 0.5).requires_grad_(True)
    xyz = (torch.rand(rows, 3, device=dev) * 2.6 - 1.3).requires_grad_(True)
    t = torch.tensor([0.3], device=dev)
    for _ in range(3):
        y = net.forward_fused(feat, xyz, t, 10, F)
        y.sum().backward()
    torch.cuda.synchronize()
    _lib.profile_enable(True); _lib.profile_collect()
    n = 5
    for _ in range(n):
        y = net.forward_fused(feat, xyz, t, 10, F)
        y.sum().backward()
    torch.cuda.synchronize()
    p = _lib.profile_collect(); _lib.profile_enable(False)
    fwd = p[pre + "_fwd"][1] / p[pre + "_fwd"][0]
    bd = p[pre + "_bwd_data"][1] / p[pre + "_bwd_data"][0]
    bw = p[pre + "_bwd_weight"][1] / n
    print(f"rows {rows:8d}: fwd {fwd*1e3:9.1f} us ({rows*flop_row/fwd/1e9:7.1f} TF/s)  bwd_data {bd*1e3:9.1f} us ({rows*flop_row/bd/1e9:7.1f} TF/s)  "
          f"bwd_weight(5 launches) {bw*1e3:9.1f} us ({rows*flop_row/bw/1e9:7.1f} TF/s)"
          + (f"  [big 3 layers {p[pre + '_bwd_weight_big'][1] / n * 1e3:.1f} us]" if pre + "_bwd_weight_big" in p else ""), flush=True)
