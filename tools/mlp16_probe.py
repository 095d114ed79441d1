#!/usr/bin/env python
"""Diagnostic: run the 16-bit MLP forward (training + inference mode) and backward a few times (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gaussianprediction_amd as gpa
dev = "cuda"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
net = gpa.Deformable_Field(104, output_dim=7, d=4, w=256, precision="fp16").to(dev)
feat = (torch.rand(rows, 32, device=dev) - 0.5)
xyz = (torch.rand(rows, 3, device=dev) * 2.6 - 1.3)
t = torch.tensor([0.3], device=dev)
with torch.no_grad():
    for _ in range(3):
        y = net.forward_fused(feat, xyz, t, 10, 6)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(10):
        y = net.forward_fused(feat, xyz, t, 10, 6)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"inference fwd rows={rows}: {dt*1e3:.3f} ms  {rows*450048/dt/1e12:.1f} TF/s")
