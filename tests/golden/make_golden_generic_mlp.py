#!/usr/bin/env python
"""Generates tests/golden/deformable_field_generic.npz by RUNNING the reference's Deformable_Field (scene/deformable_field.py, loaded by
file path) in this container for shapes OFF its operating point: other depths / widths (--d / --w) and the use_softmax / split_xyz variants.
Run here only (needs /root/reference):   python tests/golden/make_golden_generic_mlp.py
Only data is written: parameter names, inputs, outputs, gradients.  Parameters are filled from numpy (PCG64) in named_parameters() order,
uniform in +-1/sqrt(fan_in), so nothing depends on torch's initialisers."""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
CASES = {   # tag: (input_dim, output_dim, d, w, use_softmax, split_xyz, rows, seed)
    "a": (40, 7, 2, 64, False, False, 97, 31),
    "b": (30, 10, 8, 32, False, False, 33, 32),
    "c": (36, 5, 3, 48, True, False, 50, 33),
    "d": (20, 3, 1, 16, False, True, 21, 34),
    "e": (24, 2, 2, 24, True, True, 9, 35),
    "f": (45, 7, 3, 100, False, False, 130, 36),
}


def fill(net, seed):
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for _, p in net.named_parameters():
            fan_in = p.shape[1] if p.dim() == 2 else p.shape[0]
            p.copy_(torch.tensor(rng.uniform(-1, 1, size=tuple(p.shape)).astype(np.float32) / np.float32(np.sqrt(max(fan_in, 1)))))


def main():
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_deformable_field", os.path.join(REF, "scene/deformable_field.py"))
    df = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(df)
    out = {}
    for tag, (d_in, d_out, d, w, sm, split, M, seed) in CASES.items():
        net = df.Deformable_Field(d_in, output_dim=d_out, d=d, w=w, use_softmax=sm, split_xyz=split)
        fill(net, seed)
        rng = np.random.default_rng(seed + 100)
        x = torch.tensor(rng.uniform(-1, 1, size=(M, d_in)).astype(np.float32), requires_grad=True)
        y = net(x)
        gy = torch.tensor(rng.normal(size=tuple(y.shape)).astype(np.float32))
        (y * gy).sum().backward()
        out[f"{tag}_meta"] = np.array([d_in, d_out, d, w, int(sm), int(split), M, seed])
        out[f"{tag}_names"] = np.array([k for k, _ in net.named_parameters()])
        out[f"{tag}_x"], out[f"{tag}_gy"] = x.detach().numpy(), gy.numpy()
        out[f"{tag}_y"], out[f"{tag}_dx"] = y.detach().numpy(), x.grad.numpy()
        for k, p in net.named_parameters():
            out[f"{tag}_grad_{k}"] = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    np.savez_compressed(os.path.join(OUT, "deformable_field_generic.npz"), **out)
    print("wrote deformable_field_generic.npz:", {t: out[f"{t}_y"].shape for t in CASES})


if __name__ == "__main__":
    main()
