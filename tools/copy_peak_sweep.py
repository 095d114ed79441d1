#!/usr/bin/env python
"""Which copy shape reaches this part's stream-copy peak?  (round-2 verdict: the library's microbenchmark measured 4.5-4.8 TB/s
where /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy.)  Sweeps variant x grid x buffer size of
gp_microbench_copy (gp_debug_option(2, variant | gridshift << 3)) and torch's own copy_; prints one JSON line per case."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianprediction_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
st = _lib.stream_ptr(dev)
names = {5: "nt load + nt store, grid-stride (round 2 default)", 1: "temporal load + store, grid-stride", 2: "one float4 per thread, one-shot grid",
         3: "contiguous chunk per workgroup, nt", 4: "nt load + temporal store, grid-stride"}
rows = []
for mib in (64, 256, 1024, 4096):
    nbytes = mib << 20
    src = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    src[::4097] = 7                              # (not all-zero pages)
    dst = torch.empty_like(src)
    for var in (5, 1, 2, 3, 4):
        for gshift in ((0,) if var == 2 else (0, 1, 2)):
            L.gp_debug_option(2, var | (gshift << 3))
            for _ in range(2):
                _lib.check(L.gp_microbench_copy(_lib.ptr(dst), _lib.ptr(src), C.c_size_t(nbytes), st), "copy")
            torch.cuda.synchronize()
            _lib.profile_enable(1); _lib.profile_collect()
            for _ in range(10):
                _lib.check(L.gp_microbench_copy(_lib.ptr(dst), _lib.ptr(src), C.c_size_t(nbytes), st), "copy")
            torch.cuda.synchronize()
            n, tot = _lib.profile_collect()["mb_copy"]
            _lib.profile_enable(0)
            rows.append(dict(MiB=mib, variant=var, what=names[var], grid_x=8 << gshift, GBps=round(2 * nbytes / (tot / n * 1e-3) / 1e9, 1)))
            print(json.dumps(rows[-1]), flush=True)
    L.gp_debug_option(2, 0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(src); torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        dst.copy_(src)
    b.record(); torch.cuda.synchronize()
    rows.append(dict(MiB=mib, variant="torch.copy_", GBps=round(2 * nbytes * 10 / (a.elapsed_time(b) * 1e-3) / 1e9, 1)))
    print(json.dumps(rows[-1]), flush=True)
    del src, dst
best = max((r for r in rows if r["MiB"] >= 1024), key=lambda r: r["GBps"])
print(json.dumps({"best_at_or_above_1GiB": best}))
