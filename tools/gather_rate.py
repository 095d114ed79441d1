#!/usr/bin/env python
"""Random-gather rate of this GPU from a table of the hash grid's size (97.6 MB: beyond the L2s, inside the 256 MB memory-side
cache) and from a 1 GiB one: what the hash-grid encode / table-gradient kernels are bounded by on their un-shared (fine) levels."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianprediction_amd import _lib

dev = torch.device("cuda", 0)
L = _lib.lib()
st = _lib.stream_ptr(dev)
sink = torch.zeros(4, device=dev)
g = torch.Generator().manual_seed(0)
n_req = 32_000_000
for table_bytes in (97_600_000, 1 << 30):
    src = torch.zeros(table_bytes // 16 * 16, dtype=torch.uint8, device=dev)
    for rec in (16, 32, 64):
        n_rec = table_bytes // rec
        idx = torch.randint(0, n_rec, (n_req,), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
        f = lambda: _lib.check(L.gp_microbench_gather(_lib.ptr(src), C.c_int(rec), C.c_int(rec), _lib.ptr(idx), C.c_size_t(n_req), _lib.ptr(sink), st), "gather")
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"table_MB": round(table_bytes / 1e6, 1), "rec_bytes": rec, "requests": n_req, "ms": round(ms, 4),
                          "G_requests_per_s": round(n_req / ms / 1e6, 2), "useful_GB_per_s": round(n_req * rec / ms / 1e6, 1)}), flush=True)
