#!/usr/bin/env python
"""Launches kernels of KNOWN byte counts (streaming copy / read, random gathers of 16 / 48 / 64-byte records) so that a
`rocprofv3 --pmc FETCH_SIZE` (and, separately, `--pmc WRITE_SIZE`) pass over this script calibrates what the counters report
for each access shape on this box (tools/pmc_hbm.py reads the result next to the bench's counters).
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_cal_fetch -o cal --output-format csv -- python tools/pmc_calibrate.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianprediction_amd import _lib

dev = torch.device("cuda", 0)
L = _lib.lib()
st = _lib.stream_ptr(dev)
sink = torch.zeros(4, device=dev)
nbytes = 1 << 30
src = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
n_rec = 4_000_000                      # 4 M requests, like the composite's record fetch at configs[2]
g = torch.Generator().manual_seed(0)
perm = torch.randperm(n_rec, generator=g).to(torch.int32).to(dev)          # every record exactly once, random order
rep4 = torch.randint(0, n_rec // 4, (n_rec,), generator=g).to(torch.int32).to(dev)   # 1 M records, each ~4 times (the composite's reuse)
known = {}
for rep in range(3):
    _lib.check(L.gp_microbench_copy(_lib.ptr(dst), _lib.ptr(src), C.c_size_t(nbytes), st), "copy")
    _lib.check(L.gp_microbench_read(_lib.ptr(src), C.c_size_t(nbytes), _lib.ptr(sink), st), "read")
    for name, rec, stride, idx in (("gather_64B_lines_once", 64, 64, perm), ("gather_16B_of_64B_lines_once", 16, 64, perm),
                                   ("gather_48B_records_once", 48, 48, perm), ("gather_48B_records_4x_reuse", 48, 48, rep4)):
        # distinct kernel names are not available (one kernel): the variants are told apart by launch order in the CSV
        _lib.check(L.gp_microbench_gather(_lib.ptr(src), C.c_int(rec), C.c_int(stride), _lib.ptr(idx), C.c_size_t(n_rec), _lib.ptr(sink), st),
                   "gather")
        known[name] = {"requests": n_rec, "rec_bytes": rec, "stride_bytes": stride}
torch.cuda.synchronize()
known["copy"] = {"read_bytes": nbytes, "write_bytes": nbytes}
known["read"] = {"read_bytes": nbytes}
known["_order"] = ["copy", "read", "gather_64B_lines_once", "gather_16B_of_64B_lines_once", "gather_48B_records_once",
                   "gather_48B_records_4x_reuse"]
os.makedirs("gpurun_out/r2", exist_ok=True)
json.dump(known, open("gpurun_out/r2/pmc_calibrate_known.json", "w"), indent=1)
print(json.dumps(known))
