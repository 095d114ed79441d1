"""Measured device peaks: stream-copy / read bandwidth and dense MFMA rates of the GPU this
process runs on, from the library's own microbenchmark kernels (include/gp_hip.h, gp_microbench_*), timed with the
library's hipEvent brackets.  Reported beside the vendor peaks; never part of a timed region."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def measure(device="cuda:0", gib=1.0, reps=10, mfma_iters=4096) -> dict:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.GpHipError("peaks.measure needs a GPU device")
    L = _lib.lib()
    nbytes = int(gib * (1 << 30)) // 16 * 16
    src = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    flop = C.c_double(0.0)
    st = _lib.stream_ptr(dev)

    def run():
        _lib.check(L.gp_microbench_copy(_lib.ptr(dst), _lib.ptr(src), C.c_size_t(nbytes), st), "gp_microbench_copy")
        _lib.check(L.gp_microbench_read(_lib.ptr(src), C.c_size_t(nbytes), _lib.ptr(sink), st), "gp_microbench_read")
        for dt in (0, 1, 2):
            _lib.check(L.gp_microbench_mfma(C.c_int(dt), C.c_int(mfma_iters), _lib.ptr(sink), C.byref(flop), st),
                       "gp_microbench_mfma")
            flops[dt] = flop.value

    flops = {}
    _lib.profile_enable(0)
    run()                                   # warm-up (code objects, clocks)
    torch.cuda.synchronize(dev)
    _lib.profile_enable(1)
    _lib.profile_collect()
    for _ in range(reps):
        run()
    torch.cuda.synchronize(dev)
    prof = _lib.profile_collect()
    _lib.profile_enable(0)

    def ms(name):
        n, tot = prof[name]
        return tot / n

    out = {"copy_GBps": round(2 * nbytes / (ms("mb_copy") * 1e-3) / 1e9, 1),
           "read_GBps": round(nbytes / (ms("mb_read") * 1e-3) / 1e9, 1),
           "bytes": nbytes, "reps": reps}
    for dt, name in ((0, "f32"), (1, "f16"), (2, "bf16")):
        out[f"mfma_{name}_TFLOPs"] = round(flops[dt] / (ms(f"mb_mfma_{name}") * 1e-3) / 1e12, 1)
    return out
