"""GPU: the peak microbenchmarks (gp_microbench_*) copy exactly and report sane rates."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stream_copy_is_exact_and_rejects_misaligned():
    from gaussianprediction_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    n = (3 << 20) + 48                      # not a multiple of the unrolled stride
    src = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev)
    dst = torch.zeros_like(src)
    _lib.check(L.gp_microbench_copy(_lib.ptr(dst), _lib.ptr(src), C.c_size_t(n), _lib.stream_ptr(dev)), "copy")
    torch.cuda.synchronize()
    assert torch.equal(dst, src)
    assert L.gp_microbench_copy(_lib.ptr(dst), _lib.ptr(src), C.c_size_t(n - 1), _lib.stream_ptr(dev)) != 0
    assert b"16-byte" in L.gp_last_error()


def test_measured_peaks_are_plausible():
    from gaussianprediction_amd import peaks
    p = peaks.measure("cuda:0", gib=0.5, reps=3, mfma_iters=1024)
    # MI355X: 8 TB/s HBM3E vendor peak, 157 TF/s fp32-matrix, 2.5 PF/s fp16/bf16 dense
    assert 1000.0 < p["copy_GBps"] < 8000.0
    assert 1000.0 < p["read_GBps"] < 8000.0
    assert 30.0 < p["mfma_f32_TFLOPs"] < 200.0
    assert 500.0 < p["mfma_f16_TFLOPs"] < 3000.0
    assert 500.0 < p["mfma_bf16_TFLOPs"] < 3000.0
