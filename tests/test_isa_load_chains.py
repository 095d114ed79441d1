"""No GPU: the hot kernels whose global loads were found SERIALIZED in round 3 (a rolled loop or an early-out around conditional
loads: load -> s_waitcnt vmcnt(0) -> use -> next load, N independent loads = N dependent round trips to memory) are compiled to
gfx950 assembly and checked to keep their loads batched -- a change that re-introduces such a chain costs tens of microseconds per
step and is invisible in the source.  tools/isa_load_audit.py is the same scan over every kernel."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")


@pytest.mark.parametrize("src,kernels", [
    ("loss_adam_kernels.hip", ["gp_l1_ssim_fwd_kernel", "gp_l1_ssim_bwd_kernel", "gp_loss_finalize_reg_kernel", "gp_loss_finalize_kernel"]),
    ("deform_mlp16.hip", ["gp_mlp16_bwd_data_split_kernel"]),
    ("sort_scan.hip", ["gp_radix_hist_kernelILi8E", "gp_radix_scatter_kernelILi8E", "gp_radix_hist_kernelILi16E"]),
    ("deform_kernels.hip", ["gp_blend_fwd6_kernel", "gp_blend_bwd6_kernel", "gp_blend_fwd6_i16_kernel", "gp_blend_bwd6_i16_kernel", "gp_act_fwd_kernel", "gp_act_bwd_kernel"]),
    ("weights_kernels.hip", ["gp_knn_kernelILi35ELi6E"]),
    # (the projection forward: a FULL workgroup stages its SH span with all twelve loads in flight -- stage_sh_full; the rolled form with its
    # one-load tail is left for the last, partial workgroup only and accounts for one more wait behind a load)
    ("raster_kernels.hip", ["gp_tile_ranges_kernel", ("gp_preprocess_fwd_split_kernel", 3)]),
    ("bin_kernels.hip", ["gp_bin_count_kernelILi2E", "gp_bin_scan_kernel", "gp_bin_scatter_kernelILi8E"]),
])
def test_hot_kernels_keep_their_loads_batched(src, kernels):
    from isa_load_audit import CSRC, audit
    stats = audit(os.path.join(CSRC, src), HIPCC)
    for want in kernels:
        want, limit = want if isinstance(want, tuple) else (want, 2)
        hits = [(k, v) for k, v in stats.items() if want in k]
        assert hits, f"{want} not found in {src}"
        for k, (loads, waits, tight) in hits:
            assert loads >= 2 and tight <= limit, f"{k}: {tight} of {waits} full waits sit right behind one of its {loads} loads (serialized loads?)"
