"""GPU: the rasterizer's capacity mode (no host sync: binning sized by a caller-supplied capacity, sentinel-padded sort)
is exact when the capacity holds, is safe and flagged when it does not, and TrainStep's speculative protocol built on it
reproduces the exact-mode optimisation."""
import numpy as np
import pytest
import torch

from util import rel_l2

pytestmark = pytest.mark.gpu


def _render(pc, cam, it, binning=None):
    from gaussianprediction_amd.renderer import render
    from types import SimpleNamespace
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    t = torch.tensor([0.3], device="cuda")
    return render(cam, pc, pipe, torch.zeros(3, device="cuda"), time=t, it=it, binning=binning)


def _grads(pc):
    return {n: p.grad.detach().clone() for n, p in pc.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("slack", [0, 1, 5000])
def test_capacity_mode_equals_exact_mode(slack):
    from test_gpu_render import build
    pc, cam, *_ = build(N=3000, K=60, W=120, H=90)
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    a = _render(pc, cam, 50000, binning=(0, status))                     # exact mode also reports R
    (a["render"] * torch.linspace(0.5, 1.5, 120, device="cuda")).sum().backward()
    ga = _grads(pc)
    R = int(status[0])
    assert R > 0 and int(status[1]) == 0
    for p in pc.parameters():
        p.grad = None
    status2 = torch.full((2,), -1, dtype=torch.int32, device="cuda")
    b = _render(pc, cam, 50000, binning=(R + slack, status2))            # capacity == R is the tightest legal value
    (b["render"] * torch.linspace(0.5, 1.5, 120, device="cuda")).sum().backward()
    gb = _grads(pc)
    assert status2.tolist() == [R, 0]
    assert torch.equal(a["render"], b["render"])                         # the forward is deterministic: bit-exact
    assert torch.equal(a["radii"], b["radii"])
    assert ga.keys() == gb.keys()
    for k in ga:                                                          # backward sums with atomics: order-dependent rounding only
        assert rel_l2(ga[k].cpu().numpy(), gb[k].cpu().numpy()) < 2e-6, k


def test_overflow_is_flagged_safe_and_skips_the_update():
    from test_gpu_render import build, make_args
    from gaussianprediction_amd.train_step import TrainStep
    pc, cam, *_ = build(N=3000, K=60, W=120, H=90, args=make_args())
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    _render(pc, cam, 50000, binning=(0, status))
    R = int(status[0])
    gt = torch.rand(3, 90, 120, generator=torch.Generator().manual_seed(1)).cuda()
    ts = TrainStep(pc, [cam], [gt], 50000)
    before = [p.detach().clone() for p in pc.parameters()]
    st2 = torch.zeros(2, dtype=torch.int32, device="cuda")
    from gaussianprediction_amd._lib import TorchAllocator
    for t in TorchAllocator._temp_arena.values():                         # poison the scratch arena: a stale id read from an
        t.fill_(0x7F)                                                     # unwritten binning slot would fault instead of passing by luck
    ts._step(0, (R // 3, st2), st2[1:2])                                  # a third of the room: overflow
    torch.cuda.synchronize()
    assert int(st2[1]) == 1 and int(st2[0]) > R // 3                      # (R of the camera's own time stamp, not of t = 0.3)
    for p, q in zip(pc.parameters(), before):                             # Adam saw the flag: nothing moved
        assert torch.equal(p.detach(), q)
    ts._step(0, None, None)                                               # the exact path still works afterwards
    torch.cuda.synchronize()
    assert any(not torch.equal(p.detach(), q) for p, q in zip(pc.parameters(), before))


def test_speculative_train_step_matches_exact_and_recovers_from_overflow():
    from test_gpu_render import build, make_args
    from gaussianprediction_amd.cameras import orbit_cameras
    from gaussianprediction_amd.train_step import TrainStep
    outs = []
    for mode in ("exact", "speculative", "overflowing"):
        pc, cam, *_ = build(N=2000, K=40, W=96, H=80, args=make_args())
        cams = orbit_cameras(5, 4.0, 0.6911, 96, 80, device="cuda")[:3]
        gts = [torch.rand(3, 80, 96, generator=torch.Generator().manual_seed(5 + i)).cuda() for i in range(3)]
        ts = TrainStep(pc, cams, gts, 50000, speculative=mode != "exact")
        if mode == "overflowing":
            ts.SPEC_MARGIN, ts.SPEC_PAD = 0.5, 0                          # every capacity-mode frame overflows and is redone
        def exact_loss():        # all three views rendered in exact mode, no update (a frame that overflowed returns the loss of
            from gaussianprediction_amd.renderer import render            # its TRUNCATED render: not a measure of progress)
            with torch.no_grad():
                return sum(float(ts.loss_of(render(cams[v], pc, ts.pipe, ts.bg, time=ts.times[v], it=50000)["render"], gts[v])) for v in range(3))
        l0 = exact_loss()
        losses = [float(ts.step(i)[0]) for i in range(16)]
        torch.cuda.synchronize()
        outs.append((losses, pc._xyz.detach().clone(), pc._features_dc.detach().clone(), getattr(ts, "redone", 0),
                     ts.optimizer.step_count, l0, exact_loss()))
    (la, xa, fa, _, na, a0, a1), (lb, xb, fb, rb, nb, _, _), (lc, xc, fc, rc, nc, c0, c1) = outs
    assert rb == 0 and na == nb == 16
    # 16 Adam steps amplify the backward's atomic-order rounding noise, and the amplification is bimodal: two runs of the SAME mode
    # agree to ~3e-6 (loss) or to ~7e-5, depending on whether the noise flips one discrete decision (a threshold pixel, a radius) on
    # the way -- measured for exact/exact, exact/speculative and speculative/speculative pairs alike (tools history, round 2);
    # the bars sit above the upper mode
    assert np.allclose(la, lb, rtol=1e-3, atol=1e-6), (la, lb)
    assert rel_l2(xa.cpu().numpy(), xb.cpu().numpy()) < 3e-4
    assert rel_l2(fa.cpu().numpy(), fb.cpu().numpy()) < 3e-3
    assert rc > 0                                                         # overflows happened, were detected and redone
    assert torch.isfinite(xc).all() and torch.isfinite(fc).all()
    assert a1 < a0 and c0 - c1 > 0.5 * (a0 - a1)                          # and the optimisation progresses (the last SPEC_SLOTS frames await their redo)


def test_side_stream_sh_adam_matches_the_single_stream_step():
    """TrainStep(overlap_sh_adam=True): the SH tensors are updated on a second stream during the backward; render()
    orders itself behind that update through the event left on the model."""
    from test_gpu_render import build, make_args
    from gaussianprediction_amd.train_step import TrainStep
    outs = []
    for overlap in (False, True):
        pc, cam, *_ = build(N=2000, K=40, W=96, H=80, args=make_args())
        gt = torch.rand(3, 80, 96, generator=torch.Generator().manual_seed(5)).cuda()
        ts = TrainStep(pc, [cam], [gt], 50000, overlap_sh_adam=overlap)
        losses = [float(ts.step(0)[0]) for _ in range(8)]
        ts.wait_side()
        torch.cuda.synchronize()
        outs.append((losses, pc._features_dc.detach().clone(), pc._features_rest.detach().clone(), pc._xyz.detach().clone(),
                     ts.optimizer.step_count))
    (la, da, ra, xa, na), (lb, db, rb, xb, nb) = outs
    assert na == nb == 8
    assert np.allclose(la, lb, rtol=2e-4, atol=1e-6), (la, lb)
    assert rel_l2(da.cpu().numpy(), db.cpu().numpy()) < 1e-3
    assert rel_l2(ra.cpu().numpy(), rb.cpu().numpy()) < 1e-3
    assert rel_l2(xa.cpu().numpy(), xb.cpu().numpy()) < 1e-4


def _f2k(z):
    import struct
    return struct.unpack("<I", struct.pack("<f", z))[0]


def test_depth_key_promise_is_exact_when_kept_and_flagged_when_broken():
    """gp_raster_settings.depth_key_bits: a kept promise (24 bits above the smallest visible key) gives the exact mode's frame bit for
    bit with three sort passes; a broken one -- a visible key below the base, or beyond base + 2^bits -- raises the overflow word."""
    from test_gpu_render import build
    pc, cam, *_ = build(N=3000, K=60, W=120, H=90)
    status = torch.zeros(8, dtype=torch.int32, device="cuda")
    a = _render(pc, cam, 50000, binning=(0, status[0:3], None, status[4:6]))
    R, lo, hi = int(status[0]), int(status[4]) & 0xFFFFFFFF, int(status[5]) & 0xFFFFFFFF
    assert R > 0 and int(status[1]) == 0 and lo < hi
    # the reported range is the visible Gaussians' view-space depth range
    with torch.no_grad():
        xyz, *_ = pc(torch.tensor([0.3], device="cuda"), 50000)
        z = (torch.cat([xyz, torch.ones_like(xyz[:, :1])], 1) @ cam.world_view_transform)[:, 2][a["visibility_filter"]]
    assert abs(np.float32(z.min().item()) / np.array([lo], np.uint32).view(np.float32)[0] - 1) < 1e-5
    assert abs(np.float32(z.max().item()) / np.array([hi], np.uint32).view(np.float32)[0] - 1) < 1e-5
    assert hi - lo < (1 << 24) and hi - lo >= (1 << 16)
    for bits, base, broken in ((24, lo, False), (24, hi - (1 << 24) + 1, False), (31, _f2k(0.2), False),
                               (24, lo + 1, True), (16, lo, True), (8, hi - 255 + 1, True), (24, hi + 1, True)):
        st = torch.full((3,), -1, dtype=torch.int32, device="cuda")
        b = _render(pc, cam, 50000, binning=(R, st, (bits, base)))
        assert int(st[0]) == R or broken
        assert int(st[1]) == int(broken), (bits, base, st.tolist())
        if not broken:
            for k in ("render", "radii", "depth", "tidx"):
                assert torch.equal(a[k], b[k]), (k, bits, base)
    # the scratch word carries the NUMBER of the call whose promise broke: reusing the block for a kept promise is clean without clearing it
    st = torch.zeros(3, dtype=torch.int32, device="cuda")
    _render(pc, cam, 50000, binning=(R, st, (16, lo)))
    assert int(st[1]) == 1
    b = _render(pc, cam, 50000, binning=(R, st, (24, lo)))
    assert st[:2].tolist() == [R, 0] and torch.equal(a["render"], b["render"])
    c = _render(pc, cam, 50000, binning=(0, st, (24, lo)))               # exact binning under a promise
    assert st[:2].tolist() == [R, 0] and torch.equal(a["render"], c["render"])
    with pytest.raises(RuntimeError):
        _render(pc, cam, 50000, binning=(R, st[:2], (24, lo)))           # no scratch word


def test_train_step_speculates_on_the_depth_key_range():
    """TrainStep sizes the promise from the ranges its exact-mode set-up steps report; a promise that breaks is redone in exact mode,
    which widens the range."""
    from test_gpu_render import build, make_args
    from gaussianprediction_amd.cameras import orbit_cameras
    from gaussianprediction_amd.train_step import TrainStep
    res = []
    for spec in (True, False):
        pc, cam, *_ = build(N=2000, K=40, W=96, H=80, args=make_args())
        cams = orbit_cameras(5, 4.0, 0.6911, 96, 80, device="cuda")[:3]
        gts = [torch.rand(3, 80, 96, generator=torch.Generator().manual_seed(5 + i)).cuda() for i in range(3)]
        ts = TrainStep(pc, cams, gts, 50000, speculative=True)
        ts.depth_key_speculation = spec
        losses = [float(ts.step(i)[0]) for i in range(14)]
        torch.cuda.synchronize()
        res.append((losses, ts.last_depth_key_promise, ts.redone, ts))
    (la, pa, ra, ts), (lb, pb, rb, _) = res
    lo, hi = (np.array([k], np.uint32).view(np.float32)[0] for k in (ts._key_lo, ts._key_hi))
    assert pb is None and pa is not None and pa[0] == 24 and ra == 0 and rb == 0, (pa, pb, ra, rb, lo, hi)
    assert pa[1] < ts._key_lo and ts._key_hi < pa[1] + (1 << 24)          # the window holds the range, with room on both sides
    assert abs((ts._key_lo - pa[1]) - (pa[1] + (1 << 24) - 1 - ts._key_hi)) <= 1
    assert np.allclose(la, lb, rtol=1e-3, atol=1e-6), (la, lb)
    # break it: pretend only the nearest Gaussian was ever seen -- the window around it ends at twice its depth
    assert hi > 2.2 * lo
    ts._key_hi = ts._key_lo
    n0 = ts.optimizer.step_count
    for i in range(14, 14 + 3 * ts.SPEC_SLOTS):
        ts.step(i)
    torch.cuda.synchronize()
    assert ts.redone > 0                                                  # detected, repeated in exact mode ...
    assert ts._key_hi - ts._key_lo >= (1 << 16)                           # ... whose report widened the range again
    assert ts.last_depth_key_promise is not None and ts.last_depth_key_promise[0] == 24
    assert ts.optimizer.step_count > n0
    for p in ts.pc.parameters():
        assert torch.isfinite(p).all()
