"""-m gpu: the training-side surface of the reference's GaussianModel on the HIP path -- `forward(..., return_weights=True)`
[REF scene/gaussian_model.py:231,299-303; eval.py:126], checkpoint tuple interchange through the fused Adam
[REF train.py:199-201, scene/gaussian_model.py:96-104], `--batch` accumulation on one GPU [REF train.py:113-133],
furthest-point sampling [REF utils/fps.py:71-88] and a 2-rank view-parallel step (RCCL when two GPUs are visible, gloo with
both ranks on the one GPU otherwise)."""
import math
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gaussianprediction_amd as gpa  # noqa: E402
from gaussianprediction_amd.cameras import orbit_cameras  # noqa: E402
from gaussianprediction_amd.io_formats import load_checkpoint, save_checkpoint  # noqa: E402
from gaussianprediction_amd.renderer import render  # noqa: E402
from gaussianprediction_amd.scene_synth import SceneSpec, make_gaussians, make_keypoints  # noqa: E402
from gaussianprediction_amd.train_step import TrainStep  # noqa: E402
from gaussianprediction_amd.training import default_training_args, furthest_point_sampling  # noqa: E402
from oracle import deform_oracle as do  # noqa: E402


def margs(**kw):
    a = dict(beta=0.1, d=4, w=256, feature_dim=32, second_stage_iteration=30000, third_stage_iteration=40000,
             jointly_iteration=1000, nearest_num=6, norm_rotation=True, step_opacity=False, step_opacity_iteration=5000,
             opacity_type="implicit", xyz_noise_iteration=0, max_points=48, adaptive_points_num=0)
    a.update(kw)
    return SimpleNamespace(**a)


def build(n=4000, K=48, dev="cuda", W=160, H=128, seed=11, **kw):
    args = margs(**kw)
    raw = make_gaussians(SceneSpec(n_gaussians=n, extent=(1.3, 1.3, 1.3), scale_lo=0.01, scale_hi=0.08, seed=seed), device=dev)
    kp, kpf, idx, rw = make_keypoints(raw["xyz"], raw["motion_feature"], K, args.nearest_num)
    torch.manual_seed(1234)                      # the MLP's default nn.Linear init draws from the global generator
    pc = gpa.GaussianModel(3, args)
    tf = 10 if args.step_opacity else 6
    pc.set_inputDim(2 * tf, 60)
    pc.create_from_tensors(raw["xyz"], raw["features_dc"], raw["features_rest"], raw["scaling"], raw["rotation"], raw["opacity"],
                           raw["motion_feature"] * 50, kp, kpf * 50)
    pc.set_keypoint_weights(rw, idx)
    cams = orbit_cameras(4, 4.0, 0.69, W, H, device=dev)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    with torch.no_grad():
        gts = [render(c, pc, pipe, torch.zeros(3, device=dev), time=torch.tensor([0.3], device=dev), it=50000)["render"] * 0.9 for c in cams]
    return pc, cams, gts, raw, rw, idx, args


def test_forward_return_weights_is_the_reference_dense_form():
    pc, cams, gts, raw, rw, idx, args = build(n=1500, K=40)
    t = torch.tensor([0.4], device="cuda")
    with torch.no_grad():
        out4 = pc(t, 50000)
        out6 = pc(t, 50000, return_weights=True)
        assert len(out4) == 4 and len(out6) == 6
        for a, b in zip(out4, out6[:4]):
            assert torch.equal(a, b)
        wx, wr = out6[4], out6[5]
        ox, orr = do.fill_nearest(rw.cpu(), idx.cpu(), 40, 6)                     # the reference's dense scatter [REF :214-229]
        assert wx.shape == (1500, 40) and torch.allclose(wx.cpu(), ox, atol=1e-6) and torch.allclose(wr.cpu(), orr, atol=1e-6)
        assert torch.allclose(pc.weights_sum.cpu(), ox.abs() + orr.abs(), atol=1e-6)
        assert pc.kpts_xyz_motion.shape == (40, 3) and pc.kpts_rotation_motion.shape == (40, 4)
        assert torch.allclose(pc.kpts_rotation_motion.norm(dim=-1), torch.ones(40, device="cuda"), atol=1e-5)
        # the blended displacement equals the dense matmul with those weights [REF :272-273]
        assert torch.allclose(out4[0] - pc._xyz, wx @ pc.kpts_xyz_motion, atol=1e-5)
        assert len(pc(t, 20000, return_weights=True)) == 4                        # stage 1: never six [REF :302]
    pc2, *_ = build(n=1500, K=40, step_opacity=True)
    with torch.no_grad():
        o4 = pc2(t, 50000)
        o6 = pc2(t, 50000, return_weights=True)
        assert len(o6) == 6 and torch.equal(o6[3], pc2.get_opacity) and not torch.equal(o4[3], o6[3])   # [REF :299-300]


def test_furthest_point_sampling_kernel_matches_the_host_restatement():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5000, 3, generator=g)
    from host_checkers import fps_host
    ref = fps_host(x, 64)                           # the tests' torch restatement
    out = furthest_point_sampling(x.cuda(), 64)
    assert out.dtype == torch.int64 and torch.equal(out.cpu(), ref)
    assert int(out[0]) == 0 and len(set(out.tolist())) == 64
    assert furthest_point_sampling(x.cuda()[:1], 5).tolist() == [0]


def test_checkpoint_tuple_round_trips_through_the_fused_adam(tmp_path):
    """Load a reference-layout checkpoint (written by plain torch.optim.Adam with the reference's group names), run one
    step on the HIP path, save, and let torch.optim.Adam load the result."""
    pc, cams, gts, raw, rw, idx, args = build(n=3000)
    pc.training_args = default_training_args()
    ref_opt = torch.optim.Adam(pc._stage_groups(3), lr=0.0, eps=1e-15)
    g = torch.Generator(device="cuda").manual_seed(2)
    for _ in range(3):
        for grp in ref_opt.param_groups:
            for p in grp["params"]:
                p.grad = 1e-3 * torch.randn(p.shape, generator=g, device="cuda")
        ref_opt.step()
    path = os.path.join(tmp_path, "chkpnt50000.pth")
    torch.save((pc.state_dict(), ref_opt.state_dict(), 50000), path)               # [REF train.py:199-201]
    m, opt_state, it = load_checkpoint(path, args, device="cuda")
    assert it == 50000
    m.set_keypoint_weights(rw, idx)
    m.restore(opt_state, default_training_args(), it)                               # [REF train.py:48-57]
    assert type(m.optimizer).__name__ == "FusedAdam" and m.optimizer.step_count == 3
    names = [gg["name"] for gg in m.optimizer.param_groups]
    assert names == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "s_xyz", "s_motion_feature", "df_mlp"]
    ts = TrainStep(m, cams, gts, it)
    before = m._xyz.detach().clone()
    loss, pkg = ts.step(0)
    assert math.isfinite(float(loss)) and m.optimizer.step_count == 4 and not torch.equal(before, m._xyz.detach())
    out = os.path.join(tmp_path, "chkpnt50001.pth")
    save_checkpoint(m, m.optimizer.state_dict(), it + 1, out)
    sd_model, sd_opt, it2 = torch.load(out, weights_only=False)
    fresh = torch.optim.Adam([{"params": [torch.nn.Parameter(p.detach().clone()) for p in gg["params"]], "lr": 0.0, "name": gg["name"]}
                              for gg in m.optimizer.param_groups], lr=0.0, eps=1e-15)
    fresh.load_state_dict(sd_opt)
    st = fresh.state[fresh.param_groups[0]["params"][0]]
    assert float(st["step"]) == 4.0 and st["exp_avg"].shape == m._xyz.shape and it2 == 50001
    assert set(sd_model) == set(pc.state_dict())


def test_batch_accumulation_equals_the_sum_of_single_view_losses():
    """--batch B on one GPU [REF train.py:113-133]: gradients of sum_b loss_b, radii = max, visibility = any, and the
    last-view quirk of the densification input."""
    pc, cams, gts, raw, rw, idx, args = build(n=3000)
    ts = TrainStep(pc, cams, gts, 50000, lrs=dict(xyz=0.0, f_dc=0.0, opacity=0.0, scaling=0.0, rotation=0.0, kpts=0.0, mlp=0.0), batch=2)
    # reference accumulation by hand, plain autograd
    for p in pc.parameters():
        if p.grad is not None:
            p.grad.zero_()
    losses, pk = [], []
    for v in (2, 3):
        pkg = render(cams[v], pc, ts.pipe, ts.bg, time=ts.times[v], it=50000)
        losses.append(ts.loss_of(pkg["render"], gts[v])); pk.append(pkg)
    torch.stack(losses).sum().backward()
    want = {k: getattr(pc, k).grad.clone() for k in ("_xyz", "_features_rest", "_opacity", "super_gaussians")}
    want_mlp = [p.grad.clone() for p in pc.df_model.parameters()]
    ts.bucket.zero()
    captured = {}
    orig = ts.optimizer.step

    def spy(*a, **k):                               # the gradients as Adam sees them (it zeroes them in the same launch)
        captured.update({kk: getattr(pc, kk).grad.clone() for kk in want})
        captured["mlp"] = [p.grad.clone() for p in pc.df_model.parameters()]
        return orig(*a, **k)
    ts.optimizer.step = spy
    loss, pkg = ts.step(1)                          # step 1 with batch 2 = views 2, 3
    assert abs(float(loss) - float(torch.stack(losses).sum())) < 1e-5
    for k in want:
        e = float((captured[k] - want[k]).norm() / want[k].norm().clamp_min(1e-30))
        assert e < 1e-5, (k, e)
    for a, b in zip(captured["mlp"], want_mlp):
        assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < 1e-4
    assert torch.equal(pkg["radii"], torch.maximum(pk[0]["radii"], pk[1]["radii"]))
    assert torch.equal(pkg["visibility_filter"], pk[0]["visibility_filter"] | pk[1]["visibility_filter"])
    # the tensor handed on is the LAST view's (its .grad holds that view alone); the batch sum is beside it [REF train.py:123-127,167]
    assert torch.allclose(pkg["viewspace_points"].grad, pk[1]["viewspace_points"].grad, atol=1e-7)
    assert torch.allclose(pkg["viewspace_point_tensor_grad"], pk[0]["viewspace_points"].grad + pk[1]["viewspace_points"].grad, atol=1e-7)


# ---- two ranks ------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank_main(rank, world, port, out_dir, backend, step_opacity):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pc, cams, gts, raw, rw, idx, args = build(n=3000, dev=f"cuda:{dev}", step_opacity=step_opacity)
    zero = dict(xyz=0.0, f_dc=0.0, opacity=0.0, scaling=0.0, rotation=0.0, kpts=0.0, mlp=0.0)
    ts = TrainStep(pc, cams, gts, 50000, lrs=zero, sharded=False)
    assert ts.reducer.enabled
    got = {}
    orig = ts.optimizer.step

    def spy(*a, **k):
        got["flat"] = ts.bucket.flat.clone()
        return orig(*a, **k)
    for step in range(2):                            # second step: the overwrite-sink (stale gradient) protocol is live
        ts.optimizer.step = spy
        loss, pkg = ts.step(step * world + rank)     # rank r renders view world*step + r
        torch.save({"flat": got["flat"].cpu(), "radii": pkg["radii"].cpu(), "offsets": ts.bucket.offsets,
                    "names": [n for n, _ in pc.named_parameters() if _.requires_grad]}, os.path.join(out_dir, f"r{rank}_s{step}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("step_opacity", [False, True])
def test_two_rank_view_parallel_step_equals_batch_accumulation(tmp_path, step_opacity):
    """Each rank renders its own view; the gradient Adam sees on every rank is the SUM over ranks == the single-process
    `--batch 2` gradient.  With the lifecycle opacity `_xyz` has two gradient producers (blend backward + the second MLP
    pass), the case the reducer must not reduce early."""
    import torch.multiprocessing as mp
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path), backend, step_opacity), nprocs=2, join=True)
    pc, cams, gts, raw, rw, idx, args = build(n=3000, step_opacity=step_opacity)
    zero = dict(xyz=0.0, f_dc=0.0, opacity=0.0, scaling=0.0, rotation=0.0, kpts=0.0, mlp=0.0)
    ts = TrainStep(pc, cams, gts, 50000, lrs=zero, batch=2)
    for step in range(2):
        got = {}
        orig = ts.optimizer.step

        def spy(*a, **k):
            got["flat"] = ts.bucket.flat.clone()
            return orig(*a, **k)
        ts.optimizer.step = spy
        loss, pkg = ts.step(step)                    # views 2*step, 2*step + 1
        ts.optimizer.step = orig
        r0 = torch.load(os.path.join(tmp_path, f"r0_s{step}.pt"))
        r1 = torch.load(os.path.join(tmp_path, f"r1_s{step}.pt"))
        assert torch.equal(r0["flat"], r1["flat"])   # every rank holds the same reduced gradient
        want = got["flat"].cpu()
        offs = r0["offsets"] + [want.numel()]
        for k in range(len(offs) - 1):
            a, b = r0["flat"][offs[k]:offs[k + 1]], want[offs[k]:offs[k + 1]]
            e = float((a - b).norm() / b.norm().clamp_min(1e-30))
            assert e < 2e-5, (step, k, e)
        assert torch.equal(r0["radii"], pkg["radii"].cpu())


def _rank_main_params(rank, world, port, out_dir, backend, sharded, step_opacity):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pc, cams, gts, raw, rw, idx, args = build(n=3000, dev=f"cuda:{dev}", step_opacity=step_opacity)
    pc.bucket_small_numel = 8000          # (3000 Gaussians: the per-Gaussian tensors get regions of their own, as at 1 M)
    ts = TrainStep(pc, cams, gts, 50000, sharded=sharded, chain_sh=os.environ.get("GP_TEST_NO_CHAIN") != "1")
    assert ts.sharded == sharded and type(ts.reducer).__name__ == ("ShardedExchange" if sharded else "OverlappedGradReducer")
    chained = []
    if sharded and ts.chain_sh:
        orig_chain = ts.reducer.chain
        ts.reducer.chain = lambda region, h: (lambda ok: (chained.append(region[0]) if ok else None, ok)[1])(orig_chain(region, h))
    for step in range(3):
        ts.step(step * world + rank)
    if sharded and ts.chain_sh:          # the SH regions' Adam + all-gather ran on the exchange's side stream (TrainStep._chain_sh)
        assert len(chained) == 3 and len(set(chained)) == 1 and len(ts._chained_params) == 2, chained     # (one region: dc + rest)
    ts.sync_params()
    sd = pc.optimizer.state_dict()                       # (sharded: a collective)
    torch.save({"params": {n: p.detach().cpu() for n, p in pc.named_parameters()}, "state": {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in sd["state"].items()},
                "bytes": getattr(ts.reducer, "bytes_sent_per_step", None), "n": pc.bucket.flat.numel()}, os.path.join(out_dir, f"p{int(sharded)}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("step_opacity", [False, True])
def test_two_rank_sharded_adam_equals_replicated_adam(tmp_path, step_opacity):
    """reduce-scatter -> Adam on 1/world of every region -> all-gather leaves every rank with the parameters (and, gathered, the
    moments) that all-reduce + replicated Adam produce."""
    import torch.multiprocessing as mp
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    for sharded in (False, True):
        mp.spawn(_rank_main_params, args=(2, _free_port(), str(tmp_path), backend, sharded, step_opacity), nprocs=2, join=True)
    rep = [torch.load(os.path.join(tmp_path, f"p0_{r}.pt")) for r in range(2)]
    sh = [torch.load(os.path.join(tmp_path, f"p1_{r}.pt")) for r in range(2)]
    for name in rep[0]["params"]:
        a, b = rep[0]["params"][name], sh[0]["params"][name]
        assert torch.equal(sh[0]["params"][name], sh[1]["params"][name]), name          # replicas stay replicas
        # the backward accumulates with atomics (run-to-run differences in the last bits of a gradient), and Adam with eps = 1e-15
        # moves an element whose gradient is rounding noise by a full +-lr: a few such elements may differ by a couple of steps
        diff = (a - b).abs()
        frac = float((diff > 1e-6).float().mean())
        print(f"[two-rank] {name}: fraction moved apart {frac:.4f}, max {float(diff.max()):.3e}")
        assert frac < 5e-2 and float(diff.max()) < 0.05 * 3 + 1e-6, (name, frac, float(diff.max()))      # (measured: <= 0.5 % except _opacity
        # under the lifecycle opacity, 1.5 - 2.2 % from run to run: Gaussians whose opacity gradient is rounding noise)
    assert set(rep[0]["state"]) == set(sh[0]["state"]) and len(sh[0]["state"]) > 5
    for k in rep[0]["state"]:
        a, b = sh[0]["state"][k]["exp_avg"], rep[0]["state"][k]["exp_avg"]
        rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
        print(f"[two-rank] exp_avg {k}: rel {rel:.3e}")
        assert rel < 1e-2, (k, rel)      # (parameters drift apart by rounding-noise steps, see above)
        assert float(sh[0]["state"][k]["step"]) == 3.0
    assert sh[0]["bytes"] == 4 * sh[0]["n"]          # world 2: reduce-scatter + all-gather move half the flat buffer each, per rank


def _rank_main_knn(rank, world, port, out_dir, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussianprediction_amd import gaussian_model as gm
    from gaussianprediction_amd.weights_ops import WeightsModel
    pc, cams, gts, raw, rw, idx, args = build(n=3000, dev=f"cuda:{dev}", knn_type="hybird", feature_amplify=5.0)
    pc.weights_model = WeightsModel(2 * args.nearest_num, device=f"cuda:{dev}")
    pc.set_keypoint_weights(None, None)            # stage 3 as the reference runs it: weights model + neighbour search inside forward
    pc.bucket_small_numel = 8000                   # grouped regions: with two ranks one of them owns no slice of _xyz / the keypoints
    calls, orig = [], gm.knn_keypoints
    gm.knn_keypoints = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    ts = TrainStep(pc, cams, gts, 50000, sharded=True)
    owned = {id(p) for p in pc.optimizer.owner}
    for step in range(3):
        ts.step(step * world + rank)
    ts.sync_params()
    with torch.no_grad():
        pc(torch.tensor([0.3], device=f"cuda:{dev}"), 50000)          # one more forward on the final parameters
    torch.cuda.synchronize()
    torch.save({"calls": len(calls), "knn": pc._knn_cache[1].cpu(), "xyz": pc._xyz.detach().cpu(),
                "owns_xyz": id(pc._xyz) in owned, "owns_kp": id(pc.super_gaussians) in owned}, os.path.join(out_dir, f"knn_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_step_recomputes_the_neighbour_search_on_every_rank(tmp_path):
    """The neighbour-search cache is keyed on parameter versions; under the sharded optimizer a rank that holds no slice of
    `_xyz` / the keypoints receives their new values through the all-gather only.  Every rank must still see the change: one
    search per step on each rank, and identical indices at the end (round-3 advisor finding: ranks without a slice kept stale
    indices)."""
    import torch.multiprocessing as mp
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    mp.spawn(_rank_main_knn, args=(2, _free_port(), str(tmp_path), backend), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f"knn_{k}.pt")) for k in range(2)]
    assert not (r[0]["owns_xyz"] and r[1]["owns_xyz"]) or not (r[0]["owns_kp"] and r[1]["owns_kp"])   # (the hazardous layout is the one tested)
    for k in range(2):
        assert r[k]["calls"] == 4, (k, r[k]["calls"])       # 3 training steps + the final forward: the parameters changed each time
    assert torch.equal(r[0]["xyz"], r[1]["xyz"]) and torch.equal(r[0]["knn"], r[1]["knn"])


def _rank_main_rccl_single(rank, world, port, out_dir, sharded):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", GP_DIST_FORCE_SINGLE="1")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pc, cams, gts, raw, rw, idx, args = build(n=3000, dev="cuda:0")
    pc.bucket_small_numel = 8000          # (3000 Gaussians: the per-Gaussian tensors get regions of their own, as at 1 M)
    ts = TrainStep(pc, cams, gts, 50000, sharded=sharded)
    assert ts.reducer.enabled and ts.reducer.nccl if sharded else ts.reducer.enabled
    assert type(ts.reducer).__name__ == ("ShardedExchange" if sharded else "OverlappedGradReducer")
    chained, events = [], []
    if sharded:      # the SH regions leave the compute stream after their reduce-scatter (Adam + all-gather on the side stream) ...
        orig_chain, orig_late = ts.reducer.chain, pc._param_late_event
        ts.reducer.chain = lambda region, h: (lambda ok: (chained.append(region[0]) if ok else None, ok)[1])(orig_chain(region, h))
        pc._param_late_event = lambda: (lambda ev: (events.append(ev), ev)[1])(orig_late())
    losses = [float(ts.step(step)[0]) for step in range(4)]
    if sharded:      # ... every step; and from the second step on render() hands the rasterizer an event for them
        assert len(chained) == 4 and len(set(chained)) == 1, chained            # (one region holds both SH tensors)
        assert events[0] is None and all(e is not None for e in events[1:]), events
    ts.sync_params()
    torch.cuda.synchronize()
    torch.save({"params": {n: p.detach().cpu() for n, p in pc.named_parameters()}, "losses": losses}, os.path.join(out_dir, f"single{int(sharded)}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded", [True, False])
def test_rccl_code_path_on_a_one_rank_group(tmp_path, sharded):
    """The RCCL branch of the exchange (in-place reduce_scatter_tensor / all_gather_into_tensor on views of the flat buffers, the
    post-accumulate hooks, asynchronous handles awaited on the compute stream) run for real on a ONE-rank "nccl" group
    (GP_DIST_FORCE_SINGLE): a SUM over one rank must reproduce the plain single-process steps.  This is what a one-GPU box can
    check of the multi-GPU legs; the two-rank tests above use RCCL themselves as soon as two devices are visible."""
    import torch.multiprocessing as mp
    mp.spawn(_rank_main_rccl_single, args=(1, _free_port(), str(tmp_path), sharded), nprocs=1, join=True)
    got = torch.load(os.path.join(tmp_path, f"single{int(sharded)}.pt"))
    pc, cams, gts, raw, rw, idx, args = build(n=3000)
    ts = TrainStep(pc, cams, gts, 50000)
    losses = [float(ts.step(step)[0]) for step in range(4)]
    assert all(abs(a - b) < 1e-4 * max(1.0, abs(b)) for a, b in zip(got["losses"], losses)), (got["losses"], losses)
    for name, p in pc.named_parameters():
        diff = (got["params"][name] - p.detach().cpu()).abs()
        assert float((diff > 1e-6).float().mean()) < 2e-2 and float(diff.max()) < 0.05 * 4 + 1e-6, (name, float(diff.max()))


def test_sh_adam_fused_into_the_backward_equals_the_separate_step():
    """One view, one rank: the rasterizer backward applies Adam to the SH coefficients itself (gp_adam_fuse; their gradient is never
    written) and the optimizer launch covers the other tensors.  Parameters and moments must equal the separate-step run up to the
    run-to-run noise of the backward's float atomics (the arithmetic is one shared device function)."""
    from gaussianprediction_amd import grad_sink
    runs = {}
    for fuse in (True, False):
        pc, cams, gts, raw, rw, idx, args = build(n=3000)
        ts = TrainStep(pc, cams, gts, 50000, fuse_sh_adam=fuse)
        seen = []
        orig = ts.optimizer.step

        def spy(*a, _orig=orig, _seen=seen, **k):
            _seen.append(k.get("exclude"))
            return _orig(*a, **k)
        ts.optimizer.step = spy
        d0 = pc._features_rest.detach().clone()
        for step in range(3):
            ts.step(step)
        torch.cuda.synchronize()
        assert all((e is not None) == fuse for e in seen) and len(seen) == 3
        if fuse:
            assert grad_sink.is_stale(pc._features_rest.grad) and grad_sink.is_stale(pc._features_dc.grad)
        assert not torch.equal(pc._features_rest.detach(), d0)
        mom = pc.adam_moments()
        runs[fuse] = (pc, {n: p.detach().clone() for n, p in pc.named_parameters()},
                      {n: tuple(t.clone() for t in mom[id(p)]) for n, p in pc.named_parameters() if id(p) in mom}, ts.optimizer.step_count)
    (pa, A, MA, sa), (pb, B, MB, sb) = runs[True], runs[False]
    assert sa == sb == 3
    for name in ("_features_dc", "_features_rest", "_xyz", "_opacity"):
        diff = (A[name] - B[name]).abs()
        assert float((diff > 1e-6).float().mean()) < 2e-2, (name, float(diff.max()))
        for k in range(2):
            a, b = MA[name][k], MB[name][k]
            assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < 1e-3, (name, k)
    # a later ordinary backward (no harness) must not accumulate into the stale buffers
    pc = pa
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    pkg = gpa.render(cams[0], pc, pipe, torch.zeros(3, device="cuda"), time=torch.tensor([0.3], device="cuda"), it=50000)
    pkg["render"].sum().backward()
    g1 = pc._features_rest.grad.clone()
    pc._features_rest.grad.zero_(); pc._features_dc.grad.zero_()
    pkg = gpa.render(cams[0], pc, pipe, torch.zeros(3, device="cuda"), time=torch.tensor([0.3], device="cuda"), it=50000)
    pkg["render"].sum().backward()
    assert float((g1 - pc._features_rest.grad).norm() / pc._features_rest.grad.norm().clamp_min(1e-30)) < 1e-4


def test_stage_transitions_and_teaching_path_inside_forward():
    """The reference switches stages INSIDE forward [REF scene/gaussian_model.py:246-250]: at second_stage_iter + 1 the keypoints are
    initialised by k-means and the stage-2 optimizer is built, at third_stage_iter + 1 the stage-3 one; with densify_from_teaching the
    stage-1 "teacher" motion of every Gaussian is compared with the blended one and proposes new keypoints [REF :274-283, 306-312]."""
    args = margs(max_points=24, adaptive_points_num=8, densify_from_teaching=True, adaptive_from_iter=3000, adaptive_end_iter=10000,
                 adaptive_interval=200, teaching_threshold=0.0, knn_type="hybird", feature_amplify=5.0, densify_from_grad="True")
    raw = make_gaussians(SceneSpec(n_gaussians=4000, extent=(1.3, 1.3, 1.3), scale_lo=0.01, scale_hi=0.08, seed=11), device="cuda")
    torch.manual_seed(1234)
    pc = gpa.GaussianModel(3, args)
    pc.set_inputDim(12, 60)
    pc.create_from_tensors(raw["xyz"], raw["features_dc"], raw["features_rest"], raw["scaling"], raw["rotation"], raw["opacity"],
                           raw["motion_feature"] * 50, with_weights_model=True)
    pc.training_setup(default_training_args())
    assert [g["name"] for g in pc.optimizer.param_groups][-2:] == ["df_mlp", "motion_feature"] and not pc.second_stage
    t = torch.tensor([0.4], device="cuda")
    with torch.no_grad():
        pc(t, 30000)                                     # still stage 1
        assert not hasattr(pc, "super_gaussians")
        out = pc(t, 30001)                               # -> k-means keypoints + training2stage_setup
    assert pc.second_stage and pc.super_gaussians.shape == (24, 3) and pc.super_gaussians_feature.shape == (24, 32)
    assert [g["name"] for g in pc.optimizer.param_groups] == ["s_xyz", "s_motion_feature", "weight_mlp", "df_mlp"]
    assert len(out) == 4 and torch.isfinite(out[0]).all()
    # every keypoint is the mean position of a non-empty cluster of Gaussians: inside their bounding box
    lo, hi = raw["xyz"].min(0).values, raw["xyz"].max(0).values
    assert bool(((pc.super_gaussians.detach() >= lo) & (pc.super_gaussians.detach() <= hi)).all())
    # teaching window [adaptive_from_iter, adaptive_end_iter) + second_stage_iter: statistics every frame, candidates on the interval
    with torch.no_grad():
        pc(t, 33001)
        assert float(pc.motion_denom.max()) == 1.0 and pc.new_xyz is None
        pc(t, 33200)                                     # 33200 % 200 == 0, threshold 0: every Gaussian is a candidate
    assert pc.new_xyz is not None and pc.new_xyz.shape == (8, 3)          # 4000 // 100 = 40, clipped to the 8 free slots
    from gaussianprediction_amd import densify as dn
    grown = dn.keypoint_growth_step(pc, 33200, default_training_args(), args)
    assert grown and pc.super_gaussians.shape == (32, 3) and pc.new_xyz is None
    assert pc.optimizer.param_groups[0]["params"][0] is pc.super_gaussians
    with torch.no_grad():
        out = pc(t, 33201)                               # kNN / weights recomputed for 32 keypoints
    assert torch.isfinite(out[0]).all()
    with torch.no_grad():
        pc(t, 40001)                                     # -> training3stage_setup
    assert pc.third_stage and [g["name"] for g in pc.optimizer.param_groups][:6] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_adam_kernel_equals_torch_adam_on_the_summed_loss(tmp_path, world):
    """tests/test_dist_gloo.py's sharded-optimizer check (reduce-scatter -> Adam on this rank's 1/world of every region -> all-gather
    == torch.optim.Adam on the SUM of the ranks' losses, parameters and gathered moments) with the buckets on the GPU: the slices
    are updated by gp_adam_step_multi, not by the CPU tests' restatement.  All ranks share the one device; collectives run on gloo."""
    from test_dist_gloo import check_sharded_against_torch_adam
    check_sharded_against_torch_adam(tmp_path, world, device="cuda:0")


def test_keypoint_weights_and_knn_are_evaluated_once_per_parameter_state():
    """The reference evaluates the hash-grid weights model and the kNN on every forward [REF scene/gaussian_model.py:257-260];
    neither depends on the frame time.  Evaluation frames and the views of one `--batch` step share one evaluation; anything
    that changes an input (an optimizer step: the kernels bump the tensors' version counters) invalidates it.  Results are the
    uncached ones, bit for bit."""
    from gaussianprediction_amd import weights_ops
    from gaussianprediction_amd.weights_ops import WeightsModel
    pc, cams, gts, raw, rw, idx, args = build(n=3000, knn_type="hybird", feature_amplify=5.0)
    pc.weights_model = WeightsModel(2 * args.nearest_num, log2_hashmap_size=12, device="cuda")
    pc.set_keypoint_weights(None, None)
    calls = {"knn": 0, "wm": 0}
    import gaussianprediction_amd.gaussian_model as gm
    orig_knn, orig_fwd = gm.knn_keypoints, WeightsModel.forward
    gm.knn_keypoints = lambda *a, **k: (calls.__setitem__("knn", calls["knn"] + 1), orig_knn(*a, **k))[1]
    WeightsModel.forward = lambda self, x: (calls.__setitem__("wm", calls["wm"] + 1), orig_fwd(self, x))[1]
    try:
        pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
        bg = torch.zeros(3, device="cuda")
        with torch.no_grad():
            imgs = [render(cams[k % 4], pc, pipe, bg, time=torch.tensor([0.1 * k], device="cuda"), it=50000)["render"].clone() for k in range(5)]
        assert calls == {"knn": 1, "wm": 1}, calls                       # five frames, one evaluation
        pc._knn_cache = pc._rw_cache = None
        with torch.no_grad():
            again = render(cams[3], pc, pipe, bg, time=torch.tensor([0.3], device="cuda"), it=50000)["render"]
        assert torch.equal(again, imgs[3]) and calls == {"knn": 2, "wm": 2}
        # training, one view per step: every step changes the inputs -> one evaluation per step (and the autograd graph is never shared)
        ts = TrainStep(pc, cams, gts, 50000)
        v0 = pc._xyz._version
        for s in range(3):
            ts.step(s)
        # (the first step still sees the state of the last evaluation frame: its neighbour search is served from the cache, the
        # weights model -- now under autograd -- is not)
        assert pc._xyz._version > v0 and calls == {"knn": 4, "wm": 5}, calls
        # --batch 2: both views of a step share the evaluation; the gradient of the weights model is the sum over the views
        ts2 = TrainStep(pc, cams, gts, 50000, batch=2)
        for g_ in pc.optimizer.param_groups:
            g_["lr"] = 0.0                              # (the step must leave the parameters where the per-view check below finds them)
        got = {}
        orig_step = ts2.optimizer.step
        ts2.optimizer.step = lambda *a, **k: (got.__setitem__("g", pc.weights_model.params.grad.clone()), orig_step(*a, **k))[1]
        ts2.step(0)
        assert calls == {"knn": 5, "wm": 6}, calls
        gsum = torch.zeros_like(got["g"])
        for v in (0, 1):
            pc.weights_model.params.grad.zero_()
            pkg = render(cams[v], pc, pipe, bg, time=ts2.times[v], it=50000)
            ts2.loss_of(pkg["render"], gts[v]).backward()
            gsum += pc.weights_model.params.grad
        rel = float((got["g"] - gsum).norm() / gsum.norm())
        assert rel < 1e-4, rel
    finally:
        gm.knn_keypoints, WeightsModel.forward = orig_knn, orig_fwd


def test_create_from_pcd_and_the_checkpoint_sizing_path():
    """[REF scene/gaussian_model.py:327-392; train.py:50-57, eval.py:238-245]: Gaussians from a point cloud (distCUDA2 -> initial
    scale), and the way the reference sizes a model before `load_state_dict`: N_pcd_init copies of the first point,
    final_kpts_num keypoints."""
    from types import SimpleNamespace as NS
    from gaussianprediction_amd.weights_ops import dist_cuda2
    from oracle import weights_oracle as wo
    rng = np.random.default_rng(4)
    pts = rng.uniform(-1.3, 1.3, size=(5000, 3)).astype(np.float32)
    pts[100] = pts[7]                                       # a coincident pair: distance exactly 0, counted (self is excluded by index)
    d2 = dist_cuda2(torch.tensor(pts).cuda()).cpu().numpy()
    want = wo.knn3_mean_dist2(pts.astype(np.float64))
    assert np.allclose(d2, want, rtol=2e-5, atol=1e-9) and d2[7] < want.mean()
    for n in (1, 3, 4, 257):                               # fewer than three other points -> FLT_MAX terms, as the published kernel
        got = dist_cuda2(torch.tensor(pts[:n]).cuda()).cpu().numpy()
        assert got.shape == (n,) and (np.isfinite(got).all() and got.max() < 1e30) == (n >= 4)
    args = margs(max_points=48)
    pc = gpa.GaussianModel(3, args)
    pc.set_inputDim(12, 60)
    cols = rng.uniform(0, 1, size=(5000, 3)).astype(np.float32)
    pc.create_from_pcd(NS(points=pts, colors=cols, normals=np.zeros_like(pts)), spatial_lr_scale=2.5)
    assert pc.N_pcd_init == 5000 and pc.spatial_lr_scale == 2.5 and pc.active_sh_degree == 0
    assert pc._xyz.shape == (5000, 3) and pc._features_dc.shape == (5000, 1, 3) and pc._features_rest.shape == (5000, 15, 3)
    assert torch.allclose(pc._features_dc[:, 0].cpu(), (torch.tensor(cols) - 0.5) / 0.28209479177387814, atol=1e-6)
    assert float(pc._features_rest.abs().max()) == 0.0
    assert np.allclose(pc._scaling.detach().cpu().numpy(), np.log(np.sqrt(np.maximum(want, 1e-7)))[:, None].repeat(3, 1), atol=2e-5)
    assert torch.equal(pc._rotation.detach().cpu(), torch.tensor([1.0, 0, 0, 0]).repeat(5000, 1))
    assert torch.allclose(torch.sigmoid(pc._opacity.detach()), torch.full((5000, 1), 0.1, device="cuda"), atol=1e-6)
    assert pc.super_gaussians.shape == (48, 3) and float(pc.motion_feature.abs().max()) <= 1e-3 and pc.weights_model is not None
    # render + one training step straight from the point cloud (stage 0/1: static warm-up then the MLP)
    cams = orbit_cameras(2, 4.0, 0.69, 96, 80, device="cuda")
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    img = render(cams[0], pc, pipe, torch.zeros(3, device="cuda"), time=torch.tensor([0.2], device="cuda"), it=100)["render"]
    assert torch.isfinite(img).all() and float(img.sum()) > 0
    # the reference's checkpoint path: size an empty model, then load the trained state
    sd = pc.state_dict()
    pc2 = gpa.GaussianModel(3, args)
    pc2.set_inputDim(12, 60)
    pc2.N_pcd_init, pc2.final_kpts_num = 5000, 48
    pc2.create_from_pcd(NS(points=pts[:10], colors=cols[:10], normals=None), spatial_lr_scale=2.5)
    assert pc2._xyz.shape == (5000, 3) and torch.equal(pc2._xyz[0], pc2._xyz[4999])          # N_pcd_init copies of the first point
    missing, unexpected = pc2.load_state_dict(sd, strict=False)
    assert not unexpected and torch.equal(pc2._xyz.detach(), pc._xyz.detach()) and torch.equal(pc2.weights_model.params.detach(), pc.weights_model.params.detach())


def _rank_main_spec(rank, world, port, out_dir, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pc, cams, gts, raw, rw, idx, args = build(n=3000, dev="cuda:0")
    pc.bucket_small_numel = 8000
    ts = TrainStep(pc, cams, gts, 50000, speculative=mode != "exact")
    if mode == "overflowing" and rank == 1:
        ts.SPEC_MARGIN, ts.SPEC_PAD = 0.5, 0          # only rank 1's frames overflow: rank 0 must skip and redo them too
    if mode == "overflowing":
        flags = []
        orig = ts._step
        ts._step = lambda v, b, sf: (lambda out: (flags.append(None if sf is None else int(sf.item())), out)[1])(orig(v, b, sf))
    losses = [float(ts.step(i * world + rank)[0]) for i in range(14)]
    ts.sync_params()
    torch.cuda.synchronize()
    torch.save({"losses": losses, "xyz": pc._xyz.detach().cpu(), "dc": pc._features_dc.detach().cpu(), "redone": getattr(ts, "redone", 0),
                "steps": ts.optimizer.step_count, "flags": flags if mode == "overflowing" else None}, os.path.join(out_dir, f"sp_{mode}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_capacity_mode_overflow_is_agreed_on_by_all_ranks(tmp_path):
    """Capacity mode (no host read of R) under the sharded exchange: the overflow word is all-reduced (MAX) right after the forward,
    ahead of the backward, and both Adam launches of the step -- the SH region's on the side stream, the rest on the compute stream --
    take it as their skip flag.  One rank overflowing must make EVERY rank skip that step and repeat it; replicas stay replicas."""
    import torch.multiprocessing as mp
    res = {}
    for mode in ("exact", "speculative", "overflowing"):
        mp.spawn(_rank_main_spec, args=(2, _free_port(), str(tmp_path), mode), nprocs=2, join=True)
        res[mode] = [torch.load(os.path.join(tmp_path, f"sp_{mode}_{r}.pt")) for r in range(2)]
    for mode, (a, b) in res.items():
        assert torch.equal(a["xyz"], b["xyz"]) and torch.equal(a["dc"], b["dc"]), mode      # every rank holds the same parameters
        assert a["steps"] == b["steps"]
    ex, sp, ov = res["exact"][0], res["speculative"][0], res["overflowing"][0]
    assert sp["redone"] == 0 and ex["steps"] == sp["steps"] == 14
    assert np.allclose(ex["losses"], sp["losses"], rtol=2e-3, atol=1e-6)
    assert float((ex["xyz"] - sp["xyz"]).norm() / ex["xyz"].norm()) < 3e-4
    # rank 1 overflowed; rank 0 never did on its own, yet saw the flag (MAX over the ranks), skipped and redid the same steps
    assert res["overflowing"][1]["redone"] > 0 and res["overflowing"][0]["redone"] == res["overflowing"][1]["redone"]
    assert any(f == 1 for f in res["overflowing"][0]["flags"] if f is not None)
    assert ov["steps"] == 14 and torch.isfinite(ov["xyz"]).all() and ov["losses"][-1] < ov["losses"][0]


# ---- factorised SH exchange --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_sh_factor_gradient_kernel_matches_the_torch_statement(degree):
    """gp_sh_factor_gradient (sum over the views, in order, of Y_k(dir) x dL/dRGB) against the torch statement of the SH basis."""
    import host_checkers
    from gaussianprediction_amd.dist import sh_factor_gradient
    g = torch.Generator().manual_seed(degree)
    n, world = 1003, 3
    f = torch.randn(world, n, 6, generator=g)
    f[:, :, 3:6] /= f[:, :, 3:6].norm(dim=2, keepdim=True)
    f[1, ::7, 0:3] = 0.0
    want_dc, want_rest = torch.empty(n, 1, 3), torch.empty(n, 15, 3)
    host_checkers.sh_factor_gradient_host(f, degree, want_dc, want_rest)
    got_dc, got_rest = torch.full((n, 1, 3), float("nan"), device="cuda"), torch.full((n, 15, 3), float("nan"), device="cuda")
    sh_factor_gradient(f.cuda(), degree, got_dc, got_rest)
    torch.testing.assert_close(got_dc.cpu(), want_dc, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got_rest.cpu(), want_rest, rtol=1e-5, atol=1e-6)
    if degree < 3:
        assert float(got_rest[:, (degree + 1) ** 2 - 1:].abs().max()) == 0.0          # coefficients beyond the degree: written as zeros


def test_sh_factors_reproduce_the_rasterizer_backwards_sh_gradient():
    """The factors a rank sends -- dL/dRGB = (features_dc gradient) / C0 and the unit direction camera -> deformed Gaussian -- rebuild
    through gp_sh_factor_gradient the SH gradient the rasterizer backward itself wrote for that view (the basis, its signs and the
    direction convention are the kernels' own [REF utils/sh_utils.py:57-112, gaussian_renderer/__init__.py:86-91])."""
    from gaussianprediction_amd.dist import SH_C0, sh_factor_gradient
    pc, cams, gts, *_ = build(n=3000)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    pc.active_sh_degree = 3
    pkg = render(cams[1], pc, pipe, torch.zeros(3, device="cuda"), time=torch.tensor([0.3], device="cuda"), it=50000)
    (pkg["render"] * torch.randn_like(pkg["render"])).sum().backward()
    g_dc, g_rest = pc._features_dc.grad.clone(), pc._features_rest.grad.clone()
    assert float(g_rest.abs().max()) > 0
    d = pc._last_xyz_t - cams[1].camera_center.reshape(1, 3)
    f = torch.cat([g_dc.reshape(-1, 3) / SH_C0, d / d.norm(dim=1, keepdim=True)], dim=1).reshape(1, -1, 6).contiguous()
    r_dc, r_rest = torch.empty_like(g_dc), torch.empty_like(g_rest)
    sh_factor_gradient(f, 3, r_dc, r_rest)
    torch.testing.assert_close(r_dc, g_dc, rtol=1e-5, atol=1e-7 * float(g_dc.abs().max()) + 1e-12)
    assert float((r_rest - g_rest).norm() / g_rest.norm()) < 1e-5
    assert float((r_rest - g_rest).abs().max()) < 1e-4 * float(g_rest.abs().max())


def _rank_main_factor(rank, world, port, out_dir, backend, factorised):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pc, cams, gts, raw, rw, idx, args = build(n=3000, dev=f"cuda:{dev}")
    ts = TrainStep(pc, cams, gts, 50000, sharded=False, factorised_sh=factorised)
    assert type(ts.reducer).__name__ == "OverlappedGradReducer" and (ts.reducer._factor is not None) == factorised
    for step in range(3):
        ts.step(step * world + rank)
    torch.cuda.synchronize()
    torch.save({"params": {n: p.detach().cpu() for n, p in pc.named_parameters()},
                "recv": getattr(ts.reducer, "factor_bytes_received_per_step", 0)}, os.path.join(out_dir, f"fx{int(factorised)}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_factorised_sh_exchange_equals_the_all_reduce(tmp_path):
    """Three view-parallel steps with the SH gradients exchanged as factors (one all-gather of 24 B per Gaussian) leave both ranks with
    bit-identical parameters, equal to the all-reduce path's (Adam with eps = 1e-15 moves an element whose gradient is rounding noise
    by a full lr: the same allowance as the sharded-vs-replicated test)."""
    import torch.multiprocessing as mp
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    for factorised in (False, True):
        mp.spawn(_rank_main_factor, args=(2, _free_port(), str(tmp_path), backend, factorised), nprocs=2, join=True)
    rep = [torch.load(os.path.join(tmp_path, f"fx0_{r}.pt")) for r in range(2)]
    fac = [torch.load(os.path.join(tmp_path, f"fx1_{r}.pt")) for r in range(2)]
    assert fac[0]["recv"] == 24 * 3000
    for name in rep[0]["params"]:
        assert torch.equal(fac[0]["params"][name], fac[1]["params"][name]), name          # replicas stay replicas
        diff = (rep[0]["params"][name] - fac[0]["params"][name]).abs()
        frac = float((diff > 1e-6).float().mean())
        print(f"[factorised] {name}: fraction moved apart {frac:.4f}, max {float(diff.max()):.3e}")
        assert frac < 5e-2 and float(diff.max()) < 0.05 * 3 + 1e-6, (name, frac, float(diff.max()))
