"""CPU, gloo, world_size 2: view-parallel training stays rank-consistent THROUGH densify / prune / opacity reset / the k-means
stage hook / keypoint growth, and equals the single-process `--batch 2` run [REF train.py:113-133, 164-192;
scene/gaussian_model.py:663-754].

What makes that true (gaussianprediction_amd/train_step.py, training.py):
  * radii = MAX, visibility = ANY over the ranks' views (all-reduce), as over the views of a batch;
  * `viewspace_points.grad` -- the tensor add_densification_stats reads -- is the LAST view's on every rank (the reference
    hands it the loop variable of the batch loop: train.py:167), broadcast from the last rank;
  * densify_and_split draws from a generator every rank seeds identically; the k-means keypoints are rank 0's;
  * after every surgery the ranks compare N, K and an exact parameter checksum (assert_ranks_agree).

The whole iteration runs on CPU tensors: the host side is the product's (TrainStep, ShardedExchange / OverlappedGradReducer,
FusedAdam's bookkeeping, training.py); render, loss, the Adam arithmetic and furthest-point sampling -- HIP kernels in the
product -- are the tests' restatements (tests/host_render.py, tests/host_checkers.py).  The GPU form of the same schedule, on the
kernels: tests/test_gpu_train_loop.py::test_two_rank_schedule_*."""
import os
import socket
from random import Random
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

W, H, N0, VIEWS, LAST = 24, 20, 48, 8, 40


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def margs():
    return SimpleNamespace(beta=0.1, d=4, w=256, feature_dim=32, jointly_iteration=2, second_stage_iteration=26, third_stage_iteration=1000,
                           nearest_num=4, norm_rotation=True, step_opacity=False, step_opacity_iteration=5000, opacity_type="implicit",
                           xyz_noise_iteration=0, max_points=6, adaptive_points_num=5, adaptive_from_iter=2, adaptive_end_iter=14,
                           adaptive_interval=4, densify_from_grad="True", densify_from_teaching=False, teaching_threshold=0.2,
                           max_gaussian_size=260)


def opt_args():
    from gaussianprediction_amd.training import default_training_args
    return default_training_args(iterations=LAST, densify_from_iter=4, densification_interval=5, opacity_reset_interval=12,
                                 densify_until_iter=24, densify_grad_threshold=1e-7, position_lr_max_steps=LAST)


def build_model():
    import gaussianprediction_amd as gpa
    from gaussianprediction_amd.cameras import orbit_cameras
    from gaussianprediction_amd.scene_synth import SceneSpec, make_gaussians
    a = margs()
    raw = make_gaussians(SceneSpec(n_gaussians=N0, extent=(0.8, 0.8, 0.8), scale_lo=0.01, scale_hi=0.12, seed=7))
    torch.manual_seed(3)                               # (the MLP's default initialisation)
    pc = gpa.GaussianModel(3, a)
    pc.set_inputDim(12, 60)
    pc.create_from_tensors(raw["xyz"], raw["features_dc"], raw["features_rest"], raw["scaling"], raw["rotation"], raw["opacity"],
                           raw["motion_feature"], torch.ones(a.max_points, 3), torch.ones(a.max_points, 32))
    pc.deterministic_surgery = True                    # the single-process run draws its split samples from the same seeded stream
    cams = orbit_cameras(VIEWS, 4.0, 0.5, W, H)
    g = torch.Generator().manual_seed(5)
    gts = [torch.rand(3, H, W, generator=g) * 0.6 + 0.2 for _ in cams]
    return pc, cams, gts, a


def run_schedule(pc, cams, gts, a, world, rank, batch, sharded=None, log=None):
    """Iterations 1..LAST of the compressed schedule through TrainStep + densify.py; rank r of `world` renders view
    step * world + r (world 1, batch 2: views 2 step, 2 step + 1 -- the same pairs)."""
    from gaussianprediction_amd import densify as dn
    from gaussianprediction_amd.train_step import TrainStep
    opt = opt_args()
    ts = TrainStep(pc, cams, gts, 1, lambda_dssim=opt.lambda_dssim, batch=batch, schedule=True, training_args=opt, sharded=sharded)
    rnd, pending = Random(0), []
    events = dict(densify=0, prune=0, reset=0, grown=0)
    for it in range(1, LAST + 1):
        ts.iteration = it
        if not pending:
            pending = list(range(VIEWS // 2))
        pair = pending.pop(rnd.randint(0, len(pending) - 1))
        loss, pkg = ts.step(pair * world + rank if world > 1 else pair, hold=dn.held_groups(pc, it, opt))
        with torch.no_grad():
            if it < opt.densify_until_iter:
                dn.track_view(pc, pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
                n_clone, n_src, n_pruned = dn.densification_step(pc, it, opt, 2.0, max_gaussian_size=a.max_gaussian_size)
                events["densify"] += n_clone is not None
                events["prune"] += n_pruned is not None
                events["reset"] += it % opt.opacity_reset_interval == 0
            events["grown"] += bool(dn.keypoint_growth_step(pc, it, opt, a, pkg["visibility_filter"], pkg["radii"], pkg["viewspace_points"]))
        if log is not None:
            log.append((it, float(loss), pc.get_xyz.shape[0], pc.super_gaussians.shape[0]))
    ts.sync_params()
    return events


def snapshot(pc, events, log):
    sd = pc.optimizer.state_dict()                     # (sharded: a collective)
    return {"params": {n: p.detach().clone() for n, p in pc.named_parameters()},
            "state": {k: {kk: vv.clone() for kk, vv in v.items()} for k, v in sd["state"].items()},
            "groups": [g["name"] for g in sd["param_groups"]], "lag": dict(pc.optimizer.lag), "step": pc.optimizer.step_count,
            "stats": {k: getattr(pc, k).clone() for k in ("xyz_gradient_accum", "denom", "max_radii2D")}, "events": events, "log": log}


def _worker(rank, world, port, out_dir, sharded, small_numel=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import host_checkers
    import host_render
    host_checkers.install()
    host_render.install()
    pc, cams, gts, a = build_model()
    if small_numel is not None:                        # the per-Gaussian tensors get regions of their own, as at 1 M Gaussians: their
        pc.bucket_small_numel = small_numel            # reduce-scatters leave from the gradient hooks, inside backward
    log = []
    events = run_schedule(pc, cams, gts, a, world, rank, 1, sharded=sharded, log=log)
    if small_numel is not None and sharded:
        assert len(pc.bucket.regions) >= 4, pc.bucket.regions
    assert pc._vp is not None and pc._vp[1:] == (rank, world)
    torch.save(snapshot(pc, events, log), os.path.join(out_dir, f"vp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded,small_numel", [(True, None), (True, 140), (False, None)])
def test_two_ranks_stay_identical_through_surgery_and_equal_the_batch_run(tmp_path, sharded, small_numel):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), sharded, small_numel), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"vp{k}.pt"), weights_only=False) for k in range(world)]
    ev = r[0]["events"]
    # the schedule really exercised the surgery: >= 3 densify + prune events, an opacity reset, the k-means hook, keypoint growth
    assert ev["densify"] >= 3 and ev["prune"] >= 3 and ev["reset"] >= 1 and ev["grown"] >= 1, ev
    n_hist = [e[2] for e in r[0]["log"]]
    assert max(n_hist) > N0 and len(set(n_hist)) >= 4, n_hist
    assert r[0]["log"][-1][3] > margs().max_points                        # keypoints were added
    # ---- the two ranks: bit-identical parameters, Adam moments, step counts, statistics, N and K at every iteration
    assert r[0]["events"] == r[1]["events"]
    assert [e[2:] for e in r[0]["log"]] == [e[2:] for e in r[1]["log"]]
    assert r[0]["params"].keys() == r[1]["params"].keys()
    for k in r[0]["params"]:
        assert torch.equal(r[0]["params"][k], r[1]["params"][k]), k
    assert r[0]["state"].keys() == r[1]["state"].keys() and r[0]["lag"] == r[1]["lag"] and r[0]["step"] == r[1]["step"]
    for k in r[0]["state"]:
        for kk in ("exp_avg", "exp_avg_sq", "step"):
            assert torch.equal(r[0]["state"][k][kk], r[1]["state"][k][kk]), (k, kk)
    for k in r[0]["stats"]:
        assert torch.equal(r[0]["stats"][k], r[1]["stats"][k]), k
    # ---- and the single-process --batch 2 run over the same view pairs [REF train.py:113-133]
    import host_render
    from gaussianprediction_amd import train_step
    saved = (train_step.render, train_step.l1_ssim_loss)
    host_render.install()
    try:
        pc, cams, gts, a = build_model()
        log = []
        events = run_schedule(pc, cams, gts, a, 1, 0, 2, log=log)
        one = snapshot(pc, events, log)
    finally:
        train_step.render, train_step.l1_ssim_loss = saved
    assert one["events"] == ev
    assert [e[2:] for e in one["log"]] == [e[2:] for e in r[0]["log"]]    # N and K after every iteration
    assert one["lag"] == r[0]["lag"] and one["step"] == r[0]["step"]
    # Parameters: the sum of two views' gradients is the same number however it is formed; what differs between the two runs is the
    # order in which shared leaves accumulate inside one backward (and the thread partition of the matrix products) -- a few ulp of
    # gradient.  Adam with eps = 1e-15 [REF scene/gaussian_model.py:472] turns an ulp of a near-zero gradient into a visible fraction of
    # one learning-rate step (and the size of that effect varies from run to run with the CPU's thread timing), so the bar is in units of
    # each group's learning rate: nothing farther than 2 lr (one step moves a parameter by ~lr; a row of a DIFFERENT Gaussian would be
    # off by the scene scale, 1e3 lr), and the typical element exact.
    lr_of = {}
    for g in pc.optimizer.param_groups:
        for p_ in g["params"]:
            lr_of[id(p_)] = max(float(g["lr"]), 1e-6)
    name_lr = {n: lr_of.get(id(p_), 1e-3) for n, p_ in pc.named_parameters()}
    for k in one["params"]:
        d = (r[0]["params"][k] - one["params"][k]).abs()
        assert float(d.max()) <= 2.0 * name_lr[k] + 2e-6, (k, float(d.max()), name_lr[k])
        assert float(d.median()) <= 1e-6, (k, float(d.median()))
    # per-iteration losses: each rank reports its own view's; their sum is the batch loss
    for e1, e0a, e0b in zip(one["log"], r[0]["log"], r[1]["log"]):
        assert abs(e1[1] - (e0a[1] + e0b[1])) < 1e-4 * max(1.0, abs(e1[1])), (e1, e0a, e0b)


def _worker_diverge(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import host_checkers
    host_checkers.install()
    pc, cams, gts, a = build_model()
    pc.set_view_parallel(None, rank, world)
    pc.training_setup(opt_args())
    pc.assert_ranks_agree("start")                     # identical replicas: passes
    msg = ""
    with torch.no_grad():
        if rank == 1:
            pc._xyz[3, 1] += 1e-7                      # one ulp-scale difference in one element on one rank
    try:
        pc.assert_ranks_agree("after a rank-local edit")
    except RuntimeError as e:
        msg = str(e)
    # the seeded split stream: the same numbers on both ranks, different ones for the next surgery
    g0 = torch.normal(torch.zeros(5), torch.ones(5), generator=pc._surgery_generator())
    pc._surgery_no += 1
    g1 = torch.normal(torch.zeros(5), torch.ones(5), generator=pc._surgery_generator())
    torch.save({"msg": msg, "g0": g0, "g1": g1}, os.path.join(out_dir, f"dv{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_divergent_replicas_are_detected_and_split_streams_agree(tmp_path):
    world = 2
    mp.spawn(_worker_diverge, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"dv{k}.pt")) for k in range(world)]
    for k in range(world):                             # raised on EVERY rank (the comparison is an all-reduce), naming the cause
        assert "diverged" in r[k]["msg"] and "after a rank-local edit" in r[k]["msg"], r[k]["msg"]
    assert torch.equal(r[0]["g0"], r[1]["g0"]) and torch.equal(r[0]["g1"], r[1]["g1"]) and not torch.equal(r[0]["g0"], r[0]["g1"])
