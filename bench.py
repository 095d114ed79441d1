#!/usr/bin/env python
"""bench.py -- headline benchmark of the dynamic-Gaussian render hot path on MI355X.

Workload (BASELINE.json metric "train-step ms & rendered views/s @1M Gaussians 1352x1014", i.e.
configs[2]: HyperNeRF-like, 1M Gaussians, 1352x1014, full train step on one GPU): one STEP = one pass
of the hot path over one camera view per GPU --
    GaussianModel.forward (stage 3: fused PE+MLP over K=250 keypoints, sparse blend over N, quaternion
    compose, activations) -> preprocess -> depth sort -> tile binning -> composite forward -> loss
    (0.8 L1 + 0.2 (1-SSIM)) -> composite backward -> preprocess backward -> deformation backward ->
    Adam step.
Synthetic seed-fixed scene + cameras (datasets are not available offline), random-init MLP weights.
N>1: one process per GPU, view-parallel, RCCL all-reduce(SUM) of the flat gradient bucket ("weak"
scaling: one view per GPU per step).  value = views/s of full train steps over the whole job.

Contract: python bench.py --gpus N --steps K --warmup W  ->  ONE JSON line on rank 0.

Launch forms.  Under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: WORLD_SIZE is set)
this process is one of the N ranks.  As plain `python bench.py --gpus N` with N > 1 (WORLD_SIZE unset) it STARTS the N ranks
itself -- it re-executes under `torch.distributed.run` on 127.0.0.1 with a free port and relays rank 0's line -- and it fails
loudly when the box shows fewer than N devices: a request for N GPUs never degrades to fewer ranks.  `--dry-launch` runs the
same launch with gloo and no device work (the launcher's CPU test).
"""
import argparse
import json
import math
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable (float4 copy: the guide and
                               # this package's own microbenchmark agree, profiles/r03_copy_peak_sweep.jsonl)
SHADER_CLOCK_GHZ = 2.4         # MI355X_MICROARCH.md; 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles per SIMD


def build_workload(args, device):
    import gaussianprediction_amd as gpa
    from gaussianprediction_amd.cameras import orbit_cameras
    from gaussianprediction_amd.scene_synth import SceneSpec, make_gaussians, make_keypoints
    margs = SimpleNamespace(beta=0.1, d=4, w=256, feature_dim=32, second_stage_iteration=30000, third_stage_iteration=40000,
                            jointly_iteration=1000, nearest_num=args.nearest_num, norm_rotation=True,
                            step_opacity=bool(getattr(args, "step_opacity", False)), step_opacity_iteration=5000,
                            opacity_type="implicit", xyz_noise_iteration=0)
    time_freq = args.time_freq
    extent = getattr(args, "extent", (1.5, 1.5, 0.5))
    raw = make_gaussians(SceneSpec(n_gaussians=args.gaussians, extent=extent, scale_lo=args.scale_lo,
                                   scale_hi=args.scale_hi, seed=2024), device=device)
    kp, kpf, idx, raw_w = make_keypoints(raw["xyz"], raw["motion_feature"], args.keypoints, margs.nearest_num)
    torch.manual_seed(2024)
    pc = gpa.GaussianModel(3, margs)
    pc.set_inputDim(2 * time_freq, 60)
    pc.create_from_tensors(raw["xyz"], raw["features_dc"], raw["features_rest"], raw["scaling"], raw["rotation"], raw["opacity"],
                           raw["motion_feature"], kp, kpf)
    pc.set_keypoint_weights(raw_w, idx)
    fovx = getattr(args, "fovx", None) or 2 * math.atan(1.0 / (2 * 0.9))           # focal ~ 0.9 W
    cams = orbit_cameras(8, 4.0, fovx, args.width, args.height, arc_deg=getattr(args, "arc_deg", 40.0),
                         elevation_deg=getattr(args, "elevation_deg", 5.0), device=device)
    # ground truth = the scene itself rendered at a slightly later time + pixel noise: a small, realistic
    # residual (a pure-noise target would make Adam fling the Gaussians out of view within a few steps and
    # the workload R would not be stationary over the timed region)
    from gaussianprediction_amd.renderer import render
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    g = torch.Generator().manual_seed(7)
    gts = []
    with torch.no_grad():
        for cam in cams:
            t = torch.from_numpy(cam.time).float().to(device) + 0.02
            img = render(cam, pc, pipe, torch.zeros(3, device=device), time=t, it=args.iteration)["render"]
            gts.append((img + 0.02 * torch.randn(img.shape, generator=g).to(device)).clamp(0, 1))
    return pc, cams, gts, margs


def cpu_baseline(args, pc, cams, gts, margs, seconds_budget=30.0):
    """The oracle (a CPU port of the same path) timed on this box's host cores, on a bounded sample: ONE view of the same step,
    leg by leg -- deformation forward + backward (oracle/deform_oracle.py, the reference's dense scatter + matmul form, torch on
    CPU), rasterizer forward + backward (C oracle, OpenMP: at 16 threads, at every core and -- budget permitting -- at one; the
    best of them counts, its loops stop scaling long before 256 threads), L1 + SSIM forward + backward (torch on CPU) and one Adam
    step over all parameters (torch.optim.Adam on CPU copies).  value = 1 / the sum of the legs."""
    from oracle import deform_oracle as do
    from oracle.oracle import RasterOracle, RasterSettings
    cam = cams[0]
    legs = {}
    dev = pc.get_xyz.device
    t_cpu = torch.from_numpy(cam.time).float()
    # ---- deformation, stage 3 [REF scene/gaussian_model.py:231-304]
    try:
        P = {"xyz": pc._xyz, "rotation": pc._rotation, "scaling": pc._scaling, "opacity": pc._opacity, "motion_feature": pc.motion_feature,
             "super_gaussians": pc.super_gaussians, "super_gaussians_feature": pc.super_gaussians_feature}
        P = {k: v.detach().float().cpu().requires_grad_(True) for k, v in P.items()}
        sd = {k: v.detach().float().cpu().requires_grad_(True) for k, v in pc.df_model.state_dict().items()}
        oa = SimpleNamespace(**vars(margs))
        oa.xyz_freq, oa.time_freq = 10, args.time_freq
        rw, idx = pc.raw_weights.detach().float().cpu(), pc.knn_idx.detach().cpu()
        t0 = time.perf_counter()
        x_, q_, s_, o_ = do.deform_forward(P, sd, t_cpu, args.iteration, oa, rw, idx)
        (x_.sum() + q_.sum() + s_.sum() + o_.sum()).backward()
        legs["deform_fwd_bwd"] = time.perf_counter() - t0
        del x_, q_, s_, o_
    except Exception as e:
        legs["deform_fwd_bwd"] = None
        legs["deform_error"] = str(e)[:200]
    # ---- rasterizer
    with torch.no_grad():
        xyz, q, s, o = pc(t_cpu.to(dev), args.iteration)
        shs = pc.get_features
    n64 = lambda x: x.detach().float().cpu().numpy().astype(np.float64)
    st = RasterSettings(image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
                        tanfovy=math.tan(cam.FoVy * 0.5), bg=np.zeros(3), scale_modifier=1.0,
                        viewmatrix=n64(cam.world_view_transform), projmatrix=n64(cam.full_proj_transform), sh_degree=3,
                        campos=n64(cam.camera_center))
    cores = os.cpu_count() or 1
    A = dict(means3D=n64(xyz), opacities=n64(o), shs=n64(shs), scales=n64(s), rotations=n64(q))
    g, image = None, None
    timings, spent = {}, 0.0
    for threads in sorted({min(16, cores), cores}) + [1]:
        if threads in timings:
            continue
        if threads == 1 and timings and spent + 14.0 * min(timings.values()) > seconds_budget:
            continue                               # a single thread would blow the budget (it scales ~linearly up to 16)
        orc = RasterOracle("f32", threads=threads)
        t0 = time.perf_counter()
        sres = orc.forward(st, A["means3D"], A["opacities"], shs=A["shs"], scales=A["scales"], rotations=A["rotations"])
        if g is None:
            g = np.random.default_rng(0).normal(size=sres["out_color"].shape)
            image = torch.tensor(np.asarray(sres["out_color"], dtype=np.float32))
        orc.backward(sres, g)
        timings[threads] = time.perf_counter() - t0
        spent += timings[threads]
    best = min(timings, key=timings.get)
    legs["raster_fwd_bwd"] = timings[best]
    # ---- loss [REF train.py:105-108, utils/loss_utils.py]
    img = image.clone().requires_grad_(True)
    gt = gts[0].detach().float().cpu()
    t0 = time.perf_counter()
    loss = 0.8 * do.l1_loss(img, gt) + 0.2 * (1.0 - do.ssim(img, gt))
    loss.backward()
    legs["l1_ssim_fwd_bwd"] = time.perf_counter() - t0
    # ---- Adam over every parameter [REF train.py:196-197]
    cp = [torch.nn.Parameter(p.detach().float().cpu().clone()) for p in pc.bucket.params]
    for p_ in cp:
        p_.grad = torch.full_like(p_, 1e-3)
    opt = torch.optim.Adam(cp, lr=1e-4, eps=1e-15)
    t0 = time.perf_counter()
    opt.step()
    legs["adam_step"] = time.perf_counter() - t0
    total = sum(v for k, v in legs.items() if isinstance(v, float))
    return {"value": 1.0 / total, "unit": "views/s", "cores": best, "kind": "port", "legs_s": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in legs.items()},
            "sample": "1 view of the same train step, leg by leg (legs_s): deformation fwd+bwd (torch CPU, the reference's dense form), "
                      "raster fwd+bwd (C oracle, OpenMP; seconds by thread count: "
                      + ", ".join(f"{k}: {v:.2f}" for k, v in sorted(timings.items()))
                      + f"; `cores` = the best), L1+SSIM fwd+bwd and one Adam step (torch CPU, its own thread pool); host has {cores} hardware threads"}


import contextlib


def no_gc():
    """Timed loops run behind `gc.freeze()`: the process's long-lived heap (torch, numpy, the scene) moves to the
    permanent generation, so a generation-2 pass that falls into a timed loop walks only what the loop itself allocated.  Without it
    such a pass is ~100 ms here -- one of them inside the 20-step loop of the host-bound graph path doubled that leg's reading (a
    step of 98 ms among steps of 4.8 ms: GP_BENCH_DEBUG=1), and inside the headline's region it would drain the launch queue.  The
    collector itself stays ON and nothing is collected in front of a loop (either way the 20-step dense variant read 1.10 instead of
    1.04 ms per step; GP_BENCH_GC=1 leaves everything alone)."""
    import gc

    @contextlib.contextmanager
    def cm():
        if os.environ.get("GP_BENCH_GC") == "1":      # (A/B: leave the collector alone)
            yield
            return
        gc.freeze()         # (no collect() in front: measured, it costs the loop behind it ~1 ms once -- 20-step dense variant 1.04 -> 1.10 ms per step)
        try:
            yield
        finally:
            gc.unfreeze()
    return cm()


@contextlib.contextmanager
def stdout_to_stderr():
    """Communicator start-up prints banners on the C library's stdout (RCCL's version line, gloo's "[Gloo] Rank ..."); stdout
    carries exactly ONE line here (the JSON result), so file descriptor 1 points at stderr while a process group comes up."""
    sys.stdout.flush()
    fd1 = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)          # the banner sits in the C library's stdio buffer: push it out while fd 1 is stderr
        except Exception:
            pass
        os.dup2(fd1, 1)
        os.close(fd1)


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N ranks of this script under torch.distributed.run on this
    node (one process per GPU, rendezvous on 127.0.0.1 at a free port) and exit with its return code.  Never fewer ranks than
    asked for."""
    import socket
    import subprocess
    if not args.dry_launch:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but this box shows {have} HIP device(s); refusing to run fewer ranks\n")
            sys.exit(2)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, GP_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # (dmabuf IPC: RCCL across processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stderr.write("[bench] self-launch: " + " ".join(cmd) + "\n")
    sys.exit(subprocess.call(cmd, env=env))


def dry_launch(args):
    """The launch path without device work: the ranks rendezvous over gloo, prove they see each other with one all-reduce, and
    rank 0 prints the one line (tests/test_bench_launch.py)."""
    os.environ["GP_DIST_BACKEND"] = "gloo"
    from gaussianprediction_amd.dist import init_from_env
    with stdout_to_stderr():
        rank, local, world = init_from_env("gloo")
        if world != args.gpus:
            raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
        seen = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(seen)
            dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry launch (no device work)", "value": None, "n_gpus": world, "dry_launch": True,
                          "config": {"ranks_seen_by_rccl": dist.get_world_size() if dist.is_initialized() else 1,
                                     "backend": "gloo", "rank_id_sum": float(seen.item()),
                                     "self_launched": os.environ.get("GP_BENCH_SELF_LAUNCHED") == "1"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def _key_depth(key):
    """view-space depth of a depth key (the key is the float's bit pattern)"""
    import struct
    return struct.unpack("<f", struct.pack("<I", int(key) & 0xFFFFFFFF))[0]


def dense_variant(args, device, steps=20):
    """The same scene with 1.5 x larger splats (scale 0.0045 .. 0.018: R / N at the UPPER end of SURVEY.md 8d's 4 .. 9 range, where
    the bytes per (pixel, splat) pair fall and the composite's HBM fraction with them): set-up + warm-up as the headline, `steps`
    timed steps, then the per-kernel pass.  Untimed as far as the headline is concerned (it runs after it)."""
    from gaussianprediction_amd import _lib
    from gaussianprediction_amd.rasterizer import raster_forward_debug
    from gaussianprediction_amd.renderer import _settings
    from gaussianprediction_amd.train_step import TrainStep
    a2 = argparse.Namespace(**vars(args))
    a2.scale_lo, a2.scale_hi = 1.5 * args.scale_lo, 1.5 * args.scale_hi
    pc, cams, gts, _ = build_workload(a2, device)
    ts = TrainStep(pc, cams, gts, a2.iteration, lrs=dict(xyz=1.6e-6 * 5.0), speculative=not args.exact_binning)
    pre = len(cams) + TrainStep.SPEC_SLOTS + 1 + 5
    for i in range(pre):
        ts.step(i)
    torch.cuda.synchronize()
    with no_gc():
        t0 = time.perf_counter()
        for i in range(steps):
            ts.step(pre + i)
        torch.cuda.synchronize()
        ms = 1000.0 * (time.perf_counter() - t0) / steps
    _lib.profile_enable(2)
    _lib.profile_collect()
    for i in range(5):
        ts.step(pre + steps + i)
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(0)
    with torch.no_grad():
        cam = cams[(pre + steps + 4) % len(cams)]
        xyz, q, s, o = pc(torch.from_numpy(cam.time).float().to(device), a2.iteration)
        dbg = raster_forward_debug(_settings(cam, pc, ts.bg, 1.0), xyz, o, shs=pc.get_features, scales=s, rotations=q)
    R, n_vis = int(dbg["R"]), int((dbg["radii"] > 0).sum())
    pairs = None
    try:        # (pixel, splat) pairs of the same view, as the headline's roofline object counts them: the fraction below is only readable beside them
        import ctypes as C
        L = _lib.lib()
        _lib.check(L.gp_debug_option(0, 3), "opt")
        cnt = (C.c_uint64 * 4)()
        _lib.check(L.gp_debug_counters(cnt), "counters")
        with torch.no_grad():
            raster_forward_debug(_settings(cam, pc, ts.bg, 1.0), xyz, o, shs=pc.get_features, scales=s, rotations=q)
        _lib.check(L.gp_debug_counters(cnt), "counters")
        pairs = (int(cnt[0]), int(cnt[1]))
    except Exception:
        pairs = None
    finally:
        try:
            _lib.check(_lib.lib().gp_debug_option(0, 0), "opt")
        except Exception:
            pass
    W, H = args.width, args.height
    T, P = ((W + 15) // 16) * ((H + 15) // 16), W * H
    k = {name: prof[name][1] / 5 for name in ("composite_fwd", "composite_bwd") if name in prof}
    out = {"scale_lo": a2.scale_lo, "scale_hi": a2.scale_hi, "R": R, "R_per_gaussian": round(R / max(args.gaussians, 1), 3), "visible": n_vis,
           "steps": steps, "ms_per_step": round(ms, 3), "frames_repeated_after_overflow": getattr(ts, "redone", 0)}
    if pairs is not None:
        # the denser scene's pixels saturate earlier: tiles stop reading their lists, so 44 R over-counts the bytes the kernel needs --
        # contributing pairs per list entry say by how much (headline: see roofline.contributing_pairs / R)
        out["contributing_pairs"], out["evaluated_pairs"] = pairs
        out["contributing_pairs_per_list_entry"] = round(pairs[0] / max(R, 1), 2)
    if "composite_fwd" in k:
        out["composite_fwd_ms"] = round(k["composite_fwd"], 4)
        out["composite_fwd_frac"] = round((44 * R + 8 * T + 28 * P) / (k["composite_fwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if pairs is not None:
            out["composite_fwd_ns_per_1e3_contributing_pairs"] = round(k["composite_fwd"] * 1e6 / max(pairs[0], 1) * 1e3, 3)
    if "composite_bwd" in k:
        out["composite_bwd_ms"] = round(k["composite_bwd"], 4)
        out["composite_bwd_frac"] = round((44 * R + 8 * T + 24 * P + 40 * n_vis) / (k["composite_bwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    del ts, pc
    torch.cuda.empty_cache()
    return out


def kernel_source_stamp():
    """Identifies the composite kernels' source: a traffic figure from a PMC run is only quoted for the kernel it measured."""
    return _sha16(os.path.join(ROOT, "gaussianprediction_amd", "csrc", "raster_kernels.hip"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1352)
    ap.add_argument("--height", type=int, default=1014)
    ap.add_argument("--keypoints", type=int, default=250)
    ap.add_argument("--nearest_num", type=int, default=6)
    ap.add_argument("--time_freq", type=int, default=8)
    ap.add_argument("--iteration", type=int, default=50000)
    ap.add_argument("--scale_lo", type=float, default=0.003)
    ap.add_argument("--scale_hi", type=float, default=0.012)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-variant", action="store_true", help="skip the 20 extra steps of the scene with 1.5 x larger splats (dense_variant)")
    ap.add_argument("--no-weights-model-step", action="store_true", help="skip the secondary measurement of the step that also "
                    "runs the per-frame hash-grid weights model + kNN")
    ap.add_argument("--exact-binning", action="store_true",
                    help="size the binning buffers by reading R back every step (one host sync per step) instead of the "
                         "capacity mode with the high-water-mark protocol of TrainStep(speculative=True)")
    ap.add_argument("--no-fused", action="store_true",
                    help="take every step through the autograd graph (render -> loss -> backward -> FusedAdam.step) instead of the one-call "
                         "fused step (gp_train_step_run): the A/B of TrainStep(fused=...)")
    ap.add_argument("--no-early-adam", action="store_true",
                    help="fused step: ONE optimizer launch behind the backward instead of the per-Gaussian tensors' Adam update riding in the "
                         "launch of the keypoint MLP's data backward (gp_step_update.adam_early_mask; A/B: profiles/r06_adam_rider_ab.txt).  The "
                         "rider travels with the 16-row MLP kernels only (more than 512 keypoints, or --debug-option 13=1)")
    ap.add_argument("--debug-option", action="append", default=[], metavar="KEY=VALUE",
                    help="gp_debug_option(KEY, VALUE) before the run (the library's A/B knobs, e.g. 8=2: the three-pass 11-bit depth sort); repeatable")
    ap.add_argument("--render-only", action="store_true", help="time eval-style forward renders instead of train steps")
    ap.add_argument("--roofline-sample-period", type=int, default=8,
                    help="the roofline kernel is bracketed by hipEvents at every P-th step of the timed region (a bracket is a ~10 us bubble "
                         "on the stream; 1 = every step)")
    ap.add_argument("--no-chain-sh", action="store_true",
                    help="N > 1, sharded exchange: keep the SH regions' Adam + all-gather on the compute stream (A/B of TrainStep._chain_sh)")
    ap.add_argument("--replicated-adam", action="store_true",
                    help="N > 1: all-reduce + replicated Adam (round 1) instead of reduce-scatter -> sharded Adam -> all-gather")
    ap.add_argument("--factorised-sh", action="store_true",
                    help="N > 1: replicated Adam with the SH gradients exchanged as (dL/dRGB, view direction) factors -- one all-gather of 24 B per "
                         "Gaussian and rank instead of the 192-B gradient's all-reduce (implies --replicated-adam; DESIGN.md section 6)")
    ap.add_argument("--time-waits", action="store_true",
                    help="N > 1, sharded exchange: bracket every wait of the compute stream for a collective with events and report "
                         "config.exposed_wait_ms_per_step (adds a few stream bubbles: an A/B aid, not the headline run)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch path only: N ranks rendezvous over gloo, one all-reduce, one JSON line, no device work")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])             # does not return
    if args.dry_launch:
        return dry_launch(args)

    from gaussianprediction_amd import _lib
    from gaussianprediction_amd.dist import init_from_env
    from gaussianprediction_amd.train_step import TrainStep
    with stdout_to_stderr():
        rank, local, world = init_from_env()
        if world != args.gpus:                      # (a launcher that started another number of ranks than --gpus names)
            raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
        assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
        if dist.is_available() and dist.is_initialized():
            dist.barrier()                          # (creates the communicator now)
            torch.cuda.synchronize()
    for kv in args.debug_option:
        k, v = kv.split("=")
        _lib.check(_lib.lib().gp_debug_option(int(k), int(v)), "gp_debug_option")
    pc, cams, gts, margs = build_workload(args, device)
    # learning rates of the reference at iteration 50000 (position lr has decayed to position_lr_final,
    # [REF arguments/__init__.py:75-76, scene/gaussian_model.py:474-491])
    ts = TrainStep(pc, cams, gts, args.iteration, lrs=dict(xyz=1.6e-6 * 5.0), speculative=not args.exact_binning,
                   sharded=False if (args.replicated_adam or args.factorised_sh) else None, chain_sh=not args.no_chain_sh, fused=not args.no_fused,
                   factorised_sh=args.factorised_sh)
    ts.early_adam = not args.no_early_adam

    def one_step(i):
        view = i * world + rank            # rank r renders view world*i + r
        if args.render_only:
            with torch.no_grad():
                cam = cams[view % len(cams)]
                t = torch.from_numpy(cam.time).float().to(device)
                from gaussianprediction_amd.renderer import render
                return render(cam, pc, ts.pipe, ts.bg, time=t, it=args.iteration)
        return ts.step(view)

    # set-up, not warm-up: TrainStep sizes its binning buffers from the largest R it has seen, which it learns by running
    # every camera once in exact mode (one host read of R per step).  Those steps happen here, so that BOTH the warm-up and the
    # timed steps below run in the mode `config.binning` names, whatever --warmup is.
    preroll = 0
    if not args.render_only and not args.exact_binning:
        preroll = len(cams) + TrainStep.SPEC_SLOTS + 1
        for i in range(preroll):
            ts.step(i * world + rank)
    for i in range(args.warmup):
        one_step(preroll + i)
    torch.cuda.synchronize()
    if rank == 0:
        from gaussianprediction_amd.rasterizer import raster_forward_debug as _rfd
        from gaussianprediction_amd.renderer import _settings as _st
        with torch.no_grad():
            cam0 = cams[((preroll + args.warmup) * world) % len(cams)]
            x0, q0, s0, o0 = pc(torch.from_numpy(cam0.time).float().to(device), args.iteration)
            main._R0 = _rfd(_st(cam0, pc, ts.bg, 1.0), x0, o0, shs=pc.get_features, scales=s0, rotations=q0)["R"]
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # timed region: only the roofline kernel (composite forward) is bracketed by hipEvents -- every
    # bracket costs ~10 us of stream bubble, so the full per-kernel table is taken in a separate pass
    _lib.check(_lib.lib().gp_debug_option(12, int(args.roofline_sample_period)), "gp_debug_option")
    _lib.profile_enable(1)
    _lib.profile_collect()
    if args.time_waits and getattr(ts.reducer, "exposed_wait_ms", None):
        ts.reducer.time_waits = True
        ts.reducer.exposed_wait_ms(1)                  # (drop what the warm-up recorded)
    torch.cuda.synchronize()
    with no_gc():
        t0 = time.perf_counter()
        pkg = None
        for i in range(args.steps):
            out = one_step(preroll + args.warmup + i)
            pkg = out if args.render_only else out[1]
        ts.sync_params()                            # (N > 1: the last step's parameter all-gather belongs to the timed work)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    waits = ts.reducer.exposed_wait_ms(args.steps) if (args.time_waits and getattr(ts.reducer, "exposed_wait_ms", None)) else None
    if waits is not None:
        ts.reducer.time_waits = False
    prof = _lib.profile_collect()
    _lib.profile_enable(0)
    _lib.check(_lib.lib().gp_debug_option(12, 0), "gp_debug_option")
    # the headline's depth sort runs three passes under a promise about the key range, which holds for scenes whose visible depths span less
    # than a factor of four (config.depth_sort): the same step WITHOUT the promise (four passes, what a deeper scene pays), 10 untimed-for-
    # the-headline steps, so that the scene-dependent part of the number is visible in the record
    no_promise_ms, four_pass_sort_ms = None, None
    if world == 1 and not args.render_only and getattr(ts, "speculative", False) and getattr(ts, "last_depth_key_promise", None):
        try:
            ts.depth_key_speculation = False
            for i in range(3):
                one_step(preroll + args.warmup + args.steps + i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(10):
                one_step(preroll + args.warmup + args.steps + 3 + i)
            torch.cuda.synchronize()
            no_promise_ms = 1000.0 * (time.perf_counter() - t1) / 10
            _lib.profile_enable(2); _lib.profile_collect()
            for i in range(5):
                one_step(preroll + args.warmup + args.steps + 13 + i)
            torch.cuda.synchronize()
            pp = _lib.profile_collect(); _lib.profile_enable(0)
            if "depth_sort" in pp:
                four_pass_sort_ms = pp["depth_sort"][1] / 5
        finally:
            ts.depth_key_speculation = True
            for i in range(2):
                one_step(preroll + args.warmup + args.steps + 18 + i)       # (the promise is back for whatever follows)
            torch.cuda.synchronize()
    # eval-style forward renders (the "rendered views/s" half of the metric, [REF eval.py:208-224]); separate
    # timed loop, reported as an extra field
    eval_fps, eval_ms = None, None
    if not args.render_only:
        from gaussianprediction_amd.renderer import render as _render
        n_eval = max(10, min(args.steps, 50))
        with torch.no_grad(), no_gc():
            ev = lambda i: _render(cams[(i * world + rank) % len(cams)], pc, ts.pipe, ts.bg,              # noqa: E731
                                   time=ts.times[(i * world + rank) % len(cams)], it=args.iteration)
            for i in range(3):
                ev(i)
            torch.cuda.synchronize()
            # (a) the reference's own loop [REF eval.py:208-215]: synchronize, t0, render, synchronize, t1 -- per-view latency
            lat = []
            for i in range(n_eval):
                torch.cuda.synchronize()
                t_a = time.perf_counter()
                ev(i)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t_a)
            eval_ms = 1000.0 * float(np.mean(lat))
            # (b) back-to-back renders, one synchronisation at the end -- pipelined throughput.  render() in its exact mode still reads
            # the instance count back in the middle of every frame (as the reference's rasterizer does); SpeculativeRenderer is the
            # harness-side form without it: capacity mode from the first frame's count, overflow words checked once at the end
            # (both back-to-back loops: the median of three repetitions -- a single 50-frame loop is ~20 ms, one host hiccup is half of that)
            reps = []
            for _ in range(3):
                te = time.perf_counter()
                for i in range(n_eval):
                    ev(i)
                torch.cuda.synchronize()
                reps.append(n_eval / (time.perf_counter() - te))
            eval_fps_exact = sorted(reps)[1]
            from gaussianprediction_amd.renderer import SpeculativeRenderer
            sr = SpeculativeRenderer(pc, ts.pipe, ts.bg)
            evs = lambda i: sr(cams[(i * world + rank) % len(cams)], time=ts.times[(i * world + rank) % len(cams)], it=args.iteration)   # noqa: E731
            for i in range(len(cams)):              # (the exact first frame + one capacity-mode frame per view: the high-water mark settles)
                evs(i)
            sr.flush()
            torch.cuda.synchronize()
            reps, eval_again = [], 0
            for _ in range(3):
                te = time.perf_counter()
                for i in range(n_eval):
                    evs(i)
                eval_again += sr.flush()
                torch.cuda.synchronize()
                reps.append(n_eval / (time.perf_counter() - te))
            eval_fps = sorted(reps)[1]
    _lib.profile_enable(2)                      # untimed pass for the per-kernel table
    for i in range(min(args.steps, 5)):
        one_step(preroll + args.warmup + args.steps + i)
    torch.cuda.synchronize()
    prof_all = _lib.profile_collect()
    _lib.profile_enable(0)
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / args.steps
        views_per_s = world * args.steps / elapsed
        # algorithmic bytes of the composite forward pass: 44 R + 8 T + 28 P
        W, H = args.width, args.height
        T = ((W + 15) // 16) * ((H + 15) // 16)
        P = W * H
        # R of the last rendered view (re-render to read it)
        from gaussianprediction_amd.rasterizer import raster_forward_debug
        from gaussianprediction_amd.renderer import _settings
        with torch.no_grad():
            cam = cams[((preroll + args.warmup + args.steps - 1) * world) % len(cams)]
            t = torch.from_numpy(cam.time).float().to(device)
            xyz, q, s, o = pc(t, args.iteration)
            dbg = raster_forward_debug(_settings(cam, pc, ts.bg, 1.0), xyz, o, shs=pc.get_features, scales=s, rotations=q)
        R, n_vis = dbg["R"], int((dbg["radii"] > 0).sum())
        R0 = getattr(main, "_R0", None)
        nprof = min(args.steps, 5)
        kern = {k: {"launches_per_step": round(v[0] / nprof, 2), "ms_per_step": round(v[1] / nprof, 4)} for k, v in sorted(prof_all.items())}
        roof = None
        if "composite_fwd" in prof:
            avg_ms = prof["composite_fwd"][1] / prof["composite_fwd"][0]
            bytes_alg = 44 * R + 8 * T + 28 * P
            ach = bytes_alg / (avg_ms * 1e-3) / 1e9
            roof = {"kernel": "composite_fwd", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes": bytes_alg, "avg_ms": round(avg_ms, 4),
                    # hipEvent pairs around the kernel's launches INSIDE the timed region: every `event_sample_period`-th step
                    "event_samples": int(prof["composite_fwd"][0]), "event_sample_period": int(args.roofline_sample_period),
                    # `bound` is the contract's roofline (SURVEY 8d prices the kernel in bytes); what the kernel WAITS for is the vector ALU
                    # (valu_issue_frac / valu_busy below): the fraction of the HBM roof is therefore capped by instructions, not by traffic
                    "binding_resource": "valu"}
            # HBM bytes per launch from the PMC passes (tools/pmc_hbm.py): quoted only while the file describes THIS kernel
            # source (it is stamped with the source's hash); otherwise null -- a stale constant is not a measurement
            tf = os.path.join(ROOT, "profiles", "composite_fwd_traffic.json")
            roof["traffic_source"] = None
            roof["traffic_measured_in_this_run"] = False          # (`traffic` is a stored PMC result of the same kernel source)
            try:
                # contributing (pixel, splat) pairs, counted by the kernel's counting variant on one untimed render of the last view,
                # and the instruction floor they imply: every contributing pair costs at least the straight-line visit of the
                # kernel (19 vector-ALU instructions per 64 pairs: the visit of a fast block, raster_kernels.hip CF2_VISIT), on 1024 SIMDs issuing
                # one wave64 instruction per 4 cycles
                import ctypes as C
                L = _lib.lib()
                _lib.check(L.gp_debug_option(0, 3), "opt")
                cnt = (C.c_uint64 * 4)()
                _lib.check(L.gp_debug_counters(cnt), "counters")          # (clears what earlier renders left)
                with torch.no_grad():
                    raster_forward_debug(_settings(cam, pc, ts.bg, 1.0), xyz, o, shs=pc.get_features, scales=s, rotations=q)
                _lib.check(L.gp_debug_counters(cnt), "counters")
                _lib.check(L.gp_debug_option(0, 0), "opt")
                MIN_VALU = 19
                roof["contributing_pairs"], roof["evaluated_pairs"] = int(cnt[0]), int(cnt[1])
                roof["min_valu_per_pair"] = MIN_VALU
                roof["valu_lower_bound_ms"] = round(int(cnt[0]) * MIN_VALU / 64.0 * 4.0 / 1024.0 / (SHADER_CLOCK_GHZ * 1e9) * 1e3, 4)
            except Exception as e:
                roof["contributing_pairs"] = f"failed: {e}"
                try:
                    _lib.check(_lib.lib().gp_debug_option(0, 0), "opt")
                except Exception:
                    pass
            if os.path.exists(tf):
                try:
                    tj = json.load(open(tf))
                    if tj.get("kernel_source_sha16") == kernel_source_stamp():
                        roof["traffic"] = tj.get("hbm_bytes_per_launch")
                        # (the same counters read as if every request were a 128-byte one: what the traffic could be at most if the
                        # gather calibration of the PMC file's header did not hold)
                        roof["traffic_upper_bound"] = tj.get("hbm_bytes_per_launch_upper_bound")
                        roof["traffic_source"] = tj.get("source")
                        vj = tj.get("valu")
                        if vj:
                            # the kernel is declared "hbm"-bound by the contract's formula, but what it waits for is the vector
                            # ALU: the instruction floor (every SIMD issuing one wave64 VALU instruction per 4 cycles) and how busy
                            # the VALUs were, from the SQ counters of the same kernel source
                            cycles = avg_ms * 1e-3 * SHADER_CLOCK_GHZ * 1e9
                            roof["valu_floor_ms"] = round(4.0 * vj["SQ_INSTS_VALU"] / 1024.0 / (SHADER_CLOCK_GHZ * 1e9) * 1e3, 4)
                            roof["valu_issue_frac"] = round(roof["valu_floor_ms"] / avg_ms, 4)     # the roof the kernel actually runs under
                            roof["valu_busy"] = round(min(1.0, 4.0 * vj["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles)), 3)
                            roof["valu_insts_per_launch"] = int(vj["SQ_INSTS_VALU"])
                            roof["valu_source"] = vj.get("source")
                    else:
                        roof["traffic_source"] = "profiles/composite_fwd_traffic.json describes another kernel source: not quoted"
                except Exception:
                    pass
        # the other HBM-bound kernels of the step against the same roofline (algorithmic bytes: SURVEY.md 8d / DESIGN.md 4;
        # durations from the untimed per-kernel pass), reported beside the contract's `roofline` object
        others = {}
        N = args.gaussians
        # Adam: 28 B per parameter the launch ACTUALLY updates (p, g, m, v read; p, m, v written).  In single-view single-rank
        # steps the two SH tensors (48 of the 59 floats per Gaussian) are updated inside the rasterizer backward instead
        # (gp_adam_fuse) and their bytes belong to preprocess_bwd's line.
        sh_fused = bool(getattr(ts, "fuse_sh_adam", False) and not ts.reducer.enabled and ts.batch == 1)
        sh_ids = {id(pc._features_dc), id(pc._features_rest)}
        n_adam = sum(int(p.numel()) for p in ts.bucket.params if not (sh_fused and id(p) in sh_ids))
        n_sh = sum(int(p.numel()) for p in ts.bucket.params if id(p) in sh_ids)
        # preprocess backward per Gaussian: reads means 12 + scales 12 + rotations 16 + SH 192 + radii 4 + clamp flags 1 + the
        # 40 useful bytes of its accumulator line; writes d(means3D 12, means2D 12, opacity 4, scales 12, rotations 16) and
        # dSH 192 -- or, with the SH Adam inside, no dSH but m, v read and p, m, v written for the 48 SH floats (5 x 192)
        pb_plain = (277 + 56 + 192) * N
        pb_fused = (277 + 56) * N + 20 * n_sh
        alg = {"composite_bwd": 44 * R + 8 * T + 24 * P + 40 * n_vis, "preprocess_fwd": 312 * N, "adam": 28 * n_adam,
               "preprocess_bwd": pb_fused if sh_fused else pb_plain}
        for k, nbytes in alg.items():
            if k in kern and kern[k]["ms_per_step"] > 0:
                ms_k = kern[k]["ms_per_step"]
                if k == "preprocess_fwd" and "sh_color" in kern:       # (view-parallel: SH -> RGB runs as a kernel of its own)
                    ms_k += kern["sh_color"]["ms_per_step"]
                gbs = nbytes / (ms_k * 1e-3) / 1e9
                others[k] = {"algorithmic_bytes": nbytes, "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
        if "adam" in others:
            others["adam"]["parameters_updated"] = n_adam
        if "preprocess_bwd" in others:
            others["preprocess_bwd"]["includes_fused_sh_adam"] = sh_fused
            others["preprocess_bwd"]["algorithmic_bytes_without_the_sh_adam"] = pb_plain
        result = {
            "metric": f"rendered views/s ({'eval render' if args.render_only else 'full train step'}) "
                      f"@{args.gaussians / 1e6:g}M Gaussians {args.width}x{args.height}",
            "value": round(views_per_s, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: HyperNeRF-like 1M Gaussians, 1352x1014, stage-3 full train step "
                                   "(deform, raster, L1+SSIM loss, Adam: all HIP kernels)"
                       if not args.render_only else "configs[2] eval render (forward only)",
                       "gaussians": args.gaussians, "width": W, "height": H, "keypoints": args.keypoints,
                       "nearest_num": args.nearest_num, "time_freq": args.time_freq, "iteration": args.iteration,
                       "tiles": T, "pixels": P, "R": R, "R_before_timed_region": R0, "R_per_gaussian": round(R / max(args.gaussians, 1), 3),
                       "visible": n_vis, "parallelism": f"view-parallel x{world}",
                       "ranks_seen_by_rccl": dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1,
                       "self_launched": os.environ.get("GP_BENCH_SELF_LAUNCHED") == "1",
                       "gradient_exchange": None if not ts.reducer.enabled else (
                           ("all-reduce(SUM) of the flat gradient bucket + replicated Adam" if not getattr(ts, "factorised_sh", False) else
                            "replicated Adam; SH gradients as (dL/dRGB, view direction) factors: one all-gather of 24 B per Gaussian and rank, summed "
                            "over the views in rank order on every rank (gp_sh_factor_gradient); all-reduce(SUM) of everything else") if not ts.sharded else
                           "reduce-scatter(SUM) per region -> Adam on 1/N of every region -> asynchronous all-gather of the parameters"
                           + ("; SH regions: Adam + all-gather on a side stream from the moment their reduce-scatter lands, awaited by the "
                              "next forward in front of its SH->RGB kernel only" if getattr(ts, "chain_sh", False) else "")
                           + (" [ONE-rank group, GP_DIST_FORCE_SINGLE: the code path, not a scaling number]" if world == 1 else "")),
                       "xgmi_bytes_sent_per_rank_per_step": None if world == 1 else (
                           getattr(ts.reducer, "bytes_sent_per_step", None) or
                           (int(2 * 4 * ts.bucket.flat.numel() * (world - 1) / world) if not getattr(ts, "factorised_sh", False) else
                            # the all-reduce of everything but the SH tensors + the factor all-gather (every rank sends its 24 B per Gaussian to world - 1 peers)
                            int(2 * 4 * (ts.bucket.flat.numel() - pc._features_dc.numel() - pc._features_rest.numel()) * (world - 1) / world)
                            + 24 * int(pc._features_dc.shape[0]) * (world - 1))),
                       "exposed_wait_ms_per_step": waits,          # (--time-waits: compute-stream time inside waits for collectives, rank 0)
                       "step_driver": (f"one library call per step (gp_train_step_run): {getattr(ts, 'fused_steps', 0)} of the steps since the set-up"
                                       + ("; the per-Gaussian tensors' Adam update rides in the launch of the keypoint MLP's data backward "
                                          "(kernels_ms.adam = that launch + the MLP tensors' update; no mlp_bwd_data entry)"
                                          if (getattr(getattr(getattr(ts, "_fused_plan", None), "upd", None), "adam_early_mask", 0) and world == 1) else "")
                                       if getattr(ts, "fused_steps", 0) else "autograd graph (render -> loss -> backward -> FusedAdam.step)"),
                       "binning": "exact (R read back every step)" if args.exact_binning else
                                  f"capacity mode in warm-up and timed steps (no host sync; {preroll} exact-mode set-up steps before the "
                                  f"warm-up; {getattr(ts, 'redone', 0)} frames repeated after overflow)",
                       "depth_sort": "32-bit keys, four 8-bit passes" if not getattr(ts, "last_depth_key_promise", None) else
                                     (f"{ts.last_depth_key_promise[0]}-bit keys above a promised base ({-(-ts.last_depth_key_promise[0] // 8)} passes; the "
                                      "projection kernel checks the promise, a broken one repeats the frame like a binning overflow; visible depths "
                                      f"seen in the set-up steps: {_key_depth(ts._key_lo):.3f} .. {_key_depth(ts._key_hi):.3f})"),
                       "keypoint_weights": "raw_weights / knn_idx are inputs of the step (BASELINE.json north_star); the reference "
                                           "recomputes them per frame (hash-grid weights model + kNN): see train_step_with_weights_model_ms"},
            "ms_per_step_without_depth_promise": None if no_promise_ms is None else round(no_promise_ms, 4),
            "depth_sort_four_pass_ms": None if four_pass_sort_ms is None else round(four_pass_sort_ms, 4),
            "roofline": roof,
            "roofline_other_kernels": others,
            "eval_render_ms_per_view_synced": None if eval_ms is None else round(eval_ms, 3),      # eval.py's own timing loop
            "eval_render_views_per_s_synced": None if eval_ms is None else round(1000.0 / eval_ms, 2),
            "eval_render_views_per_s_pipelined_per_gpu": None if eval_fps is None else round(eval_fps, 2),   # SpeculativeRenderer: no host sync per frame
            "eval_render_views_per_s_back_to_back_exact_mode": None if eval_fps is None else round(eval_fps_exact, 2),   # render() as is (reads R back every frame)
            "eval_render_frames_rendered_again_after_overflow": None if eval_fps is None else int(eval_again),
            "kernels_ms": kern,
        }
        if roof is not None and world == 1:
            try:                                 # measured device peaks beside the vendor number; untimed
                from gaussianprediction_amd import peaks
                mp = peaks.measure(device, gib=1.0, reps=5, mfma_iters=2048)
                roof["peak_measured"] = {"stream_copy": mp["copy_GBps"], "stream_read": mp["read_GBps"], "unit": "GB/s"}
                roof["frac_of_measured_copy"] = round(roof["achieved"] / mp["copy_GBps"], 4)
                result["device_peaks_measured"] = mp
            except Exception as e:
                roof["peak_measured"] = f"failed: {e}"
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(args, pc, cams, gts, margs)
            except Exception as e:  # the baseline must never kill the bench line
                result["cpu_baseline"] = {"value": None, "unit": "views/s", "cores": os.cpu_count(), "kind": "port",
                                          "sample": f"failed: {e}"}
        if world == 1 and not args.render_only and not args.no_dense_variant:
            try:
                result["dense_variant"] = dense_variant(args, device)
            except Exception as e:
                result["dense_variant"] = f"failed: {e}"
        if world == 1 and not args.render_only and not args.no_weights_model_step:
            # the step the REFERENCE runs in stage 3 also evaluates the hash-grid weights model and the kNN every frame (and
            # optimizes the former) [REF scene/gaussian_model.py:257-260,402]; north_star takes their outputs as inputs, so the
            # headline does not contain them -- this is the same step with them inside, on this package's own kernels
            try:
                from gaussianprediction_amd.weights_ops import WeightsModel
                margs.knn_type, margs.feature_amplify = "hybird", 5.0
                pc.weights_model = WeightsModel(2 * args.nearest_num, device=device)
                pc.set_keypoint_weights(None, None)
                pc.optimizer = None
                ts2 = TrainStep(pc, cams, gts, args.iteration, lrs=dict(xyz=1.6e-6 * 5.0))
                for i in range(5):
                    ts2.step(i)
                torch.cuda.synchronize()
                host = []
                with no_gc():
                    t1 = time.perf_counter()
                    for i in range(20):
                        th = time.perf_counter()
                        ts2.step(5 + i)
                        host.append(time.perf_counter() - th)
                    torch.cuda.synchronize()
                    result["train_step_with_weights_model_ms"] = round(1000.0 * (time.perf_counter() - t1) / 20, 3)
                if os.environ.get("GP_BENCH_DEBUG"):
                    ms = torch.cuda.memory_stats()
                    print("[weights-model leg] host ms per step:", [round(1e3 * h, 2) for h in host], "reserved GB",
                          round(torch.cuda.memory_reserved() / 2**30, 2), "alloc retries", ms.get("num_alloc_retries"),
                          "device mallocs", ms.get("num_device_alloc"), "device frees", ms.get("num_device_free"), file=sys.stderr, flush=True)
            except Exception as e:
                result["train_step_with_weights_model_ms"] = f"failed: {e}"
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
