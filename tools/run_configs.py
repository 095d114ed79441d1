#!/usr/bin/env python
"""Runs the five BASELINE.json configs (single-GPU forms) and prints one JSON line each -> BASELINE.md section 5.
    python tools/run_configs.py [c1 c2 c3 c4 c5a c5b]"""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np
import torch
import bench
from gaussianprediction_amd import _lib
from gaussianprediction_amd.train_step import TrainStep
from gaussianprediction_amd.renderer import render

DN = dict(extent=(1.3, 1.3, 1.3), fovx=0.6911, arc_deg=360.0, elevation_deg=20.0)   # D-NeRF-like
CONFIGS = {
    "c1": dict(gaussians=10_000, width=400, height=400, keypoints=100, nearest_num=6, time_freq=6, iteration=0, scale_lo=0.01, scale_hi=0.06, mode="render", **DN),
    "c2": dict(gaussians=200_000, width=800, height=800, keypoints=100, nearest_num=6, time_freq=6, iteration=20000, scale_lo=0.005, scale_hi=0.02, mode="fwdbwd", **DN),
    "c3": dict(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000, scale_lo=0.003, scale_hi=0.012, mode="train"),
    "c4": dict(gaussians=1_000_000, width=1352, height=1014, keypoints=300, nearest_num=6, time_freq=10, iteration=50000, scale_lo=0.003, scale_hi=0.012, mode="train", step_opacity=True),
    "c5a": dict(gaussians=2_000_000, width=800, height=800, keypoints=512, nearest_num=8, time_freq=6, iteration=20000, scale_lo=0.003, scale_hi=0.010, mode="train", **DN),
    "c5b": dict(gaussians=2_000_000, width=800, height=800, keypoints=512, nearest_num=8, time_freq=6, iteration=50000, scale_lo=0.003, scale_hi=0.010, mode="train", **DN),
}


def run(name, steps=15, warmup=5, precision="fp32s"):
    cfg = dict(CONFIGS[name])
    mode = cfg.pop("mode")
    args = SimpleNamespace(**cfg)
    dev = torch.device("cuda", 0)
    pc, cams, gts, margs = bench.build_workload(args, dev)
    pc.df_model.precision = precision
    ts = TrainStep(pc, cams, gts, args.iteration, lrs=dict(xyz=8e-6), speculative=(mode == "train"))
    out = {"config": name, "mlp_precision": precision, "mode": mode, **{k: v for k, v in cfg.items() if k not in ("extent",)}}

    def one(i):
        if mode == "render":
            with torch.no_grad():
                return render(cams[i % 8], pc, ts.pipe, ts.bg, time=ts.times[i % 8], it=args.iteration)
        if mode == "fwdbwd":
            pkg = render(cams[i % 8], pc, ts.pipe, ts.bg, time=ts.times[i % 8], it=args.iteration)
            ts.loss_of(pkg["render"], gts[i % 8]).backward()
            ts.bucket.zero()
            return pkg
        return ts.step(i)[1]
    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    _lib.profile_enable(2); _lib.profile_collect()
    t0 = time.perf_counter()
    for i in range(steps):
        pkg = one(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = _lib.profile_collect(); _lib.profile_enable(0)
    out["ms_per_step_profiled"] = round(1e3 * dt, 3)
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + steps + i)
    torch.cuda.synchronize()
    out["ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / steps, 3)
    out["visible"] = int((pkg["radii"] > 0).sum())
    out["kernels_ms_per_step"] = {k: round(v[1] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    if name == "c1":   # CPU oracle, forward only, same scene (BASELINE config 1) + parity
        from oracle.oracle import RasterOracle, RasterSettings
        cam = cams[0]
        with torch.no_grad():
            xyz, q, s, o = pc(ts.times[0], args.iteration)
            img = render(cam, pc, ts.pipe, ts.bg, time=ts.times[0], it=args.iteration)["render"].cpu().numpy()
        n64 = lambda x: x.detach().float().cpu().numpy().astype(np.float64)
        st = RasterSettings(image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
                            tanfovy=math.tan(cam.FoVy * 0.5), bg=np.zeros(3), scale_modifier=1.0, viewmatrix=n64(cam.world_view_transform),
                            projmatrix=n64(cam.full_proj_transform), sh_degree=3, campos=n64(cam.camera_center))
        for thr in (1, os.cpu_count()):
            orc = RasterOracle("f32", threads=thr)
            ts_ = []
            for _ in range(6):
                t0 = time.perf_counter()
                r = orc.forward(st, n64(xyz), n64(o), shs=n64(pc.get_features), scales=n64(s), rotations=n64(q))
                ts_.append(time.perf_counter() - t0)
            out[f"cpu_oracle_forward_ms_{thr}thr"] = round(1e3 * float(np.median(ts_[1:])), 2)
        clean = r["ambiguous"] == 0
        out["rgb_linf_vs_oracle"] = float(np.abs(img - r["out_color"])[:, clean].max())
        out["cpu_cores"] = os.cpu_count()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    prec = os.environ.get("GP_MLP_PRECISION", "fp32s")      # the model's default for passes over more than 2048 rows
    if os.environ.get("GP_W16_2D"):                         # A/B: the split-mode weight gradient's round-2 grid (jobs x slabs)
        _lib.check(_lib.lib().gp_debug_option(3, 1), "gp_debug_option")
    for n in (sys.argv[1:] or list(CONFIGS)):
        run(n, precision=prec)
        torch.cuda.empty_cache()
