"""Where the back-to-back eval loop's time goes: host time per SpeculativeRenderer call (no synchronisation inside), frame by frame,
and the loop's rate for several ring sizes / warm-up lengths.  python tools/probe/eval_pipeline_probe.py   (bench.py's scene)"""
import json, sys, time
from types import SimpleNamespace
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from gaussianprediction_amd.renderer import SpeculativeRenderer

args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
pc, cams, gts, margs = bench.build_workload(args, dev)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
bg = torch.zeros(3, device=dev)
times = [torch.from_numpy(c.time).float().to(dev) for c in cams]
out = {}
with torch.no_grad():
    for slots, warm in ((32, 8), (32, 40), (8, 8), (8, 20), (16, 40)):
        sr = SpeculativeRenderer(pc, pipe, bg, slots=slots)
        for i in range(warm):
            sr(cams[i % 8], time=times[i % 8], it=args.iteration)
        sr.flush(); torch.cuda.synchronize()
        for rep in range(3):
            host = []
            t0 = time.perf_counter()
            for i in range(50):
                a = time.perf_counter()
                sr(cams[i % 8], time=times[i % 8], it=args.iteration)
                host.append(time.perf_counter() - a)
            sr.flush(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            h = np.array(host) * 1e3
            out[f"slots{slots}_warm{warm}_rep{rep}"] = {"views_per_s": round(50 / dt, 1), "host_ms_median": round(float(np.median(h)), 3),
                                                        "host_ms_max": round(float(h.max()), 3), "host_ms_first10": [round(float(x), 2) for x in h[:10]],
                                                        "mem_reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2)}
            print(f"slots{slots}_warm{warm}_rep{rep}", json.dumps(out[f"slots{slots}_warm{warm}_rep{rep}"]), flush=True)
