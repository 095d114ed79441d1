#!/usr/bin/env python
import json, sys
d = json.load(open(sys.argv[1]))
print(f"ms_per_step {d['ms_per_step']}  value {d['value']} {d['unit']}  R={d['config'].get('R')}  roofline={d.get('roofline')}")
tot = 0
for k, v in sorted(d["kernels_ms"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    tot += v["ms_per_step"]
    print(f"  {k:16s} launches/step {v['launches_per_step']:5.1f}  per-step {v['ms_per_step']:.4f} ms")
print(f"  sum of instrumented kernels per step: {tot:.3f} ms")
if "cpu_baseline" in d:
    print("  cpu_baseline:", d["cpu_baseline"])
