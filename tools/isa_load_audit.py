#!/usr/bin/env python
"""Compile the kernel sources to gfx950 assembly and list kernels whose global loads are waited for one at a time
(`s_waitcnt vmcnt(0)` a few instructions behind a load): a rolled loop or an early-out branch around conditional loads turns N
independent loads into N dependent round trips to memory.  Round 3 found the L1+SSIM halo staging (14 / 25 trips per workgroup),
the projection backward's per-Gaussian inputs and SH Adam stream, the blend forward's keypoint gathers, the radix histogram and the
neighbour search's row loads this way: step 1.45 -> 1.33 ms.
    python tools/isa_load_audit.py [file.hip ...]          (no GPU needed)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussianprediction_amd", "csrc")


def audit(path, hipcc=None):
    """{kernel symbol: (global loads, full waits, full waits right behind a load)} of one .hip file compiled for gfx950."""
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-I", os.path.join(ROOT, "include"),
                        "-I", CSRC, "-S", "--cuda-device-only", "-o", out, path], check=True, stderr=subprocess.DEVNULL)
        name, stats, last = None, {}, -100
        for i, l in enumerate(open(out)):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                name, last = m.group(1), -100
                stats[name] = [0, 0, 0]
                continue
            if name is None:
                continue
            if "global_load" in l or "buffer_load" in l:
                stats[name][0] += 1
                last = i
            if "s_waitcnt" in l and "vmcnt(0)" in l:
                stats[name][1] += 1
                stats[name][2] += i - last <= 6
            if "s_endpgm" in l:
                name = None
    return {k: tuple(v) for k, v in stats.items()}


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and not f.startswith("gp_capi")]
    for f in files:
        for k, (nl, nw, ns) in audit(f).items():
            if nl and ns >= 3:
                print(f"{os.path.basename(f):24s} {k[:64]:64s} loads {nl:4d}  full waits {nw:3d}  right behind a load {ns:3d}")
