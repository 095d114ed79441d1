"""-m gpu: the ONE JSON line bench.py prints -- the fields the driver and the judge read (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, the `roofline` and `cpu_baseline` objects) are
present, typed and mutually consistent.  Run on a small scene (the contract is the same code path as the headline run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--gaussians", "60000", "--width", "400",
                        "--height", "304", "--keypoints", "100", *extra], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                         # exactly one line on stdout
    return json.loads(lines[0])


def test_bench_prints_one_well_formed_line():
    d = _bench("--no-weights-model-step")
    assert d["metric"].startswith("rendered views/s") and d["unit"] == "views/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["ms_per_step"] > 0 and abs(d["value"] - 1000.0 / d["ms_per_step"]) <= 0.01 * d["value"]      # whole-job throughput = views / time
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and (r["traffic"] is None or r["traffic"] > 0)
    assert r["avg_ms"] > 0 and r["avg_ms"] < d["ms_per_step"]                                             # the roofline kernel is part of the step
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_ms"] * 1e-3) / 1e9) <= 0.01 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str) and c["unit"]
    ks = d["kernels_ms"]
    for name in ("composite_fwd", "composite_bwd", "preprocess_fwd", "preprocess_bwd", "depth_sort", "adam", "l1_ssim_fused"):
        assert ks[name]["ms_per_step"] > 0, name
    # one stream: the kernels fit in the step.  (Each entry of the table carries its hipEvent bracket, ~3 us x 27 launches -- a quarter of
    # this small scene's step, which itself is timed with one bracket per eight steps: hence the margin.)
    assert sum(v["ms_per_step"] for v in ks.values()) <= 1.35 * d["ms_per_step"]
    assert d["dense_variant"]["R_per_gaussian"] > d["config"]["R_per_gaussian"]
    assert d["dense_variant"]["contributing_pairs"] > 0 and d["dense_variant"]["evaluated_pairs"] >= d["dense_variant"]["contributing_pairs"]
    assert r["binding_resource"] == "valu"
    # the headline's depth sort runs under a key-range promise that depends on the scene: the same step without it is in the record
    if "promised base" in d["config"]["depth_sort"]:
        assert d["ms_per_step_without_depth_promise"] > 0 and d["depth_sort_four_pass_ms"] > 0
