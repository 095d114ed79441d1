#!/usr/bin/env python
"""HBM traffic per kernel from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE, WRITE_SIZE; CSV output), and the
calibration of those counters from two more passes over tools/pmc_calibrate.py (kernels of known byte counts).

MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of what a wide coalesced read stream fetches; other
access shapes are uncalibrated.  The composite kernels GATHER 48-byte records, so the factor is measured for gathers too and the
table states which factor was applied to which kernel.
    python tools/pmc_hbm.py <bench fetch dir> <bench write dir> <calib fetch dir> <calib write dir> profiles/rNN_pmc_hbm.txt \\
                            profiles/composite_fwd_traffic.json"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows_of(d, counter):
    out = []
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                out.append((int(row.get("Dispatch_Id", 0)), row["Kernel_Name"].split("(")[0], float(row["Counter_Value"])))
    out.sort()
    return out


def mean_by_kernel(rows):
    acc = defaultdict(list)
    for _, k, v in rows:
        acc[k].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def calibration(dfetch, dwrite):
    """{variant: (known read bytes, FETCH_SIZE KiB, known write bytes, WRITE_SIZE KiB)} -- variants identified by launch order."""
    order = ["gather_64B_lines_once", "gather_16B_of_64B_lines_once", "gather_48B_records_once", "gather_48B_records_4x_reuse"]
    n_req = 4_000_000
    known = {"gather_64B_lines_once": 64 * n_req, "gather_16B_of_64B_lines_once": 16 * n_req, "gather_48B_records_once": 48 * n_req,
             "gather_48B_records_4x_reuse": 48 * n_req}
    F, W = rows_of(dfetch, "FETCH_SIZE"), rows_of(dwrite, "WRITE_SIZE")
    cal = {}
    gat = [v for _, k, v in F if "gp_mb_gather" in k]
    per = len(order)
    for i, name in enumerate(order):
        vals = gat[i::per]
        if vals:
            cal[name] = {"requested_bytes": known[name], "FETCH_KiB": sum(vals) / len(vals)}
    for kern, name, rb in (("gp_mb_copy", "copy", 1 << 30), ("gp_mb_read", "read", 1 << 30)):
        vals = [v for _, k, v in F if kern in k]
        if vals:
            cal[name] = {"requested_bytes": rb, "FETCH_KiB": sum(vals) / len(vals)}
    wv = [v for _, k, v in W if "gp_mb_copy" in k]
    if wv:
        cal["copy"]["written_bytes"] = 1 << 30
        cal["copy"]["WRITE_KiB"] = sum(wv) / len(wv)
    for c in cal.values():
        c["fetch_counter_bytes_over_requested"] = round(c["FETCH_KiB"] * 1024 / c["requested_bytes"], 4)
    return cal


def main(dfetch, dwrite, cfetch, cwrite, out_txt, out_json):
    F, W = mean_by_kernel(rows_of(dfetch, "FETCH_SIZE")), mean_by_kernel(rows_of(dwrite, "WRITE_SIZE"))
    cal = calibration(cfetch, cwrite)
    stream_factor = 1.0 / cal["read"]["fetch_counter_bytes_over_requested"] if "read" in cal else 2.0
    # What the calibration shows: FETCH_SIZE = (number of L2 -> fabric read requests) x 64 B.  A wide coalesced stream issues 128-byte
    # requests (tallied at 64: x2, 'read').  A gather issues one 64-byte request per line it touches -- 'gather_16B_of_64B_lines_once':
    # 4 M lanes, 4 M distinct lines, counter = 4.09 M x 64 B -- so for gathers the counter IS the byte count (x1); and the L2 does not
    # merge misses on a line in flight ('gather_64B_lines_once': four 16-byte loads per line -> ~3 requests per line).
    # The composite kernels read their records by gather and little else: x1 is applied to them (their streamed share -- the id
    # lists, 4 B per instance -- would count double: an upper bound with x2 on everything is given beside it).
    per_req = cal["gather_16B_of_64B_lines_once"]["FETCH_KiB"] * 1024 / 4_000_000 if "gather_16B_of_64B_lines_once" in cal else 64.0
    gather_factor = 1.0 if 56.0 <= per_req <= 72.0 else stream_factor
    gather_kernels = ("gp_composite_fwd", "gp_composite_bwd")
    rows = []
    for k in F:
        f, n = F[k]
        w = W.get(k, (0.0, 0))[0]
        fac = gather_factor if any(g in k for g in gather_kernels) else stream_factor
        rows.append((1024 * (fac * f + w) / 1e6, k, n, f, w, fac))
    rows.sort(reverse=True)
    with open(out_txt, "w") as o:
        o.write("# rocprofv3 --pmc FETCH_SIZE  and (separate pass)  --pmc WRITE_SIZE  -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline\n")
        o.write("# units: KiB per dispatch (counter value); hbm_bytes = 1024 * (factor * FETCH_SIZE + WRITE_SIZE)\n")
        o.write("# calibration (tools/pmc_calibrate.py, same box, separate passes): FETCH_SIZE*1024 / bytes requested --\n")
        for name, c in cal.items():
            o.write(f"#   {name:32s} requested {c['requested_bytes'] / 1e6:9.1f} MB   FETCH_SIZE*1024 {c['FETCH_KiB'] * 1024 / 1e6:9.1f} MB   ratio "
                    f"{c['fetch_counter_bytes_over_requested']:.3f}" + (f"   WRITE_SIZE*1024 {c['WRITE_KiB'] * 1024 / 1e6:9.1f} MB" if "WRITE_KiB" in c else "") + "\n")
        o.write(f"# factor applied: streaming kernels x{stream_factor:.3f} (from 'read'), record-gather kernels ({', '.join(gather_kernels)}) "
                f"x{gather_factor:.3f} ('gather_16B_of_64B_lines_once': {per_req:.1f} B counted per 64-byte request)\n")
        o.write(f"{'kernel':42s} {'n':>4s} {'FETCH_SIZE_KiB':>16s} {'WRITE_SIZE_KiB':>16s} {'factor':>7s} {'hbm_MB_corrected':>18s}\n")
        for mb, k, n, f, w, fac in rows[:40]:
            o.write(f"{k[:42]:42s} {n:4d} {f:16.1f} {w:16.1f} {fac:7.3f} {mb:18.1f}\n")
    k = next(kk for kk in F if kk.startswith("gp_composite_fwd"))
    f, w = F[k][0], W[k][0]
    src = os.path.join(ROOT, "gaussianprediction_amd", "csrc", "raster_kernels.hip")
    json.dump({"kernel": k, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "read_correction": round(gather_factor, 4),
               "read_correction_from": "gather_16B_of_64B_lines_once (tools/pmc_calibrate.py): the counter tallies each L2->fabric read "
                                       "request at 64 B, which is what a gathered line moves; wide streams use 128-B requests (x2)",
               "calibration": cal,
               "hbm_bytes_per_launch": int(1024 * (gather_factor * f + w)), "hbm_bytes_per_launch_upper_bound": int(1024 * (2 * f + w)),
               "raw_bytes_per_launch": int(1024 * (f + w)),
               "kernel_source_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
               "source": out_txt + " (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, same bench command)"},
              open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:7])
