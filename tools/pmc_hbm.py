#!/usr/bin/env python
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output), with the gfx950 read
correction (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half of a wide coalesced read; calibrated here on the
Adam kernel, which streams exactly 16 B in / 16 B out per parameter).  Writes the table and composite_fwd_traffic.json.
    python tools/pmc_hbm.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/rNN_pmc_hbm.txt profiles/composite_fwd_traffic.json"""
import csv, glob, json, sys
from collections import defaultdict

def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

def main(dfetch, dwrite, out_txt, out_json):
    F, W = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    rows = []
    for k in F:
        f, n = F[k]
        w = W.get(k, (0.0, 0))[0]
        rows.append((1024 * (2 * f + w) / 1e6, k, n, f, w))
    rows.sort(reverse=True)
    with open(out_txt, "w") as o:
        o.write("# rocprofv3 --pmc FETCH_SIZE  and (separate pass)  --pmc WRITE_SIZE  -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline\n")
        o.write("# units: KiB per dispatch (counter value); hbm_bytes = 1024 * (2*FETCH_SIZE + WRITE_SIZE)  (gfx950: FETCH_SIZE = 1/2 of wide reads)\n")
        if "gp_adam_multi_kernel" in F:
            o.write(f"# calibration: gp_adam_multi_kernel FETCH*1024 = {F['gp_adam_multi_kernel'][0] * 1024 / 1e9:.3f} GB, WRITE*1024 = "
                    f"{W.get('gp_adam_multi_kernel', (0, 0))[0] * 1024 / 1e9:.3f} GB (it reads and writes 16 B per parameter each way)\n")
        o.write(f"{'kernel':42s} {'n':>4s} {'FETCH_SIZE_KiB':>16s} {'WRITE_SIZE_KiB':>16s} {'hbm_MB_corrected':>18s}\n")
        for mb, k, n, f, w in rows[:40]:
            o.write(f"{k[:42]:42s} {n:4d} {f:16.1f} {w:16.1f} {mb:18.1f}\n")
    k = "gp_composite_fwd_kernel"
    f, w = F[k][0], W[k][0]
    json.dump({"kernel": k, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "read_correction": 2.0,
               "hbm_bytes_per_launch": int(1024 * (2 * f + w)), "raw_bytes_per_launch": int(1024 * (f + w)),
               "source": out_txt + " (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, same bench command)"},
              open(out_json, "w"), indent=1)

if __name__ == "__main__":
    main(*sys.argv[1:5])
