#!/usr/bin/env python
"""MFMA utilisation of the large-row MLP kernels from tools/collect_mlp_evidence.sh's output:
    python tools/mlp_evidence_table.py gpurun_out/<tag> > profiles/rNN_mlp_mfma.jsonl
One JSON line per (precision, kernel, launch shape): average duration from the rocprofv3 kernel trace of the counter pass itself,
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)  (the counter counts cycles, summed over the SIMDs:
MI355X_MICROARCH.md), valu_busy = 4 x SQ_ACTIVE_INST_VALU / the same denominator (quad-cycles), and -- for the launches that cover
all five layers of one pass -- the FLOP rate against the dense 16-bit MFMA peak (2.5 PFLOP/s vendor, 2.2 measured; the split-fp16
mode spends three MFMA products per fp32-grade product)."""
import collections, csv, json, os, sys
root = sys.argv[1]
CLK, SIMDS = 2.4e9, 1024
D_IN, W, OUT = 32 + 60 + 12, 256, 7
FLOP_ROW = 2 * (D_IN * W + 3 * W * W + W * OUT)
for prec in ("fp32s", "fp16"):
    f = os.path.join(root, f"pmc_{prec}", "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "gp_mlp16" not in r["Kernel_Name"]:
            continue
        d = disp[r["Dispatch_Id"]]
        d["name"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d["grid"] = int(r["Grid_Size"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    groups = collections.defaultdict(list)
    for d in disp.values():
        groups[(d["name"], d["grid"])].append(d)
    for (name, grid), ds in sorted(groups.items()):
        n = len(ds)
        ns = sum(d["ns"] for d in ds) / n
        cyc = ns * 1e-9 * CLK * SIMDS
        mean = lambda k: sum(d.get(k, 0.0) for d in ds) / n
        out = {"precision": prec, "kernel": name, "grid_threads": grid, "launches": n, "avg_us_under_pmc": round(ns / 1e3, 1),
               "mfma_busy": round(mean("SQ_VALU_MFMA_BUSY_CYCLES") / cyc, 4), "valu_busy": round(4 * mean("SQ_ACTIVE_INST_VALU") / cyc, 4),
               "mfma_mops_f16": mean("SQ_INSTS_VALU_MFMA_MOPS_F16"), "waves": mean("SQ_WAVES")}
        print(json.dumps(out))
    sw = os.path.join(root, f"sweep_{prec}.txt")
    if os.path.exists(sw):
        for line in open(sw):
            if line.startswith("rows"):
                rows = int(line.split()[1].rstrip(":"))
                parts = line.replace("(", " ").replace(")", " ").split()
                us = [float(parts[i + 1]) for i, p in enumerate(parts) if p in ("fwd", "bwd_data") or p == "launches"]
                names = ["fwd", "bwd_data", "bwd_weight"]
                mult = 3 if prec == "fp32s" else 1      # MFMA products per useful product
                for nm, u in zip(names, us):
                    tf = rows * FLOP_ROW / (u * 1e-6) / 1e12
                    print(json.dumps({"precision": prec, "pass": nm, "rows": rows, "us": u, "useful_TFLOPs": round(tf, 1),
                                      "mfma_TFLOPs_issued": round(tf * mult, 1), "frac_of_mfma_peak_vendor_2500": round(tf * mult / 2500, 4),
                                      "frac_of_mfma_peak_measured_2200": round(tf * mult / 2200, 4)}))
