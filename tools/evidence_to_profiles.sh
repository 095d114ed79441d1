#!/bin/bash
# tools/evidence_to_profiles.sh gpurun_out/<tag> r02_v3   ->  profiles/r02_v3_{bench,bench_under_rocprof}.json, _kernel_stats.txt,
# _pmc_hbm.txt, _pmc_sq.txt and profiles/composite_fwd_traffic.json (stamped with the current kernel source)
set -eu
IN=$1; TAG=$2
cp "$IN/bench.json" "profiles/${TAG}_bench.json"
cp "$IN/bench_under_rocprof.json" "profiles/${TAG}_bench_under_rocprof.json"
python tools/pmc_hbm.py "$IN/pmc_bench_FETCH_SIZE" "$IN/pmc_bench_WRITE_SIZE" "$IN/pmc_cal_FETCH_SIZE" "$IN/pmc_cal_WRITE_SIZE" "profiles/${TAG}_pmc_hbm.txt" profiles/composite_fwd_traffic.json
python - "$IN" "$TAG" <<'PY'
import csv, sys
inp, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f"{inp}/stats/s_kernel_stats.csv")))
out = ["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weights-model-step   (MI355X)",
       f"{'kernel':62s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
for r in rows[:48]:
    out.append(f"{r['Name'][:62]:62s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} "
               f"{float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
open(f"profiles/{tag}_kernel_stats.txt", "w").write("\n".join(out) + "\n")
PY
{ echo "# rocprofv3 --pmc <8 SQ counters> (two passes) -- python tools/composite_lab.py --fwd 2,0 --reps 3   (per-dispatch means; SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles; VALU busy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * kernel cycles at 2.4 GHz))";
  python tools/pmc_table.py "$IN/pmc_sq*/*counter_collection.csv" gp_composite; } > "profiles/${TAG}_pmc_sq.txt"
python - "$TAG" <<'PY'
# the VALU figures bench.py prints beside the HBM fraction: SQ counters of the shipped composite forward, same kernel source
import json, re, sys
tag = sys.argv[1]
txt = open(f"profiles/{tag}_pmc_sq.txt").read()
blk = txt[txt.index("== gp_composite_fwd_sb_kernel"):]
nxt = blk.find("\n== ", 3)
blk = blk if nxt < 0 else blk[:nxt]
val = lambda name: float(re.search(name + r"\s+mean\s+([0-9.]+)", blk).group(1))
j = json.load(open("profiles/composite_fwd_traffic.json"))
j["valu"] = {"SQ_INSTS_VALU": val("SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU": val("SQ_ACTIVE_INST_VALU"), "SQ_BUSY_CYCLES": val("SQ_BUSY_CYCLES"),
             "source": f"profiles/{tag}_pmc_sq.txt (rocprofv3 --pmc SQ counters over tools/composite_lab.py; same kernel source)",
             "note": "SQ_ACTIVE_INST_VALU counts quad-cycles: VALU-busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles); floor = 4 cycles per wave64 VALU instruction x SQ_INSTS_VALU / 1024 SIMDs at the shader clock"}
json.dump(j, open("profiles/composite_fwd_traffic.json", "w"), indent=1)
PY
ls -la profiles | grep "$TAG"
