#!/bin/bash
# Collects the measurements DESIGN.md section 5 quotes, on a GPU box:  tools/collect_evidence.sh gpurun_out/<tag>
#   bench line (200 steps) | rocprofv3 kernel trace + stats of a 20-step bench | HBM counters (FETCH_SIZE, WRITE_SIZE: separate
#   passes) of the bench and of the calibration kernels | SQ counters of the composite kernels (two passes of 8).
# Counter passes never combine --pmc with a trace domain.  Post-process on any machine with tools/evidence_to_profiles.sh.
set -u
OUT=${1:-gpurun_out/evidence}
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-weights-model-step --no-dense-variant"
timeout 600 python bench.py --steps 200 --warmup 30 > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o s --output-format csv -- $B --steps 20 --warmup 5 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d "$OUT/pmc_bench_$c" -o b --output-format csv -- $B --steps 3 --warmup 2 > "$OUT/pmc_bench_$c.log" 2>&1
  timeout 300 rocprofv3 --pmc $c -d "$OUT/pmc_cal_$c" -o c --output-format csv -- python tools/pmc_calibrate.py > "$OUT/pmc_cal_$c.log" 2>&1
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES \
  -d "$OUT/pmc_sq1" -o sq1 --output-format csv -- python tools/composite_lab.py --fwd 2,0 --reps 3 > "$OUT/pmc_sq1.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE \
  -d "$OUT/pmc_sq2" -o sq2 --output-format csv -- python tools/composite_lab.py --fwd 2,0 --reps 3 > "$OUT/pmc_sq2.log" 2>&1
ls "$OUT"
tail -c 600 "$OUT/bench.json"
