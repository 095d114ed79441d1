#!/bin/bash
# SQ counters of the composite backward's two kernels (quadrant kernel, gp_debug_option(7, 0); sub-block kernel, 3) on the bench workload:
#   tools/probe/bwd_pmc.sh gpurun_out/<tag>      (two counter passes; never combined with a trace domain)
set -u
OUT=${1:-gpurun_out/bwd_pmc}
mkdir -p "$OUT"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES \
  -d "$OUT/sq1" -o sq1 --output-format csv -- python tools/composite_lab.py --fwd 0 --bwdk 0,3 --reps 3 > "$OUT/sq1.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM \
  -d "$OUT/sq2" -o sq2 --output-format csv -- python tools/composite_lab.py --fwd 0 --bwdk 0,3 --reps 3 > "$OUT/sq2.log" 2>&1
for f in $(find "$OUT" -name "*counter_collection.csv"); do echo "== $f"; python tools/probe/pmc_kernels.py "$f" composite_bwd; done | tee "$OUT/summary.txt"
