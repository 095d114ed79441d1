#!/bin/bash
# MFMA utilisation of the large-row Deformable_Field kernels (BASELINE configs[1] / [3] / [4]: the rows are the Gaussians):
#   tools/collect_mlp_evidence.sh gpurun_out/<tag>   -> <tag>/sweep_<prec>.txt (hipEvent times, TF/s) and <tag>/pmc_<prec>/ (SQ counters)
# Post-process with tools/mlp_evidence_table.py.  Counter passes never combine --pmc with a trace domain.
set -u
OUT=${1:-gpurun_out/mlp_evidence}
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for prec in fp32s fp16; do
  timeout 200 python tools/mlp_sweep.py $prec 1048576 2097152 > "$OUT/sweep_$prec.txt" 2>&1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES \
    -d "$OUT/pmc_$prec" -o p --output-format csv -- python tools/mlp_sweep.py $prec 1048576 2097152 > "$OUT/pmc_$prec.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$prec" -o s --output-format csv -- python tools/mlp_sweep.py $prec 1048576 2097152 > "$OUT/stats_$prec.log" 2>&1
done
ls "$OUT"
