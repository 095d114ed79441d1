#!/bin/bash
# SQ counters of the large-row split-fp16 MLP kernels (several passes: the SQ block takes 8 counters at a time):
#   tools/probe/mlp16_pmc.sh gpurun_out/<tag> [precision] [ablate bits]
set -u
OUT=${1:-gpurun_out/mlp16_pmc}; PREC=${2:-fp32s}
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_WAVE_DEP_WAIT SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d "$OUT/p$i" -o p --output-format csv -- python tools/mlp_sweep.py $PREC 1048576 > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/probe/pmc_kernels.py "$f" mlp16 >> "$OUT/summary.txt"
done
cat "$OUT/summary.txt"
