#!/bin/bash
# A/B: the per-Gaussian half of the composite forward's sub-block test computed by the projection kernel (shipped) against per instance in
# the staging lanes (-DGP_SB_HOIST=0, the round-5 form); alternating builds on one box.   tools/probe/sb_hoist_ab.sh gpurun_out/<tag>
OUT=${1:-gpurun_out/sb_hoist}; mkdir -p "$OUT"
for v in 2 0 1 2 0 1; do
  GP_EXTRA_HIP_FLAGS="-DGP_SB_HOIST=$v" python -c "import __graft_entry__ as g; g.build(force=True)" > "$OUT/build_$v.log" 2>&1
  echo "== GP_SB_HOIST=$v" >> "$OUT/summary.txt"
  timeout 200 python tools/composite_lab.py --fwd 0 --bwd 0 --reps 20 2>/dev/null | grep -o '"composite_fwd": [0-9.]*\|"preprocess_fwd": [0-9.]*' >> "$OUT/summary.txt"
done
python -c "import __graft_entry__ as g; g.build(force=True)" > "$OUT/build_final.log" 2>&1
cat "$OUT/summary.txt"
