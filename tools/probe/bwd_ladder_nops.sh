#!/bin/bash
# Bound for "pair the composite backward's DPP ladders so that nothing waits" (round-5 verdict, item 4a): the kernel with the ladders' wait
# states REMOVED (wrong sums, timing only) against the shipped one, alternating builds on one box.
#   tools/probe/bwd_ladder_nops.sh gpurun_out/<tag>
OUT=${1:-gpurun_out/bwd_ladder_nops}; mkdir -p "$OUT"
for v in 1 0 1 0; do
  GP_EXTRA_HIP_FLAGS="-DGP_CB_LADDER_NOPS=$v" python -c "import __graft_entry__ as g; g.build(force=True)" > "$OUT/build_$v.log" 2>&1
  echo "== GP_CB_LADDER_NOPS=$v" >> "$OUT/summary.txt"
  timeout 200 python tools/composite_lab.py --fwd 0 --bwd 0 --reps 20 2>/dev/null | grep -o '"composite_bwd": [0-9.]*' >> "$OUT/summary.txt"
done
python -c "import __graft_entry__ as g; g.build(force=True)" > "$OUT/build_final.log" 2>&1
cat "$OUT/summary.txt"
