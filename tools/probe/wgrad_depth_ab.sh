#!/bin/bash
# A/B of the weight-gradient kernels' prefetch depth (W16_DEPTH k-steps of operands in flight per wave): rebuilds the library on the box.
#   tools/probe/wgrad_depth_ab.sh gpurun_out/<tag>
OUT=${1:-gpurun_out/wgrad_depth}; mkdir -p "$OUT"
for d in 4 8 4 8; do
  GP_EXTRA_HIP_FLAGS="-DW16_DEPTH=$d" python -c "import __graft_entry__ as g; g.build(force=True)" > "$OUT/build_$d.log" 2>&1
  for prec in fp32s fp16; do
    echo "== W16_DEPTH=$d $prec" >> "$OUT/summary.txt"
    timeout 200 python tools/mlp_sweep.py $prec 1048576 2>&1 | grep rows >> "$OUT/summary.txt"
  done
done
python -c "import __graft_entry__ as g; g.build(force=True)" > "$OUT/build_final.log" 2>&1
cat "$OUT/summary.txt"
