#!/bin/bash
# tools/ab_bench.sh <dirA> <dirB> [rounds]: alternate `bench.py` of two checkouts on ONE box (boxes differ by several percent in
# HBM speed, so step times of different gpurun calls do not compare); prints ms_per_step and the per-kernel table of each run.
A=${1:-ab_prev}; B=${2:-.}; N=${3:-2}
for i in $(seq 1 $N); do
  for d in "$A" "$B"; do
    (cd "$d" && python bench.py --no-cpu-baseline --no-weights-model-step 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$d', d['ms_per_step'], 'sum', round(sum(v['ms_per_step'] for v in k.values()),4), ' '.join(f'{n}={v[\"ms_per_step\"]:.4f}' for n,v in sorted(k.items())))"
  done
done
