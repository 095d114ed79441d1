#!/bin/bash
# The first contact with a multi-GPU node in one command: 1 / 2 / 4 / 8 ranks x the three gradient-exchange forms, every run with the
# compute stream's exposed waits per collective kind (--time-waits), one table at the end.
#   tools/scale_sweep.sh [out dir] [max ranks]       (each bench.py call launches its own ranks and refuses fewer devices than asked)
# Forms: sharded (reduce-scatter -> 1/N Adam -> all-gather, SH region chained on a side stream: the default), sharded-nochain (everything
# awaited on the compute stream), replicated (all-reduce + replicated Adam), factorised (replicated Adam, SH gradients as 24-B factors).
OUT=${1:-gpurun_out/scale_sweep}; MAXN=${2:-8}
mkdir -p "$OUT"
for n in 1 2 4 8; do
  [ "$n" -gt "$MAXN" ] && break
  for form in sharded sharded-nochain replicated factorised; do
    [ "$n" -eq 1 ] && [ "$form" != "sharded" ] && continue
    case $form in
      sharded) flags="--time-waits";;
      sharded-nochain) flags="--time-waits --no-chain-sh";;
      replicated) flags="--replicated-adam";;
      factorised) flags="--factorised-sh";;
    esac
    timeout 900 python bench.py --gpus $n --steps 30 --warmup 10 --no-cpu-baseline --no-dense-variant --no-weights-model-step $flags \
      > "$OUT/n${n}_${form}.json" 2> "$OUT/n${n}_${form}.err" || echo "n=$n $form: exit $?" >> "$OUT/failures.txt"
  done
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "n*_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    n, form = os.path.basename(f)[:-5].split("_", 1)
    rows.append((int(n[1:]), form, d["ms_per_step"], d["value"], d["config"].get("xgmi_bytes_sent_per_rank_per_step"), d["config"].get("exposed_wait_ms_per_step")))
base = next((r[3] for r in rows if r[0] == 1), None)
print(f"{'ranks':>5} {'form':16} {'ms/step':>8} {'views/s':>9} {'x of 1 rank':>11} {'MB sent/rank':>12}  exposed waits (ms/step)")
for n, form, ms, v, b, w in rows:
    print(f"{n:5d} {form:16} {ms:8.3f} {v:9.1f} {(v / base if base else float('nan')):11.2f} {(b or 0) / 1e6:12.1f}  {w}")
PY
