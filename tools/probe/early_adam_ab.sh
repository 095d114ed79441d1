# A/B on ONE box, alternating: the optimizer's rider in the MLP data backward's launch (default) against one optimizer launch behind the
# backward (--no-early-adam), with the feature-split small-row MLP kernels (default) and with the 16-row kernels (--debug-option 13=1)
for i in 1 2; do
  for f in "" "--no-early-adam" "--debug-option 13=1" "--debug-option 13=1 --no-early-adam"; do
    python bench.py --no-cpu-baseline --no-weights-model-step $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('[$f]', d['ms_per_step'], 'sum', round(sum(v['ms_per_step'] for v in k.values()),4), ' '.join(f'{n}={v[\"ms_per_step\"]:.4f}' for n,v in sorted(k.items()) if n.startswith(('adam','mlp'))))"
  done
done
