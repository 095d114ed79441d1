for i in 1 2; do
  for f in "" "--early-adam" "--early-adam --debug-option 12=1" "--early-adam --debug-option 12=2"; do
    python bench.py --no-cpu-baseline --no-weights-model-step $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('[$f]', d['ms_per_step'], 'sum', round(sum(v['ms_per_step'] for v in k.values()),4), ' '.join(f'{n}={v[\"ms_per_step\"]:.4f}' for n,v in sorted(k.items()) if n.startswith(('adam','mlp'))))"
  done
done
