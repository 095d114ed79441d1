# what the composite backward's accumulator atomics / pixel walk cost: composite_lab with gp_debug_option(1, bits)
# (1 = flush without atomics, 2 = no pixel walk).  Gradients of the ablated variants are wrong by construction.
for b in 0 1 2 3; do
  timeout 120 python tools/composite_lab.py --fwd 0 --bwd $b --reps 10 2>/dev/null | grep -o '"composite_bwd": [0-9.]*' | sed "s/^/bwd variant $b: /"
done
