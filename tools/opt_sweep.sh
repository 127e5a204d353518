#!/bin/bash
# Run on the GPU box: bench line summary for the values of one library tunable.
#   tools/opt_sweep.sh refill_min 16 32 64        tools/opt_sweep.sh ln_fold 0 1
KEY=$1; shift
for v in "$@"; do
  python bench.py --steps 3 --warmup 1 --cpu-sample 0 ${BENCH_ARGS} --opt $KEY=$v 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$KEY=$v', d['value'], d['device_ms_per_step'], 'march frac', d['roofline']['frac'], 'vit TF', d['roofline_vit']['achieved'])"
done
