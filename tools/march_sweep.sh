#!/bin/bash
# Run on the GPU box: k_march time per step for the values of one library tunable.
#   tools/march_sweep.sh refill_min 8 16 24 32 40 48 64
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['device_ms_per_step']['march'], d['roofline']['lane_utilisation'], d['value'])"; }
KEY=$1; shift
for v in "$@"; do
  python bench.py --steps 2 --warmup 1 --cpu-sample 0 --opt $KEY=$v 2>&1 | tail -1 | show "$KEY=$v"
done
