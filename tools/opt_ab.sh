#!/bin/bash
# Same-box A/B of library OPTIONS (run on the GPU box): tools/opt_ab.sh "" "gemm_nsplit=1" "refill_min=32" ... ("" = defaults);
# BENCH_ARGS="--config 4 --slice-of 64 --vit-fp8" for another workload
for r in $(seq 1 ${ROUNDS:-2}); do
  for o in "$@"; do
    python bench.py --steps ${STEPS:-6} --warmup 2 --cpu-sample 0 --power-seconds 0 ${BENCH_ARGS} ${o:+--opt $o} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$o]', d['value'], d['device_ms_per_step'], 'lane_util', r.get('lane_utilisation'), 'wave_iters', d['render_stats_per_step']['wave_iters'])"
  done
done
