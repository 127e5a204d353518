#!/bin/bash
# Same-box A/B of library builds (run on the GPU box): copy the builds to compare into ab/ (git-ignored *.so; they
# travel with the gpurun snapshot), then  tools/ab.sh lib_a.so lib_b.so ...  — alternating, ROUNDS times each.
# Boxes differ by +-1.5 % in the clock their power manager grants; runs on one box repeat to +-0.1 ms.
for r in $(seq 1 ${ROUNDS:-3}); do
  for l in "$@"; do
    cp ab/$l dream2real_amd/libd2r.so
    python bench.py --steps ${STEPS:-8} --warmup 2 --cpu-sample 0 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['device_ms_per_step'])"
  done
done
