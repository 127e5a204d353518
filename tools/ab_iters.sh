#!/bin/bash
# Same-box A/B of library builds in ab/ with the marcher's per-iteration cost: ms per launch, wave iterations, ns per wave iteration x 3072 waves
for l in "$@"; do
  cp ab/$l dream2real_amd/libd2r.so
  python bench.py --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 --power-seconds 0 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['device_ms_per_step']['march']; it=d['render_stats_per_step']['wave_iters']; s=d['render_stats_per_step']['samples']
print('$l', 'march ms', m, 'wave_iters', it, 'samples', s, 'us per 1000 wave-iterations', round(m*1e3/it*1e3, 3))"
done
