#!/bin/bash
# Round 4: the drop-in API (ImaginationEngine.dream_best_pose) timed on the reference's own workload and on configs[1],
# scores-only and with cb_render/*.png written.  Run on the GPU box; writes gpurun_out/r04_api_*.json.
mkdir -p gpurun_out
df -h /tmp | tail -1
a() { python bench.py --api --steps 1 --warmup 1 "$@" 2>gpurun_out/r04_api_err.log | tail -1; }
a                                   > gpurun_out/r04_api_ref_scores.json
a --api-save 1                      > gpurun_out/r04_api_ref_png.json
a --config 1 --steps 3              > gpurun_out/r04_api_cfg1_scores.json
a --config 1 --steps 3 --api-save 1 > gpurun_out/r04_api_cfg1_png.json
a --config 1 --steps 3 --api-phys 0 > gpurun_out/r04_api_cfg1_nophys.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_api_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config']['poses_valid'], d['peak_host_rss_gb'], d['device_ms_per_step'])
    except Exception as e: print(f, 'FAILED', e)
PY
tail -5 gpurun_out/r04_api_err.log
