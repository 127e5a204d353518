#!/bin/bash
# Round 4: the two-stream pipeline of d2r_render_score (render half of chunk i+1 on a second stream while the ViT scores
# chunk i), same-box A/B.  Run on the GPU box; writes gpurun_out/r04_overlap_*.json (one bench line each).
mkdir -p gpurun_out
b() { python bench.py --steps ${STEPS:-6} --warmup 2 --cpu-sample 0 --power-seconds 0 "$@" 2>/dev/null | tail -1; }
for r in 1 2; do
  b --chunk 4096                              > gpurun_out/r04_overlap_c4096_single_$r.json
  b --chunk 1024 --opt overlap=0              > gpurun_out/r04_overlap_c1024_off_$r.json
  b --chunk 1024 --opt overlap=1              > gpurun_out/r04_overlap_c1024_on_$r.json
  b --chunk 2048 --opt overlap=0              > gpurun_out/r04_overlap_c2048_off_$r.json
  b --chunk 2048 --opt overlap=1              > gpurun_out/r04_overlap_c2048_on_$r.json
done
# four chunks of 4096 per step (configs[2]: 16 384 poses)
b --config 2 --steps 3 --opt overlap=0        > gpurun_out/r04_overlap_cfg2_off.json
b --config 2 --steps 3 --opt overlap=1        > gpurun_out/r04_overlap_cfg2_on.json
for m in 32 64 128; do
  b --config 2 --steps 3 --opt overlap=1 --opt march_blocks=$m > gpurun_out/r04_overlap_cfg2_on_mb$m.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_overlap_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['device_ms_per_step'])
    except Exception as e: print(f, 'FAILED', e)
PY
