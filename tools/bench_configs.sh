#!/bin/bash
# Run on the GPU box: one bench line per workload of interest.
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', d['value'], 'cand/s;', d['device_ms_per_step'], 'march frac', d['roofline']['frac'], 'vit TF', d['roofline_vit']['achieved'], 'chunk', d['config']['chunk'])"; }
python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | show "configs[1] shopping 4096 B/16"
python bench.py --steps 1 --warmup 1 --cpu-sample 0 --scene pool_triangle --poses-per-gpu 16384 2>&1 | tail -1 | show "configs[2] pool_triangle 16384 B/16"
python bench.py --steps 1 --warmup 1 --cpu-sample 0 --clip vit_l14 2>&1 | tail -1 | show "ViT-L/14 640x360 4096"
python bench.py --steps 1 --warmup 1 --cpu-sample 0 --clip vit_l14_336 --width 336 --height 336 --poses-per-gpu 1024 2>&1 | tail -1 | show "reference shapes 336x336 L/14-336 1024"
python bench.py --steps 2 --warmup 1 --cpu-sample 0 --width 160 --height 90 --poses-per-gpu 1024 2>&1 | tail -1 | show "160x90 1024 B/16"
