#!/bin/bash
# Run on the GPU box: schedule variants of k_gemm8f (D2R_F8_VAR masks, clip_dev.h) — cycle stamps and the fp8 bench line of each.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for m in ${*:-0 1 2 4 6}; do
  make -C dream2real_amd/csrc DEV=1 EXTRA="-DD2R_GEMM_STAMPS -DD2R_F8_VAR=$m" -j16 2>&1 | grep -E "error|Error"
  echo "== D2R_F8_VAR=$m"
  timeout 300 python tools/gemm_stamps.py 2048 vit_l14 fp8 2>&1 | grep "fp8:"
  python bench.py --config 4 --slice-of 16 --steps 2 --warmup 1 --cpu-sample 0 --power-seconds 0 --vit-fp8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['device_ms_per_step'])"
done
