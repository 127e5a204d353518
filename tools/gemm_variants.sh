#!/bin/bash
# Run on the GPU box: rebuild libd2r with each "make" argument string and report CLIP ms/step.
#   tools/gemm_variants.sh 'EXTRA=-DD2R_GEMM_PF=2' 'GEMM_ABLATE=14 EXTRA=-DD2R_GEMM_LDPAD=64' ...
for V in "" "$@"; do
  make -C dream2real_amd/csrc -j3 $V 2>&1 | grep -E " error"
  echo -n "variant [$V]  "
  python bench.py --steps 2 --warmup 1 --cpu-sample 0 ${BENCH_ARGS} 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cand/s', d['value'], 'clip ms/step', d['device_ms_per_step']['clip'], ' vit TF', d['roofline_vit']['achieved'])"
done
make -C dream2real_amd/csrc -j3 2>&1 | grep -E " error"
