#!/bin/bash
# Run on the GPU box: rebuild libd2r with each k_gemm ablation mask and report CLIP ms/step.
for A in ${MASKS:-0 1 2 3 4 7 8 12}; do
  make -C dream2real_amd/csrc -j3 GEMM_ABLATE=$A 2>&1 | grep -E " error" 
  echo -n "ablate=$A  "
  python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('clip ms/step', d['device_ms_per_step']['clip'], ' vit TF', d['roofline_vit']['achieved'])"
done
make -C dream2real_amd/csrc -j3 GEMM_ABLATE=0 2>&1 | grep -E " error"
