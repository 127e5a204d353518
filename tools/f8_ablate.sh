#!/bin/bash
# Run on the GPU box (through gpurun): what a tile's time is made of in the fp8 GEMMs — development builds with D2R_F8_EXP masks
# (clip_dev.h) and the per-section cycle stamps.  Results of the ablated builds are garbage; only the stamps are read.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for m in ${*:-0 1 3 7 8 16 32 64}; do
  make -C dream2real_amd/csrc DEV=1 EXTRA="-DD2R_GEMM_STAMPS -DD2R_F8_EXP=$m" -j16 2>&1 | grep -E "error|Error"
  echo "== D2R_F8_EXP=$m"
  timeout 300 python tools/gemm_stamps.py 2048 vit_l14 fp8 2>&1 | grep "fp8:"
done
