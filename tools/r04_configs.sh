#!/bin/bash
# Run on the GPU box (through gpurun): one bench line + one kernel-trace summary per BASELINE.json config, the
# reference's own shapes, and MFMA-utilisation passes on the ViT kernels.  Outputs under gpurun_out/r04/;
# copy what is to be judged into profiles/.
#   tools/r04_configs.sh [tag]
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
line() { grep '^{"metric"' | tail -1; }
# bench lines (CPU baseline + parity sample included)
python bench.py --config 0 --steps 5 --warmup 2                 2>$OUT/cfg0.err | line > $OUT/${TAG}_cfg0_bench.json
python bench.py --config 1 --steps 5 --warmup 2                 2>$OUT/cfg1.err | line > $OUT/${TAG}_cfg1_bench.json
python bench.py --config 2 --steps 3 --warmup 1 --cpu-sample 16 2>$OUT/cfg2.err | line > $OUT/${TAG}_cfg2_bench.json
python bench.py --config 3 --steps 2 --warmup 1 --cpu-sample 16 2>$OUT/cfg3.err | line > $OUT/${TAG}_cfg3_bench.json
python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 8  2>$OUT/cfg4.err | line > $OUT/${TAG}_cfg4_bench.json
python bench.py --clip vit_l14_336 --width 336 --height 336 --poses-per-gpu 1024 --steps 3 --warmup 1 --cpu-sample 4 2>$OUT/ref.err | line > $OUT/${TAG}_refshapes_bench.json
for f in $OUT/${TAG}_*_bench.json; do python - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], d["value"], "cand/s", d["device_ms_per_step"], "march", d["roofline"]["frac"], "vit TF", d["roofline_vit"]["achieved"],
      "parity", d.get("parity_vs_oracle"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done
# kernel traces (no CPU leg under the profiler)
PROF_TIMEOUT=500 tools/profile.sh ${TAG}/cfg2 --config 2 --steps 2 --warmup 1 --cpu-sample 0 --power-seconds 0 > /dev/null
PROF_TIMEOUT=600 tools/profile.sh ${TAG}/cfg3 --config 3 --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 > /dev/null
PROF_TIMEOUT=600 tools/profile.sh ${TAG}/cfg4 --config 4 --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 > /dev/null
PROF_TIMEOUT=500 tools/profile.sh ${TAG}/refshapes --clip vit_l14_336 --width 336 --height 336 --poses-per-gpu 1024 --steps 2 --warmup 1 --cpu-sample 0 --power-seconds 0 > /dev/null
PROF_TIMEOUT=500 tools/profile.sh ${TAG}/cfg1 --config 1 --steps 5 --warmup 2 --cpu-sample 0 --power-seconds 0 > /dev/null
ls $OUT
# MFMA utilisation of the ViT kernels (GEMMs, attention) at 197 tokens (configs[1]) and 257 tokens (configs[4] slice): own passes
PMC_PASS_TIMEOUT=300 tools/pmc.sh ${TAG}/pmc_vit197 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" -- --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 > $OUT/pmc_vit197.log 2>&1
PMC_PASS_TIMEOUT=400 tools/pmc.sh ${TAG}/pmc_vit257 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES" -- --config 4 --slice-of 64 --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 > $OUT/pmc_vit257.log 2>&1
# (the marcher's counters: tools/r04_pmc_march.sh)
