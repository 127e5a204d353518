#!/bin/bash
# Round 5, round-end measurements on the GPU box (through gpurun): every BASELINE.json config + the reference's shapes (bench lines with
# CPU baseline and parity sample; kernel traces), the --text tower line, the configs[4] line with the fp16 render, the API lines, the ViT MFMA
# passes.  Outputs under gpurun_out/r05/; copy what is to be judged into profiles/.  (The marcher's regimes: tools/r05_march_regimes.sh;
# parity under trained-like statistics: tests/diag/adversarial_parity.py, tests/diag/trained_field_parity.py.)
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
line() { grep '^{"metric"' | tail -1; }
python bench.py --config 1 --steps 20 --warmup 5                2>$OUT/cfg1.err | line > $OUT/${TAG}_cfg1_bench.json
python bench.py --config 0 --steps 5 --warmup 2                 2>$OUT/cfg0.err | line > $OUT/${TAG}_cfg0_bench.json
python bench.py --config 2 --steps 3 --warmup 1 --cpu-sample 16 2>$OUT/cfg2.err | line > $OUT/${TAG}_cfg2_bench.json
python bench.py --config 3 --steps 2 --warmup 1 --cpu-sample 16 2>$OUT/cfg3.err | line > $OUT/${TAG}_cfg3_bench.json
python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 8  2>$OUT/cfg4.err | line > $OUT/${TAG}_cfg4_bench.json
python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 8 --opt mlp_f16=0 2>$OUT/cfg4bf16.err | line > $OUT/${TAG}_cfg4_bf16render_bench.json
python bench.py --clip vit_l14_336 --width 336 --height 336 --poses-per-gpu 1024 --steps 3 --warmup 1 --cpu-sample 4 2>$OUT/ref.err | line > $OUT/${TAG}_refshapes_bench.json
python bench.py --config 1 --steps 5 --warmup 2 --cpu-sample 0 --power-seconds 0 --text tower 2>$OUT/cfg1tt.err | line > $OUT/${TAG}_cfg1_texttower_bench.json
python bench.py --config 1 --steps 5 --warmup 2 --cpu-sample 32 --power-seconds 0 --opt mlp_f16=1 2>$OUT/cfg1f16.err | line > $OUT/${TAG}_cfg1_fp16render_bench.json
for f in $OUT/${TAG}_*_bench.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], d["value"], "cand/s", d["device_ms_per_step"], "march", d["roofline"]["frac"], d["roofline"].get("brick_config"), "vit TF", d["roofline_vit"]["achieved"],
          "parity", d.get("parity_vs_oracle"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
# API lines
a() { python bench.py --api --steps 1 --warmup 1 "$@" 2>$OUT/api.err | tail -1; }
a                                   > $OUT/${TAG}_api_ref_scores.json
a --api-save 1                      > $OUT/${TAG}_api_ref_png.json
a --config 1 --steps 3              > $OUT/${TAG}_api_cfg1_scores.json
# kernel traces (no CPU leg under the profiler)
PROF_TIMEOUT=500 tools/profile.sh ${TAG}/cfg1 --config 1 --steps 5 --warmup 2 --cpu-sample 0 --power-seconds 0 > /dev/null
PROF_TIMEOUT=600 tools/profile.sh ${TAG}/cfg4 --config 4 --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 > /dev/null
PROF_TIMEOUT=500 tools/profile.sh ${TAG}/refshapes --clip vit_l14_336 --width 336 --height 336 --poses-per-gpu 1024 --steps 2 --warmup 1 --cpu-sample 0 --power-seconds 0 > /dev/null
PROF_TIMEOUT=500 tools/profile.sh ${TAG}/cfg2 --config 2 --steps 2 --warmup 1 --cpu-sample 0 --power-seconds 0 > /dev/null
# ViT: MFMA busy, L2 hit / miss, fabric traffic of the four GEMM products (refresh of profiles/r03_pmc_march_traffic.md's GEMM rows)
PMC_PASS_TIMEOUT=300 tools/pmc.sh ${TAG}/pmc_vit197 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" -- --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 > $OUT/pmc_vit197.log 2>&1
ls $OUT
