#!/bin/bash
# Round 6, round-end measurements on the GPU box (through gpurun): the GPU suite, every BASELINE.json config + the reference's shapes (bench lines
# with CPU baseline and parity sample), the lens A/B, the bf16-operand MLP line, kernel traces, the PMC passes behind profiles/r06_pmc.json, the
# vendor's plain GEMMs on the same box.  Outputs under gpurun_out/<tag>/; copy what is to be judged into profiles/.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
line() { grep '^{"metric"' | tail -1; }
if [ "${R06_SUITE:-1}" = "1" ]; then (timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $OUT/${TAG}_gpu_suite.txt; fi
python bench.py --config 1 --steps 20 --warmup 5                2>$OUT/cfg1.err | line > $OUT/${TAG}_cfg1_bench.json
python bench.py --config 1 --steps 10 --warmup 3 --cpu-sample 0 --power-seconds 0 --lens none     2>$OUT/cfg1nl.err | line > $OUT/${TAG}_cfg1_nolens_bench.json
python bench.py --config 1 --steps 10 --warmup 3 --cpu-sample 32 --power-seconds 0 --opt mlp_f16=0 2>$OUT/cfg1bf.err | line > $OUT/${TAG}_cfg1_bf16mlp_bench.json
python bench.py --config 0 --steps 5 --warmup 2                 2>$OUT/cfg0.err | line > $OUT/${TAG}_cfg0_bench.json
python bench.py --config 2 --steps 3 --warmup 1 --cpu-sample 16 2>$OUT/cfg2.err | line > $OUT/${TAG}_cfg2_bench.json
python bench.py --config 3 --steps 2 --warmup 1 --cpu-sample 16 2>$OUT/cfg3.err | line > $OUT/${TAG}_cfg3_bench.json
python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 8  2>$OUT/cfg4.err | line > $OUT/${TAG}_cfg4_bench.json
python bench.py --clip vit_l14_336 --width 336 --height 336 --poses-per-gpu 1024 --steps 3 --warmup 1 --cpu-sample 4 2>$OUT/ref.err | line > $OUT/${TAG}_refshapes_bench.json
for f in $OUT/${TAG}_*_bench.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    pr = d["roofline"].get("products") or {}
    print(sys.argv[1].split("/")[-1], d["value"], "cand/s", d["device_ms_per_step"], "vit frac", d["roofline"]["frac"], {k: (v["avg_launch_ms"], v["frac"]) for k, v in pr.items() if isinstance(v, dict)},
          "march", d["march"]["avg_launch_ms"], d["march"]["hash_fetch_algorithmic_ratio"], d["march"]["brick_config"], "parity", d.get("parity_vs_oracle"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
# the render's distance to the emulation of tiny-cuda-nn's half arithmetic (+ the round-5 trained-field table on this library) -> gpurun_out/r06_trained_field_parity.json
timeout 900 python tests/diag/trained_field_parity.py > $OUT/${TAG}_trained_field_parity.log 2>&1
cp gpurun_out/r06_trained_field_parity.json $OUT/${TAG}_trained_field_parity.json 2>/dev/null
# API lines
a() { python bench.py --api --steps 1 --warmup 1 "$@" 2>$OUT/api.err | tail -1; }
a                                   > $OUT/${TAG}_api_ref_scores.json
a --config 1 --steps 3              > $OUT/${TAG}_api_cfg1_scores.json
# the vendor library's plain GEMMs on this box (the comparison VERDICT r05 asks to keep beside the kernel trace)
python tools/gemm_ref_bench.py > $OUT/${TAG}_gemm_ref.txt 2>&1
# kernel traces (no CPU leg under the profiler)
PROF_TIMEOUT=500 tools/profile.sh ${TAG}/cfg1 --config 1 --steps 5 --warmup 2 --cpu-sample 0 --power-seconds 0 --product-steps 0 > /dev/null
PROF_TIMEOUT=600 tools/profile.sh ${TAG}/cfg4 --config 4 --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 --product-steps 0 > /dev/null
# counters (each group its own pass): the marcher's and the vision tower's, on configs[1] itself -> r06_pmc.json
if [ "${R06_PMC:-1}" = "1" ]; then
  PMC_PASS_TIMEOUT=300 tools/pmc.sh ${TAG}/pmc "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
      -- --config 1 --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 --product-steps 0 > $OUT/pmc.log 2>&1
  python tools/r06_pmc_report.py $OUT/pmc 2 > $OUT/${TAG}_pmc_report.txt 2>&1
fi
ls $OUT
