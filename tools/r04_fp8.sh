#!/bin/bash
# Run on the GPU box (through gpurun): the fp8 ViT (library option vit_fp8, bench.py --vit-fp8) beside the bf16 one on the same box —
# BASELINE.json configs[4] as worded ("fp16 render + fp8 MFMA ViT"), configs[1] and the reference's shapes for comparison — and a
# kernel trace of the configs[4] slice.  Outputs under gpurun_out/r04_fp8/; copy what is to be judged into profiles/.
TAG=${1:-r04_fp8}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
line() { grep '^{"metric"' | tail -1; }
python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 8            2>$OUT/cfg4_bf16.err | line > $OUT/${TAG}_cfg4_bf16_bench.json
python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 8 --vit-fp8  2>$OUT/cfg4_fp8.err  | line > $OUT/${TAG}_cfg4_fp8_bench.json
python bench.py --config 1 --steps 5 --warmup 2 --cpu-sample 32 --vit-fp8 2>$OUT/cfg1_fp8.err  | line > $OUT/${TAG}_cfg1_fp8_bench.json
python bench.py --clip vit_l14_336 --width 336 --height 336 --poses-per-gpu 1024 --steps 3 --warmup 1 --cpu-sample 4 --vit-fp8 2>$OUT/ref_fp8.err | line > $OUT/${TAG}_refshapes_fp8_bench.json
for f in $OUT/${TAG}_*_bench.json; do python - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], d["value"], "cand/s", d["ms_per_step"], d["device_ms_per_step"], "vit TF", d["roofline_vit"]["achieved"], "of", d["roofline_vit"]["peak"],
      "parity", d.get("parity_vs_oracle", {}).get("max_cosine_err"), "power", (d.get("power") or {}).get("avg_w"))
PY
done
PROF_TIMEOUT=600 tools/profile.sh ${TAG}/cfg4_fp8 --config 4 --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 --vit-fp8 > /dev/null
ls $OUT
