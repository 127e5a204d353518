#!/bin/bash
# Run on the GPU box (through gpurun): one bench line + one kernel-trace summary per BASELINE.json config, the
# reference's own shapes, and the FETCH_SIZE / WRITE_SIZE passes on the round's k_march.  Outputs under gpurun_out/r03/;
# copy what is to be judged into profiles/.
#   tools/r03_configs.sh [tag]
TAG=${1:-r03}
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
# HBM-side traffic of k_march on the default workload: FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md)
PMC_PASS_TIMEOUT=300 tools/pmc.sh ${TAG}/pmc_traffic "FETCH_SIZE" "WRITE_SIZE" -- --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 > $OUT/pmc_traffic.log 2>&1
tail -30 $OUT/pmc_traffic.log
