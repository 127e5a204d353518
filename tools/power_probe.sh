#!/bin/bash
# Run on the GPU box: sample socket power and shader clock (rocm-smi) while bench.py runs, to see whether the
# ViT GEMMs run against the power limit (the shader clock sits near 1.6 GHz under that load).
python bench.py --steps ${STEPS:-100} --warmup 2 --cpu-sample 0 "$@" > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
: > gpurun_out/power_samples.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | sed -e 's/GPU\[0\]\t\t: //' | tr '\n' ' ' >> gpurun_out/power_samples.txt
  echo >> gpurun_out/power_samples.txt
done
wait $BP
echo "samples: $(wc -l < gpurun_out/power_samples.txt); the 12 with the highest power:"
sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz).*Power (W): \([0-9.]*\).*/\2 W  sclk \1 MHz/' gpurun_out/power_samples.txt | sort -rn | head -12
python - <<'P'
import json
d = json.loads(open("gpurun_out/power_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["device_ms_per_step"])
P
rocm-smi --showmaxpower 2>/dev/null | grep -i "max"
