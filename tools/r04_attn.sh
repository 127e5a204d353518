#!/bin/bash
# Round 4: the attention kernel with leftover query tiles on small workgroups (257 tokens: 8 + 1 tiles; 577: 16 + 3), kernel
# durations from rocprofv3 --kernel-trace --stats (three layers, full width; tools/attn_probe2.py).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
run() {   # clip n opts...
  rm -rf /tmp/attn_r04; mkdir -p /tmp/attn_r04
  (cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/attn_r04 -o r -- python $ROOT/tools/attn_probe2.py "$@" > /tmp/attn_r04/log 2>&1)
  grep -E "wall per forward" /tmp/attn_r04/log
  python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/attn_r04/**/r_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_attention_s" in r["Name"]:
            print("   %-40s calls %s avg %.1f us max %.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
for rep in 1 2; do
  run vit_l14 2048 attn_rem=0
  run vit_l14 2048 attn_rem=1
  run vit_l14_336 512 attn_rem=0
  run vit_l14_336 512 attn_rem=4
  run vit_tiny 4096 attn_rem=0
  run vit_tiny 4096 attn_rem=1
done
