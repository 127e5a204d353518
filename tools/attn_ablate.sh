#!/bin/bash
# Run on the GPU box: rebuild libd2r with each attention ablation mask (k_attention_s: 16 no DMA requests, 32 no softmax,
# 64 no PV, 128 no S MFMAs, 256 no per-tile barriers) and report the kernel's average duration (2048 images x 12 heads
# per launch).  The Makefile rebuilds every object when the flags change (_build/flags.stamp).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for A in ${MASKS:-0 16 32 64 128 256 224 272 96 480}; do
  make -C $ROOT/dream2real_amd/csrc -j3 ATTN_ABLATE=$A 2>&1 | grep -E " error"
  rm -rf /tmp/attn_abl; mkdir -p /tmp/attn_abl
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/attn_abl -o r -- python $ROOT/tools/attn_probe.py > /tmp/attn_abl/log 2>&1)
  echo -n "attn_ablate=$A  "
  python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/attn_abl/**/r_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_attention" in r["Name"]:
            print("k_attention calls %s avg %.1f us max %.1f us" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
make -C $ROOT/dream2real_amd/csrc -j3 ATTN_ABLATE=0 2>&1 | grep -E " error"
