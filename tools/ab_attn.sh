#!/bin/bash
# Same-box A/B of library builds in ab/ on the streamed attention kernel alone: tools/attn_probe.py (2048 images x 12 heads per launch,
# ATTN_PROBE_LAYERS layers, two forward passes) under rocprofv3, the kernel's average / min / max duration per build.
#   tools/ab_attn.sh base.so attnrot1.so ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export ATTN_PROBE_LAYERS=${ATTN_PROBE_LAYERS:-6}
for r in $(seq 1 ${ROUNDS:-2}); do
for l in "$@"; do
  cp $ROOT/ab/$l $ROOT/dream2real_amd/libd2r.so
  rm -rf /tmp/attn_ab; mkdir -p /tmp/attn_ab
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/attn_ab -o r -- python $ROOT/tools/attn_probe.py > /tmp/attn_ab/log 2>&1)
  echo -n "$l  "
  python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/attn_ab/**/r_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_attention_s" in r["Name"]:
            print("%s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3), end="; ")
print()
PY
done
done
