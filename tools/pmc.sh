#!/bin/bash
# PMC passes (each its own run, --kernel-trace only) over a short bench; summarises per kernel.
#   tools/pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ... -- [bench args]
set -e
TAG=$1; shift
PASSES=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do PASSES+=("$1"); shift; done
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "${PASSES[@]}"; do
  # hard limit per pass: a counter set that does not fit the PMC slots (SQ 8, TCC 4 with FETCH_SIZE
  # costing 3, GRBM 2) makes rocprofv3 abort and then wait forever on the unfinished dispatch
  timeout -k 10 ${PMC_PASS_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o r -- python $ROOT/bench.py "$@" > $OUT/p$i.log 2>&1 || echo "pass $i ($P) failed or timed out"
  i=$((i+1))
done
python - <<PY
import csv, glob, collections, os
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/r_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:40]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$ROOT/gpurun_out/${TAG}_pmc.md", "w") as fo:
    fo.write("# rocprofv3 --pmc (separate passes), per-dispatch averages\n\n")
    for name, cs in sorted(agg.items(), key=lambda kv: -sum(len(v) for v in kv[1].values())):
        if not any(k in name for k in ("k_march", "k_gemm", "k_attention", "k_preprocess", "k_raygen", "k_head", "k_layernorm")):
            continue
        fo.write(f"## {name}\n\n| counter | dispatches | mean | \n|---|---|---|\n")
        for c, v in sorted(cs.items()):
            fo.write(f"| {c} | {len(v)} | {sum(v)/len(v):.4g} |\n")
        fo.write("\n")
print(open("$ROOT/gpurun_out/${TAG}_pmc.md").read()[:6000])
PY
find $OUT -name "*.csv" -size +2M -delete
