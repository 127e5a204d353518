#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of the default bench command.
#   tools/profile.sh <tag> [bench args...]
# Writes gpurun_out/<tag>/ (csv) and gpurun_out/<tag>_kernel_stats.md; copy the .md/.csv you
# want judged into profiles/.
set -e
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $ROOT/bench.py "$@" > $OUT/bench.log 2>&1 || true
rm -f $OUT/*.db $OUT/r_kernel_trace.csv
python - <<PY
import csv, sys
rows = list(csv.DictReader(open("$OUT/r_kernel_stats.csv")))
with open("$ROOT/gpurun_out/${TAG}_kernel_stats.md", "w") as f:
    f.write("# timeout -k 10 ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats -- python bench.py $*\n\n")
    f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:24]:
        f.write(f"| {r['Name'][:60]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    f.write("\nbench line under the profiler:\n\n\`\`\`\n" + [l for l in open("$OUT/bench.log").read().splitlines() if l.startswith('{"metric"')][-1][:3000] + "\n\`\`\`\n")
print(open("$ROOT/gpurun_out/${TAG}_kernel_stats.md").read()[:3500])
PY
