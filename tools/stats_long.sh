#!/bin/bash
# Runs on the GPU box: per-kernel time table (rocprofv3 --kernel-trace --stats) of tools/exp_long.py.
# Usage: tools/stats_long.sh <tag> [exp_long.py arguments]
TAG=${1:-long}; shift || true
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/stats_$TAG"
rm -rf "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/tools/exp_long.py" --no-check "$@" > "$OUT.log" 2>&1
grep -v "^W2026\|amdgpu.ids" "$OUT.log" | tail -8
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/gpurun_out/stats_${TAG}_kernel_stats.csv"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:70].ljust(72), r['Calls'].rjust(5), "avg us %9.1f" % (float(r['AverageNs']) / 1e3), "total ms %8.2f" % (float(r['TotalDurationNs']) / 1e6))
PY
