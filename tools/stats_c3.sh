#!/bin/bash
# Runs on the GPU box: per-kernel time table (rocprofv3 --kernel-trace --stats) of bench.py --config C3, the kernels of the timed steps
# only (everything that builds the device copy left out of the table).  Usage: tools/stats_c3.sh <tag> [bench args]
TAG=${1:-c3}; shift || true
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/stats_$TAG"
rm -rf "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --config C3 --steps 6 --warmup 2 --no-cpu-baseline "$@" > "$OUT.log" 2>&1
tail -1 "$OUT.log" | cut -c1-400
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/gpurun_out/stats_${TAG}_kernel_stats.csv"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(x in n for x in ('dtab_', 'seed_pos', 'pack_text', 'FillFunctor', 'path_', 'cover')): continue
    print(n[:80].ljust(82), r['Calls'].rjust(5), "avg us %9.1f" % (float(r['AverageNs']) / 1e3), "total ms %8.2f" % (float(r['TotalDurationNs']) / 1e6))
PY
