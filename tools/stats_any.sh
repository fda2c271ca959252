#!/bin/bash
# Runs on the GPU box: per-kernel time table (rocprofv3 --kernel-trace --stats) of any of the tools.
# Usage: tools/stats_any.sh <python script> [env assignments are inherited]
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/stats_any"
rm -rf "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/$1" > "$OUT.log" 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:70].ljust(72), r['Calls'].rjust(5), "avg us %9.1f" % (float(r['AverageNs']) / 1e3), "total ms %8.2f" % (float(r['TotalDurationNs']) / 1e6))
PY
