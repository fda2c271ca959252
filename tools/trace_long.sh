#!/bin/bash
# Runs on the GPU box: time line of the kernels of tools/exp_long.py (rocprofv3 --kernel-trace): the last batches of every variant,
# start / end of each kernel relative to the first of them, which queue it ran on.  Usage: tools/trace_long.sh <tag> [exp_long.py arguments]
TAG=${1:-long}; shift || true
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/trace_$TAG"
rm -rf "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/tools/exp_long.py" --no-check "$@" > "$OUT.log" 2>&1
grep -v "^W2026\|amdgpu.ids" "$OUT.log" | tail -8
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "$ROOT/gpurun_out/trace_${TAG}_timeline.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the variants run one after the other: split where map_long_kernel launches are more than 50 ms apart
main = [i for i, r in enumerate(rows) if 'map_long_kernel' in r['Kernel_Name']]
groups, cur = [], [main[0]]
for a, b in zip(main, main[1:]):
    if int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']) > 50_000_000:
        groups.append(cur); cur = []
    cur.append(b)
groups.append(cur)
for g in groups:
    if len(g) < 6: continue
    lo, hi = g[-5], g[-1]
    t0 = int(rows[lo]['Start_Timestamp'])
    t_end = int(rows[hi]['End_Timestamp']) + 600_000
    print("---- variant: %d launches of the kernel; the last four batches" % len(g))
    for r in rows[lo:]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > t_end: break
        import re as _re
        name = (_re.findall(r'([A-Za-z_0-9]+_kernel|__amd_rocclr_[A-Za-z]+)', r['Kernel_Name']) or [r['Kernel_Name'][:34]])[0][:34]
        print("%9.1f .. %9.1f us (%7.1f)  q%-3s %s  grid %s wg %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name.ljust(34),
                                                                  r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))))
PY
tail -70 "$ROOT/gpurun_out/trace_${TAG}_timeline.txt"
