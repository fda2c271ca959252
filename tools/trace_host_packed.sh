#!/bin/bash
# Runs on the GPU box: timeline (kernels, memory copies) of the packed host batch path (PACKED=1 tools/bench_host.py).
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/trace_host_packed"
rm -rf "$OUT"
export PACKED=1 SLABS=${SLABS:-32}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench_host.py > $ROOT/gpurun_out/trace_host_packed.log 2>&1
tail -3 $ROOT/gpurun_out/trace_host_packed.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
rows = []
for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'].split('(')[0][-40:]))
for f in glob.glob(os.path.join(sys.argv[1], '**', '*memory_copy_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', r.get('Name', '?'))))
rows.sort()
# the last matches call: the last 4 s... print a summary of the last 60 ms of activity by kind
end = rows[-1][1]
acc = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    acc[n][0] += e - s; acc[n][1] += 1
for n, (t, c) in sorted(acc.items(), key=lambda x: -x[1][0])[:25]:
    print(f"{n:46s} {t/1e6:10.3f} ms total {c:6d} x  {t/c/1e3:9.1f} us each")
PY
