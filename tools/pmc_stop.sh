#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp; export TMPDIR=/tmp
for S in ${STOPS:-0}; do
  OUT="$ROOT/gpurun_out/pmc_stop_$S"; rm -rf "$OUT"; mkdir -p "$OUT"
  KBO_MAP_STOP=$S CHECK=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT" -- python3 "$ROOT/tools/exp_map.py" > "$OUT.log" 2>&1
  python3 - "$OUT" $S <<'PY'
import csv, glob, sys, os, collections
acc = collections.defaultdict(float); cnt = collections.defaultdict(set); dur=[]
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'map_reads' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']].add(r['Dispatch_Id'])
for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'map_reads' in r['Kernel_Name']: dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print("stop", sys.argv[2], "kernel us %.1f" % (sum(dur)/len(dur)/1e3 if dur else 0), {k: round(acc[k]/len(cnt[k])/15625) for k in sorted(acc)})
PY
done
