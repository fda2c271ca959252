#!/bin/bash
# Runs on the GPU box: map_reads_kernel's duration (kernels serialised by the counter pass) and instruction counters per wave for
# several settings of an experiment switch, back to back on one box.  Usage: VAR=KBO_MAP_X VALS="0 1" tools/pmc_ab.sh [reps]
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp; export TMPDIR=/tmp
for rep in $(seq 1 ${1:-2}); do
for V in $VALS; do
  OUT="$ROOT/gpurun_out/pmc_ab_$V"; rm -rf "$OUT"; mkdir -p "$OUT"
  env $VAR=$V CHECK=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT" -- python3 "$ROOT/tools/exp_map.py" > "$OUT.log" 2>&1
  python3 - "$OUT" "$VAR=$V" <<'PY'
import csv, glob, sys, os, collections
acc = collections.defaultdict(float); cnt = collections.defaultdict(set); dur=[]
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'map_reads' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']].add(r['Dispatch_Id'])
for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'map_reads' in r['Kernel_Name']: dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print(sys.argv[2], "kernel us %.1f" % (sum(dur)/len(dur)/1e3 if dur else 0), {k: round(acc[k]/len(cnt[k])/15625) for k in sorted(acc)})
PY
done
done
