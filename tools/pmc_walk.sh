#!/bin/bash
# PMC counters for the walk kernel only (tools/sweep_walk.py, one configuration).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_walk_${1:-x}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  ONLY=1 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$N" -- python3 "$ROOT/tools/sweep_walk.py" > "$OUT/$N.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
acc=collections.defaultdict(float); cnt=collections.defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1],'**','*counter_collection.csv'),recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ms_walk' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']].add(r['Dispatch_Id'])
for k in sorted(acc): print(f"{k:34s} {acc[k]/len(cnt[k]):16.0f}  (avg over {len(cnt[k])} launches)")
PY
