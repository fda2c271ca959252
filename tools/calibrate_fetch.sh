#!/bin/bash
# Calibrates rocprofv3 FETCH_SIZE for this path's access pattern (random 16-byte gathers):
# tools/ubench/gather issues a known number of lane loads from tables of known size.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/calib; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $ROOT/tools/ubench/gather > $OUT/gather.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
rows=[]
for f in glob.glob(os.path.join(sys.argv[1],'fetch','**','*counter_collection.csv'),recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='FETCH_SIZE': rows.append((int(r['Dispatch_Id']), r['Kernel_Name'][:40], int(r['Grid_Size']), float(r['Counter_Value'])))
rows.sort()
for r in rows: print(r)
PY
grep "16B x1 " $OUT/gather.log
