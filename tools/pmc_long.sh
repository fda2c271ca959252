#!/bin/bash
# Runs on the GPU box: PMC counters per kernel (separate passes, --kernel-trace only) of tools/exp_long.py.
# Usage: tools/pmc_long.sh <tag> [exp_long.py arguments]
# (KBO_LONG_X - phases of map_long_kernel left out - needs a library built with -DKBO_LONG_EXPERIMENTS: `make -C kbo_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -DKBO_LONG_EXPERIMENTS"` after touching long_kernels.hip)
TAG=${1:-x}; shift || true
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/pmc_long_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
# PMC_SQ_ONLY=1: the instruction counts only
if [ -n "$PMC_SQ_ONLY" ]; then
  GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY")
else
  GROUPS_=("TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
           "GRBM_GUI_ACTIVE")
fi
for C in "${GROUPS_[@]}"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$N" -- python3 "$ROOT/tools/exp_long.py" --no-check --steps 5 "$@" > "$OUT/$N.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections, re
acc = collections.defaultdict(float); cnt = collections.defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(map_long\w+|long_\w+|ms_walk\w*)', r['Kernel_Name'])
        if not m: continue
        name = m.group(1)
        key = (name, r['Counter_Name'])
        acc[key] += float(r['Counter_Value']); cnt[key].add(r['Dispatch_Id'])
for k in sorted(acc): print(f"{k[0]:28s} {k[1]:30s} {acc[k]/len(cnt[k]):16.0f}  (avg over {len(cnt[k])} launches)")
PY
