#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes of
# the default bench.py workload.  Usage: tools/profile_bench.sh <tag> [bench args...]
# Results land in gpurun_out/prof_<tag>/ ; copy the summaries to profiles/ afterwards.
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline $*"
# (the stats pass with the default number of steps: the timed region's launches - two pipelines sharing the device - are then
# nine in ten of the kernel's dispatches, and its average duration is the one bench.py measures live)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline $* > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$N.log" 2>&1
done
find "$OUT" -name "*.csv" | head -50
python3 "$ROOT/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
