#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + FETCH_SIZE / WRITE_SIZE / L2 hit passes of bench.py at one of its larger
# presets (C3: kbo find, 100 Mbp, 10 M reads - SURVEY.md 8(d)'s designated roofline run; C4: kbo map, 250 Mbp, 100 M reads), the index
# built once and loaded from a cache file by every pass, then the bench line itself (no profiler, oracle legs on).
# C5: kbo call, 3 Gbp, k = 63, 125 k reads of 10 kbp - the passes see the first pass only (--no-whole-call); the whole call's kernels:
# tools/stats_any.sh tools/bench_call.py.
# Usage: tools/profile_cfg.sh <tag> <C3|C4|C5> [bench args...]
set -u
TAG=${1:-r05_c3}; CFG=${2:-C3}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CACHE=/tmp/$CFG.kbohip
ARGS="--config $CFG --steps 3 --warmup 1 --no-cpu-baseline --index-cache $CACHE $*"
if [ "$CFG" = C5 ]; then ARGS="$ARGS --no-whole-call"; fi   # (the profiler passes: the first pass's kernels only)
python3 "$ROOT/bench.py" --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --index-cache $CACHE > /dev/null 2>&1   # (writes the cache)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$N.log" 2>&1
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
tail -40 "$OUT/summary.txt"
cp "$OUT/summary.txt" "$ROOT/gpurun_out/${TAG}_summary.json"
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); cp "$f" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
