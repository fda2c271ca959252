#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + FETCH_SIZE / WRITE_SIZE / L2 hit passes of
# bench.py at the C3 shape (100 Mbp index, 10 M x 150 bp reads) — SURVEY.md §8(d)'s designated roofline run.
# Usage: tools/profile_c3.sh <tag> [bench args...]
set -u
TAG=${1:-r01_c3}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--config C3 --steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$N.log" 2>&1
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
tail -2 "$OUT/stats.log"
