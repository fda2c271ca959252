#!/bin/bash
# Runs on the GPU box (through gpurun): BASELINE config C5 at its real size - kbo call, 3 Gbp index, k = 63, one GPU's share of the reads
# (125 000 x 10 kbp) - once with every oracle leg (the bench line: sites of every read, 40 reads of the whole call, cpu baseline; builds
# the index and writes its cache file), then the profiler passes over the first pass's kernels from that cache file: kernel trace +
# stats, FETCH_SIZE, WRITE_SIZE, TCC hit / miss (one counter group per pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").  The cache file
# goes to /dev/shm when that has room (27 GB of path cover + the rows: a disk reads them for three minutes per pass).
# Usage: tools/profile_c5.sh <tag>      -> gpurun_out/<tag>_bench_c5.json, <tag>_c5_summary.json, <tag>_c5_kernel_stats.csv, <tag>_c5_phases.txt
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_c5
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SHM_FREE_GB=$(df -BG --output=avail /dev/shm 2>/dev/null | tail -1 | tr -dc 0-9)
CDIR=/tmp; if [ "${SHM_FREE_GB:-0}" -ge 80 ]; then CDIR=/dev/shm; fi
CACHE=$CDIR/C5_$$.kbohip
echo "cache file: $CACHE (/dev/shm free: ${SHM_FREE_GB:-?} GB)" > "$ROOT/gpurun_out/${TAG}_c5_phases.txt"
nproc >> "$ROOT/gpurun_out/${TAG}_c5_phases.txt"; free -g | head -2 >> "$ROOT/gpurun_out/${TAG}_c5_phases.txt"
T0=$(date +%s)
python3 "$ROOT/bench.py" --config C5 --steps 3 --warmup 1 --index-cache $CACHE > "$OUT/line.json" 2> "$OUT/line.err"
echo "bench line: $(( $(date +%s) - T0 )) s, rc $?" >> "$ROOT/gpurun_out/${TAG}_c5_phases.txt"
grep "bench C5" "$OUT/line.err" >> "$ROOT/gpurun_out/${TAG}_c5_phases.txt"
cp "$OUT/line.json" "$ROOT/gpurun_out/${TAG}_bench_c5.json"
ARGS="--config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-whole-call --index-cache $CACHE"
T0=$(date +%s)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
echo "stats pass: $(( $(date +%s) - T0 )) s" >> "$ROOT/gpurun_out/${TAG}_c5_phases.txt"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  T0=$(date +%s)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$N.log" 2>&1
  echo "pmc pass $N: $(( $(date +%s) - T0 )) s" >> "$ROOT/gpurun_out/${TAG}_c5_phases.txt"
done
rm -f $CACHE
python3 "$ROOT/tools/summarize_prof.py" "$OUT" > "$ROOT/gpurun_out/${TAG}_c5_summary.json" 2> "$OUT/summ.err"
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); cp "$f" "$ROOT/gpurun_out/${TAG}_c5_kernel_stats.csv"
# (the traces are large: only the summaries travel back)
rm -rf "$OUT/stats" "$OUT"/pmc_*/
cat "$ROOT/gpurun_out/${TAG}_c5_phases.txt"
