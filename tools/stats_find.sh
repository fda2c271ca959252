cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stats_find
rm -rf $OUT
FIND=1 SLABS=32 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_host.py > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -d, -f1-6 $f | head -14
