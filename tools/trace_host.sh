#!/bin/bash
# Runs on the GPU box: timeline (HIP API calls, kernels, memory copies) of tools/bench_host.py, for
# checking how well the slabs of the host batch path overlap.  No counters are collected.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
GRAFT_REPO_ROOT="$ROOT"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/trace_host"
rm -rf "$OUT"
SLABS=${SLABS:-32} FIND=${FIND:-} rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_host.py > $GRAFT_REPO_ROOT/gpurun_out/trace_host.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/trace_host.log
find $OUT -name "*.csv" | head
