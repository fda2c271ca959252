#!/bin/bash
# Runs on the GPU box: bench.py's line under several settings of KBO_TAIL_CUS (compute units the second passes' streams of kbo_map_stream_*
# are confined to), back to back on one box.  Usage: tools/ab_tail_cus.sh "<bench args>" <values...>
ARGS="$1"; shift
mkdir -p gpurun_out/tailcus
for v in "$@"; do
  KBO_TAIL_CUS=$v python bench.py $ARGS --no-extras --no-cpu-baseline > gpurun_out/tailcus/cus_$v.json 2> gpurun_out/tailcus/cus_$v.err
  python - gpurun_out/tailcus/cus_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); ro = d["roofline"]
print("KBO_TAIL_CUS", sys.argv[2], d["value"], "Mbp/s", d["ms_per_step"], "ms/step kernel", ro.get("kernel_ms"), "redo", ro.get("redo_pass_ms"))
PY
done
