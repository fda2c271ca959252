#!/bin/bash
# Runs on the GPU box: bench.py --config C3 with a filter of F bases in front of the depth table (KBO_DEPTH_FILTER; 0 = none), index from one cache file
mkdir -p gpurun_out/c3f
python bench.py --config C3 --steps 1 --warmup 0 --no-extras --no-cpu-baseline --index-cache /tmp/C3.kbohip > /dev/null 2>&1
for f in "$@"; do
  KBO_DEPTH_FILTER=$f python bench.py --config C3 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --index-cache /tmp/C3.kbohip > gpurun_out/c3f/f$f.json 2> gpurun_out/c3f/f$f.err
  python - gpurun_out/c3f/f$f.json $f <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); ro = b["roofline"]
print("filter", sys.argv[2], b["value"], b["ms_per_step"], "kernel", ro.get("kernel_ms"), "redo", ro.get("redo_pass_ms"), "alone", (ro.get("alone") or {}).get("kernel_ms"), "one", (b.get("one_batch_at_a_time") or {}).get("value"))
PY
done
