#!/bin/bash
# Runs on the GPU box: the default bench line (no extras) under several settings of one environment variable, twice each.
# Usage: tools/ab_env.sh VAR value1 value2 ...   ("-" = unset)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  python3 $ROOT/bench.py --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d.get('one_batch_at_a_time') or {}
print('$VAR=$v', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel(shared)', d['kernels_ms'].get('map_reads_kernel'), 'alone: step', o.get('ms_per_step'), 'kernel', o.get('map_reads_kernel_ms'), 'second', o.get('second_pass_ms'))
"
done
done
