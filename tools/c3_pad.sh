#!/bin/bash
# Runs on the GPU box: C3 (kbo find, 100 Mbp, 10 M reads) with fewer resident waves of map_reads_kernel (KBO_MAP_LDS_PAD bytes of
# LDS per wave more than it needs): what occupancy is worth at that index size.  Usage: tools/c3_pad.sh [pads...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
python3 $ROOT/bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --index-cache /tmp/c3.kbohip > /dev/null 2>&1
for pad in "$@"; do
  KBO_MAP_LDS_PAD=$pad python3 $ROOT/bench.py --config C3 --steps 6 --warmup 2 --no-cpu-baseline --index-cache /tmp/c3.kbohip 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pad', $pad, 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernels', d['kernels_ms'], 'serial', (d.get('one_batch_at_a_time') or {}).get('ms_per_step'), (d.get('one_batch_at_a_time') or {}).get('map_reads_kernel_ms'))
"
done
