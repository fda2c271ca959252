#!/bin/bash
# Runs on the GPU box (through gpurun), once the code is frozen: for C2, C3, C4 the rocprofv3 passes (tools/profile_bench.sh /
# tools/profile_cfg.sh), their fold into profiles/traffic_latest.json (tools/traffic_from_summary.py: the build's fingerprint goes in),
# then the bench line of the same build, which quotes that traffic.  Everything the repository keeps lands in gpurun_out/final/
# under the names profiles/ uses.  Usage: tools/final_profiles.sh <round tag, e.g. r05> [C2 C3 C4]
set -u
R=${1:-r05}; shift || true
CFGS=${*:-C2 C3 C4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
F=$ROOT/gpurun_out/final
mkdir -p "$F"
cd "$ROOT"
for CFG in $CFGS; do
  c=$(echo $CFG | tr A-Z a-z)
  if [ "$CFG" = "C2" ]; then
    tools/profile_bench.sh ${R}_c2 > "$F/prof_${R}_c2.log" 2>&1
    OUT=gpurun_out/prof_${R}_c2
    cp $OUT/summary.txt "$F/${R}_c2_summary.json"
    cp "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" "$F/${R}_c2_kernel_stats.csv"
    cp "$F/${R}_c2_summary.json" profiles/${R}_c2_summary.json
    python3 tools/traffic_from_summary.py profiles/${R}_c2_summary.json "5000000x1000000x150x0.01:map" "profiles/${R}_c2_summary.json (tools/profile_bench.sh ${R}_c2: rocprofv3 --pmc passes of bench.py; the counter passes serialise the kernels)"
    python3 bench.py > "$F/${R}_bench_c2.json" 2> "$F/${R}_bench_c2.err"
    python3 bench.py --steps 20 --warmup 5 --no-extras > "$F/${R}_bench_c2_20steps.json" 2>/dev/null
  else
    tools/profile_cfg.sh ${R}_$c $CFG > "$F/prof_${R}_$c.log" 2>&1
    cp gpurun_out/${R}_${c}_summary.json "$F/"; cp gpurun_out/${R}_${c}_kernel_stats.csv "$F/"
    cp "$F/${R}_${c}_summary.json" profiles/${R}_${c}_summary.json
    if [ "$CFG" = "C3" ]; then KEY="100000000x10000000x150x0.01:map"; N=2; else KEY="250000000x100000000x150x0.01:map"; N=13; fi
    python3 tools/traffic_from_summary.py profiles/${R}_${c}_summary.json "$KEY" "profiles/${R}_${c}_summary.json (tools/profile_cfg.sh ${R}_$c $CFG: rocprofv3 --pmc passes of bench.py --config $CFG; the counter passes serialise the kernels)" $N
    suffix=""; [ "$CFG" = "C4" ] && suffix="_n1"
    python3 bench.py --config $CFG --index-cache /tmp/$CFG.kbohip > "$F/${R}_bench_${c}${suffix}.json" 2> "$F/${R}_bench_${c}${suffix}.err"
  fi
  cp profiles/traffic_latest.json "$F/traffic_latest.json"
  for f in "$F"/${R}_bench_${c}*.json; do echo "$f: $(cut -c1-170 $f | tail -1)"; done
done
