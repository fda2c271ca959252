python -m pytest tests/test_gpu_map_long.py -x -q -m gpu 2>&1 | tail -2
echo "WPE=5"; tools/stats_long.sh a5 --variants ont,1pct,big --steps 10 2>&1 | grep "Gbp\|map_long"
echo "WPE=4"; KBO_HIP_LIB=$PWD/kbo_amd/libkbo_hip_w4.so tools/stats_long.sh a4 --variants ont,1pct,big --steps 10 2>&1 | grep "Gbp\|map_long"
