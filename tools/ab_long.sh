# timing only: occupancy variants and pieces per wave
python -m pytest tests/test_gpu_map_long.py -x -q -m gpu 2>&1 | tail -3
for w in 4 5 6 8; do for ppw in 8; do echo "WPE=$w PPW=$ppw"; KBO_LONG_PPW=$ppw KBO_HIP_LIB=$PWD/kbo_amd/libkbo_hip_w$w.so python tools/exp_long.py --variants 1pct,clean --no-check --steps 10 2>&1 | grep "Gbp"; done; done
for ppw in 1 4 16; do echo "WPE=5 PPW=$ppw"; KBO_LONG_PPW=$ppw KBO_HIP_LIB=$PWD/kbo_amd/libkbo_hip_w5.so python tools/exp_long.py --variants 1pct,clean --no-check --steps 10 2>&1 | grep "Gbp"; done
