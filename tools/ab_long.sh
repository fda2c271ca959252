# timing only: phases of map_long_kernel left out (KBO_LONG_X), the kernel alone (rocprofv3 stats)
for x in 16 32 1 6 4 0; do echo "X=$x"; KBO_LONG_X=$x tools/stats_long.sh ab$x --variants ${1:-1pct} --steps 10 2>&1 | grep "map_long_kernel"; done
for ppw in 4 16; do echo "PPW=$ppw"; KBO_LONG_PPW=$ppw tools/stats_long.sh abp$ppw --variants ${1:-1pct} --steps 10 2>&1 | grep "map_long_kernel"; done
