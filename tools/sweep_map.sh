# experiments on map_reads_kernel (C2): characters through LDS or straight to memory, resident waves (LDS padding)
for I in 1 0; do for P in 0 2000 4500 8000; do echo "INLDS=$I PAD=$P: $(CHECK=${CHECK:-0} KBO_MAP_INLDS=$I KBO_MAP_LDS_PAD=$P python tools/exp_map.py 2>&1 | tail -1)"; done; done
