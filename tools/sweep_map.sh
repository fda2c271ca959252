for S in 1 0; do for P in 0 3000 6500 10000; do echo "SEED3=$S PAD=$P: $(CHECK=0 KBO_MAP_SEED3=$S KBO_MAP_LDS_PAD=$P python tools/exp_map.py 2>&1 | tail -1)"; done; done
