# experiments on kbo_map_batch_dev at C2: anchors on the 15-base table (seeds + present windows), bases per piece of the redo pass
for A in 0 1; do for PC in 32 16 8; do echo "ANCHORS=$A PIECE=$PC: $(CHECK=0 KBO_DEPTH_TABLE_ANCHORS=$A KBO_REDO_PIECE=$PC python tools/exp_map.py 2>&1 | tail -1)"; done; done
