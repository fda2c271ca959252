for G in 50000 500000 2000000 5000000 20000000 50000000; do echo "== G=$G"; G=$G timeout 300 python tools/sweep_walk.py 2>&1 | grep "threads= 64 waves/CU=\(16\|32\)"; done
