for x in 0 1 2 4 6 16 32; do echo "X=$x"; KBO_LONG_X=$x python tools/exp_long.py --variants 1pct,clean --no-check --steps 10 2>&1 | grep "Gbp"; done
