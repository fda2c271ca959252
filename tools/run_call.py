#!/usr/bin/env python3
"""bench.py --call without arguments on the command line (for tools/stats_any.sh / tools/pmc_any.sh, which run a script)."""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(root, "bench.py"), "--call", "--no-cpu-baseline", "--steps", "5", "--warmup", "2"] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
