#!/usr/bin/env python3
"""How many reads map_reads_kernel leaves to its second pass, by kind of read, on the C2 index: python tools/dbg_witness.py
(KBO_MAP_X=32: present windows flag their read as before the rule that asks the windows around them)"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch

    import kbo_amd
    from kbo_amd import batch, synth
    dev0 = torch.device("cuda:0")
    g = synth.genome(5_000_000)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
    sbwt.to_device(-1)
    other = synth.genome(3_000_000, seed=99)
    n = 200_000
    for name, (concat, offsets) in (("1% subs", synth.reads(g, n, 150, 0.01, seed=1)), ("5% subs", synth.reads(g, n, 150, 0.05, seed=2)),
                                    ("unrelated", synth.reads(other, n, 150, 0.0, seed=3))):
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=True, want_ms=False)
        dev.run()
        torch.cuda.synchronize()
        fl = dev.plan_flags()
        print("%-10s fused %s flagged %d of %d (%.3f %%)" % (name, dev.fused, int(np.count_nonzero(fl)), n, 100.0 * np.count_nonzero(fl) / n), flush=True)


if __name__ == "__main__":
    main()
