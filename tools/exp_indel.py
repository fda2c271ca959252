"""kbo_map_batch_dev on the C2 index with reads that carry one insertion or deletion at a chosen place: how many the kernel leaves
to the plain walk, by place and kind (counters of the kernel)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, ROOT)
import kbo_amd
from kbo_amd import batch, synth
g = synth.genome(5_000_000)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
L = kbo_amd.lib()
dev0 = torch.device("cuda:0")
rng = np.random.default_rng(3)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
N, LEN = 200_000, 150
for kind in ("del", "ins"):
    for lo, hi in ((10, 20), (20, 40), (40, 80), (80, 115), (115, 132), (132, 141)):
        start = rng.integers(0, len(g) - LEN - 8, N)
        pos = rng.integers(lo, hi, N)
        size = rng.integers(1, 4, N)
        i = np.arange(LEN)[None, :]
        if kind == "del":
            reads = g[start[:, None] + i + np.where(i >= pos[:, None], size[:, None], 0)]
        else:
            reads = g[start[:, None] + i - np.where(i >= pos[:, None], np.minimum(size[:, None], i - pos[:, None]), 0)]
            new = (i >= pos[:, None]) & (i < (pos + size)[:, None])
            reads = np.where(new, acgt[rng.integers(0, 4, (N, LEN))], reads)
        concat = np.ascontiguousarray(reads.reshape(-1))
        offsets = np.arange(N + 1, dtype=np.uint64) * np.uint64(LEN)
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=True, want_ms=False)
        L.kbo_set_plan(1, 0, 0)
        L.kbo_set_plan_stats(1)
        dev.run(); torch.cuda.synchronize()
        st = dev.plan_stats()
        L.kbo_set_plan_stats(0)
        print(kind, "at", lo, "..", hi, "flagged %.1f %%" % (100.0 * st["tab_flagged"] / N), "mismatches per read %.2f" % (st["mismatches"] / N), "seed look-ups per read %.2f" % (st["seed_lookups"] / N), flush=True)
        del dev
