"""flagged reads with one insertion, a few of them in detail"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, ROOT)
import kbo_amd
from kbo_amd import batch, synth
from oracle import binding as ora
g = synth.genome(5_000_000)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
rows, Carr, lcs = sbwt.export_parts()
oi = ora.Index.from_parts(31, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
dev0 = torch.device("cuda:0")
rng = np.random.default_rng(3)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
N, LEN = 20_000, 150
start = rng.integers(0, len(g) - LEN - 8, N)
pos = rng.integers(40, 80, N)
size = rng.integers(1, 4, N)
i = np.arange(LEN)[None, :]
reads = g[start[:, None] + i - np.where(i >= pos[:, None], np.minimum(size[:, None], i - pos[:, None]), 0)]
new = (i >= pos[:, None]) & (i < (pos + size)[:, None])
reads = np.where(new, acgt[rng.integers(0, 4, (N, LEN))], reads)
concat = np.ascontiguousarray(reads.reshape(-1)); offsets = np.arange(N + 1, dtype=np.uint64) * np.uint64(LEN)
dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=False, want_ms=False)
dev.run(); torch.cuda.synchronize()
fl = dev.plan_flags()
print("flagged", int((fl != 0).sum()), "of", N, "by size", [int(((fl != 0) & (size == d)).sum()) for d in (1, 2, 3)], [int((size == d).sum()) for d in (1, 2, 3)])
shown = 0
for r in np.flatnonzero(fl != 0)[:12]:
    rd = reads[r]
    d, lo, hi = oi.matching_statistics(rd.tobytes())
    p, sz = int(pos[r]), int(size[r])
    print("read", r, "ins at", p, "size", sz, "inserted", rd[p:p + sz].tobytes(), "genome before/after", g[start[r] + p - 3:start[r] + p + 3].tobytes())
    print("  ms", [int(v) for v in d[p - 4:p + 36]])
