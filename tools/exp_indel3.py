"""which reads with one insertion / deletion the one kernel still leaves to the plain walk: by kind, size and position"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, ROOT)
import kbo_amd
from kbo_amd import batch, synth
g = synth.genome(5_000_000)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
dev0 = torch.device("cuda:0")
rng = np.random.default_rng(3)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
N, LEN = 200_000, 150
SUB = float(os.environ.get("SUB", "0.01"))
start = rng.integers(8, len(g) - LEN - 8, N)
pos = rng.integers(1, LEN - 1, N)
size = rng.integers(1, 4, N)
ins = rng.random(N) < 0.5
i = np.arange(LEN)[None, :]
shift = np.where(i >= pos[:, None], np.where(ins[:, None], -np.minimum(size[:, None], i - pos[:, None]), size[:, None]), 0)
reads = g[start[:, None] + i + shift]
new = ins[:, None] & (i >= pos[:, None]) & (i < (pos + size)[:, None])
reads = np.where(new, acgt[rng.integers(0, 4, (N, LEN))], reads)
hit = rng.random((N, LEN)) < SUB
nsub = hit.sum(1)
reads = np.where(hit, acgt[(np.searchsorted(acgt, reads) + rng.integers(1, 4, (N, LEN))) % 4], reads)
concat = np.ascontiguousarray(reads.reshape(-1)); offsets = np.arange(N + 1, dtype=np.uint64) * np.uint64(LEN)
dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=False, want_ms=False)
dev.run(); torch.cuda.synchronize()
fl = dev.plan_flags() != 0
print("flagged %d of %d (%.1f %%)" % (fl.sum(), N, 100.0 * fl.mean()))
for name, sel in (("insertions", ins), ("deletions", ~ins)):
    print(name, "by size", ["%.1f%%" % (100.0 * fl[sel & (size == d)].mean()) for d in (1, 2, 3)])
print("by position (tens):", " ".join("%d:%.0f" % (b, 100.0 * fl[(pos // 10) == b // 10].mean()) for b in range(0, 150, 10)))
print("by substitutions 0..5+:", ["%.1f%%" % (100.0 * fl[np.minimum(nsub, 5) == c].mean()) for c in range(6)])
