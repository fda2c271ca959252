"""kbo_call_batch with the device's second pass against the host's (KBO_CALL_DEVICE_SECOND is read once per process: the host's
result comes from a child process), every variant of every read, several slab sizes."""
import os, sys, subprocess, pickle, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, ROOT)
import kbo_amd
from kbo_amd import batch, synth
G, R, L, K = 20_000_000, 6000, 10_000, 63
g = synth.genome(G)
rng = np.random.default_rng(7)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
starts = rng.integers(0, G - L - 64, R)
reads = np.stack([g[a:a + L] for a in starts])
hit = rng.random((R, L)) < 0.01
reads[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
for r in range(0, R, 3):
    p = int(rng.integers(200, L - 200)); reads[r, p:L - 3] = reads[r, p + 3:].copy()
    q = int(rng.integers(200, L - 200)); reads[r, q + 2:] = reads[r, q:L - 2].copy(); reads[r, q:q + 2] = acgt[rng.integers(0, 4, 2)]
concat = reads.reshape(-1); offsets = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=K, num_threads=16))
sbwt.to_device()
opts = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=K, build_select=True))


def run(mb):
    kbo_amd.lib().kbo_set_slab_bytes(mb << 20)
    res = batch.call_batch_arrays(sbwt, concat, offsets, opts)
    return [[(p, q, r) for p, q, r in batch.variants_of(res, s)] for s in range(R)]


if len(sys.argv) > 1:
    pickle.dump({mb: run(mb) for mb in (16, 64)}, open(sys.argv[1], "wb"))
    sys.exit(0)
env = dict(os.environ, KBO_CALL_DEVICE_SECOND="0")
subprocess.run([sys.executable, __file__, "/tmp/host_second.pkl"], env=env, check=True)
host = pickle.load(open("/tmp/host_second.pkl", "rb"))
print("host second pass: slab 16 == slab 64:", host[16] == host[64], sum(len(v) for v in host[16]), "variants")
for mb in (16, 64, 64, 64, 32, 128):
    got = run(mb)
    bad = [s for s in range(R) if got[s] != host[16][s]]
    print("device second pass, slab", mb, ":", sum(len(v) for v in got), "variants;", len(bad), "reads differ", bad[:5])
    for s in bad[:2]:
        a, b = set(got[s]), set(host[16][s])
        print("   read", s, "only device", sorted(a - b)[:4], "only host", sorted(b - a)[:4])
