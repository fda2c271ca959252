"""kbo_call_batch (host second pass) under different chunk sizes / plan on-off: every variant against the default's."""
import os, sys, subprocess, pickle, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
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
    if os.environ.get("INDEL"):
        for r in range(0, R, 3):
            p = int(rng.integers(200, L - 200)); reads[r, p:L - 3] = reads[r, p + 3:].copy()
            q = int(rng.integers(200, L - 200)); reads[r, q + 2:] = reads[r, q:L - 2].copy(); reads[r, q:q + 2] = acgt[rng.integers(0, 4, 2)]
    concat = reads.reshape(-1); offsets = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    if os.environ.get("RAGGED"):
        lens = rng.integers(3000, L + 1, R)
        concat = np.concatenate([reads[r, :lens[r]] for r in range(R)])
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=K, num_threads=16))
    if os.environ.get("NOPLAN"):
        kbo_amd.lib().kbo_set_plan(0, 0, 0)
    sbwt.to_device()
    if os.environ.get("SLAB_MB"):
        kbo_amd.lib().kbo_set_slab_bytes(int(os.environ["SLAB_MB"]) << 20)
    opts = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=K, build_select=True))
    out = []
    for rep in range(2):
        res = batch.call_batch_arrays(sbwt, concat, offsets, opts)
        out.append([[(p, q, r) for p, q, r in batch.variants_of(res, s)] for s in range(R)])
    pickle.dump(out, open(sys.argv[1], "wb"))
    sys.exit(0)


def child(tag, **env):
    e = dict(os.environ, KBO_CALL_DEVICE_SECOND="0", **{k: str(v) for k, v in env.items()})
    subprocess.run([sys.executable, __file__, "/tmp/c_%s.pkl" % tag], env=e, check=True)
    return pickle.load(open("/tmp/c_%s.pkl" % tag, "rb"))


base = child("base")
print("default, substitutions only: run 1 == run 2:", base[0] == base[1], sum(len(v) for v in base[0]), "variants")
ref = child("ragged_np", RAGGED=1, NOPLAN=1)[0]
for tag, env in (("ragged", dict(RAGGED=1)), ("ragged_c300", dict(RAGGED=1, KBO_WALK_CHUNK=300)), ("ragged_c457", dict(RAGGED=1, KBO_WALK_CHUNK=457)),
                 ("ragged_s64", dict(RAGGED=1, SLAB_MB=64)), ("ragged_s100", dict(RAGGED=1, SLAB_MB=100))):
    got = child(tag, **env)
    for rep in range(2):
        bad = [s for s in range(len(ref)) if got[rep][s] != ref[s]]
        print(tag, "run", rep, ":", sum(len(v) for v in got[rep]), "variants;", len(bad), "reads differ", bad[:6])
