"""A sharded index at its real size: n_sets >= 2^32 (kbo_hip.h "Sharded indexes").

    G=2200000000 READS=1000000 python tools/shard_scale.py

builds a G-base synthetic genome WITH its reverse complements through kbo_index_build (2 G >= 3.76 * 10^9 rows: the build shards
by itself - nothing is forced unless SHARDS is set), walks READS 150-base reads of both strands (a tenth error-free, the rest
with 1 % substitutions, some with N) through kbo_ms_batch / kbo_matches_batch / kbo_find_batch, and checks

  1. the error-free reads: MS depth = min(i + 1, k) at every base, whichever strand - known without any index;
  2. a SAMPLE of all reads against the oracle: every shard's parts adopted by the oracle, expected depth = the maximum over
     the shards of the oracle's depths (the property tests/test_capi_host.py proves against the oracle's index of everything
     at small size), expected characters = the oracle's derandomize + translate of it with the threshold of the union's
     n_kmers; run lengths = the oracle's of those characters.

Test infrastructure (it imports oracle/); prints one summary line per step.
"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd  # noqa: E402
from kbo_amd import batch, synth  # noqa: E402
import oracle.binding as oracle  # noqa: E402


def rss_gb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def main():
    G = int(float(os.environ.get("G", "2200000000")))
    R = int(float(os.environ.get("READS", "1000000")))
    sample = int(os.environ.get("SAMPLE", "100000"))
    forced = int(os.environ.get("SHARDS", "0"))
    contig = int(float(os.environ.get("CONTIG", "100000000")))
    k, rl = 31, 150
    threads = len(os.sched_getaffinity(0))
    rng = np.random.default_rng(5)
    t0 = time.time()
    g = synth.genome(G, seed=77)
    seqs = [g[a:min(a + contig, G)] for a in range(0, G, contig)]
    print(f"genome: {G} bases in {len(seqs)} contigs ({time.time() - t0:.0f} s)", flush=True)
    L = kbo_amd.lib()
    t0 = time.time()
    L.kbo_set_index_shards(forced)
    sbwt, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, add_revcomp=True, num_threads=threads))
    L.kbo_set_index_shards(0)
    print(f"built: {sbwt.shards()} shards, n_sets {sbwt.n_sets()} (2^32 = {1 << 32}), n_kmers {sbwt.n_kmers()}, "
          f"{time.time() - t0:.0f} s, peak RSS {rss_gb():.0f} GB", flush=True)
    assert sbwt.shards() >= 2
    if not forced:
        assert sbwt.n_sets() >= (1 << 32)
    t0 = time.time()
    sbwt.to_device(-1)
    print(f"on the device: {sum(sbwt.device_bytes()) / 1e9:.1f} GB of rank blocks + entries, {sbwt.device_plan_bytes() / 1e9:.1f} GB of "
          f"path cover ({time.time() - t0:.0f} s, peak RSS {rss_gb():.0f} GB)", flush=True)

    # reads: whole inside a contig, alternate strands, a tenth error-free
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    c_of = rng.integers(0, len(seqs), R)
    clen = np.array([len(s) for s in seqs])
    at = (rng.random(R) * (clen[c_of] - rl)).astype(np.int64) + c_of * contig
    concat = g[(at[:, None] + np.arange(rl)[None, :]).ravel()].reshape(R, rl)
    exact = np.arange(R) % 10 == 0
    hit = (rng.random((R, rl)) < 0.01) & ~exact[:, None]
    concat[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]   # (a quarter of these put the same base back)
    rev = (np.arange(R) + np.arange(R) // 10) % 2 == 1
    concat[rev] = comp[concat[rev]][:, ::-1]
    with_n = (np.arange(R) % 23 == 5) & ~exact
    concat[with_n, rng.integers(0, rl, int(with_n.sum()))] = ord("N")
    concat = np.ascontiguousarray(concat.ravel())
    offsets = (np.arange(R + 1, dtype=np.uint64) * rl)

    for _ in range(2):  # (the second pass is the steady state)
        t0 = time.time()
        chars = batch.matches_batch(sbwt, concat, offsets)
        dt = time.time() - t0
    print(f"kbo_matches_batch: {R} reads, {R * rl / dt / 1e9:.1f} Gbp/s host to host over {sbwt.shards()} shards", flush=True)
    d, _, _ = batch.ms_batch(sbwt, concat, offsets)
    rles, ro = batch.find_batch(sbwt, concat, offsets, kbo_amd.FindOpts(max_gap_len=2))

    # 1. error-free reads of either strand
    ramp = np.minimum(np.arange(rl) + 1, k).astype(np.uint8)
    dm = d.reshape(R, rl)
    assert np.array_equal(dm[exact], np.broadcast_to(ramp, (int(exact.sum()), rl)))
    assert (chars.reshape(R, rl)[exact] == ord("M")).all()
    print(f"1. {int(exact.sum())} error-free reads ({int((exact & rev).sum())} of the reverse strand): depth = min(i + 1, k) everywhere, "
          "all 'M'", flush=True)

    # 2. a sample against the oracle over the shards
    n = min(sample, R)
    sub, sub_off = concat[:n * rl], offsets[:n + 1]
    t0 = time.time()
    exp_d = np.zeros(n * rl, dtype=np.uint8)
    for i in range(sbwt.shards()):
        sh = sbwt.shard(i)
        rows, Carr, lcs = sh.export_parts()
        ora = oracle.Index.from_parts(k, sh.n_sets(), sh.n_kmers(), rows, Carr, lcs)
        del rows, lcs
        _, di = ora.matches_batch(sub, sub_off, 1e-7, n_threads=threads, want_d=True)
        np.maximum(exp_d, di, out=exp_d)
        del ora
    assert np.array_equal(d[:n * rl], exp_d)
    thr = oracle.random_match_threshold(k, sbwt.n_kmers(), 4, 1e-7)
    n_chars = min(n, 20000)
    for s in range(n_chars):
        e = oracle.translate_ms_vec(oracle.derandomize_ms_vec(exp_d[s * rl:(s + 1) * rl].astype(np.uint64), k, thr), k, thr)
        assert chars[s * rl:(s + 1) * rl].tobytes().decode() == e, s
    er, eo = oracle.run_lengths_batch(chars, offsets, 2)
    assert np.array_equal(np.asarray(ro, dtype=np.uint64), eo) and np.array_equal(np.asarray(rles, dtype=np.uint64).reshape(-1, 7), er)
    print(f"2. {n} reads: depth equal to the maximum of the oracle's over the {sbwt.shards()} shards at every base; {n_chars} reads' "
          f"characters equal to the oracle's derandomize + translate (threshold {thr}); run lengths of all {R} reads equal "
          f"({time.time() - t0:.0f} s, peak RSS {rss_gb():.0f} GB)", flush=True)
    print("shard_scale ok", flush=True)


if __name__ == "__main__":
    main()
