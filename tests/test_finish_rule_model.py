"""The rule finish_reads_kernel's lanes rest on (kbo_amd/csrc/map_kernels.hip: a lane walks a piece of a read behind k - 1 bases that only
bring its state up), checked on the CPU with the oracle alone: the k-bounded matching statistics of bases [own0, own1) of a read are those
of the same bases of the string that starts k - 1 bases in front of own0 (or at the read's head) - a match is at most k bases long, so
no base further to the left can take part in one that ends at own0 or behind it.  The statistics themselves:
sbwt::StreamingIndex::matching_statistics as /root/reference/src/index.rs:243-256 calls it.  A byte that is no base (N, lower case) ends
every match in the whole read and in the piece alike.  Lanes per read as the kernel has them: 64, 16 and 4."""
import numpy as np
import pytest


@pytest.mark.parametrize("k", [31, 11, 5])
def test_a_piece_behind_k_minus_1_bases_has_the_reads_own_values(oracle, k):
    rng = np.random.default_rng(40 + k)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, 30_000)]
    genome[5_000:5_400] = genome[1_000:1_400]  # (a repeat: matches that are not the read's own place)
    ora = oracle.Index.build([genome.tobytes()], k=k)
    reads = []
    for _ in range(120):
        L = int(rng.integers(3, 161))
        a = int(rng.integers(0, len(genome) - L))
        r = genome[a:a + L].copy()
        hit = rng.random(L) < 0.06
        r[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
        if rng.random() < 0.2:
            r[int(rng.integers(0, L))] = ord("N")
        if rng.random() < 0.1:
            r[int(rng.integers(0, L))] |= 0x20
        reads.append(r)
    n_pieces = 0
    for s, r in enumerate(reads):
        L = len(r)
        whole = ora.matching_statistics(r.tobytes())[0]
        for lanes in (64, 16, 4):
            per = (L + lanes - 1) // lanes
            for lane in range(lanes):
                own0, own1 = min(lane * per, L), min(lane * per + per, L)
                if own0 >= own1:
                    continue
                j0 = max(0, own0 - (k - 1))
                part = ora.matching_statistics(r[j0:own1].tobytes())[0]
                assert np.array_equal(part[own0 - j0:], whole[own0:own1]), (k, s, own0, own1, j0, part.tolist(), whole[own0:own1].tolist())
                n_pieces += 1
    assert n_pieces > 1000
