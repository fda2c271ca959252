#!/usr/bin/env python3
"""CPU model of map_long_kernel (kbo_amd/csrc/long_kernels.hip): kbo::map / matches for sequences of any length, piece by piece,
against the oracle (oracle/kbo_oracle.c), base by base.

A sequence is cut into PIECES: `own` bases [s, s + n) inside a region [s - CB, s + n + CA) of at most 1024 bases.  Inside a
region the read is covered by STRETCHES - intervals that equal a path of the index's text on one diagonal - kept as two bit
planes (consecutive diagonals alternate planes, so that two stretches can overlap); everything that follows is a function of
the planes alone:

  cov(i)   i lies in a stretch of more than t bases                      G(i)   ... at depth > t (i - b >= t)
  chars    'M' where cov; else 'X' if cov(i + 1) and (i <= 1 or cov(i - 1)) else '-';
           'R','R' at i, i + 1 where G(i), not G(i + 1), cov(i + 1)          (translate.rs:195-203, :282-288)
  U(e)     the window of `order` bases ending at e lies in no single stretch
  proof    every maximal run of U: its first end, its last end, and every cov-th in between (cov = t - order + 2) is looked up in
           the depth table; all absent => no string of t + 1 bases outside a single stretch is in the index => the matching
           statistic is min(k, depth in the stretch) wherever that exceeds t, and at most t elsewhere; derandomize_ms_vec
           (derandomize.rs:269-288) then gives x[i] = min(k, i + 1 - b) with b the start of the next stretch of more than t
           bases that ends behind i.  A window that is present but not extended to the left by the read's base is fine when the
           window one base on is absent or not extended either.

python tools/model/long_form.py [trials] [order offset]"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import binding as ora  # noqa: E402

ACGT = b"ACGT"
CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(ACGT):
    CODE[_c] = _i


class Index:
    def __init__(self, contigs, k, order, seed_d):
        self.k, self.order, self.D = k, order, seed_d
        self.oi = ora.Index.build([bytes(c) for c in contigs], k=k)
        self.t = ora.random_match_threshold(k, self.oi.n_kmers, 4, 1e-7)
        # the text: the contigs' ACGT runs back to back, a mark (255) in front of each
        parts = []
        for c in contigs:
            parts.append(np.array([255], dtype=np.uint8))
            parts.append(CODE[np.frombuffer(bytes(c), dtype=np.uint8)])
        parts.append(np.array([255] * 4, dtype=np.uint8))
        self.text = np.concatenate(parts)
        self.present = set()
        self.present1 = set()
        self.seeds = {}
        tx = self.text
        n = len(tx)
        # strings of order / order + 1 / D bases without a mark, as tuples of bytes -> python bytes keys
        tb = tx.tobytes()
        for i in range(n):
            for L, dst in ((order, self.present), (order + 1, self.present1)):
                if i + L <= n:
                    s = tb[i:i + L]
                    if b"\xff" not in s:
                        dst.add(s)
            if i + seed_d <= n:
                s = tb[i:i + seed_d]
                if b"\xff" not in s:
                    e = i + seed_d - 1
                    if s in self.seeds:
                        self.seeds[s] = (self.seeds[s][0], True)
                    else:
                        self.seeds[s] = (e, False)


DEBUG = False


class Stats:
    def __init__(self):
        self.pieces = self.flagged = self.lookups = self.second = self.bases = self.seed_lookups = self.iters = 0
        self.plain_pieces = self.plain_flagged = 0
        self.reasons = {}


def find_planes(ix, qc, st, TH=6, J=None, RUN=10):
    """qc: codes of the region (255 = no base).  -> planes P[2] (1 = in no stretch of that plane)"""
    R = len(qc)
    order, D = ix.order, ix.D
    if J is None:
        J = order - 3
    tx = ix.text
    nT = len(tx)
    qb = qc.tobytes()
    P = [np.ones(R, dtype=np.uint8), np.ones(R, dtype=np.uint8)]
    endz = [-1, -1]

    def compare(delta):
        idx = np.arange(R) + delta
        ok = (idx >= 0) & (idx < nT)
        tv = np.where(ok, tx[np.clip(idx, 0, nT - 1)], 255)
        return ((tv != qc) | (qc == 255) | (tv == 255)).astype(np.uint8)

    def seed_at(e):  # window ending at e -> diagonal or None
        if e - D + 1 < 0 or e >= R:
            return None
        s = qb[e - D + 1:e + 1]
        st.seed_lookups += 1
        hit = ix.seeds.get(s)
        if hit is None:
            return None
        return hit[0] - e, hit[1]

    def seed_search(c, lanes):
        amb = None
        for j in lanes:
            e = c + D - 1 + D * j
            if e >= R:
                break
            h = seed_at(e)
            if h is None:
                continue
            if not h[1]:
                return h[0], e - D + 1
            if amb is None:
                amb = (h[0], e - D + 1)
        return amb

    def left_start(mm, A, lower):
        # the assignment reaches back from A over sparse mismatches, up to a stretch of TH mismatches in 16 bases
        cs = np.concatenate(([0], np.cumsum(mm)))
        for j in range(A - 1, lower - 1, -1):
            lo = max(lower, j - 15)
            if mm[j] and cs[j + 1] - cs[lo] >= TH:
                return j + 1
        return lower

    c = 0
    cur = 0
    delta = None
    mm = None
    start = A = 0
    for _ in range(96):
        st.iters += 1
        if delta is None:
            hit = seed_search(c, range(4))
            if hit is None:
                hit = seed_search(c + 4 * D, range(64))
            if hit is None:
                break
            delta, A = hit
            mm = compare(delta)
            start = left_start(mm, A, max(0, endz[cur] + 1, endz[1 - cur] - J))
        # where the diagonal is lost: the first window of 16 bases from A on with TH mismatches
        cs = np.concatenate(([0], np.cumsum(mm)))
        f = R
        for i in range(A, R):
            hi = min(R, i + 16)
            if cs[hi] - cs[i] >= TH:
                f = i + int(np.argmax(mm[i:hi]))
                break
        seg = slice(start, f)
        P[cur][seg] = np.where(mm[seg] == 0, 0, P[cur][seg])
        endz[cur] = f
        if f >= R:
            break
        # the next diagonal: 16 bases a little further on, against the text beside the current diagonal
        # of the 64 diagonals beside this one, the one on which the read goes on soonest: the first run of RUN matching bases among the
        # 32 behind f (ties: the longer run, then the nearer diagonal)
        found = None
        best = None
        if f + 1 + RUN <= R:
            hi = min(R, f + 33)
            for s in range(-32, 32):
                idx = np.arange(f + 1, hi) + delta + s
                ok = (idx >= 0) & (idx < nT)
                tv = np.where(ok, tx[np.clip(idx, 0, nT - 1)], 255)
                m = ((tv != qc[f + 1:hi]) | (qc[f + 1:hi] == 255) | (tv == 255)).astype(np.uint8)
                run = 0
                for j in range(len(m) - 1, -1, -1):  # run[j] = zeros from j on
                    run = 0 if m[j] else run + 1
                    m[j] = min(run, 255)
                at = next((j for j in range(len(m)) if m[j] >= RUN), None)
                if at is None:
                    continue
                key = (at, -int(m[at]), abs(s), s)
                if best is None or key < best[0]:
                    best = (key, s, at)
        if best is not None:
            found = (delta + best[1], f + 1 + best[2])
        if found is None:
            hit = seed_search(f + 4, range(4))
            if hit is not None:
                found = hit
        if found is None:
            delta = None
            c = f + 4 + 4 * D
            continue
        d2, A2 = found
        if d2 == delta:  # the same diagonal after all (a cluster of substitutions): on with it
            start = f
            A = A2
            continue
        mm2 = compare(d2)
        other = 1 - cur
        b2 = left_start(mm2, A2, max(0, f - J, endz[other] + 1))
        if DEBUG:
            print("    switch f", f, "shift", d2 - delta, "A2", A2, "b2", b2, "mm1", ''.join(map(str, mm[max(0, f - 20):f + 40])), "mm2", ''.join(map(str, mm2[max(0, f - 20):f + 40])))
        delta, mm, cur, start, A = d2, mm2, other, b2, A2
    return P


def runs_all_ones(z, L):
    """e -> z[e - L + 1 .. e] all ones (False where the window leaves the array)"""
    n = len(z)
    cs = np.concatenate(([0], np.cumsum(z)))
    out = np.zeros(n, dtype=bool)
    if n >= L:
        out[L - 1:] = (cs[L:] - cs[:n - L + 1]) == L
    return out


def analyse(ix, qc, P, g0, seqlen, own0, own1, st, grid_off=0):
    """region = sequence bases [g0, g0 + R); own bases (region coordinates) [own0, own1) -> (chars of the own bases, flagged)"""
    R = len(qc)
    k, t, order = ix.k, ix.t, ix.order
    covw = t - order + 2
    Z = [1 - P[0], 1 - P[1]]
    inA = [runs_all_ones(z, t + 1) for z in Z]  # ending-at form: the t + 1 bases ending at e lie in one stretch
    G = inA[0] | inA[1]
    cov = np.zeros(R + 2, dtype=bool)  # cov[i + 1] for region position i; cov[0], cov[R + 1] = outside
    for e in np.nonzero(G)[0]:
        cov[e - t + 1:e + 2] = True
    chars = np.full(R, ord('-'), dtype=np.uint8)
    for i in range(R):
        gi = g0 + i
        if cov[i + 1]:
            chars[i] = ord('M')
        elif cov[i + 2] and (gi <= 1 or cov[i]):
            chars[i] = ord('X')
    for i in range(R - 1):
        if G[i] and not G[i + 1] and cov[i + 2]:
            chars[i] = ord('R')
            if 2 <= g0 + i + 1 < seqlen - 1:
                chars[i + 1] = ord('R')
    # the proof
    inO = runs_all_ones(Z[0], order) | runs_all_ones(Z[1], order)
    U = ~inO
    U[:order - 1] = False
    qb = bytes(np.where(qc == 255, 0, qc).astype(np.uint8))  # (codes; windows with a byte that is no base are never looked up)
    bad = np.concatenate(([0], np.cumsum(qc == 255)))
    flagged = False
    reason = None

    def look(e):
        """-> (present, extended by the read's base in front)"""
        st.lookups += 1
        if bad[e + 1] - bad[e - order + 1] > 0:
            return False, False
        w = qb[e - order + 1:e + 1]
        if w not in ix.present:
            return False, False
        if e - order < 0 or qc[e - order] == 255:
            return True, False
        return True, qb[e - order:e + 1] in ix.present1

    # every run of U: its first end u0 and every c-th from there (c = t - order: a window may then be present as long as nothing
    # deeper than order + 1 ends there: present and extended to the left by the read's base -> the window one base back must not
    # be both).  The kernel knows the distance to u0 exactly up to 48 bases back: beyond M = c * (48 / c) it takes every c-th
    # position of the piece instead.  The run's last end u1 - the window that starts at the last base in front of the next
    # stretch: when it is one of those points and present, the window one base on (inside that stretch) must not be extended by
    # that base; when it is not, that window one base on is looked up INSTEAD, for the same bit.
    c = covw - 2
    M = c * (48 // c)
    e = order - 1
    while e < R:
        if not U[e]:
            e += 1
            continue
        u0 = e
        while e + 1 < R and U[e + 1]:
            e += 1
        u1 = e

        def is_point(x):
            rl = x - u0
            return (rl % c == 0) if rl <= M else ((x + grid_off) % c == 0)
        pts = [x for x in range(u0, u1 + 1) if is_point(x)]
        ext_end = None
        if not is_point(u1):
            if u1 + 1 < R:
                ext_end = u1 + 1
            else:
                pts.append(u1)
        for x in pts:
            pr, ext = look(x)
            if not pr:
                continue
            if ext:
                # order + 1 bases end at x.  The windows of t + 1 bases that hold them end at x .. x + c - 1.  A string of
                # order + 1 bases ending at x - j that is absent rules out those ending at x + c - j or before; one ending at
                # x + i, those ending at x + i or later: any pair with i + j <= c + 1 (j = 1 alone) will do.  (A string that
                # starts in front of the region or ends behind it is in none of the windows that matter.)
                def both(e_):
                    if e_ - order < 0 or e_ >= R:
                        return False
                    st.second += 1
                    pr_, ext_ = look(e_)
                    return pr_ and ext_
                j = 1
                while j <= c and both(x - j):
                    j += 1
                ok = j <= c
                if ok and j > 1:
                    i = 1
                    while i <= c + 1 - j and both(x + i):
                        i += 1
                    ok = i <= c + 1 - j
                if not ok:
                    flagged, reason = True, "deeper than order + 1"
                    break
            if x == u1 and x + 1 < R:
                st.second += 1
                pr2, ext2 = look(x + 1)
                if pr2 and ext2:
                    flagged, reason = True, "behind the last window"
                    break
        if not flagged and ext_end is not None:
            pr2, ext2 = look(ext_end)  # inside the next stretch: is it extended by the base in front of that stretch?
            if pr2 and ext2:
                flagged, reason = True, "behind the last window"
        if flagged:
            break
        e += 1
    if flagged:
        st.reasons[reason] = st.reasons.get(reason, 0) + 1
    return chars[own0:own1], flagged


def run_sequence(ix, seq, st, n_region=1008, grid_off=0):
    """-> (chars, mask of the bases of flagged pieces)"""
    k = ix.k
    L = len(seq)
    qc_all = CODE[np.frombuffer(bytes(seq), dtype=np.uint8)]
    CB, CA = k, k + 1
    n_own = n_region - CB - CA
    out = np.zeros(L, dtype=np.uint8)
    fl = np.zeros(L, dtype=bool)
    s = 0
    while s < L:
        n = min(n_own, L - s)
        g0, g1 = max(0, s - CB), min(L, s + n + CA)
        qc = qc_all[g0:g1]
        P = find_planes(ix, qc, st)
        ch, flagged = analyse(ix, qc, P, g0, L, s - g0, s - g0 + n, st, grid_off=grid_off)
        st.pieces += 1
        st.bases += n
        if flagged:
            st.flagged += 1
            fl[s:s + n] = True
        out[s:s + n] = ch
        s += n
    return out, fl


def mutate(rng, src, sub, indel, big=0.0):
    out = bytearray()
    i = 0
    while i < len(src):
        r = rng.random()
        if r < sub:
            out.append(rng.choice([c for c in ACGT if c != src[i]]))
            i += 1
        elif r < sub + indel:
            n = rng.randint(1, 3) if rng.random() > big else rng.randint(4, 60)
            if rng.random() < 0.5:
                i += n  # deletion
            else:
                out.extend(rng.choice(ACGT) for _ in range(n))
        else:
            out.append(src[i])
            i += 1
    return bytes(out)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(11)
    total_bad = 0
    for (k, glen, n_contigs) in ((31, 120_000, 3), (15, 30_000, 2), (63, 60_000, 1)):
        contigs = []
        for c in range(n_contigs):
            g = bytearray(rng.choice(ACGT) for _ in range(glen // n_contigs))
            # repeats: a copy of 300 bases from elsewhere, a tandem array, a homopolymer
            a, b = rng.randrange(len(g) - 400), rng.randrange(len(g) - 400)
            g[b:b + 300] = g[a:a + 300]
            t0 = rng.randrange(len(g) - 500)
            g[t0:t0 + 240] = bytes(g[t0:t0 + 12]) * 20
            h0 = rng.randrange(len(g) - 100)
            g[h0:h0 + 40] = b"A" * 40
            contigs.append(bytes(g))
        n_rows = glen
        order = min(k - 1, max(4, int(np.ceil(np.log(n_rows) / np.log(4) + 3.2)))) + (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        seed_d = min(order - 1, max(4, int(np.log(n_rows) / np.log(4) + 3)))
        ix = Index(contigs, k, order, seed_d)
        if not (order <= ix.t < k):
            print("k", k, "skipped: order", order, "t", ix.t)
            continue
        print("k", k, "t", ix.t, "order", order, "D", seed_d, "cov", ix.t - order + 2)
        for name, sub, indel, big in (("1% subs", 0.01, 0.0, 0.0), ("5% subs", 0.05, 0.0, 0.0), ("ONT-like", 0.025, 0.0125, 0.0),
                                      ("1% subs + rare big indels", 0.01, 0.002, 0.5), ("clean", 0.0, 0.0, 0.0)):
            st = Stats()
            bad = 0
            for tr in range(trials):
                ci = rng.randrange(len(contigs))
                src = contigs[ci]
                L = rng.choice([200, 700, 1500, 3000, 10000])
                a = rng.randrange(0, len(src) - L)
                seq = bytearray(mutate(rng, src[a:a + L], sub, indel, big))
                kind = rng.random() if (tr % 3) == 0 else 1.0
                if kind < 0.15:  # a stretch of something else in the middle, or N
                    p = rng.randrange(len(seq))
                    n = rng.randint(1, 400)
                    seq[p:p + n] = bytes(rng.choice(ACGT) for _ in range(n)) if rng.random() < 0.5 else b"N" * n
                elif kind < 0.25:  # a join of two places
                    b2 = rng.randrange(0, len(src) - 500)
                    seq = seq[:len(seq) // 2] + bytearray(src[b2:b2 + 500])
                elif kind < 0.3:
                    seq = bytearray(rng.choice(ACGT) for _ in range(L))  # unrelated
                seq = bytes(seq)
                if len(seq) < 3:
                    continue
                want = np.frombuffer(ix.oi.matches(seq, 1e-7), dtype=np.uint8)
                p0, f0 = st.pieces, st.flagged
                got, fl = run_sequence(ix, seq, st, grid_off=tr % 16)
                if kind >= 0.3:
                    st.plain_pieces += st.pieces - p0
                    st.plain_flagged += st.flagged - f0
                diff = (want != got) & ~fl
                if diff.any():
                    bad += 1
                    p = int(np.nonzero(diff)[0][0])
                    if bad <= 3:
                        print("  MISMATCH", name, "len", len(seq), "at", p)
                        print("   want", want[max(0, p - 40):p + 40].tobytes().decode())
                        print("   got ", got[max(0, p - 40):p + 40].tobytes().decode())
            total_bad += bad
            print("  %-28s pieces %5d flagged %5.2f%% (plain sequences %5.2f%%)  look-ups/kb %6.1f second %5.2f seeds/kb %5.1f iters/piece %4.1f  bad %d  %s" % (
                name, st.pieces, 100.0 * st.flagged / max(1, st.pieces), 100.0 * st.plain_flagged / max(1, st.plain_pieces), 1000.0 * st.lookups / max(1, st.bases),
                1000.0 * st.second / max(1, st.bases), 1000.0 * st.seed_lookups / max(1, st.bases), st.iters / max(1, st.pieces), bad, st.reasons))
    print("bad", total_bad)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
