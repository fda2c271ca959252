#!/usr/bin/env python3
"""The closed form of map_reads_kernel's direct instantiation (kbo_amd/csrc/map_kernels.hip) against the literal recurrences
(derandomize.rs:233-246, 282-285; translate.rs:180-216, 263-293), on the CPU: random reads described by their breaks - mismatch
positions, optionally one junction of two diagonals - with random values (<= t) wherever the matching statistic is not the
distance to the last break.  python tools/model/direct_form.py [trials]"""
import random
import sys


def literal(a, k, t):
    n = len(a)
    x = [0] * n
    x[n - 1] = a[n - 1] if a[n - 1] > t else 0
    for i in range(n - 2, -1, -1):
        c, nx = a[i], x[i + 1]
        x[i] = k if c == k else (c if (c > t and nx < c) else nx - 1)
    res = [' '] * n
    pos = 0
    while pos < n:  # translate.rs:275-290
        prev = x[pos - 1] if pos > 1 else k
        nxt = x[pos + 1] if pos < n - 1 else x[pos]
        cur = x[pos]
        if cur > t and 0 < nxt < t:
            a1, a2 = 'R', 'R'
        elif cur <= 0:
            a1, a2 = ('X' if (nxt == 1 and prev > 0) else '-'), ' '
        else:
            a1, a2 = 'M', ' '
        if not (pos > 1 and res[pos - 1] == 'R' and res[pos] == 'R'):
            res[pos] = a1
            if a2 == 'R' and pos + 1 < n - 1:
                res[pos + 1] = 'R'
        pos += 1
    return ''.join(res)


def direct(n, breaks, junction, K, T, ov=0):
    """breaks: sorted positions; junction: index into breaks or None; ov: bases in front of the junction that lie on BOTH diagonals
    (the second diagonal's ramp starts that much earlier).  Mirrors the kernel's loop."""
    at = ['M'] * n
    e, dn, at_end, cand = n, 0, True, False
    e_junction = False
    for q in range(len(breaks) - 1, -2, -1):
        m = breaks[q] if q >= 0 else -1
        j0 = 1 if (q < 0 or q == junction) else 0
        if q >= 0 and q == junction:
            m -= ov
            j0 += ov
        L = e - m
        jl = L - 1
        if jl < j0:  # an empty segment (a head in front of a mismatch at base 0, a junction right in front of the next break)
            continue
        if L > K:
            d = 0
        elif at_end:
            d = 0 if jl > T else -jl
        else:
            din = dn - L
            d = 0 if (din <= -2 and jl > T) else din
        xt = K if L > K else jl + d
        if cand:
            at[e] = 'X' if (K if e <= 1 else xt) > 0 else '-'
        cand = False
        # the junction to the right starts with x = 1 (its d is 0) and this segment ends above the threshold: translate's 'R','R'
        if not at_end and 0 < dn < T and xt > T and e_junction:
            at[e - 1] = 'R'
            if 2 <= e < n - 1:
                at[e] = 'R'
        if True:
            if -d == j0:
                nxt = 1 if -d + 1 <= jl else (0 if at_end else dn)
                if nxt == 1:
                    cand = True
                else:
                    at[m - d] = '-'
            elif -d > j0:
                for j in range(j0, min(-d, jl) + 1):
                    at[m + j] = '-'
                pz = m - d  # the base with x = 0: an 'X' when its successor has x = 1 and it is one of the read's two first bases (prev = k)
                nz = 1 if -d + 1 <= jl else ((0 if at_end else dn) if -d == jl else 0)
                if nz == 1 and 0 <= pz <= 1:
                    at[pz] = 'X'
        dn = d + j0
        e = m + j0
        e_junction = q >= 0 and q == junction
        at_end = False
    if cand:
        at[e] = 'X'
    return ''.join(at)


def make_a(n, breaks, junction, K, T, rng, ov=0):
    """matching statistics consistent with the breaks: distance to the last break where that exceeds what runs through a break
    (any value <= T there), never more than possible (<= i + 1, rises by at most one)"""
    a = []
    last = -1  # virtual break behind base -1
    bi = 0
    for i in range(n):
        while bi < len(breaks) and breaks[bi] + (1 if bi == junction else 0) <= i:
            last = breaks[bi] - (ov if bi == junction else 0)  # (the second diagonal matched `ov` bases in front of the junction too)
            bi += 1
        j = i - last
        base = min(j, K)
        v = base
        if rng.random() < 0.5 and last >= 0:  # something through the break: at most T, at least the ramp
            v = max(base, min(rng.randint(0, T), i + 1, K, (a[-1] + 1) if a else 1))
            if v > T:
                v = base
        if last >= 0 and j == 0 and not (junction is not None and breaks[junction] - ov == last):
            v = min(rng.randint(0, T), i + 1, (a[-1] + 1) if a else 1)  # the mismatch itself: whatever runs through it
        a.append(v)
    return a


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    rng = random.Random(7)
    bad = 0
    for tr in range(trials):
        K = rng.choice([11, 19, 31, 31, 31, 63])
        T = rng.randint(max(2, K // 3), K - 1)
        n = rng.randint(3, 160)
        nb = rng.choice([0, 1, 1, 2, 2, 3, 5, 9])
        breaks = sorted(rng.sample(range(n), min(nb, n)))
        junction, ov = None, 0
        if breaks and rng.random() < 0.4:
            qj = rng.randrange(len(breaks))
            if breaks[qj] + 1 < n:
                junction = qj
                prev = breaks[qj - 1] if qj > 0 else -1
                ov = rng.randint(0, max(0, min(12, breaks[qj] - prev - 1)))  # the overlap lies behind the previous break
        a = make_a(n, breaks, junction, K, T, rng, ov)
        want = literal(a, K, T)
        got = direct(n, breaks, junction, K, T, ov)
        if want != got:
            bad += 1
            if bad <= 8:
                print("K", K, "T", T, "n", n, "breaks", breaks, "junction", junction, "ov", ov)
                print(" a   ", a)
                print(" want", want)
                print(" got ", got)
    print("trials", trials, "bad", bad)


if __name__ == "__main__":
    main()
