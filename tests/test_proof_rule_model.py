"""A brute-force model (CPU, no library) of the rule map_reads_kernel uses for the window of `order` bases that STARTS at a mismatch m
and is in the index (kbo_amd/csrc/map_kernels.hip, the proof's loop): when the read's base in front of the window does not extend
it (the entry's left-extension bit) and the read's base at m does not extend the window one base further on (that window's entry,
or that window is absent), no string of the read that runs through m and ends at or behind the window's last base is in the index
with more than `order` bases.  The same for a read without a seed, where every window is judged like that.  Checked against all
substrings of a random text with many chance repeats.  Semantics of the values it protects: /root/reference/src/derandomize.rs:221-288
(values above the threshold are what the closed form must not miss)."""
import numpy as np


def _substrings(text, max_len):
    s = set()
    for L in range(1, max_len + 1):
        for a in range(len(text) - L + 1):
            s.add(text[a:a + L])
    return s


def test_window_that_starts_at_a_mismatch():
    rng = np.random.default_rng(11)
    order, max_len = 4, 9
    accepted = refused = 0
    for trial in range(30):
        text = "".join(rng.choice(list("ACGT"), 260, p=[0.4, 0.3, 0.2, 0.1]))  # skewed: windows of 4 bases repeat by chance
        index = _substrings(text, max_len)
        for _ in range(300):
            L = int(rng.integers(order + 3, 30))
            a = int(rng.integers(0, len(text) - L))
            read = list(text[a:a + L])
            m = int(rng.integers(1, L - order))           # the mismatch; the window [m, e] lies inside the read
            read[m] = rng.choice([c for c in "ACGT" if c != read[m]])
            read = "".join(read)
            e = m + order - 1
            window = read[m:e + 1]
            if window not in index:
                continue                                   # (an absent window: the ordinary case, nothing to model)
            deeper = read[m - 1] + window in index         # entry of the window: its left-extension bit for the read's base
            nxt = read[m + 1:e + 2] if e + 1 < L else None  # the window one base on, if the read has that base
            extends = nxt is not None and nxt in index and (read[m] + nxt) in index  # its entry: present, and extended by read[m]
            ok = (not deeper) and not extends
            # brute force: the longest string through m that ends at or behind e and is in the index
            longest = 0
            for x in range(0, m + 1):
                for y in range(e, L):
                    if y - x + 1 <= max_len and read[x:y + 1] in index:
                        longest = max(longest, y - x + 1)
            if ok:
                accepted += 1
                assert longest <= order, (text, read, m, longest)
            else:
                refused += 1
    assert accepted > 200 and refused > 50  # (both branches were really exercised)


def test_read_without_a_seed_every_window_judged_that_way():
    """a read that seeds nowhere: windows of `order` bases ending at order - 1, + cov, .. and at the read's last base, cov = t - order + 2
    (any string of t + 1 bases holds one of them whole).  Every window absent, or present but neither extended to the left by the
    base in front of it nor - one base on - by its own first base: no string of t + 1 bases of the read is in the index."""
    rng = np.random.default_rng(12)
    order, t = 4, 6
    cov = t - order + 2
    passed = flagged = with_present = 0
    for trial in range(30):
        text = "".join(rng.choice(list("ACGT"), 200, p=[0.4, 0.3, 0.2, 0.1]))
        index = _substrings(text, t + 2)
        for _ in range(400):
            L = int(rng.integers(t + 1, 24))
            read = "".join(rng.choice(list("ACGT"), L, p=[0.1, 0.2, 0.3, 0.4]))  # (the other end of the alphabet: few windows present)
            ends = sorted({min(order - 1 + u * cov, L - 1) for u in range((L - order + cov - 1) // cov + 1)})
            fail = any_present = False
            for e in ends:
                s = e - order + 1
                w = read[s:e + 1]
                if w not in index:
                    continue
                any_present = True
                if s == 0:
                    fail = True      # (no base in front of it: the kernel leaves such a read to the plain walk)
                    continue
                deeper = read[s - 1] + w in index
                nxt = read[s + 1:e + 2] if e + 1 < L else None
                extends = nxt is not None and nxt in index and (read[s] + nxt) in index
                if deeper or extends:
                    fail = True
            if fail:
                flagged += 1
                continue
            passed += 1
            with_present += any_present
            for a in range(0, L - t):
                assert read[a:a + t + 1] not in index, (text, read, a)
    assert passed > 500 and flagged > 100 and with_present > 50


def _witnessed(read, e, m, order, t, index):
    """The rule map_reads_kernel (no anchors) applies to a window of `order` bases ending at e that IS in the index: the strings of
    t + 1 bases that hold it end at e .. e + c, c = t + 1 - order.  A window ending at e + i that is absent - or whose string with the
    base in front of it is - rules out those ending at e + i or later; a window ending at e - j that is absent those ending at
    e - j + c or before (one base fewer when only its string with the base in front is absent; j = 0: the window's own
    left-extension bit).  m: the break the proof is about (only strings that hold it need ruling out), None for a read without a
    seed.  -> True when every string of t + 1 bases that holds the window (and the break) is ruled out."""
    L = len(read)
    c = t + 1 - order

    def present(w):
        return read[w - order + 1:w + 1] in index

    def ext(w):  # the window with the read's base in front of it (w >= order)
        return read[w - order:w + 1] in index
    e_lo = max(e, t)
    e_hi = min(e + c, L - 1) if m is None else min(e + c, m + t, L - 1)
    if e_lo > e_hi:
        return True
    r = None
    for i in range(1, c + 2):
        w = e + i
        if w > e_hi:
            r = e_hi + 1
            break
        if not present(w) or not ext(w):
            r = w
            break
    if r is None:
        return False
    if r - 1 < e_lo:
        return True
    if e >= order and not ext(e) and e + c - 1 >= r - 1:
        return True
    for j in range(1, c + 1):
        w = e - j
        if w < order - 1 or w + c < r - 1:
            break
        if not present(w):
            return True
        if w >= order and not ext(w) and w + c - 1 >= r - 1:
            return True
    return False


def test_present_window_ruled_out_by_the_windows_around_it():
    """reads with mismatches against the text: the proof's windows per mismatch (ending at m, m + cov, .., m + order - 1); a present one
    is handed to _witnessed.  Whenever every window of every mismatch is absent or witnessed, no string of t + 1 bases that holds a
    mismatch is in the index."""
    rng = np.random.default_rng(13)
    order, t = 4, 7
    cov = t - order + 2
    passed = flagged = by_witness = 0
    for trial in range(40):
        text = "".join(rng.choice(list("ACGT"), 220, p=[0.4, 0.3, 0.2, 0.1]))
        index = _substrings(text, t + 2)
        for _ in range(300):
            L = int(rng.integers(t + 2, 40))
            a = int(rng.integers(0, len(text) - L))
            read = list(text[a:a + L])
            ms = sorted(set(int(x) for x in rng.integers(0, L, int(rng.integers(1, 4)))))
            for m in ms:
                read[m] = rng.choice([c for c in "ACGT" if c != read[m]])
            read = "".join(read)
            fail = used = False
            for m in ms:
                ends = []
                i = 0
                while i == 0 or (i - 1) * cov < order - 1:
                    e = m + min(i * cov, order - 1)
                    if e < L and e + 1 >= order:
                        ends.append(e)
                    i += 1
                for e in sorted(set(ends)):
                    if read[e - order + 1:e + 1] not in index:
                        continue
                    used = True
                    if not _witnessed(read, e, m, order, t, index):
                        fail = True
            if fail:
                flagged += 1
                continue
            passed += 1
            by_witness += used
            for s in range(0, L - t):
                if any(s <= m <= s + t for m in ms):
                    assert read[s:s + t + 1] not in index, (text, read, ms, s)
    assert passed > 1000 and flagged > 100 and by_witness > 100


def test_read_without_a_seed_present_windows_ruled_out_by_the_windows_around_them():
    rng = np.random.default_rng(14)
    order, t = 4, 7
    cov = t - order + 2
    passed = flagged = by_witness = 0
    for trial in range(40):
        text = "".join(rng.choice(list("ACGT"), 200, p=[0.4, 0.3, 0.2, 0.1]))
        index = _substrings(text, t + 2)
        for _ in range(400):
            L = int(rng.integers(t + 1, 30))
            read = "".join(rng.choice(list("ACGT"), L, p=[0.15, 0.2, 0.3, 0.35]))
            ends = sorted({min(order - 1 + u * cov, L - 1) for u in range((L - order + cov - 1) // cov + 1)})
            fail = used = False
            for e in ends:
                if read[e - order + 1:e + 1] not in index:
                    continue
                used = True
                if not _witnessed(read, e, None, order, t, index):
                    fail = True
            if fail:
                flagged += 1
                continue
            passed += 1
            by_witness += used
            for s in range(0, L - t):
                assert read[s:s + t + 1] not in index, (text, read, s)
    assert passed > 1000 and flagged > 100 and by_witness > 100
