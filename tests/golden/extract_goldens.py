#!/usr/bin/env python3
"""Transcribes the golden vectors held by the reference's own unit tests and
doctests (tmaklin/kbo v0.5.1, Rust) into tests/golden/kbo_golden.json.

The reference cannot be compiled or run in this image (no rustc), so these are
the *inputs and expected outputs written in the reference's tests* — data only,
no reference source text is kept.  Runs only where /root/reference is mounted;
the JSON it writes is committed and is what the tests read.

Every entry carries `src` = reference file:line of the test it was taken from.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kbo_golden.json")


def lines(fname, a, b):
    with open(os.path.join(REF, "src", fname)) as f:
        ls = f.readlines()
    return "".join(ls[a - 1:b])


def byte_vec(text):
    """vec![b'A',b'C',...] or b"ACGT" -> str"""
    m = re.search(r'b"([^"]*)"', text)
    if m:
        return m.group(1)
    return "".join(re.findall(r"b'(.)'", text))


def char_vec(text):
    return "".join(re.findall(r"'(.)'", text))


def let(fname, a, b, name):
    """value text of `let <name>... = <value>;` within lines a..b"""
    src = lines(fname, a, b)
    m = re.search(r"let\s+(?:mut\s+)?" + name + r"\b[^=]*=\s*(.*?);\s*$", src, re.S | re.M)
    assert m, (fname, a, b, name)
    return m.group(1)


G = {}

# ---------------------------------------------------------------- MS (index.rs)
G["ms"] = [{
    "src": "index.rs:265-273 (also doctest :229-240)",
    "ref_seqs": [byte_vec(let("index.rs", 262, 275, "reference"))],
    "k": 3,
    "query": byte_vec(let("index.rs", 262, 275, "query")),
    "expected_ms": [int(x) for x in re.findall(r"\d+", let("index.rs", 262, 275, "expected"))],
    "n_sets": 16, "n_kmers": 13,  # SURVEY.md Appendix A (derived, consistent with all goldens)
}]

# ---------------------------------------------------------------- derandomize.rs
cdf = [float(x) for x in re.findall(r"-?\d+\.\d+", let("derandomize.rs", 297, 305, "expected"))]
G["log_rm_max_cdf"] = {
    "src": "derandomize.rs:298-304", "alphabet_size": 4, "n_kmers": 20240921,
    "t": list(range(1, 32)), "expected": cdf, "tol": 1e-8,
    "doctest": {"src": "derandomize.rs:77-79", "t": 10, "alphabet_size": 4, "n_kmers": 20240921,
                "expected": -4.825812199808644, "tol": 1e-8},
}
G["random_match_threshold"] = {
    "src": "derandomize.rs:307-314", "k": 31, "n_kmers": 20240921, "alphabet_size": 4,
    "max_error_prob": [0.01 ** i for i in range(1, 6)], "expected": [15, 18, 22, 25, 28],
    "doctest": {"src": "derandomize.rs:118-125", "k": 31, "n_kmers": 20240921, "alphabet_size": 4,
                "max_error_prob": 0.01, "expected": 15},
}
G["derandomize_ms_val"] = [
    {"src": "derandomize.rs:317-328", "args": [3, 3, 2, 3], "expected": 3},
    {"src": "derandomize.rs:331-342", "args": [2, -1, 2, 3], "expected": -2},
    {"src": "derandomize.rs:345-356", "args": [3, -1, 2, 3], "expected": 3},
    {"src": "derandomize.rs:359-370", "args": [3, -1, 2, 4], "expected": 3},
]
G["derandomize_ms_vec"] = [{
    "src": "derandomize.rs:373-379 (doctest :260-266)",
    "noisy_ms": [1, 2, 2, 3, 2, 2, 3, 2, 1, 2, 3, 1, 1, 1, 2, 3, 1, 2], "k": 3, "threshold": 2,
    "expected": [0, 1, 2, 3, 1, 2, 3, 0, 1, 2, 3, -1, 0, 1, 2, 3, -1, 0],
}]

# ---------------------------------------------------------------- translate.rs
G["translate_ms_val"] = [
    {"src": "translate.rs:396-410", "args": [3, 1, 2, 2], "expected": ["R", "R"]},
    {"src": "translate.rs:413-427", "args": [3, 1, 3, 2], "expected": ["R", "R"]},
    {"src": "translate.rs:430-444", "args": [0, 1, 3, 2], "expected": ["X", " "]},
    {"src": "translate.rs:447-464", "args": [0, 1, 3, 2], "expected": ["X", " "]},
    {"src": "translate.rs:467-481", "args": [-1, 0, 3, 2], "expected": ["-", " "]},
    {"src": "translate.rs:484-498", "args": [1, 2, 3, 2], "expected": ["M", " "]},
]
G["translate_ms_vec"] = [
    {"src": "translate.rs:501-515 (doctest :239-242)",
     "derand_ms": [0, 1, 2, 3, 1, 2, 3, 0, 1, 2, 3, -1, 0, 1, 2, 3, -1, 0], "k": 3, "threshold": 2,
     "expected": "XMMRRMMXMMM--MMM--"},
    {"src": "translate.rs:518-532 (doctest :257-260)",
     "derand_ms": [1, 2, 3, 1, 2, 3, 3, 3, 3, 1, 2, 3], "k": 3, "threshold": 2,
     "expected": "MMRRMMMMRRMM"},
]

# ---------------------------------------------------------------- lib.rs doctests
small_ref = "AAAGAACCA-TCAGGGCG"
small_q = "GTGACTATGAGGAT"
G["matches"] = [{
    "src": "lib.rs:600-609", "ref_seqs": [small_ref], "k": 3, "query": small_q,
    "max_error_prob": 1e-7, "expected": "---------MMM--",
}]
G["map"] = [
    {"src": "lib.rs:647-660 (full defaults: fill_gaps, call_variants, format)",
     "query_seqs": [small_ref], "k": 3, "ref_seq": small_q, "max_error_prob": 1e-7,
     "fill_gaps": True, "call_variants": True, "format": True,
     "expected": "".join(chr(c) for c in [45] * 9 + [65, 71, 71, 45, 45])},
    {"src": "lib.rs:670-688", "query_seqs": ["CGTTGACTGGTGCCTGGGTTCTCAGAGCTGGGC"], "k": 7,
     "ref_seq": "CGTTGACTCTAGGTGCCTGGGTTCTCAGAGCTGGGC", "max_error_prob": 0.1,
     "fill_gaps": False, "call_variants": False, "format": True,
     "expected": "CGTTGACT---GGTGCCTGGGTTCTCAGAGCTGGGC"},
    {"src": "lib.rs:698-717", "query_seqs": ["CGTTGACTGGTGCCTGGGTTCTCAGAGCTGGGC"], "k": 7,
     "ref_seq": "CGTTGACTCTAGGTGCCTGGGTTCTCAGAGCTGGGC", "max_error_prob": 0.1,
     "fill_gaps": False, "call_variants": False, "format": False,
     "expected": "MMMMMMMM---MMMMMMMMMMMMMMMMMMMMMMMMM"},
]
find_src = lines("lib.rs", 786, 805)
find_seqs = re.findall(r'b"([ACGT]+)"', find_src)
assert len(find_seqs) == 3
G["find"] = [{
    "src": "lib.rs:786-805", "ref_seqs": find_seqs[:2], "k": 31, "query": find_seqs[2],
    "max_error_prob": 1e-7, "max_gap_len": 50,
    "n_kmers": 1176, "n_sets": 1237, "threshold": 16,
    "expected": [[0, 513, 512, 1, 0, 0, 0], [593, 1340, 709, 0, 0, 38, 3]],
    "rle_fields": ["start", "end", "matches", "mismatches", "jumps", "gap_bases", "gap_opens"],
}]

# ---------------------------------------------------------------- format.rs
G["run_lengths"] = [{
    "src": "format.rs:295-330", "aln": char_vec(let("format.rs", 295, 330, "input")),
    "max_gap_len": 0,
    "expected": [[5, 33, 28, 0, 0, 0, 0], [81, 207, 126, 0, 0, 0, 0], [372, 423, 51, 0, 0, 0, 0],
                 [487, 512, 25, 0, 0, 0, 0]],
}]

# ---------------------------------------------------------------- call (variant_calling.rs)


def vc_case(a, b, k, p, expected, ref_name="reference", q_name="query"):
    return {"src": f"variant_calling.rs:{a}-{b}", "k": k, "max_error_prob": p,
            "reference": byte_vec(let("variant_calling.rs", a, b, ref_name)),
            "query": byte_vec(let("variant_calling.rs", a, b, q_name)),
            "expected": expected}  # [query_pos, query_chars, ref_chars]


G["call_variants"] = [
    vc_case(311, 322, 20, 0.001, [[49, "T", "A"]]),
    vc_case(324, 336, 30, 0.001, [[29, "GCG", "AA"]]),
    vc_case(338, 348, 30, 0.001, [[29, "GCG", ""]]),
    vc_case(350, 360, 30, 0.001, [[31, "AAAA", ""]]),
    vc_case(362, 374, 20, 0.001, [[50, "G", ""]]),
    vc_case(376, 388, 20, 0.001, [[50, "A", ""]]),
    vc_case(390, 402, 20, 0.001, [[50, "", "G"]]),
    vc_case(404, 416, 20, 0.001, [[51, "", "T"]]),
    vc_case(418, 428, 30, 0.001, [[29, "", "GCG"]]),
    vc_case(430, 440, 30, 0.001, [[31, "", "AAAA"]]),
    vc_case(441, 455, 20, 0.001, [[24, "", "G"], [41, "C", "T"], [59, "C", ""]]),
]
# lib.rs:526-544 call() doctest: index built from `query`, `reference` streamed
call_doc = lines("lib.rs", 526, 544)
cd = re.findall(r'b"([ACGT]+)"', call_doc)
G["call"] = [{
    "src": "lib.rs:526-544", "k": 20, "max_error_prob": 0.001, "reference": cd[0], "query": cd[1],
    "expected": [[22, "AGG", ""], [42, "T", "C"], [60, "", "C"]],
}]


def av_case(a, b):
    src = lines("translate.rs", a, b)
    seqs = re.findall(r'b"([A-Z]+)"', src)
    return {"src": f"translate.rs:{a}-{b}", "k": 20, "threshold": 10, "max_error_prob": 0.001,
            "reference": seqs[0], "query": seqs[1], "expected": seqs[-1]}


G["add_variants"] = [av_case(535, 568), av_case(571, 604), av_case(607, 640), av_case(643, 676),
                     av_case(324, 347)]
G["add_variants"][-1]["src"] = "translate.rs:324-347 (doctest)"

# ---------------------------------------------------------------- gap_filling.rs
G["nearest_unique_context"] = [
    {"src": "gap_filling.rs:535-564", "k": 9,
     "reference": byte_vec(let("gap_filling.rs", 535, 564, "reference")),
     "query": byte_vec(let("gap_filling.rs", 535, 564, "query")),
     "search_range": [11, 16], "expected": [16, "CAGACAGCT"]},
]


def fg_case(a, b, k, threshold, p):
    return {"src": f"gap_filling.rs:{a}-{b}", "k": k, "threshold": threshold, "max_err_prob": p,
            "query": byte_vec(let("gap_filling.rs", a, b, "query")),
            "reference": byte_vec(let("gap_filling.rs", a, b, "reference")),
            "expected": char_vec(let("gap_filling.rs", a, b, "expected"))}


G["fill_gaps"] = [
    fg_case(641, 682, 7, 3, 0.001),
    fg_case(685, 726, 9, 3, 0.001),
    fg_case(729, 770, 9, 3, 0.001),
    fg_case(773, 814, 9, 3, 0.001),
    fg_case(817, 856, 9, 4, 0.001),
    fg_case(860, 891, 51, 23, 0.0000001),
    fg_case(894, 923, 31, None, 0.0000001),  # threshold from random_match_threshold(k, n_kmers, 4, p)
]
G["fill_gaps"].append({
    "src": "gap_filling.rs:403-441 (doctest)", "k": 9, "threshold": 4, "max_err_prob": 0.001,
    "query": "TTGATGTACAGACTGCGGAGAGCTG", "reference": "TTGATTAACAGGCTGCGCAGAGCTG",
    "expected": "MMMMMGTMMMMAMMMMMGMMMMMMM"})

with open(OUT, "w") as f:
    json.dump(G, f, indent=1)
print("wrote", OUT, {k: (len(v) if isinstance(v, list) else 1) for k, v in G.items()})
