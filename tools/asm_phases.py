#!/usr/bin/env python3
"""Static instruction counts of map_long_kernel by phase: compiles kbo_amd/csrc/long_kernels.hip with assembler comments in front
of its phases (the `// ---- n.` comments and a few others) and counts what lies between them in the gfx950 assembly - vector, scalar,
LDS, memory, 64-bit vector, and lane instructions (v_readlane / v_writelane: mostly spilled scalar registers).  Control flow makes this
a map, not a profile.  python tools/asm_phases.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "kbo_amd", "csrc", "long_kernels.hip")
TMP = "/tmp/kbo_asm"
MARKS = [("        // ---- 0. the region's 2-bit digits", "STAGE"), ("        // ---- 1. stretches", "STRETCHES"),
         ("        // the piece that goes on with this sequence", "PREDICT"), ("        // ---- 2. the planes -> G, cov, characters, U", "PLANES"),
         ("        // ---- 3. the proof", "POINTS"), ("            __builtin_amdgcn_wave_barrier(); // (the list goes where the text was)", "LIST"),
         ("            for (uint32_t i0 = 0; !flag && i0 < total; i0 += 64u) {", "LOOKUPS"), ("        finish_round();\n        pred_in_lds = tx_next;\n        // ---- 4.", "FINISH"),
         ("        // ---- 4. the characters of the own bases", "OUTPUT"), ("        // x of the first own base, 0 .. k (what derandomize_ms_vec gives there", "XIN")]


def main():
    os.makedirs(TMP, exist_ok=True)
    s = open(SRC).read()
    for anchor, name in MARKS:
        if anchor not in s:
            print("marker not found:", name)
            continue
        s = s.replace(anchor, '        asm volatile("; MARK_%s");\n' % name + anchor, 1)
    open(os.path.join(TMP, "long_mark.hip"), "w").write(s)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.dirname(SRC), "--offload-arch=gfx950", "-save-temps",
                    "-c", "long_mark.hip", "-o", "long_mark.o"], cwd=TMP, stderr=subprocess.DEVNULL, check=True)
    lines = open(os.path.join(TMP, "long_mark-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith("_ZN3kbo12_GLOBAL__N_115map_long_kernel")][0]
    end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
    cur, counts, order = "PROLOGUE", {}, ["PROLOGUE"]
    for l in lines[start:end]:
        m = re.search(r"; MARK_(\w+)", l)
        if m:
            cur = m.group(1)
            order.append(cur)
            continue
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        c = counts.setdefault(cur, dict(vector=0, scalar=0, lds=0, mem=0, v64=0, lane=0))
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            c["lane"] += 1
        elif op.startswith("v_"):
            c["vector"] += 1
            c["v64"] += ("b64" in op or "u64" in op)
        elif op.startswith("s_"):
            c["scalar"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            c["mem"] += 1
    for k in order:
        if k in counts:
            print("%-10s" % k, " ".join("%s %4d" % kv for kv in counts[k].items()))
    print("%-10s" % "all", " ".join("%s %4d" % (f, sum(c[f] for c in counts.values())) for f in ("vector", "scalar", "lds", "mem", "v64", "lane")))


if __name__ == "__main__":
    sys.exit(main())
