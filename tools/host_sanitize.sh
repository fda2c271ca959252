#!/bin/bash
# The host side of libkbo_hip.so (index builder, path cover, flat files, C-ABI argument handling, pack helpers, the run
# automaton of kbo_call_batch, refinement code) under AddressSanitizer + UBSan (default) or ThreadSanitizer on the CPU:
# builds kbo_amd/libkbo_hip_host{asan,tsan}.so (make -C kbo_amd/csrc host-san) and runs the CPU test suite over it.
# GPU sanitizers are not available on the pool; what needs a device (the slab pipeline) is not covered by this.
# Usage: tools/host_sanitize.sh [asan|tsan] [pytest args...]
set -u
KIND=${1:-asan}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$KIND" = tsan ]; then SAN=thread; RT=$(gcc -print-file-name=libtsan.so); else SAN=address,undefined; RT=$(gcc -print-file-name=libasan.so); fi
make -s -C "$ROOT/kbo_amd/csrc" host-san SAN=$SAN SANNAME=$KIND || exit 1
# (python does not link libstdc++: without it in the preload list the runtime cannot intercept __cxa_throw)
export LD_PRELOAD="$RT $(gcc -print-file-name=libstdc++.so.6)"
export KBO_HIP_LIB="$ROOT/kbo_amd/libkbo_hip_host$KIND.so"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0"
cd "$ROOT" && python -m pytest tests -q -s -m "not gpu" -k "not dist_gloo" "$@"
