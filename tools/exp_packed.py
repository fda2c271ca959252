"""The one kernel's packed-native form (kbo_matches_packed_dev: 2-bit words in, 2-bit words out) against its byte form
(kbo_map_batch_dev, format = 0) on the C2 batch, device-resident: time per step, one stream and two batches in flight."""
import os, sys, ctypes as C, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, ROOT)
import kbo_amd
from kbo_amd import batch, synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
device = torch.device("cuda:0")
g = synth.genome(G)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
L = kbo_amd.lib()
S, T = torch.cuda.Stream(device), torch.cuda.Stream(device)
pk, by = [], []
for b in range(2):
    concat, offsets = synth.reads(g, R, 150, 0.01, seed=100 + b)
    pk.append(batch.PackedDeviceBatch(sbwt, concat, offsets, device=device))
    by.append(batch.DeviceBatch(sbwt, concat, offsets, device=device, format=False, want_ms=False))
    if b == 0:
        from oracle import binding as ora
        rows, Carr, lcs = sbwt.export_parts()
        oi = ora.Index.from_parts(31, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
        want = oi.matches_batch(concat, offsets, 1e-7, n_threads=16)
        pk[0].run(S); torch.cuda.synchronize()
        print("packed-native equals the oracle on every read:", bool(np.array_equal(pk[0].chars(), want)), flush=True)


def timed(devs, piped, steps=40, warm=4):
    done = [None] * len(devs)

    def go(n):
        for i in range(n):
            b = i % len(devs)
            if piped:
                if done[b] is not None:
                    S.wait_event(done[b])
                devs[b].run(S, tail_stream=T)
                done[b] = done[b] or torch.cuda.Event()
                done[b].record(T)
            else:
                devs[b].run(S)
    go(warm)
    torch.cuda.synchronize()
    L.kbo_set_stage_timing(1)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(S)
    go(steps)
    S.wait_stream(T)
    e.record(S)
    torch.cuda.synchronize()
    ks, rs, n = C.c_double(), C.c_double(), C.c_int()
    L.kbo_stage_timing_read(C.byref(ks), C.byref(rs), C.byref(n))
    L.kbo_set_stage_timing(0)
    t = a.elapsed_time(e) / steps
    return "%.4f ms/step = %.1f Gbp/s (kernel %.4f, second pass %.4f)" % (t, devs[0].total / t / 1e6, ks.value / n.value, rs.value / n.value)


for name, devs in (("bytes ", by), ("packed", pk)):
    print(name, "one stream:", timed(devs, False), "| two in flight:", timed(devs, True), flush=True)
