"""Two batches in flight: batch i's second pass (kbo_map_batch_dev_tail) on a second stream beside batch i+1's kernel, against
one batch after the other on one stream.  python tools/exp_overlap.py [genome] [reads]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, ROOT)
import kbo_amd
from kbo_amd import batch, synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
NB = int(os.environ.get("NB", "2"))
device = torch.device("cuda:0")
g = synth.genome(G)
if os.environ.get("PAIR"):
    kbo_amd.lib().kbo_set_pair_steps(0, int(os.environ["PAIR"]))  # two-base steps of the plain walk on this small index too
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
devs = []
for b in range(NB):
    if os.environ.get("INDEL"):  # bench.py's indel variant: 1 % substitutions + INDEL per base start an insertion / deletion of 1 - 3 bases
        import bench
        concat, offsets = bench.indel_reads(g, R, 150, 0.01, float(os.environ["INDEL"]), seed=0x5E11C + b)
    else:
        concat, offsets = synth.reads(g, R, 150, float(os.environ.get("SUB", "0.01")), seed=100 + b)
    devs.append(batch.DeviceBatch(sbwt, concat, offsets, device=device, format=True, want_ms=False))
PRI = int(os.environ.get("PRI", "0"))
NT = int(os.environ.get("NT", "1"))
S = torch.cuda.Stream(device)
Ts = [torch.cuda.Stream(device, priority=PRI) for _ in range(NT)]
T = Ts[0]
done = [torch.cuda.Event() for _ in devs]


def serial(steps):
    for i in range(steps):
        devs[i % NB].run(S)


NS = int(os.environ.get("NS", "1"))  # kernel streams (NS = 2 with NB = 3: the next kernel starts while this one drains)
Ss = [S] + [torch.cuda.Stream(device) for _ in range(NS - 1)]


def piped(steps):
    for i in range(steps):
        b = i % NB
        s = Ss[i % NS]
        if not os.environ.get("NOWAIT"):  # (NOWAIT=1, measurement only: what the cross-stream wait in front of every kernel costs)
            s.wait_event(done[b])  # the batch's buffers are free again once its last second pass is through
        devs[b].run(s, tail_stream=Ts[i % NT])
        done[b].record(Ts[i % NT])
    for s in Ss[1:]:
        S.wait_stream(s)


def timed(fn, steps=40, warm=4):
    fn(warm)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(S)
    import time as _t
    t0 = _t.perf_counter()
    fn(steps)
    global ENQ
    ENQ = (_t.perf_counter() - t0) / steps * 1e3
    for t in Ts:
        S.wait_stream(t)
    e.record(S)
    torch.cuda.synchronize()
    return a.elapsed_time(e) / steps


for d in devs:
    d.run(S)
torch.cuda.synchronize()
ref = [d.chars[:d.total].clone() for d in devs]
print("flagged reads per batch:", [int((d.plan_flags() != 0).sum()) for d in devs])
import ctypes as C
L = kbo_amd.lib()


def staged(fn):
    L.kbo_set_stage_timing(1)
    t = timed(fn)
    a, b, n = C.c_double(), C.c_double(), C.c_int()
    L.kbo_stage_timing_read(C.byref(a), C.byref(b), C.byref(n))
    L.kbo_set_stage_timing(0)
    print("  kernel %.4f ms, second pass %.4f ms (from its start on its stream) over %d calls" % (a.value / n.value, b.value / n.value, n.value))
    return t


ts = staged(serial)
tp = staged(piped)
ok = all(bool((d.chars[:d.total] == r).all()) for d, r in zip(devs, ref))
print("host enqueue of the last loop %.4f ms per step" % ENQ)
print("serial %.4f ms/step = %.1f Gbp/s; two in flight %.4f ms/step = %.1f Gbp/s; equal %s" % (ts, devs[0].total / ts / 1e6, tp, devs[0].total / tp / 1e6, ok))
