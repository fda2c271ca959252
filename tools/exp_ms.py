"""The MS-emitting entry points at C2's shape, one batch at a time on one stream and four in flight through kbo_map_stream_*
(kbo_map_batch_dev with want_ms: MS bytes + characters; kbo_ms_batch_dev: MS bytes).  CHECK=1: every byte against the oracle.
python tools/exp_ms.py [genome] [reads]"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ROOT)
import kbo_amd
from kbo_amd import batch, synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
device = torch.device("cuda:0")
g = synth.genome(G)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
concat, offsets = synth.reads(g, R, 150, float(os.environ.get("SUB", "0.01")), seed=100)
devs = [batch.DeviceBatch(sbwt, concat, offsets, device=device, format=True, want_ms=True) for _ in range(4)]
S = torch.cuda.Stream(device)
total = devs[0].total


def timed(fn, n=40, warm=150):
    fn(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def serial(n):
    for i in range(n):
        devs[i % 4].run(S)


def walk(n):
    for i in range(n):
        devs[i % 4].walk(S)


ms = batch.MapStream(sbwt, devs[0].n_seqs, total, devs[0].max_len, pipelines=2)


def piped(n):
    for i in range(n):
        ms.submit(devs[i % 4])


for name, fn in (("kbo_map_batch_dev(want_ms), one at a time", serial), ("kbo_ms_batch_dev, one at a time", walk), ("kbo_map_stream (want_ms), four in flight", piped)):
    t = timed(fn)
    print(f"{name}: {t:.4f} ms = {total / t / 1e6:.0f} Gbp/s")
print("flagged reads:", int((devs[0].plan_flags() != 0).sum()), "fused:", devs[0].fused)
if os.environ.get("CHECK"):
    from oracle import binding as ora
    oi = ora.Index.build([g.tobytes()], k=31)
    exp_chars, exp_d = oi.matches_batch(concat, offsets, 1e-7, n_threads=16, want_d=True)
    exp_map = np.frombuffer(ora.relative_to_ref(concat, exp_chars), dtype=np.uint8)
    for d in devs:
        assert np.array_equal(d.ms[:total].cpu().numpy(), exp_d), "MS differ"
        assert np.array_equal(d.chars[:total].cpu().numpy(), exp_map), "characters differ"
    print("every MS byte and character equal to the oracle")
ms.close()
