"""The one-kernel route of kbo_map_batch_dev against the two-kernel route on the C2 batch: parity with the oracle, time per
step of both (same DeviceBatch, events on the stream).  python tools/exp_map.py [genome] [reads] [sub_rate] [format]"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kbo_amd  # noqa: E402
from kbo_amd import batch, synth  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
SUB = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
FMT = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
CHECK = int(os.environ.get("CHECK", "1"))
device = torch.device("cuda:0")
g = synth.genome(G)
t0 = time.time()
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
concat, offsets = synth.reads(g, R, 150, SUB)
dev = batch.DeviceBatch(sbwt, concat, offsets, device=device, format=FMT, want_ms=False)
print("index + copy %.1f s; layout %s" % (time.time() - t0, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in sbwt.device_layout().items()}), flush=True)
stream = torch.cuda.current_stream(device)


def timed(fn, steps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(steps):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def two():
    dev.walk(stream)
    dev.derand_translate(stream)


if CHECK:
    from oracle import binding as ora
    rows, Carr, lcs = sbwt.export_parts()
    oi = ora.Index.from_parts(31, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
    exp_chars, exp_d = oi.matches_batch(concat, offsets, 1e-7, n_threads=16, want_d=True)
    want = np.frombuffer(ora.relative_to_ref(concat, exp_chars), dtype=np.uint8) if FMT else exp_chars
    dev.chars.fill_(0xEE)
    dev.run()
    torch.cuda.synchronize()
    got = dev.chars[:dev.total].cpu().numpy()
    bad = np.flatnonzero(got != want)
    print("one kernel: fused", dev.fused, "bad bases", len(bad), "of", dev.total, flush=True)
    if len(bad):
        s = int(bad[0]) // 150
        print(" read", s, "got ", got[150 * s:150 * s + 150].tobytes())
        print(" read", s, "want", want[150 * s:150 * s + 150].tobytes())
        print(" ms", list(exp_d[150 * s:150 * s + 150]))
    st = np.frombuffer(dev.work.cpu().numpy().tobytes(), dtype=np.uint8)
    dev.chars.fill_(0xEE)
    two()
    torch.cuda.synchronize()
    got2 = dev.chars[:dev.total].cpu().numpy()
    print("two kernels: bad bases", int((got2 != want).sum()), flush=True)
t_one = timed(dev.run)
t_two = timed(two)
print("one kernel %.4f ms/step = %.1f Gbp/s; two kernels %.4f ms/step = %.1f Gbp/s" % (t_one, dev.total / t_one / 1e6, t_two, dev.total / t_two / 1e6))
