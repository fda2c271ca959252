"""When the batches of a short run through kbo_map_stream_* complete: 20 batches submitted at once behind a few warm-up ones, the host
waits for the tickets in order and notes the clock (the ramp of bench.py's --steps 20 line against its steady state).
python tools/ramp_timeline.py [steps] [warmup] [pipelines]"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ROOT)
import kbo_amd
from kbo_amd import batch, synth
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 5
PIPES = int(sys.argv[3]) if len(sys.argv) > 3 else 2
device = torch.device("cuda:0")
g = synth.genome(5_000_000)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
devs = []
for b in range(2 * PIPES):
    concat, offsets = synth.reads(g, 1_000_000, 150, 0.01, seed=100 + b)
    devs.append(batch.DeviceBatch(sbwt, concat, offsets, device=device, format=True, want_ms=False))
S = torch.cuda.Stream(device)
for d in devs:
    d.run(S)
torch.cuda.synchronize()
ms = batch.MapStream(sbwt, devs[0].n_seqs, devs[0].total, devs[0].max_len, pipelines=PIPES)
L = kbo_amd.lib()
for rep in range(3):
    if os.environ.get("STAGE"):  # bench.py's event records around every launch
        L.kbo_set_stage_timing(WARM + STEPS + 1)
    for w in range(WARM):
        ms.submit(devs[w % len(devs)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tickets = [ms.submit(devs[(WARM + s) % len(devs)]) for s in range(STEPS)]
    t_sub = time.perf_counter() - t0
    done = []
    for t in tickets:
        ms.wait(t)
        done.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    if os.environ.get("STAGE"):
        L.kbo_set_stage_timing(0)
        L.kbo_stage_timing_read(None, None, None)
    d = np.array(done) * 1e3
    print(f"rep {rep}: {STEPS} batches submitted in {t_sub * 1e3:.3f} ms, all complete at {total * 1e3:.3f} ms = {STEPS * 150 / total / 1e3:.0f} Gbp/s")
    print("  completion (ms):", " ".join(f"{x:.2f}" for x in d))
    print("  between completions:", " ".join(f"{x:.2f}" for x in np.diff(d)))
ms.close()
