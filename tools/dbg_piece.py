import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, kbo_amd
from kbo_amd import batch, synth
g = synth.genome(5_000_000)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
concat, offsets = synth.reads(g, 20000, 10000, 0.01)
dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
dev.run(); torch.cuda.synchronize()
n = dev.n_seqs; slots = dev.total // 132 + n
w = dev.dt_work.cpu().numpy().view(np.uint32)
words = n + 1 + (n + 1 + 1023) // 1024
redo = w[slots * 4 + words: slots * 4 + words + n]
print("flagged", int(redo.sum()), "of", n)
items = w[:slots*4].reshape(-1, 4)
print("items head", items[:3], "nonempty", int((items[:,2] > 0).sum()), "slots", slots)
for nm, f in (("walk", dev.walk), ("derand", dev.derand_translate)):
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(nm, min(ts))
