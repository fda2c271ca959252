import os, sys, gc, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, kbo_amd, torch
from kbo_amd import batch, synth
args = bench.parse(["--no-extras"])
g = synth.genome(args.genome)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
sbwt.to_device()
def leg(tag):
    r = bench.host_to_host_leg(args, sbwt, g)
    print(tag, r["value"], r["packed"]["value"], flush=True)
dev = torch.device("cuda:0")
S = torch.cuda.current_stream(dev)
devs = []
for b in range(2):
    concat, offsets = synth.reads(g, 1_000_000, 150, 0.01, seed=5 + b)
    devs.append(batch.DeviceBatch(sbwt, concat, offsets, device=dev, format=True, want_ms=False))
leg("two resident batches")
for d in devs:
    d.run(S)
torch.cuda.synchronize()
leg("after serial runs")
T = torch.cuda.Stream(dev)
leg("after creating a torch side stream")
bench.run_piped(devs, S, T, 50, torch)
torch.cuda.synchronize()
leg("after 50 piped steps")
del T
gc.collect(); torch.cuda.empty_cache()
leg("after dropping the stream")
kbo_amd.lib().kbo_release_scratch()
leg("after kbo_release_scratch")
