import os, sys, gc, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, kbo_amd, torch
from kbo_amd import batch, synth
MODE = sys.argv[1]
args = bench.parse(["--no-extras"])
g = synth.genome(args.genome)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
def leg(tag):
    r = bench.host_to_host_leg(args, sbwt, g)
    print(MODE, tag, r["value"], r["packed"]["value"], flush=True)
dev = torch.device("cuda:0")
if MODE == "early":
    leg("before anything")
S = torch.cuda.current_stream(dev)
devs = []
for b in range(2):
    concat, offsets = synth.reads(g, 1_000_000, 150, 0.01, seed=5 + b)
    devs.append(batch.DeviceBatch(sbwt, concat, offsets, device=dev, format=True, want_ms=False))
if MODE == "mid":
    leg("after DeviceBatch creation")
T = torch.cuda.Stream(dev)
bench.run_piped(devs, S, T, 50, torch)
torch.cuda.synchronize()
leg("at the end")
