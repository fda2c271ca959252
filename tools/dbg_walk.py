import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd
from kbo_amd import batch, synth
g = synth.genome(int(os.environ.get("G", 5_000_000)))
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
R = 1_000_000
concat, offsets = synth.reads(g, R, 150, 0.01)
dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
cnt = torch.zeros(16, dtype=torch.int32, device="cuda:0")
s = torch.cuda.current_stream()
kbo_amd.check(kbo_amd.lib().kbo_ms_batch_dev(sbwt._h, dev.q.data_ptr(), dev.off.data_ptr(), dev.n_seqs, dev.total,
              dev.ms.data_ptr(), None, cnt.data_ptr(), dev.work.data_ptr(), s.cuda_stream))
torch.cuda.synchronize()
c = cnt.cpu().numpy()
print("waves", c[3], "iters/wave", c[0] / c[3], "rare entries/wave", c[1] / c[3], "con passes/wave", c[2] / c[3])
