import os, sys, ctypes as C
import numpy as np, torch
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd
from kbo_amd import batch, synth
g = synth.genome(int(os.environ.get("G", 5_000_000)))
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
R = int(os.environ.get("R", 1_000_000))
concat, offsets = synth.reads(g, R, 150, float(os.environ.get("SUB", 0.01)))
dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
cnt = torch.zeros(16, dtype=torch.int32, device="cuda:0")
s = torch.cuda.current_stream()
kbo_amd.check(kbo_amd.lib().kbo_ms_batch_dev(sbwt._h, dev.q.data_ptr(), dev.off.data_ptr(), dev.n_seqs, dev.total,
              dev.max_len, dev.ms.data_ptr(), None, cnt.data_ptr(), dev.work.data_ptr(), dev.work_bytes, s.cuda_stream))
torch.cuda.synchronize()
c = cnt.cpu().numpy()
c = c.astype(np.int64)
iters = c[0] / c[3]
print("waves", c[3], "iters/wave %.1f" % iters, "rare entries/wave %.1f" % (c[1] / c[3]))
lanes = c[3] * 64
print("per lane: accepted %.1f  failed-extends %.1f  contraction-levels %.1f  idle %.1f  (of %.1f iterations)" %
      (c[4] / lanes, c[5] / lanes, c[6] / lanes, iters - (c[4] + c[5] + c[6]) / lanes, iters))
print("idle per lane: waiting for the item switch %.1f, out of items %.1f, other %.1f" %
      (c[7] / lanes, c[8] / lanes, iters - (c[4] + c[5] + c[6] + c[7] + c[8]) / lanes))
