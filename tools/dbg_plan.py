"""Iteration counters of the guided walk (needs the counters build: make -C kbo_amd/csrc debug;
KBO_HIP_LIB=kbo_amd/libkbo_hip_dbg.so python tools/dbg_plan.py)."""
import os, sys
import numpy as np, torch
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd
from kbo_amd import batch, synth
g = synth.genome(int(os.environ.get("G", 5_000_000)))
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
R = int(os.environ.get("R", 1_000_000))
L = kbo_amd.lib()
for sub in [float(x) for x in os.environ.get("SUB", "0.01").split(",")]:
    concat, offsets = synth.reads(g, R, int(os.environ.get("LEN", 150)), sub)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
    for rare in [int(x) for x in os.environ.get("RARE", "8").split(",")]:
        L.kbo_set_walk_rare(rare)
        cnt = torch.zeros(16, dtype=torch.int32, device="cuda:0")
        s = torch.cuda.current_stream()
        kbo_amd.check(L.kbo_ms_batch_dev(sbwt._h, dev.q.data_ptr(), dev.off.data_ptr(), dev.n_seqs, dev.total,
                      dev.max_len, dev.ms.data_ptr(), None, cnt.data_ptr(), dev.work.data_ptr(), dev.work_bytes, s.cuda_stream))
        torch.cuda.synchronize()
        c = cnt.cpu().numpy().astype(np.int64)
        waves, lanes = c[3], c[3] * 64
        it = c[0] / waves
        print(f"sub={sub} rare={rare}: waves {waves}, hot iterations per wave avg {it:.1f} max {c[13]}, bookkeeping visits {c[1]/waves:.1f}, units {c[12]} ({c[12]/R:.2f} per read), items flagged {c[7]}")
        print("   per lane-slot: accept %.1f fail %.1f contract %.1f | waiting for switch %.1f, out of units %.1f | of %.1f" %
              (c[4]/lanes, c[5]/lanes, c[6]/lanes, c[8]/lanes, c[9]/lanes, it))
        print("   per unit: accept %.1f fail %.1f contract %.1f (of which windows too short: %.2f)" % (c[4]/max(c[12],1), c[5]/max(c[12],1), c[6]/max(c[12],1), c[14]/max(c[12],1)))
