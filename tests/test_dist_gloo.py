"""Multi-rank path on CPU (gloo, world_size 2): the read sharding used by bench.py and the
max-over-ranks timing reduce.  The data path has no collective (index replicated, reads
sharded); the shards of different ranks must tile the read set exactly."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from kbo_amd import synth
from oracle import binding as ora

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = synth.genome(50_000)
n_per = 400
# same call bench.py makes: rank r owns reads [r*n_per, (r+1)*n_per)
concat, offsets = synth.reads(g, n_per, 100, 0.02, first_read=rank * n_per)
full, _ = synth.reads(g, n_per * world, 100, 0.02)
assert np.array_equal(concat, full[rank * n_per * 100:(rank + 1) * n_per * 100])
# every rank computes its shard independently (CPU oracle stands in for the GPU here)
oi = ora.Index.build([g.tobytes()], k=31)
chars = oi.matches_batch(concat, offsets, 1e-7, n_threads=1)
digest = torch.tensor([int(chars.astype(np.uint64).sum()), len(chars)], dtype=torch.int64)
gathered = [torch.zeros_like(digest) for _ in range(world)]
dist.all_gather(gathered, digest)
t = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py's max-over-ranks step time
dist.barrier()
if rank == 0:
    whole = oi.matches_batch(full, np.arange(n_per * world + 1, dtype=np.uint64) * 100, 1e-7, n_threads=2)
    print(json.dumps({"sum": int(sum(int(x[0]) for x in gathered)), "n": int(sum(int(x[1]) for x in gathered)),
                      "whole_sum": int(whole.astype(np.uint64).sum()), "whole_n": len(whole), "tmax": float(t.item())}))
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_world2_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["n"] == r["whole_n"] and r["sum"] == r["whole_sum"]  # shards tile the job exactly
    assert r["tmax"] == 2.0
