"""bench.py's host logic, no GPU: presets and the metric's config, the sharding of reads over ranks (weak for C2 / C3,
strong for C4), the index cache (built once on the host, path cover included, loaded by the other ranks), and the command
the parent of `bench.py --gpus N` starts its ranks with."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_presets_and_sharding():
    b = _bench()
    a = b.parse([])
    assert (a.genome, a.reads, a.read_len, a.k, a.find, a.scaling, a.gpus) == (5_000_000, 1_000_000, 150, 31, False, "weak", 1)
    a = b.parse(["--config", "C3"])
    assert (a.genome, a.reads, a.find, a.scaling) == (100_000_000, 10_000_000, True, "weak")
    a = b.parse(["--config", "C4", "--gpus", "8"])
    assert (a.genome, a.reads, a.find, a.scaling) == (250_000_000, 100_000_000, False, "strong")
    for world in (1, 2, 3, 4, 8):  # strong scaling: the shards tile the 100 M reads exactly
        parts = [b.shard(a, r, world) for r in range(world)]
        assert sum(n for n, _ in parts) == a.reads
        assert all(parts[r][1] == sum(n for n, _ in parts[:r]) for r in range(world))
    a = b.parse(["--gpus", "4"])
    assert [b.shard(a, r, 4) for r in range(4)] == [(1_000_000, r * 1_000_000) for r in range(4)]  # weak: fixed per rank


def test_index_cache_round_trip_and_spawn_command(tmp_path):
    b = _bench()
    cache = str(tmp_path / "idx.kbohip")
    argv = ["--gpus", "2", "--genome", "60000", "--reads", "500", "--index-cache", cache]
    a = b.parse(argv)
    g1, s1 = b.build_or_load_index(a, 2)            # builds and writes the cache (with the path cover)
    assert os.path.exists(cache) and s1.n_sets() == 60_001
    g2, s2 = b.build_or_load_index(a, 2, may_build=False)  # what the other ranks do
    assert np.array_equal(g1, g2)
    p1, p2 = s1.export_parts(), s2.export_parts()
    assert all((x == y).all() for x, y in zip(p1[0], p2[0])) and p1[1] == p2[1] and (p1[2] == p2[2]).all()
    assert all((x == y).all() for x, y in zip(s1.path_cover(), s2.path_cover()))
    cmd = b.spawn_command(a, argv, 29512)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-len(argv):] == argv                 # (the cache is already named: nothing appended)
    a2 = b.parse(["--gpus", "2"])
    cmd2 = b.spawn_command(a2, ["--gpus", "2"], 1)
    assert cmd2[-2] == "--index-cache" and cmd2[-1].endswith("kbo_bench_iid_5000000_k31.kbohip")
