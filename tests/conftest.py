import json
import os
import sys

import pytest

# (the suite keeps several streams busy - pipelines, tail streams, torch's own: more hardware queues than the runtime's four, asked for by
# the application, here the test session, before anything touches the GPU: INTEGRATION.md "Streams and hardware queues")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "kbo_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding


def shipped_defaults(L):
    """Every process-wide knob of include/kbo_hip_tuning.h back to what the library ships with."""
    L.kbo_set_plan(1, -1, 64)            # plan on, seed depth automatic (log4(rows) + 3), seed search over 64 bases
    L.kbo_set_plan_tuning(-1, 32, 50)    # unit gap automatic (log4(rows) + 9), chunks of 32, bail-out at 50 / 16 units per read
    L.kbo_set_plan_unit_cap_divisor(1)
    L.kbo_set_plan_stats(0)
    L.kbo_set_index_shards(0)
    L.kbo_set_depth_table(0)             # depth table by index size
    L.kbo_set_depth_table_anchors(-1)    # ... its anchors by its margin
    L.kbo_set_guided_walk(0, -1)         # resident waves and rank blocks / recovery lines by index size
    L.kbo_set_force_big_layout(0)
    L.kbo_set_seed_table_depth(0)
    L.kbo_set_pair_steps(24 << 20, 16)
    L.kbo_set_walk_waves_per_cu(0)
    L.kbo_set_slab_bytes(16 << 20)
    L.kbo_set_devices(None, 0)
    L.kbo_set_plan_table_budget(0)       # the tables of a copy: half of the free device memory
    L.kbo_set_plan_lazy(-1)              # plan structures of implicitly made copies: by index size
    L.kbo_set_map_long(1)                # sequences of more than 160 bases: the one kernel where it applies
    L.kbo_set_ms_one_kernel(1)           # kbo_ms_batch_dev over reads: the one kernel's MS-emitting form


@pytest.fixture(autouse=True)
def _shipped_defaults_around_every_test():
    """Tests that force a knob through its corner must not leak the setting into the tests that run after them (round 2
    left seed depth 14 / cap 40 behind for most of the suite): every test starts from, and leaves, the shipped defaults."""
    import kbo_amd
    try:
        L = kbo_amd.lib()
    except Exception:
        yield
        return
    shipped_defaults(L)
    yield
    shipped_defaults(L)
