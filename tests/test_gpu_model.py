"""The CPU model of the plan-guided stage (oracle/plan_model.c) pinned to the kernels: on the same reads, with the shipped
parameters, the model's counts of units, accepted / failed extensions, contraction levels, entry levels, seed-table
look-ups, seed extensions and mismatches must EQUAL the counters the kernels keep about themselves
(kbo_plan_stats_dev), in both forms of the guided walk.  bench.py prices the stage's roofline by the model's counts."""
import numpy as np
import pytest

import kbo_amd
from kbo_amd import batch, synth
from gpu_helpers import adopt, threads

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def guided_walk_only():
    """these tests are about the units and the guided walk: no depth table (tests/test_gpu_dtab.py has that form)"""
    kbo_amd.lib().kbo_set_depth_table(-1)


def _compare(oracle, sbwt, ora, concat, offsets, fat):
    import torch
    L = kbo_amd.lib()
    L.kbo_set_plan_stats(1)  # (the counting instantiations of the kernels; off by default)
    L.kbo_set_guided_walk(0, fat)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
    dev.ms.fill_(0xEE)
    dev.walk()
    torch.cuda.synchronize()
    st = dev.plan_stats()
    P = oracle.shipped_plan_params(sbwt.k(), sbwt.n_sets(), recovery_lines=fat, depth_table=0)
    ms, cn = ora.plan_model(sbwt.path_cover(), P, concat, offsets, n_threads=threads())
    assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), ms)
    assert st["gave_up"] == cn["gave_up"] and st["guard"] == 0
    want = {"units": cn["units_counted"], "seed_lookups": cn["seed_lookups"], "seed_extensions": cn["seed_extensions"],
            "mismatches": cn["mismatches"]}
    if not cn["gave_up"]:
        want.update({"units_walked": cn["units"], "accepted": cn["walk_accepted"], "failed": cn["walk_failed"],
                     "levels": cn["walk_contractions"], "entry_levels": cn["walk_entry_levels"]})
    got = {k: st[k] for k in want}
    assert got == want, (fat, got, want)
    return cn


@pytest.mark.parametrize("fat", [0, 1])
def test_model_counts_equal_the_kernels_counters(oracle, fat):
    g = synth.genome(2_000_000, seed=2024)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = adopt(oracle, sbwt)
    for sub, seed in ((0.01, 1), (0.03, 2), (0.0, 3)):
        concat, offsets = synth.reads(g, 60_000, 150, sub, seed=seed)
        cn = _compare(oracle, sbwt, ora, concat, offsets, fat)
        assert cn["items_flagged"] < 60
    # reads from elsewhere: nothing seeds, the plan is given up on the device, the model says so too
    other = synth.genome(300_000, seed=77)
    concat, offsets = synth.reads(other, 20_000, 150, 0.0, seed=4)
    kbo_amd.lib().kbo_set_plan(1, 0, 0)
    cn = _compare(oracle, sbwt, ora, concat, offsets, fat)
    assert cn["gave_up"] == 1


def test_model_counts_on_a_repeat_rich_multi_contig_index(oracle):
    rng = np.random.default_rng(9)
    g = synth.genome(300_000, seed=31)
    contigs = [g[:120_000].tobytes(), g[100_000:220_000].tobytes(), np.tile(g[250_000:251_000], 20).tobytes(),
               g[220_000:].tobytes() + b"NNN" + g[5_000:9_000].tobytes()]
    sbwt, _ = kbo_amd.build(contigs, kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = adopt(oracle, sbwt)
    cat = np.frombuffer(b"".join(contigs), dtype=np.uint8)
    starts = rng.integers(0, len(cat) - 150, 30_000)
    reads = np.stack([cat[a:a + 150] for a in starts]).copy()
    hit = rng.random(reads.shape) < 0.01
    reads[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(hit.sum()))]
    concat = reads.reshape(-1)
    offsets = np.arange(len(starts) + 1, dtype=np.uint64) * np.uint64(150)
    for fat in (0, 1):
        kbo_amd.lib().kbo_set_plan(1, 0, 0)
        _compare(oracle, sbwt, ora, concat, offsets, fat)
