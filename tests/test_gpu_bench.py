"""bench.py end to end on the GPU box, at reduced size: the single-process line (kbo_map_batch_dev's one kernel priced by its own
counters; with --two-kernels / without a depth table the round-3 route priced by the stage model; CPU baseline, sensitivity
variants, host-to-host rate) and `--gpus 2` started from a plain `python bench.py` (the parent builds
the index cache on the host and spawns the ranks; with KBO_BENCH_ONE_GPU=1 both ranks share cuda:0 over gloo - a functional
test of the multi-process path, its numbers mean nothing)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, tmp_path, extra_env=None):
    env = dict(os.environ, KBO_BENCH_CACHE_DIR=str(tmp_path), **(extra_env or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_line(r, n_gpus, table=True, one_kernel=True):
    assert r["n_gpus"] == n_gpus and r["unit"] == "Mbp/s" and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["bit_exact_vs_oracle"] is True
    ro = r["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert 0 < ro["frac"] <= 1.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and len(cb["runs_mbps"]) == 3
    assert r["config"]["setup_seconds"]["total"] > 0 and r["config"]["index_device_bytes"]["total"] > 0
    if one_kernel:  # kbo_map_batch_dev: the kernel's own counters price it, its duration comes from events inside the library
        assert ro["kernel"].startswith("map_reads_kernel") and "one kernel" in r["config"]["walk"]
        st = ro["stage_counters_gpu_first_slab"]
        assert st["tab_lookups"] > 0 and st["seed_lookups"] > 0 and st["mismatches"] > 0
        assert 0 < ro["kernel_ms"] <= r["kernels_ms"]["kbo_map_batch_dev"] * 1.05 and ro["redo_pass_ms"] > 0
        assert 2.0 < ro["algorithmic_bytes_per_base"] < 4.0 and ro["build_sha16"]
        assert ro["cross_check_whole_step_gbps"] <= ro["peak"]
        if n_gpus == 1:  # four batches in flight on two pipelines; the same steps on one stream beside them
            assert str(r["config"]["batches_in_flight"]).startswith("4 on 2 pipeline") and r["one_batch_at_a_time"]["value"] > 0
            # two launches share the device: the bytes are priced by the device's time per launch, never less than half a launch's duration
            assert ro["launches_sharing_the_device"] == 2 and ro["device_ms_per_launch"] >= ro["kernel_ms"] / 2 - 1e-4
            assert ro["device_ms_per_launch"] >= r["ms_per_step"] - 1e-3
            assert 0 < ro["per_launch_duration"]["frac"] <= 1.0 and 0 < ro["alone"]["frac"] <= 1.0 and ro["alone"]["kernel_ms"] > 0
        return
    assert ro["stage_model"]["ms_equal_to_gpu"] is True
    # the model's counts are the kernels' own (first slab)
    st, m = ro["stage_counters_gpu_first_slab"], ro["stage_model"]
    if table:  # a small index: the stretches behind mismatches come from the depth table
        assert "depth table" in r["config"]["walk"] and m["parameters"]["depth_table"] > 0 and st["units"] == 0
        assert st["tab_lookups"] > 0 and 8 < m["per_mismatch"]["table_lookups"] < 16 and st["tab_flagged"] == st["tab_unresolved"]
    else:
        assert abs(st["units"] / st["units_walked"] - 1) < 1e-9 and m["units_per_read"] > 1.0
    assert ro["cross_check_whole_step_gbps"] <= ro["peak"]


def test_bench_line_two_kernels(tmp_path):
    """--two-kernels: round 3's route (plan_kernel with the table look-ups, then the derandomize / translate kernel), priced by the model"""
    r = _run(["--genome", "400000", "--reads", "30000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1", "--no-extras", "--two-kernels"], tmp_path)
    _check_line(r, 1, one_kernel=False)


def test_bench_line_single_process_with_extras(tmp_path):
    r = _run(["--genome", "400000", "--reads", "30000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1.5", "--extras"], tmp_path)
    _check_line(r, 1)
    names = [v["variant"] for v in r["sensitivity"]]
    assert len(names) == 8 and any("reverse" in n for n in names) and any("repeat" in n for n in names) and any("10 kbp reads, 1%" in n for n in names)
    assert any("insertions / deletions" in n and "ONT" not in n for n in names) and any("ONT-like" in n for n in names)
    assert all(v["bit_exact_vs_oracle"] is True and v["value"] > 0 for v in r["sensitivity"])
    assert r["host_to_host"]["value"] > 0
    # the reference's call shape: one ref_seq against an index built for it, end to end (build + device copy + the call)
    osm = r["one_shot_map"]
    assert osm["bit_exact_vs_oracle"] is True and osm["same_output_with_plan_structures"] is True and osm["build_s"] > 0 and osm["first_map_s"] > 0
    # the forms of the API that return the matching statistics: timed, every MS byte against the oracle
    for key in ("kbo_ms_batch_dev", "kbo_map_batch_dev_want_ms"):
        assert r["ms_variant"][key]["value"] > 0 and r["ms_variant"][key]["bit_exact_vs_oracle"] is True


def test_bench_line_guided_walk(tmp_path):
    """--depth-table -1: units and the guided walk (what larger indexes get), priced by the same model"""
    r = _run(["--genome", "400000", "--reads", "30000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1", "--no-extras", "--depth-table", "-1"],
             tmp_path)
    _check_line(r, 1, table=False, one_kernel=False)
    assert "guided walk" in r["config"]["walk"]


def test_bench_c4_shape_strong_scaling_two_ranks(tmp_path):
    """--config C4 at reduced size: the reads are a fixed set, sharded over the ranks (strong scaling)"""
    r = _run(["--config", "C4", "--gpus", "2", "--genome", "400000", "--reads", "50001", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"],
             tmp_path, {"KBO_BENCH_ONE_GPU": "1"})
    _check_line(r, 2)
    assert r["scaling"] == "strong" and "50001 x 150 bp reads in all, 25001 per GPU" in r["config"]["workload"]


def test_bench_c4_shape_eight_ranks(tmp_path):
    """the launch the driver makes on an 8-GPU node, executed once with world = 8 (all ranks on this box's one GPU, rendezvous over
    gloo: functional only): --config C4's strong-scaling shards - contiguous, ceil(reads / 8) each, the last one short, tiling the
    read set exactly - the index cache built once by the parent, the CPU quota split cores // world"""
    r = _run(["--config", "C4", "--gpus", "8", "--genome", "300000", "--reads", "240003", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.5", "--no-extras"],
             tmp_path, {"KBO_BENCH_ONE_GPU": "1"})
    _check_line(r, 8)
    assert r["scaling"] == "strong" and "240003 x 150 bp reads in all, 30001 per GPU" in r["config"]["workload"] and "x8" in r["config"]["parallelism"]
    lo, hi = r["roofline"]["kernel_ms_per_rank"]["min"], r["roofline"]["kernel_ms_per_rank"]["max"]
    assert 0 < lo <= hi


def test_bench_gpus_2_spawns_two_ranks(tmp_path):
    r = _run(["--gpus", "2", "--genome", "400000", "--reads", "30000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1.5"],
             tmp_path, {"KBO_BENCH_ONE_GPU": "1"})
    _check_line(r, 2)
    assert r["scaling"] == "weak" and "x2" in r["config"]["parallelism"]
    assert os.path.exists(os.path.join(str(tmp_path), "kbo_bench_iid_400000_k31.kbohip"))  # built once by the parent, loaded by the ranks
    lo, hi = r["roofline"]["kernel_ms_per_rank"]["min"], r["roofline"]["kernel_ms_per_rank"]["max"]
    assert 0 < lo <= hi


def test_bench_call_line(tmp_path):
    """--call: the first pass of kbo call timed on the device (sites of every read against the oracle's first pass), and the whole
    kbo_call_batch beside it (a sample of reads against the oracle's literal call)"""
    r = _run(["--call", "--genome", "2000000", "--reads", "600", "--k", "51", "--steps", "2", "--warmup", "1"], tmp_path)
    assert r["bit_exact_vs_oracle"] is True and r["value"] > 0 and r["kernels_ms"]["ms_walk_call_mode"] > 0
    w = r["whole_call"]
    assert w["entry_point"] == "kbo_call_batch_flat" and w["variants"] > 600 and w["us_per_read"] > 0 and w["equal_to_oracle_call_on_sampled_reads"] == 40
    assert w["kbo_call_batch_same_offsets"] is True and w["phases_of_one_call_host_clock_ms"]["slabs"] >= 1


def test_bench_c5_line_at_reduced_size(tmp_path):
    """--config C5 with a small genome: the line's shape - roofline of the first pass by the oracle's counts, cpu_baseline = oracle.call on a
    sample - and its checks: the sites of EVERY read (300 of 10 kbp, k = 63) against the oracle's first pass, 40 reads of kbo_call_batch
    against the oracle's literal kbo::call"""
    r = _run(["--config", "C5", "--genome", "3000000", "--reads", "300", "--steps", "2"], tmp_path)
    assert r["bit_exact_vs_oracle"] is True and r["value"] > 0 and "k=63" in r["metric"] and "kbo call" in r["metric"]
    ro, cb = r["roofline"], r["cpu_baseline"]
    assert ro["bound"] == "hbm" and 0 < ro["frac"] < 1 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and ro["units_per_launch"] == 300 * 10000
    # (priced by the walk's own bytes, counted by its kernels; the reference algorithm's bytes beside it)
    assert sum(ro["bytes_by_part"].values()) == round(ro["algorithmic_bytes_per_base"] * 3_000_000, -3) or abs(
        sum(ro["bytes_by_part"].values()) / 3_000_000 - ro["algorithmic_bytes_per_base"]) < 1e-2
    assert ro["frac_reference_algorithm"] > 0 and ro["walk_counters"]["units"] > 0
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "oracle" in cb["sample"]
    assert r["whole_call"]["equal_to_oracle_call_on_sampled_reads"] == 40 and r["config"]["sites_per_step"] > 300


def test_bench_c5_at_its_real_size(tmp_path):
    """bench.py --config C5 as it is: the 3 Gbp index, k = 63, 125 000 reads of 10 kbp - every site of the first pass against the oracle's,
    40 sampled reads of the whole call (kbo_call_batch_flat) against the oracle's literal kbo::call.  Runs wherever the host has the
    memory (MemAvailable >= 300 GB: the index, its device layout and the oracle's adopted parts) - no environment gate since round 6 -
    under a time budget: the index build on the box's 16 CPUs' worth of time is what takes minutes, and a box that needs more than
    KBO_TEST_C5_BUDGET seconds (default 640; the round's measured run: profiles/r06_c5_phases.txt) is skipped with its phases so far
    rather than allowed to run the suite out of the driver's time.  KBO_TEST_C5_FULL=0 skips it outright."""
    if os.environ.get("KBO_TEST_C5_FULL") == "0":
        pytest.skip("KBO_TEST_C5_FULL=0")
    avail_gb = 0
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable:"):
            avail_gb = int(ln.split()[1]) / 1e6
    if avail_gb < 300:
        pytest.skip("%.0f GB of host memory available, the 3 Gbp index and its oracle copy need 300" % avail_gb)
    budget = float(os.environ.get("KBO_TEST_C5_BUDGET", "640"))
    # (no cache file: this run builds the index once and uses it once - writing 40 GB takes a disk minutes)
    env = dict(os.environ, KBO_BENCH_CACHE_DIR=str(tmp_path), KBO_BENCH_NO_CACHE="1")
    err_path = os.path.join(str(tmp_path), "c5.err")
    with open(err_path, "w") as err:
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C5", "--steps", "2", "--warmup", "1"],
                                stdout=subprocess.PIPE, stderr=err, text=True, env=env, start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=budget)
        except subprocess.TimeoutExpired:
            import signal
            os.killpg(proc.pid, signal.SIGKILL)  # (the group this test started: bench.py and nothing else)
            proc.wait()
            phases = [ln.strip() for ln in open(err_path) if ln.startswith("[bench C5")]
            pytest.skip("C5 at its real size did not finish within %.0f s on this box; phases so far: %s" % (budget, "; ".join(phases[-6:])))
    assert proc.returncode == 0, open(err_path).read()[-3000:]
    r = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert r["bit_exact_vs_oracle"] is True and "3000 Mbp" in r["metric"] and 0 < r["roofline"]["frac"] < 1 and r["cpu_baseline"]["value"] > 0
    assert r["whole_call"]["equal_to_oracle_call_on_sampled_reads"] == 40 and r["whole_call"]["kbo_call_batch_same_offsets"] is True
    assert r["whole_call"]["ms"] < 450  # (VERDICT round 5 asked for <= 0.45 s per 125 k reads; 0.91 s then)
