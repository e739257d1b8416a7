"""The bench line contract (task description, section 4): checked on the committed bench line of the round's final
profile (the newest profiles/r*_bench.json) and on bench.py's argument parser — no GPU needed."""
import ast
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_bench_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1])), files[-1]


def test_committed_bench_line_has_the_contract_fields():
    j, path = latest_bench_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, (path, k)
    assert j["metric"].startswith("witnessed VM cycles/sec") and j["unit"] == "cycles/s"
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == j["unit"]
    # whole-job throughput = cycles of K steps / time
    cycles = j["config"]["instances_per_gpu"] * j["config"]["cycles_per_instance"] * j["n_gpus"]
    assert abs(j["value"] - cycles / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6


def test_bench_accepts_the_driver_flags():
    src = open(os.path.join(ROOT, "bench.py")).read()
    ast.parse(src)
    for flag in ("--gpus", "--steps", "--warmup"):
        assert 'add_argument("%s"' % flag in src
    # rank / world come from the environment of torch.distributed.run
    for env in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR"):
        assert env in src


def test_traffic_file_matches_the_profile_summary():
    t = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))
    assert abs(t["hbm_bytes_per_launch"] - (t["fetch_bytes_corrected_x2"] + t["write_bytes"])) < 1
    assert abs(t["fetch_bytes_corrected_x2"] - 2 * 1024 * t["FETCH_SIZE_KB"]) < 1


def test_roofline_traffic_is_tied_to_the_kernel_it_was_measured_on():
    """bench.py reports `roofline.traffic` from profiles/traffic.json only while the device sources hash to the value the
    PMC passes were taken on; after any edit of the kernel the line carries null until the collection is repeated."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class A:
        cfg, instances, cycles = 2, 4096, 256
    path = os.path.join(ROOT, "profiles", "traffic.json")
    got = bench.measured_traffic(A, 20)
    if not os.path.exists(path):
        assert got is None
        return
    t = json.load(open(path))
    assert abs(t["hbm_bytes_per_launch"] - (t["fetch_bytes_corrected_x2"] + t["write_bytes"])) < 1
    if t["kernel_source_sha256"] == bench.kernel_source_hash():
        assert abs(got - t["hbm_bytes_per_launch"] * 20 / t["fused_batches"]) < 1
    else:
        assert got is None


def test_cfg3_line_has_roofline_and_cpu_baseline():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cfg3_line.json")))
    if not files:
        import pytest
        pytest.skip("no cfg-3 line collected yet")
    j = json.load(open(files[-1]))
    assert j["unit"] == "message bytes/s" and "configs[3]" in j["metric"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "keccak_f_per_s", "sha256_compressions_per_s"):
        assert k in r, k
    assert abs(r["achieved"] - 2.5 * r["message_bytes_per_launch"] / (j["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    c = j["cpu_baseline"]
    assert c["unit"] == j["unit"] and c["kind"] == "port" and c["cores"] >= 1


def test_rocprof_kernel_duration_agrees_with_the_bench_line():
    """The committed rocprofv3 --kernel-trace --stats summary of a command and the HIP-event duration its bench line
    carries describe the same launches of zkw_cycle_kernel: they must agree (within 5 %).  Checked on the newest line
    (the driver's command: profiles/rNN_driver_bench.json) — test_every_round2_bench_line_has_its_rocprof_summary covers
    the default command's pair."""
    import csv
    j, path = latest_bench_line()
    _check_pair(j, path)


def test_every_round2_bench_line_has_its_rocprof_summary():
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")) if not os.path.basename(f).startswith("r01"))
    assert len(files) >= 2, files  # the driver's command and the default command
    for f in files:
        _check_pair(json.load(open(f)), f)


def _check_pair(j, path):
    import csv
    stats = path.replace("_bench.json", "_kernel_stats.csv")
    assert os.path.exists(stats), stats
    rows = [r for r in csv.DictReader(open(stats)) if r["Name"].startswith("zkw_cycle_kernel")]
    assert rows, "zkw_cycle_kernel missing from the rocprof summary"
    avg_ms = float(rows[0]["AverageNs"]) * 1e-6
    assert abs(avg_ms - j["kernel_ms"]) / avg_ms < 0.05, (avg_ms, j["kernel_ms"])
    # and the roofline figure is algorithmic bytes per launch over that duration
    r = j["roofline"]
    assert abs(r["achieved"] - r["bytes_per_cycle"] * r["cycles_per_launch"] / (j["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6


def test_roofline_accounting_is_frozen():
    """DESIGN.md 6 (the rule): the headline workload's algorithmic bytes per VM cycle are the round-3 figure — a byte the
    kernel stops writing raises `roofline.frac`, it does not lower the denominator; the run's own bytes are a separate field."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.ALGORITHMIC_BYTES_R3 == 149.125
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "b_cycle = ALGORITHMIC_BYTES_R3 if headline_shape else b_run" in src
    assert '"bytes_per_cycle_this_run": b_run' in src and '"algorithmic_bytes_r3"' in src
    j, _ = latest_bench_line()
    if j["config"].get("instances_per_gpu") == 4096 and j["config"].get("cycles_per_instance") == 256 and "algorithmic_bytes_r3" in j["roofline"]:
        assert j["roofline"]["bytes_per_cycle"] == 149.125


def test_driver_line_carries_the_round4_fields():
    """the driver's command as the driver runs it (profiles/rNN_driver_full_line.json): five timed regions, every single-GPU
    BASELINE configuration in `other_configs` with its roofline fraction and CPU baseline, the restore outside the one-launch
    timed region, the device-side record expansion"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_driver_full_line.json")))
    if not files:
        import pytest
        pytest.skip("no full driver line collected yet")
    j = json.load(open(files[-1]))
    assert j["steps"] == 20 and j["warmup"] == 5 and j["n_gpus"] == 1
    assert j["timed_regions"] == 5 and j["value_min"] <= j["value_median"] <= j["value_max"]
    assert j["value_min"] <= j["value"] <= j["value_max"]
    assert j["config"]["restore"] == "between-uses" and j["config"]["restores_in_timed_region"] == 0 and j["config"]["cycle_kernel_launches"] == 1
    assert j["ms_per_step"] - j["kernel_ms"] / 20 <= 0.003  # the step is the kernel + the launch gap
    assert j["roofline"]["bytes_per_cycle"] == 149.125 and j["roofline"]["algorithmic_bytes_r3"] == 149.125
    assert j["roofline"]["expand_GBps"] and j["roofline"]["expand"]["fused_batches"] == 20
    oc = j["other_configs"]
    assert len(oc) == 4 and not any("error" in e for e in oc), oc
    for e, key in zip(oc, ("configs[1] literal", "configs[1] at 4096", "configs[3]", "configs[4]")):
        assert e["workload"].startswith(key)
        assert e["value"] > 0 and e["kernel_ms"] > 0 and 0 < e["roofline"]["frac"] < 1
        assert e["cpu_baseline"]["single_socket_value"] > 0 and e["checked"]["instances_failed"] == 0
    assert oc[2]["unit"] == "message bytes/s" and oc[2]["roofline"]["lone_batch_kernel_ms"]["two_lanes_per_wave"] > 0
    assert oc[3]["commit_mask"] == 7
