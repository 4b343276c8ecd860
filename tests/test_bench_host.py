"""bench.py's host-side helpers that need no GPU: the roofline record and the fall-backs of the live PMC traffic
measurement (the measurement itself runs in tests/test_gpu_pipeline.py::test_bench_contract_line)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("pp_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_live_traffic_declines_under_a_profiler_and_without_one(monkeypatch):
    b = _bench()
    # a run that is itself being profiled does not start profilers of its own
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    assert b.live_traffic(4, "pp::k_step<0, 0>") is None
    monkeypatch.delenv("ROCPROF_OUTPUT_PATH")
    # no rocprofv3 on the PATH: None (the caller then reports the committed constant, labelled static)
    monkeypatch.setenv("PATH", "/nonexistent")
    assert b.live_traffic(4, "pp::k_step<0, 0>") is None


def test_roofline_record_reports_both_peaks():
    b = _bench()
    rec = b.roofline_record({"k_step": 40.0}, 50, 177_792_000, traffic=2.0e8, traffic_source="static: x")
    assert rec["bound"] == "hbm" and rec["kernel"].startswith("pp::k_step")
    assert abs(rec["achieved"] - 177_792_000 / 40e-6 / 1e9) < 1e-6
    assert abs(rec["frac"] - rec["achieved"] / 8000.0) < 1e-12
    assert abs(rec["frac_of_measured_copy"] - rec["achieved"] / 6290.0) < 1e-12
    assert rec["pipeline_frac"] == rec["frac"] and rec["traffic"] == 2.0e8
    three = b.roofline_record({"k_split": 9.0, "k_tile": 9.0, "k_emit": 30.0}, 50, 177_792_000)
    assert three["kernel"].startswith("pp::k_emit") and three["pipeline"]["sum_us"] == 48.0
    assert three["pipeline_frac"] < three["frac"] < 1
    s_key, src = b.static_traffic("step_batch4")
    assert s_key and 1.0 < s_key / 177_792_000 < 1.35 and src.startswith("static")


def test_dropin_record_arithmetic():
    b = _bench()
    r = b.dropin_record(0.4, 10.0, "x")
    assert abs(r["speedup"] - 25.0) < 1e-12 and r["meets_50x"] is False and r["hip_ms"] == 0.4 and r["cpu_ms"] == 10.0
    assert b.dropin_record(0.1, 5.0, "x")["meets_50x"] is True          # exactly 50x counts
    assert b.dropin_record(0.1, 4.99, "x")["meets_50x"] is False


def test_per_rank_and_min_max_single_process():
    """the N = 1 forms of the per-rank helpers (the 2-rank forms run in tests/test_shard_gloo.py and, on the GPU, in
    tests/test_gpu_bench_ranks.py)"""
    from pp_amd import shard
    b = _bench()
    ctx = shard.ShardContext()
    assert b.per_rank(ctx, [1.5, 2.5]) == [[1.5, 2.5]]
    mm = b.ranks_min_max(ctx, 40.0, 177_792_000)
    assert mm["k_step_us"] == [40.0, 40.0] and mm["per_rank_k_step_us"] == [40.0]
    assert abs(mm["frac"][0] - 177_792_000 / 40e-6 / 8e12) < 1e-12 and mm["frac"][0] == mm["frac"][1]


def test_cpu_baseline_reports_both_variants():
    """bench.py's cpu_baseline: the C port (top-level value: the faster, conservative one) AND the baseline-faithful
    variant of BASELINE.md section 4, each on one core, as 4 worker processes and on all host cores (tiny budgets here)."""
    b = _bench()
    cb = b.cpu_baseline(seconds_budget=0.8, workers=4, worker_budget=0.6, c1_budget=0.8, all_budget=0.6)
    assert cb["kind"] == "port" and cb["unit"] == "sweeps/s" and cb["cores"] == 1
    assert cb["value"] == cb["port"]["one_core"]["value"] > 0
    for variant in ("port", "faithful"):
        rec = cb[variant]
        assert rec["one_core"]["cores"] == 1 and rec["one_core"]["calls"] >= 5
        assert rec["workers"]["processes"] == 4 and rec["workers"]["value"] > 0
        assert rec["all_cores"]["processes"] >= 1 and rec["all_cores"]["value"] > 0
    assert cb["c1"]["value"] > 0 and cb["c1"]["faithful"]["value"] > 0
    r = b.dropin_record(0.2, 10.0, "x", faithful_ms=20.0)
    assert r["meets_50x"] is True and abs(r["faithful_speedup"] - 100.0) < 1e-9 and r["faithful_meets_50x"] is True
