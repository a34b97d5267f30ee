"""The committed profiler records bench.py quotes must cover every launch plan the tuner can choose: round 3's driver run picked a
plan that had no `--pmc` pass behind it and reported `roofline.traffic: null`."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


def test_every_plan_candidate_has_a_pmc_traffic_entry_and_a_kernel_trace_record():
    import crdmodel_amd as crd

    plans = crd.launch_plan_candidates()
    assert len(plans) >= 32 and len(set(plans)) == len(plans) and plans[0] == (0, 0, 1, 0, 1)  # (index 0 is the plain plan)
    traffic, stats = _table("pmc_traffic.json"), _table("plan_stats.json")
    # the headline workload (BASELINE.json: FHN 8192^2 fp64) and the other two kernels BASELINE's configurations run (C4 Goldbeter fp64, C5 FHN fp32)
    for model, precision in (("fhn", "f64"), ("goldbeter", "f64"), ("fhn", "f32")):
        for plan in plans:
            key = crd.plan_key(model, precision, plan)
            assert key in traffic, "no rocprofv3 --pmc passes recorded for %s (tools/jobs/r05_sweep.sh)" % key
            rec = traffic[key]
            real = 8 if precision == "f64" else 4
            # read + write of both fields once is the least a launch can move; the aprons and strip edges add to it
            assert 4 * real <= rec["bytes_per_point"] <= 1.6 * 4 * real, (key, rec)
            assert abs(rec["write_bytes_per_point"] - 2 * real) <= 0.05 * 2 * real, (key, rec)  # every point of both fields is written exactly once
            assert os.path.exists(os.path.join(ROOT, rec["source"])), rec["source"]
            assert key in stats and stats[key]["sweep_trace_avg_us"] > 0, key


def test_bench_reads_the_committed_tables():
    import bench
    import crdmodel_amd as crd

    key = crd.plan_key("fhn", "f64", {"one_round": 0, "xcd_mapping": 2, "columns_per_lane": 2, "nontemporal_stores": 1})  # what round 3's driver box chose
    nbytes, source = bench.measured_traffic(key, 8192 * 8192)
    assert nbytes and 32.0 <= nbytes / (8192 * 8192) <= 40.0 and "rocprofv3" in source
    assert bench.measured_traffic("fused/fhn/f64/no-such-plan", 1)[0] is None
    streams = bench.committed_json("hbm_streams.json")
    assert streams and 4000 <= streams["read_only_gbs"] <= 8000 and "source" in streams


def test_pinned_bench_stats_reproduce_the_sweep_for_the_headline_workload():
    """profiles/plan_stats.json: for FHN 8192^2 fp64 every plan also has the `rocprofv3 --kernel-trace --stats` row of bench.py itself
    with that plan pinned (`bench.py --launch-plan ...`), beside the HIP-event kernel time the same run printed: the two agree."""
    import crdmodel_amd as crd

    stats = _table("plan_stats.json")
    for plan in crd.launch_plan_candidates():
        rec = stats[crd.plan_key("fhn", "f64", plan)]
        if "bench_stats_avg_us" not in rec:
            pytest.fail("no pinned bench.py --stats record for %s (tools/jobs/r05_plan_stats.sh)" % crd.plan_key("fhn", "f64", plan))
        assert rec["bench_stats_calls"] >= 100
        # (bench.py's figure is the average of a few dozen event-bracketed launches of the timed region, the profiler's of every launch
        # of the process: they agree to 2 % for most plans, 8 % at worst)
        assert abs(rec["bench_stats_avg_us"] - 1e3 * rec["bench_kernel_ms_events"]) <= 0.08 * rec["bench_stats_avg_us"], rec
        if plan[4] == 2 and plan[2] == 1:  # the plans a run actually ends up with at this size
            assert abs(rec["bench_stats_avg_us"] - 1e3 * rec["bench_kernel_ms_events"]) <= 0.03 * rec["bench_stats_avg_us"], rec
