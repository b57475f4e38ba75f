"""Host-side logic of bench.py that needs no GPU: the shape of the JSON lines it assembles around the measurements (the compact
form of an extra workload, the one-line error report of a failed multi-rank run, the host topology of `cpu_baseline`)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    base = dict(gpus=1, steps=10, warmup=3, workload="as", prior=False, quick=False, headline_only=False, dist_single=False,
                samples_total=512, kernel_point=False, eig_large=False)
    base.update(kw)
    return argparse.Namespace(**base)


def test_extras_ride_on_the_default_run_only():
    assert bench.kernel_extras_wanted(_args(), 1)
    for kw in (dict(headline_only=True), dict(workload="pod"), dict(prior=True), dict(dist_single=True), dict(samples_total=64)):
        assert not bench.kernel_extras_wanted(_args(**kw), 1)
    assert not bench.kernel_extras_wanted(_args(), 2)


def test_compact_form_of_an_extra_workload():
    line = {"value": 4.1, "unit": "GDoF*rank/s", "ms_per_step": 15.5, "median_ms_per_step": 15.4, "steps": 5, "warmup": 2,
            "config": {"workload": "config3 PODProjector"}, "literal_T_ms_per_step": 20.0,
            "roofline": {"kernel": "k_tsgemm_tn m=2048 k=138 N=500000", "bound": "mfma", "achieved": 64.0, "peak": 78.6, "unit": "TFLOP/s",
                         "frac": 0.81, "avg_launch_ms": 4.4, "traffic": None, "extra": 1},
            "parity": {"eig_rel_err_vs_oracle": 4e-13, "note": "long text", "oracle_form": "x"},
            "phases_ms_per_step": {"apply_A": 9.0},
            "cpu_baseline": {"value": 0.01, "unit": "GDoF*rank/s", "kind": "port", "cores": 64, "threads": 256, "physical_cores": 128,
                             "sockets": 2, "best_setting": "threads_all", "seconds_full_estimate": 3.8, "sample": "s",
                             "reference_style": {"best_single_setting_value": 6e-4,
                                                 "composite_of_per_component_best_settings": {"value": 7e-4}}, "blas3": {}},
            "communicator": {"ranks": 1, "transport": "rccl"}}
    c = bench._compact(line)
    assert c["workload"] == "config3 PODProjector" and c["ms_per_step"] == 15.5 and c["roofline"]["frac"] == 0.81
    assert "extra" not in c["roofline"] and "note" not in c["parity"] and c["parity"]["eig_rel_err_vs_oracle"] == 4e-13
    assert c["cpu_baseline"]["reference_style_composite_of_per_component_best_settings_value"] == 7e-4
    assert c["cpu_baseline"]["reference_style_best_single_setting_value"] == 6e-4 and c["cpu_baseline"]["physical_cores"] == 128
    assert c["communicator"]["transport"] == "rccl"
    json.dumps(c)
    bare = bench._compact({"value": 1.0, "config": {"workload": "w"}})             # nothing optional present
    assert bare["roofline"]["frac"] is None and bare["parity"] == {} and "cpu_baseline" not in bare


def test_error_line_keeps_the_contract_keys():
    e = bench._error_line(_args(gpus=8), "rank 3 exited with code 7", rank_exit_codes=[0, 0, 0, 7])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "error"):
        assert key in e
    assert e["value"] is None and e["n_gpus"] == 8 and e["rank_exit_codes"][3] == 7
    json.dumps(e)


def test_cpu_topology_is_sane():
    t = bench._cpu_topology()
    assert t["threads"] >= 1 and 1 <= t["physical_cores"] <= max(t["threads"], t["physical_cores"]) and t["sockets"] >= 1
    assert t["physical_cores"] >= t["sockets"]


def test_cpu_quota_of_the_container_is_read_from_the_cgroup(tmp_path):
    """bench.py keeps the host BLAS pool inside what the cgroup lets the process run: cgroup v2 `cpu.max`
    ("quota period" or "max period"), cgroup v1 cfs files, nothing at all."""
    f = tmp_path / "cpu.max"
    f.write_text("1600000 100000\n")
    assert bench._cpu_quota(str(f), str(tmp_path / "none")) == 16
    f.write_text("max 100000\n")
    assert bench._cpu_quota(str(f), str(tmp_path / "none")) is None
    f.write_text("50000 100000\n")
    assert bench._cpu_quota(str(f), str(tmp_path / "none")) == 1                  # half a CPU still means one thread
    v1 = tmp_path / "v1"
    v1.mkdir()
    (v1 / "cpu.cfs_quota_us").write_text("800000\n")
    (v1 / "cpu.cfs_period_us").write_text("100000\n")
    assert bench._cpu_quota(str(tmp_path / "missing"), str(v1)) == 8
    (v1 / "cpu.cfs_quota_us").write_text("-1\n")
    assert bench._cpu_quota(str(tmp_path / "missing"), str(v1)) is None

