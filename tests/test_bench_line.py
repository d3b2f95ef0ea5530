"""The ONE stdout line of bench.py must be readable by the driver: compact (round 4's 22 KB line came back `parsed: null`), built
from the full record by a pure function -- checked here on round 4's stored record and on synthetic N = 8 records -- and
`bench.py --plan` (arithmetic only) must say that the first real 8-GPU launch fits the driver's limits."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchlib.line import CONFIG_COLUMNS, MAX_LINE_BYTES, compact_line, dumps  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")


def r04_record():
    with open(os.path.join(ROOT, "profiles", "r04_default_bench_line.json")) as f:
        return json.load(f)


def test_round4_record_fits_the_line():
    d = r04_record()
    assert len(json.dumps(d)) > 20000  # the line that could not be parsed
    s = dumps(compact_line(d, "gpurun_out/bench_detail.json"))
    assert len(s) < MAX_LINE_BYTES and "\n" not in s
    line = json.loads(s)
    for k in CONTRACT:
        assert line[k] == d[k], k
    rf = line["roofline"]
    assert rf["frac"] == d["roofline"]["frac"] and rf["traffic"] == d["roofline"]["traffic"] and rf["bound"] == "hbm" and rf["peak"] == 8000.0
    assert rf["frac_reference_layout"] == d["sparse_unpacked_history"]["frac"] and rf["frac_full_history"] == d["full_trial_history"]["frac"]
    assert abs(rf["traffic_over_algorithmic"] - d["roofline"]["traffic"] / d["roofline"]["algorithmic_bytes_per_launch"]) < 1e-4
    cb = line["cpu_baseline"]
    assert cb["value"] == d["cpu_baseline"]["value"] and cb["cores"] == 1 and cb["kind"] == "port" and cb["sample"]
    assert line["configs_columns"] == CONFIG_COLUMNS and set(line["configs"]) == set(d["configs"])
    for name, row in line["configs"].items():
        assert row[0] == d["configs"][name]["frac"] and row[1] == d["configs"][name]["traffic_over_algorithmic"]
    assert line["host_path_Mpts_s"]["resident_sparse"] == d["host_path"]["sizes"]["10000000"]["registered"]["resident_sparse"]["Mpts_s"]
    assert "launch_log" not in line and "placement" not in line and "note" not in json.dumps(line)


def test_eight_gpu_record_fits_the_line():
    """the N = 8 line: per-rank kernel times, strong-scaling leg, every gather variant (with long error texts), host_path_multi"""
    d = r04_record()
    d.update({"n_gpus": 8, "per_rank_kernel_ms": [7.6123456] * 8,
              "strong_scaling": {"value": 91234.5, "unit": "Mpts/s", "points_total": 99999744, "points_per_gpu": 12499968, "ms_per_step": 1.09,
                                 "per_rank_kernel_ms": [1.0123456] * 8, "note": "x" * 400},
              "allgather": {"points_per_rank": 100000000, "shard_GB": 33.6, "free_GB_before": 200.0, "note": "y" * 300, "rccl_ms": 1500.0,
                            "rccl_recv_GBs_per_gpu": 156.8, "direct_ms": 300.1, "direct_recv_GBs_per_gpu": 783.0, "tangent_chunks": 3, "chunk_points": 33333312,
                            "p2p_error": "RuntimeError: " + "z" * 600},
              "host_path_multi": d["host_path"]})
    d.pop("configs"), d.pop("host_path"), d.pop("cpu_baseline")
    s = dumps(compact_line(d, "gpurun_out/bench_detail.json"))
    assert len(s) < MAX_LINE_BYTES
    line = json.loads(s)
    assert len(line["per_rank_kernel_ms"]) == 8 and line["strong_scaling"]["value"] == 91234.5 and "note" not in line["strong_scaling"]
    assert line["allgather"]["direct_ms"] == 300.1 and len(line["allgather"]["p2p_error"]) <= 100
    assert line["host_path_multi_Mpts_s"]["resident_sparse"] > 0 and "configs" not in line and "cpu_baseline" not in line


def test_an_oversized_line_is_cut_down_not_printed_whole():
    d = r04_record()
    d["configs"] = {f"configuration_with_a_long_name_{k:04d}": dict(d["configs"]["linear_elasticity"]) for k in range(400)}
    s = dumps(compact_line(d, None))
    assert len(s) < MAX_LINE_BYTES
    line = json.loads(s)
    assert "configs" in line["dropped"] and line["value"] == d["value"] and line["roofline"]["frac"] == d["roofline"]["frac"]


def run_plan(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plan", *args], capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_plan_of_the_eight_gpu_launch_fits_the_limits():
    """the driver's N = 8 command (config 5: 8 x 1e8 points): no GPU work, and by the plan's own arithmetic the run fits 288 GB per
    GPU and -- with margin -- the driver's 600 s, the gather leg inside its watchdog"""
    p = run_plan("--gpus", "8")
    assert p["n_gpus"] == 8 and p["points_total"] == 800_000_000
    assert p["fits_memory"] and p["peak_GB_per_gpu"] <= 288.0
    assert p["fits_driver_limit"] and p["est_total_s"] < 0.5 * p["driver_limit_s"] and p["est_total_s"] < p["wall_budget_s"]
    legs = {x["leg"]: x for x in p["legs"]}
    for name in ("setup", "placement", "timed_steps", "strong_scaling_leg", "allgather", "host_path_multi"):
        assert name in legs and legs[name]["est_s"] >= 0, name
    assert legs["allgather"]["est_s"] < 240.0  # --gather-timeout
    # every optional leg starts with more of --wall-budget left than it asks for: none is skipped by the plan's own arithmetic
    assert all(x["runs"] for x in p["legs"] if "runs" in x), [(x["leg"], x["starts_at_s"]) for x in p["legs"]]
    assert "configs" not in legs and "live_traffic" not in legs  # N = 1 legs
    # the estimated value scales with N (weak scaling, no data-path collective)
    p1 = run_plan("--gpus", "1")
    assert abs(p["est_value_Mpts_s"] / p1["est_value_Mpts_s"] - 8.0) < 1e-2


def test_plan_of_the_default_command_is_under_two_minutes_plus_margin():
    p = run_plan()
    assert p["n_gpus"] == 1 and p["fits_memory"] and p["est_total_s"] < 180.0
    legs = {x["leg"] for x in p["legs"]}
    assert {"configs", "frows", "host_path", "live_traffic", "cpu_baseline"} <= legs
    for n in ("2", "4"):
        q = run_plan("--gpus", n)
        assert q["fits_memory"] and q["fits_driver_limit"]


def test_round5_line_is_what_compact_line_makes_of_its_detail_file():
    """profiles/r05_default_bench_line.json (what the driver's command printed) == compact_line(profiles/r05_default_bench_detail.json)"""
    with open(os.path.join(ROOT, "profiles", "r05_default_bench_detail.json")) as f:
        detail = json.load(f)
    with open(os.path.join(ROOT, "profiles", "r05_default_bench_line.json")) as f:
        text = f.read().strip()
    line = json.loads(text)
    assert len(text) < MAX_LINE_BYTES and "\n" not in text
    assert json.loads(dumps(compact_line(detail, line["detail"]))) == line
    rf = line["roofline"]
    assert rf["traffic_source"] == "live_pmc" and 1.0 <= rf["traffic_over_algorithmic"] <= 1.03
    assert 0.7 < rf["frac"] < 0.85 and rf["frac_in_place"] < rf["frac"] and rf["frac_reference_layout"] < rf["frac"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    assert line["wall_s"] < 120.0  # the default command back under two minutes
    from benchlib import frows
    from benchlib.workloads import EXTRA_CONFIGS

    assert list(line["configs"]) == EXTRA_CONFIGS + list(frows.FROWS)
    assert all(isinstance(v, list) and len(v) == len(CONFIG_COLUMNS) for v in line["configs"].values())
