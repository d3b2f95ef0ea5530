"""bench.py itself under test (its N > 1 path had never run outside a manual rehearsal).

A one-GPU box rehearses N = 2 with both ranks on GPU 0 (``--backend gloo``: the RCCL gather variants are skipped, the C
ABI's IPC peer copies run) -- the same control flow as the driver's 8-GPU launch: torchrun child, placement capped by
free memory, barriers, max-over-ranks timing, the strong-scaling leg, the chunked all-gather leg.  Each run is a child
process (bench.py initialises the GPU itself); at most three processes use the GPU at a time."""

import json
import os
import socket
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


LAST_LINE = {}  # the compact stdout line of the last run_bench call (the tests below read the full record, the detail file)


def check_line(stdout, detail_path):
    """ONE JSON line on stdout, whatever happened; compact (the driver keeps an 8 KB tail: round 4's 22 KB line came back
    unparsed); it names the detail file, whose record agrees with it.  Returns (line, full record)."""
    from benchlib.line import MAX_LINE_BYTES

    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]  # (the gloo rehearsal backend prints "[Gloo] Rank ..." lines of its own)
    assert len(lines) <= 1, stdout[-3000:]
    if not lines:
        return None, None
    assert len(lines[0]) < MAX_LINE_BYTES, len(lines[0])
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert key in line, key
    assert isinstance(line["config"]["workload"], str) and line["roofline"]["bound"] in ("hbm", "pcie")
    assert line.get("detail") == detail_path, line.get("detail")
    with open(detail_path) as f:
        detail = json.load(f)
    assert detail["value"] == line["value"] and detail["ms_per_step"] == line["ms_per_step"]
    return line, detail


def run_bench(*args, timeout=420, env=None, live_traffic=False):
    import tempfile

    e = dict(os.environ, MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    if not live_traffic and "--mode" not in args:
        args = (*args, "--no-live-traffic")  # (two more child processes under rocprofv3 --pmc: one test below runs them)
    with tempfile.TemporaryDirectory(prefix="fcamd_bench_") as d:
        detail_path = os.path.join(d, "detail.json")
        r = subprocess.run([sys.executable, BENCH, *args, "--detail", detail_path], capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
        line, detail = check_line(r.stdout, detail_path)
    LAST_LINE.clear()
    LAST_LINE.update(line or {})
    return r, detail


def test_two_ranks_weak_scaling_with_strong_leg_and_gather():
    r, out = run_bench("--gpus", "2", "--backend", "gloo", "--points", "2000000", "--steps", "3", "--warmup", "2",
                       "--configs", "none", "--no-cpu-baseline", "--placement-tries", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out is not None, r.stdout[-2000:]
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["unit"] == "Mpts/s"
    assert out["config"]["points_per_gpu"] == 2_000_000 and out["config"]["points_total"] == 4_000_000
    assert len(out["per_rank_kernel_ms"]) == 2 and all(ms > 0 for ms in out["per_rank_kernel_ms"])
    # value = all ranks' points / max-over-ranks wall time of the timed steps
    assert abs(out["value"] - 4_000_000 * 3 / (out["ms_per_step"] * 3e-3) / 1e6) <= 0.01 * out["value"]
    assert 0 < out["roofline"]["frac"] < 1 and out["roofline"]["bound"] == "hbm"
    ss = out["strong_scaling"]
    assert ss["points_total"] == 2 * ss["points_per_gpu"] and ss["points_per_gpu"] == (2_000_000 // 2 // 64) * 64
    assert len(ss["per_rank_kernel_ms"]) == 2 and ss["value"] > 0
    ag = out["allgather"]
    assert "error" not in ag and "direct_ms" in ag and ag["direct_ms"] > 0, ag
    assert ag["points_per_rank"] == 2_000_000 and ag["tangent_chunks"] >= 1
    assert "rccl_ms" not in ag  # gloo rehearsal: RCCL variants are not run
    assert out.get("cpu_baseline") is None and "configs" not in out
    hm = out["host_path_multi"]  # rank 0 alone drives one context per rank's device over the host path
    assert "error" not in hm and hm["devices"] == [0, 0], hm
    assert hm["sizes"]["4000000"]["registered"]["resident_sparse"]["Mpts_s"] > 0 and "per_call_us" not in hm


def test_four_ranks_rehearsal():
    """world = 4 on one GPU -- the most the box allows inside the suite (its process guard: six GPU processes = pytest + the
    launcher's agent + four ranks; five ranks ran once on their own, round 4; the decisions of world 8 run on the CPU in
    tests/test_sharded_gloo.py): staggered peer order, three peers per rank in the IPC gather, a chunk plan forced down to
    several chunks, four device contexts in the host_path_multi leg."""
    r, out = run_bench("--gpus", "4", "--backend", "gloo", "--points", "1000000", "--steps", "2", "--warmup", "1", "--configs", "none",
                       "--no-cpu-baseline", "--placement", "first", "--gather-points", "300000")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["n_gpus"] == 4 and len(out["per_rank_kernel_ms"]) == 4 and out["config"]["points_total"] == 4_000_000
    assert out["strong_scaling"]["points_per_gpu"] == (1_000_000 // 4 // 64) * 64 and len(out["strong_scaling"]["per_rank_kernel_ms"]) == 4
    ag = out["allgather"]
    assert "error" not in ag and ag["direct_ms"] > 0 and ag["points_per_rank"] == (300_000 // 64) * 64, ag
    assert out["host_path_multi"]["devices"] == [0] * 4 and "error" not in out["host_path_multi"]


def test_host_mode_eight_contexts():
    """--mode host with the target's EIGHT device contexts (one process: no process-guard limit), all on GPU 0: the
    single-assembler path of an 8-GPU node cut as it will be cut there"""
    r, out = run_bench("--mode", "host", "--gpus", "8", "--host-devices", "0,0,0,0,0,0,0,0", "--points", "200000", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["mode"] == "host" and out["n_gpus"] == 8 and out["config"]["devices_used"] == 8
    assert out["config"]["points_total"] == 1_600_000 and out["value"] > 0


def test_two_ranks_strong_scaling():
    r, out = run_bench("--gpus", "2", "--backend", "gloo", "--points", "3000000", "--scaling", "strong", "--steps", "3", "--warmup", "1",
                       "--configs", "none", "--no-cpu-baseline", "--no-gather", "--placement", "first")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["scaling"] == "strong" and out["config"]["points_total"] == 3_000_000
    assert out["config"]["points_per_gpu"] == 1_500_032  # fcamd_shard_bounds(3e6, 2, 0): tile-aligned slot
    assert abs(out["value"] - 3_000_000 * 3 / (out["ms_per_step"] * 3e-3) / 1e6) <= 0.01 * out["value"]
    assert "strong_scaling" not in out and "allgather" not in out


def test_a_hung_gather_leg_is_a_failed_run():
    """--gather-timeout fires (here at once): rank 0 still prints the line -- with the step timing and an error entry for the
    leg -- and the processes end with a NON-ZERO exit code (a hung exchange used to be reported as rc 0)."""
    r, out = run_bench("--gpus", "2", "--backend", "gloo", "--points", "4000000", "--steps", "2", "--warmup", "1", "--configs", "none",
                       "--no-cpu-baseline", "--placement", "first", "--gather-timeout", "0.0", "--wall-budget", "1000")
    assert r.returncode != 0
    assert out is not None and out["n_gpus"] == 2 and out["value"] > 0
    assert "did not finish" in out["allgather"]["error"]


@pytest.mark.parametrize("leg,gpus", [("host_path_multi", 2), ("allgather", 2), ("host_path", 1), ("live_traffic", 1)])
def test_the_line_survives_the_death_of_rank0_in_an_optional_leg(leg, gpus):
    """rank 0 is killed (SIGKILL, fault injection) when it enters an optional leg: the measured line still comes out, once,
    marked incomplete, and the job fails"""
    extra = ["--backend", "gloo"] if gpus > 1 else []
    r, out = run_bench("--gpus", str(gpus), *extra, "--points", "1000000", "--steps", "2", "--warmup", "1", "--configs", "none",
                       "--no-cpu-baseline", "--placement", "first", env={"BENCH_DIE_IN": leg}, live_traffic=leg == "live_traffic")
    assert r.returncode != 0
    assert out is not None and out["n_gpus"] == gpus and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert leg in out["incomplete"] and leg not in out


def test_host_mode_two_contexts():
    r, out = run_bench("--mode", "host", "--gpus", "2", "--host-devices", "0,0", "--points", "300000", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["mode"] == "host" and out["n_gpus"] == 2 and out["config"]["devices_used"] == 2
    assert out["config"]["points_total"] == 600_000 and out["value"] > 0
    assert out["roofline"]["bound"] == "pcie" and out["roofline"]["achieved"] > 0
    hp = out["host_path"]
    big = hp["sizes"]["600000"]
    for key in ("pageable", "registered"):
        for leg in ("evaluate", "resident", "resident_sparse"):
            assert big[key][leg]["Mpts_s"] > 0
    assert set(hp["per_call_us"]) == {"1000", "10000"} and hp["pinned_copy_GBs"]["d2h"] > 0


def test_default_line_carries_the_host_path_block():
    """the driver's command shape at a small size: one rank, no launcher, extra configurations off"""
    r, out = run_bench("--points", "2000000", "--steps", "3", "--warmup", "1", "--configs", "none", "--no-cpu-baseline", live_traffic=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["n_gpus"] == 1 and out["scaling"] == "weak"
    import shutil

    rf = out["roofline"]
    if shutil.which("rocprofv3"):
        # roofline.traffic measured in this very run (rocprofv3 --pmc child passes): between the algorithmic bytes and 1.25 x them
        assert "measured in this run" in rf["traffic_source"], rf.get("traffic_source")
        assert rf["algorithmic_bytes_per_launch"] * 0.99 <= rf["traffic"] <= rf["algorithmic_bytes_per_launch"] * 1.25, rf
        assert sum(rf["traffic_read_write"]) == rf["traffic"]
        assert LAST_LINE["roofline"]["traffic_source"] == "live_pmc" and LAST_LINE["roofline"]["traffic"] == rf["traffic"]
        # the reference-layout forms of the same step were measured by the same child passes
        for key in ("sparse_unpacked_history", "in_place"):
            assert 0.99 <= out[key]["traffic_over_algorithmic"] <= 1.3, out[key]
    for key in ("frac_reference_layout", "frac_in_place", "frac_full_history"):
        assert 0 < LAST_LINE["roofline"][key] < 1, key
    hp = out["host_path"]
    assert "error" not in hp, hp
    assert set(hp["sizes"]) == {"1000000", "2000000"}
    assert hp["sizes"]["2000000"]["registered"]["resident_sparse"]["Mpts_s"] > hp["sizes"]["2000000"]["registered"]["resident"]["Mpts_s"] * 0.8
    assert out["roofline"]["traffic"] is None or "traffic_source" in out["roofline"]


def test_one_rank_under_the_launcher_with_rccl(tmp_path):
    """The driver's launch shape (python -m torch.distributed.run ... bench.py --gpus N) with N = 1: the process group is
    RCCL, the barriers / all-reduces of the timed region and of the per-rank kernel times run on the device."""
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), BENCH, "--gpus", "1", "--steps", "3", "--warmup", "1", "--points", "2000000",
           "--configs", "none", "--no-cpu-baseline", "--no-host-path", "--placement-tries", "2", "--no-live-traffic", "--detail", os.path.join(tmp_path, "detail.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["per_rank_kernel_ms"] and len(out["per_rank_kernel_ms"]) == 1
    assert "allgather" not in out and "strong_scaling" not in out and "host_path_Mpts_s" not in out and "host_path_multi_Mpts_s" not in out
    assert abs(out["value"] - 2_000_000 * 3 / (out["ms_per_step"] * 3e-3) / 1e6) <= 0.01 * out["value"]


def test_default_legs_at_small_size_carry_every_configuration_and_8f_row():
    """The driver's command shape (N = 1, no --workload) at 2e6 points: the line carries the five other BASELINE configurations
    and every SURVEY 8(f) row, each with a kernel time, a roofline fraction and a launch log whose phases end with the timed one (and, for the rows that have one,
    the measurement of their synthetic twin: mem_floor_ms)."""
    from benchlib import frows

    r, out = run_bench("--points", "2000000", "--steps", "3", "--warmup", "1", "--config-steps", "5", "--no-cpu-baseline", "--no-host-path",
                       "--placement-tries", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["n_gpus"] == 1 and out["value"] > 0 and 0 < out["roofline"]["frac"] < 1
    cfg = out["configs"]
    for name in ("linear_elasticity", "von_mises_plastic", "von_mises_elastic", "spring_maxwell", "spring_kelvin", *frows.FROWS):
        assert name in cfg, (name, sorted(cfg))
        c = cfg[name]
        assert "error" not in c, (name, c)
        assert c["kernel_ms_avg"] > 0 and 0 < c["frac"] < 1.2, (name, c)
        phases = [ph for ph, _ in c["launch_log"]]  # ... end with the timed phase (rows with a synthetic twin: then its measurement)
        assert "timed" in phases, (name, c["launch_log"])
        last_timed = max(i for i, ph in enumerate(phases) if ph == "timed")  # (a row measured on several sets of allocations logs one sequence per set)
        assert all(ph.startswith("mem_floor_") for ph in phases[last_timed + 1:]), (name, c["launch_log"])
    assert set(frows.SURVEY_ROW) == set(frows.FROWS)
    # the compact line: one short list per configuration, columns named once
    from benchlib.line import CONFIG_COLUMNS

    assert LAST_LINE["configs_columns"] == CONFIG_COLUMNS and set(LAST_LINE["configs"]) == set(cfg)
    for name, row in LAST_LINE["configs"].items():
        assert row[0] == cfg[name]["frac"] and row[2] == cfg[name]["kernel_ms_avg"], (name, row)
