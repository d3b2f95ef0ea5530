"""bench.py: HBM traffic of a workload's launch -- measured live by rocprofv3 --pmc child passes, or the stored figure."""

from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BENCH = os.path.join(ROOT, "bench.py")


def library_hash(kernels_only=False):
    """Content hash of the sources libfcamd.so was built from (fenics_constitutive_amd/_build.py); kernels_only:
    of the device code alone, which is what measured HBM traffic depends on."""
    try:
        from fenics_constitutive_amd import _build

        if kernels_only:
            return _build.built_kernel_hash()
        with open(_build.HASHFILE) as f:
            return f.read().strip()
    except Exception:
        return None


def read_traffic_split(workload_key, n):
    """(read bytes, written bytes) per launch of the same PMC measurement, or None"""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(tf) as f:
            e = json.load(f).get(workload_key)
        if e and int(e.get("n", 0)) == n and e.get("kernel_hash") and e.get("kernel_hash") == library_hash(kernels_only=True):
            return int(e["read_bytes"]), int(e["write_bytes"])
    except Exception:
        pass
    return None


def read_traffic(workload_key, n):
    """PMC-measured HBM bytes per launch (profiles/traffic.json, written by tools/summarize_profile.py) --
    only if they were measured with THIS build of the kernels (same hash of the device sources) at this size; a
    kernel change makes the figure stale and the line then says null."""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(tf) as f:
            e = json.load(f).get(workload_key)
        if e and int(e.get("n", 0)) == n and e.get("kernel_hash") and e.get("kernel_hash") == library_hash(kernels_only=True):
            return e.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def under_profiler():
    """is THIS process running under rocprofv3 (a nested profiler must not be started)"""
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def _pmc_pass(counter, child_args, timeout_s):
    """one `rocprofv3 --pmc COUNTER` pass over `bench.py <child_args>`: (the child's JSON line, [counter value per fcamd::evaluate
    dispatch, in dispatch order]) or None"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    d = tempfile.mkdtemp(prefix="fcamd_pmc_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, BENCH] + list(child_args)
        # a process group of its own: on a timeout the whole pass (profiler + the profiled child) is ended, nothing else
        p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             text=True, start_new_session=True)
        try:
            stdout, _ = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            import signal

            os.killpg(p.pid, signal.SIGKILL)
            p.communicate()
            return None
        lines = [ln for ln in stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        if p.returncode != 0 or not lines:
            return None
        child = json.loads(lines[-1])
        files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
        if not files:
            return None
        with open(files[-1]) as f:
            rows = [x for x in csv.DictReader(f) if "fcamd::evaluate" in x["Kernel_Name"] and x.get("Counter_Name", counter) == counter]
        rows.sort(key=lambda x: int(x["Dispatch_Id"]))
        return child, [float(x["Counter_Value"]) for x in rows]
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def slice_timed(log, vals):
    """{item: mean counter value of its "timed" launches} from the child's ordered log [item, phase, launches] and the counter
    values of all fcamd::evaluate dispatches in order.  An item whose phases run past the end of `vals`, or an "error" entry, ends
    the slicing (what came before it stands)."""
    out, i = {}, 0
    for item, phase, k in log:
        if phase == "error" or not isinstance(k, int) or i + k > len(vals):
            break
        if phase == "timed" and k > 0:
            out[item] = sum(vals[i: i + k]) / k
        i += k
    return out


def live_traffic_batch(items, n, history, extra, budget_s, headline=None, pass_budget_s=0.0):
    """HBM bytes per timed launch of every item (workloads, their reference-layout forms `name+unpacked` / `name+in_place`, SURVEY
    8(f) rows), measured NOW as MI355X_MICROARCH.md ("HBM", rocprofv3) prescribes: two child runs of this file (`--pmc-child`: all
    items in ONE process, 4 timed launches each, the allocator's arrays as they come -- the traffic of a launch does not depend on
    where its arrays lie) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes; no trace domain next to
    --pmc), the evaluate dispatches of every item's TIMED phase picked out with the child's own launch log.  Corrections: both
    counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide (16 B per lane) streaming read -> x 2.
    Returns {item: {"hbm_bytes_per_launch", "read_bytes", "write_bytes"}} (items that could not be sliced are missing), or None if
    rocprofv3 is not there, this process is itself profiled, or the budget is short."""
    import shutil

    if shutil.which("rocprofv3") is None or under_profiler() or not items:
        return None
    t_end = time.perf_counter() + budget_s
    child_args = ["--pmc-child", ",".join(items), "--points", str(n), "--history", history, "--pmc-budget", f"{pass_budget_s:.1f}"] \
        + (["--workload", headline] if headline else []) + list(extra)
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        left = t_end - time.perf_counter()
        if left < 25:
            return None
        res = _pmc_pass(counter, child_args, left)
        if res is None:
            return None
        child, vals = res
        got[counter] = slice_timed(child.get("log", []), vals)
    out = {}
    for item in items:
        if item in got["FETCH_SIZE"] and item in got["WRITE_SIZE"]:
            read_b, write_b = 2.0 * 1024.0 * got["FETCH_SIZE"][item], 1024.0 * got["WRITE_SIZE"][item]
            out[item] = {"hbm_bytes_per_launch": int(read_b + write_b), "read_bytes": int(read_b), "write_bytes": int(write_b)}
    return out
