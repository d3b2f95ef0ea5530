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


def live_traffic(name, n, history, extra, budget_s, frow=False):
    """HBM bytes per timed launch of this workload, measured NOW as MI355X_MICROARCH.md ("HBM", rocprofv3) prescribes: two
    child runs of this file (4 timed steps each, first-allocation placement -- the traffic of a launch does not depend on
    where its arrays lie) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes; no trace domain next
    to --pmc), the evaluate dispatches of the TIMED phase picked out with the child's own launch_log.  Corrections: both
    counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide (16 B per lane) streaming read -> x 2.
    None if rocprofv3 is not there, the budget is short or anything goes wrong (the stored figure is reported then)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None or under_profiler():
        return None
    t_end = time.perf_counter() + budget_s
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        left = t_end - time.perf_counter()
        if left < 25:
            return None
        d = tempfile.mkdtemp(prefix="fcamd_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, BENCH]
            if frow:  # one row of SURVEY 8(f) alone (benchlib/frows.py)
                cmd += ["--frow", name, "--points", str(n), "--steps", "4", "--warmup", "2"]
            else:
                cmd += ["--workload", name, "--points", str(n), "--history", history, "--steps", "4", "--warmup", "2", "--configs", "none",
                        "--no-host-path", "--no-cpu-baseline", "--placement", "first", "--no-live-traffic"] + list(extra)
            # a process group of its own: on a timeout the whole pass (profiler + the profiled child) is ended, nothing else
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 text=True, start_new_session=True)
            try:
                stdout, _ = p.communicate(timeout=left)
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
            vals, i, timed = [float(x["Counter_Value"]) for x in rows], 0, []
            for phase, k in child.get("launch_log", []):
                if phase == "timed":
                    timed = vals[i: i + k]
                i += k
            if len(timed) != 4 or i > len(vals):
                return None
            got[counter] = sum(timed) / len(timed)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    read_b, write_b = 2.0 * 1024.0 * got["FETCH_SIZE"], 1024.0 * got["WRITE_SIZE"]
    return {"hbm_bytes_per_launch": int(read_b + write_b), "read_bytes": int(read_b), "write_bytes": int(write_b)}
