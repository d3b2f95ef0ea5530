"""bench.py: the ONE stdout line (compact, under 4 KB: the driver keeps an 8 KB tail) and the detail file next to it.

Round 4's line was one ~22 KB JSON object: the driver could not parse it (BENCH_r04.json `parsed: null`).  The line is now
numbers only -- the contract keys, `roofline`, `cpu_baseline`, and one short list per extra configuration -- and everything
else (workload sentences, launch logs, placement candidates, host-path and crossover tables, notes) goes to the detail file
(`--detail`, default gpurun_out/bench_detail.json) whose path the line carries.  `compact_line` is a pure function of the
detail dict, so a CPU test can hold it to the size limit on a stored detail file."""

from __future__ import annotations

import json
import os

MAX_LINE_BYTES = 4096
CONFIG_COLUMNS = ["frac", "traffic_over_algorithmic", "kernel_ms_avg", "frac_first_allocation"]


def _r(x, k=4):
    return None if x is None else round(float(x), k)


def _short(text, limit):
    text = str(text)
    return text if len(text) <= limit else text[: limit - 3] + "..."


def _host_rates(hp):
    """{leg: Mpts/s} of the largest size, registered arrays -- the PCIe-inclusive figures of the ndarray entries"""
    try:
        sizes = hp["sizes"]
        big = sizes[max(sizes, key=int)]
        rows = big.get("registered") or big.get("pageable")
        out = {leg: rows[leg]["Mpts_s"] for leg in ("evaluate", "evaluate_pcie_tangent", "evaluate_le", "resident", "resident_sparse") if leg in rows}
        out["points"] = int(max(sizes, key=int))
        if "tangent_threads" in rows.get("evaluate", {}):  # the tangent rows of `evaluate` come from this many host threads, at this summed CPU time
            out["tangent_threads"] = rows["evaluate"]["tangent_threads"]
            out["tangent_cpu_ms"] = rows["evaluate"]["tangent_cpu_ms"]
        return out
    except Exception:
        return {"error": _short(hp.get("error", "no figures"), 80)} if isinstance(hp, dict) else None


def compact_line(d, detail_path=None):
    """the stdout line of a device-mode run from the full record `d` (what bench.py used to print whole)"""
    rf = d.get("roofline") or {}
    cfg = d.get("config") or {}
    alg = rf.get("algorithmic_bytes_per_launch")
    traffic = rf.get("traffic")
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": _short(cfg.get("workload_short") or cfg.get("workload", ""), 160)}
    for k in ("baseline_config", "points_per_gpu", "points_total", "plastic_fraction", "mean_newton_iters", "ever_fraction", "parallelism"):
        if cfg.get(k) is not None:
            line["config"][k] = cfg[k]
    src = str(rf.get("traffic_source") or "")
    line["roofline"] = {
        "bound": rf.get("bound"), "achieved": rf.get("achieved"), "peak": rf.get("peak"), "unit": rf.get("unit"), "frac": rf.get("frac"),
        "traffic": traffic,
        "traffic_over_algorithmic": _r(traffic / alg) if (traffic and alg) else None,
        "traffic_source": None if traffic is None else ("live_pmc" if src.startswith("measured in this run") else "stored_pmc"),
        "algorithmic_bytes_per_launch": alg,
        "kernel_ms_avg": rf.get("kernel_ms_avg"),
    }
    for k in ("frac_first_allocation", "frac_worst_candidate", "frac_vmm_set", "stream_copy_GBs", "mem_floor_ms", "kernel_over_mem_floor"):
        if rf.get(k) is not None:
            line["roofline"][k] = rf[k]
    # the same step in the reference's forms: the sparse protocol on the reference's array layout, the plain in-place call of a
    # drop-in caller, the full trial history
    for key, src_key in (("frac_reference_layout", "sparse_unpacked_history"), ("frac_in_place", "in_place"), ("frac_full_history", "full_trial_history")):
        leg = d.get(src_key)
        if isinstance(leg, dict) and leg.get("frac") is not None:
            line["roofline"][key] = leg["frac"]
            if leg.get("read_over_needed_lines") is not None:  # in place: measured bytes over the lines / granules the call must move
                line["roofline"][key + "_bytes_over_needed"] = [leg["read_over_needed_lines"], leg.get("write_over_needed_granules")]
    cb = d.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": _short(cb.get("sample", ""), 150)}
        ex = cb.get("extra") or {}
        for k_out, k_in in (("numpy_port", "numpy_port_Mpts_s"), ("comfe_rs_c_port", "comfe_rs_mises_c_port_1_thread_Mpts_s"),
                            ("all_cores", "c_port_all_cores_Mpts_s"), ("all_cores_threads", "c_port_all_cores_threads"),
                            ("all_cores_per_thread", "c_port_all_cores_per_thread_Mpts_s")):
            if ex.get(k_in) is not None:
                line["cpu_baseline"][k_out] = ex[k_in]
    elif "cpu_baseline" in d:
        line["cpu_baseline"] = None
    configs = d.get("configs")
    if isinstance(configs, dict):
        line["configs_columns"] = CONFIG_COLUMNS
        line["configs"] = {}
        for name, c in configs.items():
            if not isinstance(c, dict) or "kernel_ms_avg" not in c:
                line["configs"][name] = _short((c or {}).get("error") or (c or {}).get("skipped") or "no figures", 60) if isinstance(c, dict) else None
                continue
            line["configs"][name] = [c.get("frac"), c.get("traffic_over_algorithmic"), c.get("kernel_ms_avg"), c.get("frac_first_allocation", c.get("frac"))]
            if c.get("mem_floor_ms") is not None:  # the row's synthetic twin (same request stream, no arithmetic), measured
                line.setdefault("mem_floor_ms", {})[name] = [c["mem_floor_ms"], c.get("kernel_over_mem_floor")]
    if isinstance(d.get("host_path"), dict):
        line["host_path_Mpts_s"] = _host_rates(d["host_path"])
    if d.get("per_rank_kernel_ms") is not None:
        line["per_rank_kernel_ms"] = [_r(x, 3) for x in d["per_rank_kernel_ms"]]
    ss = d.get("strong_scaling")
    if isinstance(ss, dict):
        line["strong_scaling"] = {k: ss.get(k) for k in ("value", "unit", "points_total", "points_per_gpu", "ms_per_step")}
        if ss.get("per_rank_kernel_ms") is not None:
            line["strong_scaling"]["per_rank_kernel_ms"] = [_r(x, 3) for x in ss["per_rank_kernel_ms"]]
    ag = d.get("allgather")
    if isinstance(ag, dict):
        keep = {}
        for k, v in ag.items():
            if k.endswith(("_ms", "_recv_GBs_per_gpu")) or k in ("points_per_rank", "shard_GB", "tangent_chunks"):
                keep[k] = v
            elif k.endswith(("_error", "_skipped")) or k in ("error", "skipped"):
                keep[k] = _short(v, 100)
        line["allgather"] = keep
    hm = d.get("host_path_multi")
    if isinstance(hm, dict):
        line["host_path_multi_Mpts_s"] = _host_rates(hm) if "sizes" in hm else {k: _short(v, 100) for k, v in hm.items() if k in ("error", "skipped")}
    if d.get("incomplete"):
        line["incomplete"] = _short(d["incomplete"], 160)
    lib = d.get("library") or {}
    if lib.get("kernel_hash"):
        line["kernel_hash"] = str(lib["kernel_hash"])[:16]
    if detail_path:
        line["detail"] = detail_path
    if d.get("wall_s") is not None:
        line["wall_s"] = d["wall_s"]
    return line


def dumps(line):
    """one line, no spaces; asserts the size limit the driver's tail imposes (a longer line is cut down, never printed whole)"""
    s = json.dumps(line, separators=(",", ":"))
    if len(s) > MAX_LINE_BYTES:  # drop the optional blocks, least important first, until it fits
        for k in ("host_path_multi_Mpts_s", "host_path_Mpts_s", "allgather", "strong_scaling", "configs", "configs_columns"):
            if k in line:
                line = {kk: v for kk, v in line.items() if kk != k}
                line["dropped"] = line.get("dropped", []) + [k]
                s = json.dumps(line, separators=(",", ":"))
                if len(s) <= MAX_LINE_BYTES:
                    break
    return s


def write_detail(d, path):
    """the full record next to the line (atomic: a reader never sees half a file); returns the path written or None"""
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(d, f, indent=1)
        os.replace(tmp, path)
        return path
    except OSError:
        return None
