#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs written by tools/profile_gpu.sh / tools/profile_default.sh into the committed
summaries:

    python tools/summarize_profile.py gpurun_out/prof/<tag> <round> <workload> ["extra bench args"]

writes profiles/r<round>_<workload>_rocprof.md (kernel stats, the dispatches of the evaluate kernels sliced
into the phases bench.py logged -- `launch_log` of its JSON line --, PMC traffic) and updates
profiles/traffic.json (HBM bytes per launch + the hash of the kernel sources that were profiled: bench.py
reports `roofline.traffic` only while that hash is the running library's).  PMC corrections per
MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B-per-lane
streaming stores.  <workload> = default_bench: the driver's command (`python3 bench.py`), kernel trace only,
every configuration of the line."""
import csv
import glob
import json
import os
import sys

src, rnd, workload = sys.argv[1], sys.argv[2], sys.argv[3]
extra_args = sys.argv[4] if len(sys.argv) > 4 else ""  # bench flags beyond --workload (e.g. "--history full")
is_default = workload == "default_bench"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)  # newest run if several were merged
    return list(csv.DictReader(open(f[-1]))) if f else []


def bench_line(name):
    """the full record of a profiled bench.py run: its --detail file (round 5: the stdout line is compact and carries no launch logs)"""
    for cand in (name, name.replace(".log", "_detail.json")):
        p = os.path.join(src, cand)
        if not os.path.exists(p):
            continue
        try:
            with open(p) as f:
                d = json.load(f)
            if isinstance(d, dict) and "launch_log" in d:
                return d
        except ValueError:
            pass
        for line in open(p):
            if line.startswith("{") and '"metric"' in line and '"launch_log"' in line:
                return json.loads(line)
    return {}


def is_eval(r):
    return "fcamd::evaluate" in r["Kernel_Name"]


def phases_of(bench):
    """[(workload, phase, n launches)] in dispatch order: the headline, then the configs in their order."""
    out = [("headline", ph, k) for ph, k in bench.get("launch_log", [])]
    for name, c in (bench.get("configs") or {}).items():
        out += [(name, ph, k) for ph, k in c.get("launch_log", [])]
    return out


def slice_phases(values, bench):
    """{(workload, phase): [values]} -- `values` = one entry per evaluate-kernel dispatch, in order."""
    out, i = {}, 0
    for wl, ph, k in phases_of(bench):
        out.setdefault((wl, ph), []).extend(values[i : i + k])
        i += k
    return out, i


stats = rows("kt/*/*_kernel_stats.csv")
trace = sorted([r for r in rows("kt/*/*_kernel_trace.csv") if is_eval(r)], key=lambda r: int(r["Start_Timestamp"]))
bench = bench_line("bench_under_rocprof.json") or bench_line("kt.log")
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in trace]
by_phase, used = slice_phases(dur, bench)

out = os.path.join(ROOT, "profiles", f"r{rnd}_{workload}_rocprof.md")
with open(out, "w") as f:
    f.write(f"# rocprofv3 summary, round {rnd}, workload `{workload}`\n\n")
    if is_default:
        f.write("Command (tools/profile_default.sh): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py` "
                "-- the driver's bench command (+ `--detail`, which only names where the full record is written): the headline workload, the five other "
                "single-GPU configurations and the SURVEY 8(f) rows.\n\n")
    else:
        wl = next((workload[: -len(sfx)] for sfx in ("_full", "_unpacked", "_rows7") if workload.endswith(sfx)), workload)
        f.write("Command (tools/profile_gpu.sh): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py "
                f"--steps 10 --warmup 2 --no-cpu-baseline --workload {wl} {extra_args}` and one `--pmc FETCH_SIZE`, one "
                "`--pmc WRITE_SIZE` pass (`--steps 4 --warmup 2`).\n\n")
    f.write("bench.py logs how many evaluate launches every phase issues (`launch_log`); the dispatches of the kernel trace are "
            "sliced with it.  Phases of one workload: 1 in-place warm step, the hipMalloc tangent candidates x 4 launches, the VMM "
            "working set x 4 launches (`--placement auto`), the warm-up steps, 2 launches that read the plastic counts of the two "
            "Newton iterates, the TIMED steps, and (sparse protocol only) six launches of the mask-less kernel variant for the "
            "`full_trial_history` figure and 2 + 6 launches of the sparse protocol on the reference's array layout for the "
            f"`sparse_unpacked_history` figure.  {len(dur)} evaluate dispatches in the trace, {used} accounted for by the log.\n\n")
    f.write("## kernel stats (`*_kernel_stats.csv`, top rows; averages over ALL phases of all workloads that use the kernel)\n\n"
            "| kernel | calls | total ms | avg ms | % | min ms | max ms |\n|---|---|---|---|---|---|---|\n")
    for r in stats[:8]:
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.4f} | {r['Percentage']} | {int(r['MinNs'])/1e6:.4f} | {int(r['MaxNs'])/1e6:.4f} |\n")
    f.write("\n## timed launches per workload: rocprofv3 kernel trace vs bench.py's own HIP events (same run)\n\n"
            "| workload | timed launches | trace avg ms | trace min ms | bench.py events avg ms | frac (bench.py) | placement |\n|---|---|---|---|---|---|---|\n")
    if bench:
        t = by_phase.get(("headline", "timed"), [])
        r = bench.get("roofline", {})
        if t:
            f.write(f"| **{bench['config']['workload'].split(':')[0]}** (headline) | {len(t)} | **{sum(t)/len(t):.4f}** | {min(t):.4f} | "
                    f"{r.get('kernel_ms_avg')} | {r.get('frac')} | {bench.get('placement', {}).get('mode')} |\n")
        for name, c in (bench.get("configs") or {}).items():
            t = by_phase.get((name, "timed"), [])
            if t:
                f.write(f"| {name} | {len(t)} | {sum(t)/len(t):.4f} | {min(t):.4f} | {c.get('kernel_ms_avg')} | {c.get('frac')} | {c.get('placement_mode')} |\n")
    f.write("\n## dispatches of the evaluate kernels (`*_kernel_trace.csv`), by phase\n\n")
    if trace:
        t0 = trace[0]
        f.write(f"grid {t0['Grid_Size_X']} threads, workgroup {t0['Workgroup_Size_X']}, LDS {t0['LDS_Block_Size']} B/block, "
                f"VGPR_Count {t0['VGPR_Count']}, SGPR_Count {t0['SGPR_Count']}, scratch {t0['Scratch_Size']}\n\n")
    for (wl, ph), v in by_phase.items():
        if v:
            f.write(f"* {wl} / {ph}: " + ", ".join(f"{d:.3f}" for d in v) + "\n")
    if is_default:
        f.write(f"\nHBM traffic (PMC passes) of the headline workload: `r{rnd}_von_mises_mixed_rocprof.md`.\n")
    else:
        fetch = sorted([r for r in rows("fetch/*/*_counter_collection.csv") if is_eval(r)], key=lambda r: int(r["Dispatch_Id"]))
        write = sorted([r for r in rows("write/*/*_counter_collection.csv") if is_eval(r)], key=lambda r: int(r["Dispatch_Id"]))
        fb, wb = bench_line("fetch.log"), bench_line("write.log")
        fk = slice_phases([float(r["Counter_Value"]) for r in fetch], fb)[0].get(("headline", "timed"), [])
        wk = slice_phases([float(r["Counter_Value"]) for r in write], wb)[0].get(("headline", "timed"), [])
        fetch_b = 2.0 * 1024.0 * sum(fk) / max(len(fk), 1)
        write_b = 1024.0 * sum(wk) / max(len(wk), 1)
        n = bench.get("config", {}).get("points_per_gpu", 0)
        alg = bench.get("roofline", {}).get("algorithmic_bytes_per_launch", 0)
        f.write("\n## HBM traffic per launch (PMC, corrected; the TIMED launches of the two PMC passes)\n\n")
        f.write(f"* FETCH_SIZE avg {sum(fk)/max(len(fk),1):.1f} KiB over {len(fk)} launches -> x1024 x2 (gfx950 wide-read correction) = **{fetch_b/1e9:.3f} GB read**\n")
        f.write(f"* WRITE_SIZE avg {sum(wk)/max(len(wk),1):.1f} KiB over {len(wk)} launches -> x1024 = **{write_b/1e9:.3f} GB written**\n")
        f.write(f"* total **{(fetch_b+write_b)/1e9:.3f} GB** per launch; algorithmic bytes (bench.py) {alg/1e9:.3f} GB; ratio {(fetch_b+write_b)/max(alg,1):.3f}\n")
        if n:
            f.write(f"* per point: {fetch_b/n:.1f} B read + {write_b/n:.1f} B written = {(fetch_b+write_b)/n:.1f} B\n")
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        d = json.load(open(tj)) if os.path.exists(tj) else {}
        d[workload] = {"n": n, "hbm_bytes_per_launch": int(fetch_b + write_b), "read_bytes": int(fetch_b), "write_bytes": int(write_b),
                       "round": rnd, "source": os.path.basename(out), "kernel_hash": (bench.get("library") or {}).get("kernel_hash") or os.environ.get("TRAFFIC_KERNEL_HASH")}
        json.dump(d, open(tj, "w"), indent=1, sort_keys=True)
    notes = os.path.join(ROOT, "profiles", f"r{rnd}_{workload}_notes.md")  # hand-written analysis kept next to the generated summary
    if os.path.exists(notes):
        f.write("\n" + open(notes).read())
    if bench:
        slim = {k: v for k, v in bench.items() if k not in ("configs", "launch_log", "host_path", "cpu_baseline")}
        if bench.get("configs"):
            slim["configs"] = {k: {kk: vv for kk, vv in c.items() if kk not in ("launch_log", "workload")} for k, c in bench["configs"].items()}
        f.write("\n## bench line of the profiled run (launch logs omitted)\n\n```json\n" + json.dumps(slim) + "\n```\n")
print(out)
