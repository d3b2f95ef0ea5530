#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs written by tools/profile_gpu.sh into the committed summaries:

    python tools/summarize_profile.py gpurun_out/prof/<tag> <round> <workload>

writes profiles/r<round>_<workload>_rocprof.md (kernel stats + per-dispatch durations + PMC
traffic) and updates profiles/traffic.json (HBM bytes per launch, read by bench.py for
roofline.traffic).  PMC corrections per MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import csv
import glob
import json
import os
import sys

src, rnd, workload = sys.argv[1], sys.argv[2], sys.argv[3]
extra_args = sys.argv[4] if len(sys.argv) > 4 else ""  # bench flags beyond --workload (e.g. "--history full")
is_default = workload == "default_bench"  # tools/profile_default.sh: `python3 bench.py`, kernel trace only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)  # newest run if several were merged
    return list(csv.DictReader(open(f[-1]))) if f else []


stats = rows("kt/*/*_kernel_stats.csv")
trace = [r for r in rows("kt/*/*_kernel_trace.csv") if "fcamd::" in r["Kernel_Name"]]
fetch = [r for r in rows("fetch/*/*_counter_collection.csv") if "fcamd::" in r["Kernel_Name"]]
write = [r for r in rows("write/*/*_counter_collection.csv") if "fcamd::" in r["Kernel_Name"]]
bench = {}
bj = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(bj) and os.path.getsize(bj):
    bench = json.load(open(bj))

dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in trace]
# the first dispatch is the untimed in-place warm step of bench.py (different state); the timed
# launches are the last `steps` ones of the variant the timed region runs -- after it bench.py issues six
# launches of the mask-less variant (another template instance) for its "full_trial_history" figure
steps = bench.get("steps", 10)
names = [r["Kernel_Name"] for r in trace]
main = max(set(names), key=names.count) if names else None
timed = [d for d, nm in zip(dur, names) if nm == main][-steps:]


def main_variant(rows_):
    """Counter values of the kernel variant the timed region runs (the most frequent one), without the
    first dispatch of the run (the in-place warm step) when that is the same variant."""
    nm = [r["Kernel_Name"] for r in rows_]
    if not nm:
        return []
    top = max(set(nm), key=nm.count)
    vals = [float(r["Counter_Value"]) for r in rows_ if r["Kernel_Name"] == top]
    return vals[1:] if nm[0] == top else vals


fk = main_variant(fetch)
wk = main_variant(write)
fetch_b = 2.0 * 1024.0 * sum(fk) / max(len(fk), 1)
write_b = 1024.0 * sum(wk) / max(len(wk), 1)
n = bench.get("config", {}).get("points_per_gpu", 0)
alg = bench.get("roofline", {}).get("algorithmic_bytes_per_launch", 0)

out = os.path.join(ROOT, "profiles", f"r{rnd}_{workload}_rocprof.md")
with open(out, "w") as f:
    f.write(f"# rocprofv3 summary, round {rnd}, workload `{workload}`\n\n")
    if is_default:
        f.write("Command (tools/profile_default.sh): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py` "
                "-- the default bench command, no flags.\n\n")
    else:
        wl = workload[:-5] if workload.endswith("_full") else workload
        f.write("Command (tools/profile_gpu.sh): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py "
                f"--steps 10 --warmup 2 --no-cpu-baseline --workload {wl} {extra_args}` and one `--pmc FETCH_SIZE`, one "
                "`--pmc WRITE_SIZE` pass (`--steps 4 --warmup 2`).\n\n")
    f.write("Launch sequence of one bench run: 1 in-place warm step (kernel variant `<..., false>`), the placement candidates "
            "x 4 launches (bench.py --placement-tries, the slower candidates are part of the `kernel_stats` average below), "
            "the warm-up steps, 2 launches that read the plastic counts of the two Newton iterates, the timed steps, and "
            "(sparse protocol only) six launches of the mask-less kernel variant for the `full_trial_history` figure.\n\n")
    f.write("## kernel stats (`*_kernel_stats.csv`, top rows)\n\n| kernel | calls | total ms | avg ms | % | min ms | max ms |\n|---|---|---|---|---|---|---|\n")
    for r in stats[:6]:
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.4f} | {r['Percentage']} | {int(r['MinNs'])/1e6:.4f} | {int(r['MaxNs'])/1e6:.4f} |\n")
    f.write("\n## dispatches of the evaluate kernel (`*_kernel_trace.csv`)\n\n")
    if trace:
        t0 = trace[0]
        f.write(f"grid {t0['Grid_Size_X']} threads, workgroup {t0['Workgroup_Size_X']}, LDS {t0['LDS_Block_Size']} B/block, "
                f"VGPR_Count {t0['VGPR_Count']}, SGPR_Count {t0['SGPR_Count']}, scratch {t0['Scratch_Size']}\n\n")
    f.write("durations (ms): " + ", ".join(f"{d:.4f}" for d in dur) + "\n\n")
    if timed:
        f.write(f"timed launches (last {len(timed)}): avg **{sum(timed)/len(timed):.4f} ms**, min {min(timed):.4f} ms; "
                f"bench.py's own HIP-event average in the same run: {bench.get('roofline', {}).get('kernel_ms_avg')} ms\n\n")
    if is_default:
        f.write("HBM traffic (PMC passes) of this workload: `r%s_von_mises_mixed_rocprof.md`.\n" % rnd)
    else:
      f.write("## HBM traffic per launch (PMC, corrected)\n\n")
      f.write(f"* FETCH_SIZE avg {sum(fk)/max(len(fk),1):.1f} KiB -> x1024 x2 (gfx950 wide-read correction) = **{fetch_b/1e9:.3f} GB read**\n")
      f.write(f"* WRITE_SIZE avg {sum(wk)/max(len(wk),1):.1f} KiB -> x1024 = **{write_b/1e9:.3f} GB written**\n")
      f.write(f"* total **{(fetch_b+write_b)/1e9:.3f} GB** per launch; algorithmic bytes (bench.py) {alg/1e9:.3f} GB; ratio {(fetch_b+write_b)/max(alg,1):.3f}\n")
      if n:
        f.write(f"* per point: {fetch_b/n:.1f} B read + {write_b/n:.1f} B written = {(fetch_b+write_b)/n:.1f} B\n")
    if bench:
        f.write("\n## bench line of the profiled run\n\n```json\n" + json.dumps(bench) + "\n```\n")
tj = os.path.join(ROOT, "profiles", "traffic.json")
d = json.load(open(tj)) if os.path.exists(tj) else {}
if is_default:
    print(out)
    sys.exit(0)
d[workload] = {"n": n, "hbm_bytes_per_launch": int(fetch_b + write_b), "read_bytes": int(fetch_b), "write_bytes": int(write_b),
               "round": rnd, "source": os.path.basename(out)}
json.dump(d, open(tj, "w"), indent=1, sort_keys=True)
print(out)
