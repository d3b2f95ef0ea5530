"""One-screen summary of a bench.py JSON line (last line of the file given)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
strip = lambda x: {k: v for k, v in (x or {}).items() if k not in ("note", "launch_log", "workload", "cpu", "placement_candidate_ms", "bytes_per_point", "traffic_source")}  # noqa: E731
print("value", d["value"], "frac", r["frac"], "kernel_ms", r["kernel_ms_avg"], "alg", r["algorithmic_bytes_per_launch"], "traffic", r.get("traffic"),
      "ratio", None if not r.get("traffic") else round(r["traffic"] / r["algorithmic_bytes_per_launch"], 4), r.get("traffic_read_write"))
print({k: v for k, v in r.items() if k.startswith("frac_")}, d["placement"].get("mode"))
for k in ("full_trial_history", "sparse_unpacked_history", "strong_scaling"):
    if k in d:
        print(k, strip(d[k]))
for k, c in (d.get("configs") or {}).items():
    print(k, strip(c))
print("wall_s", d.get("wall_s"), "cpu_baseline", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "unit", "cores", "kind")})
