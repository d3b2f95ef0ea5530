"""The host (ndarray) entries with the tangent rebuilt on the CPU against the kernel's own tangent stores over PCIe.

    python tools/host_tangent_bench.py [--n 10000000] [--threads 0,4,8,16] [--chunks 0] [--out gpurun_out/host_tangent.json]

Per law (VonMises3D mixed, LinearElasticityModel, SpringMaxwellModel) and per thread count ("host_tangent_threads"; 0 = the
kernel writes the tangent over the link): Mpts/s of law.evaluate(ndarrays) in place on pageable and on registered arrays, of
ResidentState.evaluate_into (full tangent), the summed CPU time of the expansion threads, and whether the tangent is bit for bit the
kernel's.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--threads", default="0,4,8,16")
    ap.add_argument("--chunks", default="0")
    ap.add_argument("--streams", default="1", help="host_tangent_streams values to sweep")
    ap.add_argument("--skip-pageable", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--laws", default="von_mises,linear_elasticity,maxwell")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import fenics_constitutive_amd as fc
    from fenics_constitutive_amd import _capi
    from fenics_constitutive_amd.resident import ResidentState

    n = args.n
    rng = np.random.default_rng(3)
    ctx = _capi.get_context(_capi.default_device())
    out = {"n": n, "cpus_affinity": len(os.sched_getaffinity(0)), "cpus": os.cpu_count(), "auto_threads": ctx.get_option("host_tangent_threads"), "rows": []}
    FULL = fc.StressStrainConstraint.FULL
    laws = {
        "von_mises": (lambda: fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}), {"eps_n": 6, "alpha": 1}),
        "linear_elasticity": (lambda: fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, FULL), None),
        "maxwell": (lambda: fc.SpringMaxwellModel({"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}, FULL), {"strain_visco": 6, "strain": 6}),
    }
    g = rng.standard_normal(9 * n)
    g *= np.repeat(10.0 ** (rng.random(n) * 2.0 - 4.0), 9)
    a0 = rng.random(n) * 0.02

    def best(fn, reset=None):
        b = None
        for _ in range(args.reps):
            if reset:
                reset()
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            b = dt if b is None else min(b, dt)
        return b

    for name in args.laws.split(","):
        make, hd = laws[name]
        law = make()
        s = np.zeros(6 * n)
        t = np.zeros(36 * n)
        h = None if hd is None else {k: np.zeros(d * n) for k, d in hd.items()}
        t_ref = None

        def reset():
            s[:] = 0.0
            if h is not None:
                for k, v in h.items():
                    v[:] = a0 if k == "alpha" else 0.0

        for registered in ((True,) if args.skip_pageable else (False, True)):
            arrays = [g, s, t] + ([] if h is None else list(h.values()))
            if registered:
                for x in arrays:
                    ctx.register_host_buffer(x)
            try:
                for th, chunk, numa in [(int(a), int(b), int(c)) for a in args.threads.split(",") for b in args.chunks.split(",") for c in args.streams.split(",")]:
                    if True:
                        if th == 0 and (chunk != int(args.chunks.split(",")[0]) or numa != int(args.streams.split(",")[0])):
                            continue
                        ctx.set_option("host_tangent_streams", numa)
                        ctx.set_option("host_tangent_threads", th)
                        ctx.set_option("host_tangent_chunk", chunk)
                        reset()
                        t[:] = np.nan
                        law.evaluate(0.0, 1.0, g, s, t, h)  # warm
                        if t_ref is None:
                            t_ref = t.copy()
                        same = bool(np.array_equal(t.view(np.uint64), t_ref.view(np.uint64)))
                        dt = best(lambda: law.evaluate(0.0, 1.0, g, s, t, h), reset)
                        row = {"law": name, "entry": "evaluate", "registered": registered, "threads": th, "chunk": chunk, "streams": numa, "ms": round(dt * 1e3, 2),
                               "Mpts_s": round(n / dt / 1e6, 1), "identical": same, "cpu_ms": round(ctx.get_option("last_host_tangent_cpu_us") / 1e3, 1),
                               "mode": ctx.last_host_mode()}
                        out["rows"].append(row)
                        print(json.dumps(row), flush=True)
                if registered:  # the resident entry, full tangent
                    for th in [int(x) for x in args.threads.split(",")]:
                        ctx.set_option("host_tangent_threads", th)
                        ctx.set_option("host_tangent_chunk", 0)
                        reset()
                        st = ResidentState(law, n, history0=h, sparse_tangent=False, reuse_constant_tangent=False, placement="torch")
                        st.evaluate_into(0.0, 1.0, g, s, t)
                        dt = best(lambda: st.evaluate_into(0.0, 1.0, g, s, t))
                        row = {"law": name, "entry": "resident", "registered": True, "threads": th, "ms": round(dt * 1e3, 2), "Mpts_s": round(n / dt / 1e6, 1),
                               "identical": bool(np.array_equal(t.view(np.uint64), t_ref.view(np.uint64))), "cpu_ms": round(ctx.get_option("last_host_tangent_cpu_us") / 1e3, 1)}
                        out["rows"].append(row)
                        print(json.dumps(row), flush=True)
                        del st
            finally:
                if registered:
                    for x in arrays:
                        ctx.unregister_host_buffer(x)
        ctx.set_option("host_tangent_threads", -1)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
