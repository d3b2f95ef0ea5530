"""bench.py: the CPU baselines -- oracle/ (the checker) timed on the GPU box's host cores.  The ONLY part of bench.py that imports oracle/."""

from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib.workloads import LE_P, SLS_P, VM_P  # noqa: E402


def cpu_quick(wl, budget_s=1.0, ns=1_000_000):
    """A short CPU figure for one configuration: the C port (oracle/oracle.c, serial loop, 1 thread) and the NumPy
    restatement of the reference's code path on the first `ns` points of the configuration's own arrays."""
    import numpy as np

    from fenics_constitutive_amd.hostio import to_host
    from oracle import c_oracle as CO
    from oracle import numpy_oracle as NO

    ns = min(ns, wl.n)
    dims = {"eps_n": 6, "alpha": 1, "strain_visco": 6, "strain": 6, "history": 7}
    g = to_host(wl.grads[0][: 9 * ns])
    s0 = to_host(wl.stress_c[: 6 * ns])
    h0 = None if wl.hist_c is None else {k: to_host(v[: dims[k] * ns]) for k, v in wl.reference_history().items()}
    tan = np.zeros(36 * ns)
    out = {}
    for label, fn, m in (("c_port_1_thread_Mpts_s", CO.MODELS[wl.kind], ns), ("numpy_port_Mpts_s", NO.MODELS[wl.kind], min(ns, 200_000))):
        def one_pass():
            s = s0[: 6 * m].copy()
            h = None if h0 is None else {k: v[: dims[k] * m].copy() for k, v in h0.items()}
            t0 = time.perf_counter()
            fn(wl.params, 0.0, wl.del_t, g[: 9 * m], s, tan[: 36 * m], h)
            return time.perf_counter() - t0

        one_pass()  # untimed: faults in the pages of the output arrays
        reps, tt = 0, 0.0
        while tt < budget_s and reps < 50:
            tt += one_pass()
            reps += 1
        out[label] = round(m * reps / tt / 1e6, 2)
    out["sample"] = f"first {ns} points of this configuration's arrays, ~{budget_s:.0f} s each"
    return out


def usable_cores():
    """(threads, why): the cores an all-cores figure may use -- physical cores (one thread per SMT sibling group) the process may run
    on, within the CPUs' worth of run time its cgroup grants (cgroup v2 cpu.max).  Round 5's line ran 256 threads under a quota of
    16 CPUs next to the PMC children and reported the one-thread rate as "all cores"."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    groups = set()
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                groups.add(fh.read().strip())
        except OSError:
            groups.add(str(c))
    physical = max(1, len(groups))
    why = f"{physical} physical cores ({len(allowed)} logical CPUs allowed)"
    threads = physical
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            q = max(1, -(-int(quota) // int(period)))
            if q < threads:
                threads, why = q, why + f", cgroup cpu.max = {q} CPUs"
    except (OSError, ValueError):
        pass
    return threads, why


def host_sample(kind, params, grad, stress, hist, del_t):
    """the sample of the headline arrays as NumPy arrays (device -> host copies, on the calling thread)"""
    from fenics_constitutive_amd.hostio import to_host

    return (kind, params, to_host(grad), to_host(stress), None if hist is None else {k: to_host(v) for k, v in hist.items()}, del_t)


def cpu_baseline(kind, params, grad, stress, hist, del_t, budget_s=8.0, extras="full"):
    """Time the C oracle ("port": serial per-point loop, 1 thread -- what the reference does per
    MPI rank) on a bounded sample of the same workload (NumPy arrays: host_sample()).  extras: "full" adds the small-call
    crossover table (GPU calls: not while a PMC pass is running) and longer side figures, "lean" (the default command)
    keeps the side figures short."""
    import numpy as np

    from oracle import c_oracle as CO

    ns = grad.size // 9  # the sample bench.py took: the first 1e7 points of the headline arrays (n = 1e7)
    g = grad[: 9 * ns]
    s0 = stress[: 6 * ns]
    dims = {"eps_n": 6, "alpha": 1, "strain_visco": 6, "strain": 6, "history": 7}
    h0 = None if hist is None else {k: v[: dims[k] * ns] for k, v in hist.items()}
    side_s = 1.5 if extras == "full" else 0.8
    tan = np.zeros(36 * ns)
    fn = CO.MODELS[kind]

    def one_pass():
        s = s0.copy()
        h = None if h0 is None else {k: v.copy() for k, v in h0.items()}
        t0 = time.perf_counter()
        fn(params, 0.0, del_t, g, s, tan, h)
        return time.perf_counter() - t0

    one_pass()  # untimed: faults in the pages of the output arrays
    reps, t_total = 0, 0.0
    while t_total < budget_s and reps < 500:
        t_total += one_pass()
        reps += 1
    out = {
        "value": round(ns * reps / t_total / 1e6, 3),
        "unit": "Mpts/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle/oracle.c serial loop ({CO.build_flags()}), first {ns} points of the headline workload x {reps} passes ({t_total:.1f} s)",
    }
    # the reference's own NumPy code path, restated (oracle/numpy_oracle.py): a few seconds, for scale
    try:
        from oracle import numpy_oracle as NO

        def time_np(fn, m):
            s = s0[: 6 * m].copy()
            h = None if h0 is None else {k: v[: dims[k] * m].copy() for k, v in h0.items()}
            t0 = time.perf_counter()
            fn(params, 0.0, del_t, g[: 9 * m], s, tan[: 36 * m], h)
            return round(m / (time.perf_counter() - t0) / 1e6, 4)

        extra = {"numpy_port_Mpts_s": time_np(NO.MODELS[kind], min(ns, 100_000 if kind == "comfe_drucker_prager" else 500_000)),
                 "threads": "NumPy/OpenBLAS default"}
        if kind == "von_mises_3d":
            extra["python_per_point_loop_port_Mpts_s"] = time_np(NO.von_mises_3d_loop, min(ns, 20_000))
            # BASELINE config 3 compares with the comfe-rs CPU path: our C restatement of the serial
            # evaluate_model loop around MisesPlasticity3D (interfaces.rs:354-456, mises_plasticity.rs:58-126;
            # mu, kappa, y_0 as above, h = 200 as in tests/models/test_plasticity.py:26-31) on the same
            # gradients and stresses, 1 thread
            hr = np.zeros(7 * ns)
            hr.reshape(-1, 7)[:, 0] = h0["alpha"]
            rs_p = {"mu": params["p_mu"], "kappa": params["p_ka"], "y_0": params["p_y0"], "h": 200.0}
            tt, rr = 0.0, 0
            while tt < side_s and rr < 100:
                s, hh = s0.copy(), {"history": hr.copy()}
                t0 = time.perf_counter()
                CO.MODELS["comfe_mises_plasticity"](rs_p, 0.0, del_t, g, s, tan, hh)
                tt += time.perf_counter() - t0
                rr += 1
            extra["comfe_rs_mises_c_port_1_thread_Mpts_s"] = round(ns * rr / tt / 1e6, 2)
        # BASELINE configs[0]: LinearElasticityModel FULL-3D, 1e5 points, the reference's NumPy evaluate() on the CPU --
        # here its NumPy restatement (and the C port) on the SURVEY 8d cfg1 inputs (grad ~ N(0, 1e-3^2), sigma = 0, E = 42, nu = 0.3, seed 0)
        rng = np.random.default_rng(0)
        g0, t0_ = rng.normal(scale=1e-3, size=9 * 100_000), np.zeros(36 * 100_000)
        for label, f0 in (("config0_le_1e5_numpy_port_Mpts_s", NO.MODELS["linear_elasticity"]), ("config0_le_1e5_c_port_Mpts_s", CO.MODELS["linear_elasticity"])):
            best = None
            for _ in range(5):
                s_ = np.zeros(6 * 100_000)
                tq = time.perf_counter()
                f0(LE_P, 0.0, 1.0, g0, s_, t0_, None)
                dq = time.perf_counter() - tq
                best = dq if best is None else min(best, dq)
            extra[label] = round(0.1 / best, 2)
        # the same C loop on all the cores this process may use (OpenMP over points; usable_cores()), for scale only: what a
        # rank-per-core run of the reference's serial loop adds up to on this host's share of the box
        nthr, why = usable_cores()
        nthr = max(1, min(nthr, CO.max_threads()))
        CO.set_num_threads(nthr)
        try:
            one_pass()
            tt, rr = 0.0, 0
            while tt < max(side_s, 2.0) and rr < 200:
                tt += one_pass()
                rr += 1
        finally:
            CO.set_num_threads(1)
        extra["c_port_all_cores_Mpts_s"] = round(ns * rr / tt / 1e6, 1)
        extra["c_port_all_cores_threads"] = nthr
        extra["c_port_all_cores_per_thread_Mpts_s"] = round(ns * rr / tt / 1e6 / nthr, 2)
        extra["c_port_all_cores_speedup"] = round((ns * rr / tt / 1e6) / out["value"], 1) if out["value"] else None
        extra["c_port_all_cores_note"] = f"{why}; {ns} points x {rr} passes ({tt:.1f} s), nothing else running on the host or the GPU meanwhile"
        out["extra"] = extra
    except Exception as e:  # the extra figures are informational only
        out["extra"] = {"error": str(e)}
    if extras == "full":
        try:
            out["small_call_crossover"] = small_call_crossover()
        except Exception as e:  # informational
            out["small_call_crossover"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def small_call_crossover(sizes=(64, 256, 1024, 4096, 16384, 65536), reps=7):
    """Per law: the number of points below which ONE ndarray ``evaluate`` call on the GPU (launch + PCIe round trips: a floor
    of tens of microseconds) loses to the NumPy restatement of the reference's own code path on this box's host -- what a
    dolfinx rank with a few thousand quadrature points per law pays (solver/_lawonsubmesh.py:86-94).  Medians of `reps`
    calls per size; the crossover is interpolated between the two sizes where the order flips.  DeviceLaw.evaluate warns
    once below `device.SMALL_CALL_POINTS` (INTEGRATION.md states the measured table)."""
    import numpy as np

    import fenics_constitutive_amd as fc
    from fenics_constitutive_amd import device as fdev
    from oracle import numpy_oracle as NO

    FULL = fc.StressStrainConstraint.FULL
    rng = np.random.default_rng(5)
    cases = {"linear_elasticity": (fc.LinearElasticityModel(LE_P, FULL), LE_P, None, 1e-3),
             "von_mises_3d": (fc.VonMises3D(VM_P), VM_P, {"eps_n": 6, "alpha": 1}, 3e-3),
             "spring_maxwell": (fc.SpringMaxwellModel(SLS_P, FULL), SLS_P, {"strain_visco": 6, "strain": 6}, 1e-3),
             "spring_kelvin": (fc.SpringKelvinModel(SLS_P, FULL), SLS_P, {"strain_visco": 6, "strain": 6}, 1e-3)}
    out = {"sizes": list(sizes), "unit": "us per call (median)", "warn_below_points": dict(fdev.SMALL_CALL_POINTS)}
    import warnings

    for kind, (law, params, hd, scale) in cases.items():
        gpu_us, np_us = [], []
        for n in sizes:
            g = rng.normal(scale=scale, size=9 * n)
            s0 = rng.normal(size=6 * n)
            h0 = None if hd is None else {k: np.abs(rng.normal(scale=1e-3, size=d * n)) for k, d in hd.items()}
            t = np.zeros(36 * n)

            def run(fn, is_law):
                ts = []
                for _ in range(reps + 2):
                    s = s0.copy()
                    h = None if h0 is None else {k: v.copy() for k, v in h0.items()}
                    t0 = time.perf_counter()
                    if is_law:
                        fn.evaluate(0.0, 2.0, g, s, t, h)
                    else:
                        fn(params, 0.0, 2.0, g, s, t, h)
                    ts.append(time.perf_counter() - t0)
                return sorted(ts[2:])[reps // 2] * 1e6

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")  # the very warning this table calibrates
                gpu_us.append(round(run(law, True), 1))
            np_us.append(round(run(NO.MODELS[kind], False), 1))
        cross = None
        for k in range(len(sizes)):
            if gpu_us[k] <= np_us[k]:
                if k == 0:
                    cross = sizes[0]
                else:  # linear interpolation of the difference between the two sizes
                    d0, d1 = gpu_us[k - 1] - np_us[k - 1], gpu_us[k] - np_us[k]
                    cross = int(sizes[k - 1] + (sizes[k] - sizes[k - 1]) * d0 / (d0 - d1)) if d0 != d1 else sizes[k]
                break
        out[kind] = {"gpu_call_us": gpu_us, "numpy_port_us": np_us, "crossover_points": cross}
    return out
