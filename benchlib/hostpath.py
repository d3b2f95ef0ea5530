"""bench.py: the host path (PCIe-inclusive) -- the `host_path` block of the default line and `--mode host`."""

from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib.traffic import library_hash  # noqa: E402
from benchlib.workloads import METRIC, VM_P  # noqa: E402


def host_path_figures(devices=None, sizes=(1_000_000, 10_000_000), latency_sizes=(1_000, 10_000), reps=3, budget_s=12.0,
                      seed=99):
    """PCIe-inclusive figures of the HOST entries -- the call the reference times with Timer("constitutive-law-evaluation")
    (solver/_lawonsubmesh.py:86-94: evaluate on views of Function.x.array) -- VonMises3D, mixed elastic / plastic NumPy arrays:
      evaluate            the reference contract: law.evaluate(ndarrays) in place (fcamd_evaluate_host).  The tangent rows are rebuilt on
                          the host by `tangent_threads` threads (context option "host_tangent_threads", csrc/fcamd_hosttangent.cpp):
                          128 B/pt + 48 per plastic point up, 56 + 120 per plastic point down instead of <= 392;
      evaluate_pcie_tangent  the same call with "host_tangent_threads" = 0 (the kernel writes the tangent over the link: rounds 2-5);
      evaluate_le         LinearElasticityModel.evaluate(ndarrays) in place: 120 B/pt up, 48 down, the tangent filled by the threads;
      resident            ResidentState.evaluate_into (fcamd_evaluate_resident): state on the device, 72 B/pt up, 48 + 64 per plastic
                          point down, the tangent rows from the host threads (336 down with "host_tangent_threads" = 0);
      resident_sparse     the same with the sparse tangent (the product default): only the tangent rows of plastic / formerly
                          plastic points cross PCIe from the second call on;
    each with pageable arrays (page-locked by the library for the duration of the call) and with arrays registered once.
    `devices` = list of device ordinals: the single-process multi-GPU form of the same calls (fcamd_multi: every device on its
    own slice over its own PCIe link); None: one device, the plain objects.  Never part of `value` of the default line."""
    import numpy as np

    import fenics_constitutive_amd as fc
    from fenics_constitutive_amd import _capi

    t_begin = time.perf_counter()
    multi = devices is not None
    n_max = max(sizes)
    rng = np.random.default_rng(seed)
    law = fc.VonMises3D(VM_P)
    law_le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.FULL)
    registered_now = [False]
    if multi:
        law.use_devices(devices)
    g = rng.standard_normal(9 * n_max)
    g *= np.repeat(10.0 ** (rng.random(n_max) * 2.0 - 4.0), 9)
    s0 = np.zeros(6 * n_max)
    a0 = rng.random(n_max) * 0.02
    s, t = np.zeros(6 * n_max), np.zeros(36 * n_max)
    e, a = np.zeros(6 * n_max), a0.copy()
    out = {"law": "VonMises3D, grad scale log-uniform in [1e-4, 1e-2], alpha ~ U(0, 0.02)", "devices": devices or [_capi.default_device()],
           "bytes_per_point": {"evaluate_up": "128 + 48 per plastic point", "evaluate_down": "56 + 120 per plastic point (+ 8 per 64 points)",
                               "evaluate_down_pcie_tangent_max": 392, "resident_up": 72, "resident_down": "48 + 64 per plastic point",
                               "resident_down_pcie_tangent": 336}, "sizes": {}}
    if not multi:
        out["host_tangent_threads"] = law._handle(_capi.default_device()).ctx.get_option("host_tangent_threads")

    def make_state(n):
        if multi:
            from fenics_constitutive_amd.multidevice import MultiDeviceResidentState

            return [MultiDeviceResidentState(fc.VonMises3D(VM_P), n, devices=devices, history0={"eps_n": e[: 6 * n], "alpha": a0[:n]},
                                             sparse_tangent=sp) for sp in (False, True)]
        from fenics_constitutive_amd.resident import ResidentState

        return [ResidentState(law, n, history0={"eps_n": e[: 6 * n], "alpha": a0[:n]}, sparse_tangent=sp, placement="torch") for sp in (False, True)]

    def best_of(fn, k, reset=None):
        best = None
        for _ in range(k):
            if reset:
                reset()
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    def figures(n, k, lat=False):
        gs, ss, ts, es, al = g[: 9 * n], s[: 6 * n], t[: 36 * n], e[: 6 * n], a[:n]
        full, sparse = make_state(n)

        def reset():
            ss[:] = 0.0
            es[:] = 0.0
            al[:] = a0[:n]

        row = {}
        ctx = None if multi else law._handle(_capi.default_device()).ctx

        def kernel_tangent():
            ctx.set_option("host_tangent_threads", 0)
            try:
                law.evaluate(0.0, 1.0, gs, ss, ts, {"eps_n": es, "alpha": al})
            finally:
                ctx.set_option("host_tangent_threads", -1)

        def reset_le():
            ss[:] = 0.0

        legs = [("evaluate", lambda: law.evaluate(0.0, 1.0, gs, ss, ts, {"eps_n": es, "alpha": al}), reset, 568),
                ("resident", lambda: full.evaluate_into(0.0, 1.0, gs, ss, ts), None, 408),
                ("resident_sparse", lambda: sparse.evaluate_into(0.0, 1.0, gs, ss, ts), None, 408)]
        if ctx is not None and not lat and registered_now[0]:
            legs += [("evaluate_pcie_tangent", kernel_tangent, reset, 568),
                     ("evaluate_le", lambda: law_le.evaluate(0.0, 1.0, gs, ss, ts, None), reset_le, 456)]
        for name, fn, rs_, bpp in legs:
            fn()  # warm: first touch, first page lock, (sparse) the full tangent
            dt = best_of(fn, k, rs_)
            if lat:
                row[name + "_us"] = round(dt * 1e6, 1)
            else:
                row[name] = {"ms": round(dt * 1e3, 3), "Mpts_s": round(n / dt / 1e6, 1), "interface_GBs": round(n * bpp / dt / 1e9, 2)}
                if ctx is not None:  # who wrote the tangent rows of the leg's last call, and what it cost the host
                    row[name]["tangent_threads"] = ctx.get_option("last_host_tangent_threads")
                    row[name]["tangent_cpu_ms"] = round(ctx.get_option("last_host_tangent_cpu_us") / 1e3, 2)
        if not lat:
            row["plastic_fraction"] = round(law.last_stats.n_plastic / n, 4)
        for st in (full, sparse):
            if multi:
                st.close()
        return row

    pin_target = None
    try:
        for registered in (False, True):
            if registered:
                if multi:
                    pin_target = law._multi()
                else:
                    pin_target = law._handle(_capi.default_device()).ctx
                for x in (g, s, t, e, a):
                    pin_target.register_host_buffer(x)
            registered_now[0] = registered
            key = "registered" if registered else "pageable"
            for n in sizes:
                if time.perf_counter() - t_begin > budget_s and n != min(sizes):
                    out["sizes"].setdefault(str(n), {})[key] = "skipped: time budget"
                    continue
                out["sizes"].setdefault(str(n), {})[key] = figures(n, reps)
            for n in latency_sizes:
                out.setdefault("per_call_us", {}).setdefault(str(n), {})[key] = figures(n, 30, lat=True)
        # what the link gives a plain copy between the registered tangent array and device memory
        import torch

        from fenics_constitutive_amd.hostio import download, upload

        dev = torch.device("cuda", (devices or [_capi.default_device()])[0])
        m = min(n_max, 4_000_000)
        buf = torch.empty(36 * m, dtype=torch.float64, device=dev)
        upload(buf, t[: 36 * m])
        h2d = best_of(lambda: upload(buf, t[: 36 * m]), 3)
        d2h = best_of(lambda: download(t[: 36 * m], buf), 3)
        out["pinned_copy_GBs"] = {"h2d": round(288 * m / h2d / 1e9, 1), "d2h": round(288 * m / d2h / 1e9, 1),
                                  "note": "fcamd_copy_to_device / _to_host of 36 doubles x %d points between the registered tangent array and one device" % m}
        big = out["sizes"].get(str(n_max), {}).get("registered")
        if isinstance(big, dict):
            ndev = len(devices) if multi else 1
            ref = big.get("evaluate_pcie_tangent") or (big["evaluate"] if big["evaluate"].get("tangent_threads", 0) == 0 else None)
            if ref:  # the kernel's tangent over the link: the downlink is the bound
                out["evaluate_d2h_over_pinned_copy"] = round(n_max * 392 / (ref["ms"] * 1e-3) / 1e9 / (out["pinned_copy_GBs"]["d2h"] * ndev), 3)
            if big["evaluate"].get("tangent_threads", 0) > 0:  # the tangent from the host threads: the uplink is (what the kernel reads over it)
                up = n_max * (128 + 48 * big.get("plastic_fraction", 0.0)) / (big["evaluate"]["ms"] * 1e-3) / 1e9
                out["evaluate_h2d_over_pinned_copy"] = round(up / (out["pinned_copy_GBs"]["h2d"] * ndev), 3)
    finally:
        if pin_target is not None:
            for x in (g, s, t, e, a):
                try:
                    pin_target.unregister_host_buffer(x)
                except Exception:
                    pass
    out["wall_s"] = round(time.perf_counter() - t_begin, 1)
    return out


def main_host(args):
    """--mode host: the single-process multi-GPU host path (fcamd_multi).  ONE process drives --gpus devices; under
    torch.distributed.run every rank but 0 leaves at once (nothing on this path needs a process group)."""
    rank = int(os.environ.get("RANK", "0"))
    if rank != 0:
        return 0
    import torch

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    have = torch.cuda.device_count()
    if args.host_devices:
        devices = [int(x) for x in args.host_devices.split(",")]
    else:
        devices = [k % have for k in range(args.gpus)]  # fewer GPUs than asked for: contexts share devices (rehearsal)
    n_total = args.n * len(devices) if args.scaling == "weak" else args.n
    t_start = time.perf_counter()
    import numpy as np

    import fenics_constitutive_amd as fc
    from fenics_constitutive_amd import _capi

    fig = host_path_figures(devices=devices, sizes=(min(1_000_000, n_total), n_total), latency_sizes=(1_000, 10_000),
                            reps=max(2, min(args.steps, 5)), budget_s=args.wall_budget / 2)
    # the timed steps proper: the reference contract (in-place evaluate on pageable NumPy arrays) over all devices
    rng = np.random.default_rng(5)
    law = fc.VonMises3D(VM_P).use_devices(devices)
    g = rng.standard_normal(9 * n_total)
    g *= np.repeat(10.0 ** (rng.random(n_total) * 2.0 - 4.0), 9)
    a0 = rng.random(n_total) * 0.02
    s, t, e, a = np.zeros(6 * n_total), np.zeros(36 * n_total), np.zeros(6 * n_total), a0.copy()

    def step():
        law.evaluate(0.0, 1.0, g, s, t, {"eps_n": e, "alpha": a})

    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    mode, used = law._multi().last_host_mode()
    n_pl = int(law.last_stats.n_plastic)
    bytes_step = n_total * 176 + (n_total - n_pl) * 336 + n_pl * 392
    out = {"metric": METRIC, "value": round(n_total * args.steps / elapsed / 1e6, 1), "unit": "Mpts/s", "n_gpus": len(devices),
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic", "mode": "host",
           "config": {"workload": f"host path: VonMises3D FULL-3D, {n_total} quadrature points in ONE process's pageable NumPy arrays, in-place "
                                  f"evaluate (the reference contract) spread over {len(devices)} device contexts by fcamd_multi_evaluate_host -- every "
                                  f"device on its own slice over its own PCIe link, no gather; PCIe-inclusive by construction",
                      "points_total": n_total, "devices": devices, "devices_used": used, "host_mode_flags": mode,
                      "plastic_fraction": round(n_pl / n_total, 4), "parallelism": f"one process x {len(devices)} device contexts"},
           "roofline": {"bound": "pcie", "achieved": round(bytes_step * args.steps / elapsed / 1e9, 2), "unit": "GB/s",
                        "peak": None if "pinned_copy_GBs" not in fig else round((fig["pinned_copy_GBs"]["h2d"] + fig["pinned_copy_GBs"]["d2h"]) * len(set(devices)), 1),
                        "frac": None, "traffic": None,
                        "note": "achieved = interface bytes over PCIe per step (176 B/pt up; 336 down for elastic, 392 for plastic points) / step time, "
                                "both directions counted; peak = measured pinned H2D + D2H copy rate of one link x distinct devices"},
           "host_path": fig, "cpu_baseline": None, "library": {"srchash": library_hash(), "kernel_hash": library_hash(kernels_only=True)}}
    if out["roofline"]["peak"]:
        out["roofline"]["frac"] = round(out["roofline"]["achieved"] / out["roofline"]["peak"], 4)
    out["wall_s"] = round(time.perf_counter() - t_start, 1)
    from benchlib.line import compact_line, dumps, write_detail

    line = compact_line(out, write_detail(out, args.detail))  # the full record (host_path tables, notes) is the detail file
    line["mode"] = "host"
    line["config"].update({"devices_used": used, "devices": len(devices)})
    print(dumps(line), flush=True)
    return 0
