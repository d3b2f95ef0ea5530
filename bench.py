#!/usr/bin/env python3
"""Benchmark of the hot path: quadrature-point stress updates per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--n POINTS]

One "step" = one ``evaluate`` pass of one constitutive law over n synthetic quadrature points
per GPU, device-resident (inputs already in HBM when the timed region starts), committed state in,
trial state out -- the call of the product's device-resident Newton loop (ResidentState.evaluate;
for the plasticity laws with the sparse trial-history protocol, --history full for the mask-less
form).  The steps alternate between two Newton iterates of the increment.  Default workload at
N = 1: BASELINE.json's target configuration, VonMises3D return mapping, 1e8 points, mixed
elastic/plastic (SURVEY.md 8d cfg3 "mixed").  Before the warm-up the placement of the tangent array
is chosen out of a few candidate allocations.  For N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank evaluates its own contiguous shard of
n points (weak scaling, no data-path collective: SURVEY.md 8e); the optional stress/tangent
all-gather of the single-assembler mode is timed separately and reported under "allgather".

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
METRIC = "quadrature-point stress updates/sec (Mpts/s) + % HBM roofline, 1/2/4/8 GPU"  # BASELINE.json "metric"

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
RS_P = {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0}
SLS_P = {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
LE_P = {"E": 42.0, "nu": 0.3}

# workload -> (law kind, strain scale spec, bytes/pt elastic, bytes/pt plastic, history dims)
WORKLOADS = {
    "von_mises_mixed": ("von_mises_3d", "loguniform", 464, 568),
    # the same 22 % of plastic points, but in contiguous zones of 4096 points (what a mesh-ordered
    # plastic zone looks like) instead of a random mixture in every 64-point tile
    "von_mises_zoned": ("von_mises_3d", "zoned", 464, 568),
    "von_mises_plastic": ("von_mises_3d", 1e-2, 464, 568),
    "von_mises_elastic": ("von_mises_3d", 1e-4, 464, 568),
    "linear_elasticity": ("linear_elasticity", 1e-3, 456, 456),
    "spring_maxwell": ("spring_maxwell", 1e-3, 648, 648),
    "spring_kelvin": ("spring_kelvin", 1e-3, 648, 648),
    "comfe_mises_mixed": ("comfe_mises_plasticity", "loguniform", 464, 568),
    # general return mapping (Newton per plastic point, invariant coordinates)
    "drucker_prager_mixed": ("comfe_drucker_prager", "isochoric", 464, 568),
    "drucker_prager_zoned": ("comfe_drucker_prager", "isochoric_zoned", 464, 568),
    "comfe_mises_zoned": ("comfe_mises_plasticity", "zoned", 464, 568),
}
DP_P = {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02}


def make_law(kind):
    import numpy as np

    import fenics_constitutive_amd as fc

    FULL = fc.StressStrainConstraint.FULL
    if kind == "von_mises_3d":
        return fc.VonMises3D(VM_P), VM_P
    if kind == "linear_elasticity":
        return fc.LinearElasticityModel(LE_P, FULL), LE_P
    if kind == "spring_maxwell":
        return fc.SpringMaxwellModel(SLS_P, FULL), SLS_P
    if kind == "spring_kelvin":
        return fc.SpringKelvinModel(SLS_P, FULL), SLS_P
    if kind == "comfe_mises_plasticity":
        return fc.MisesPlasticityLinearHardening3D({k: np.array([v]) for k, v in RS_P.items()}), RS_P
    if kind == "comfe_drucker_prager":
        return fc.DruckerPrager3D({k: np.array([v]) for k, v in DP_P.items()}), DP_P
    raise ValueError(kind)


def synth_inputs(kind, scale_spec, n, seed, device):
    """Synthetic state, generated on the device: returns
    (grad, committed stress, committed history dict, warm-up grad)."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)

    def grad_array():
        g = torch.randn(9 * n, dtype=torch.float64, device=device, generator=gen)
        if scale_spec == "loguniform":
            sc = torch.pow(10.0, torch.rand(n, dtype=torch.float64, device=device, generator=gen) * 2.0 - 4.0)
            g.view(n, 9).mul_(sc[:, None])
        elif scale_spec == "zoned":
            zone = 4096
            nz = (n + zone - 1) // zone
            pl = torch.rand(nz, dtype=torch.float64, device=device, generator=gen) < 0.22
            sc = torch.where(pl, 1e-2, 1e-4).to(torch.float64).repeat_interleave(zone)[:n]
            g.view(n, 9).mul_(sc[:, None])
        elif scale_spec in ("isochoric", "isochoric_zoned"):
            # Drucker-Prager: mostly isochoric increments, scale log-uniform in [1e-4, 5e-3] (keeps the
            # trial states away from the tip of the classic surface); zoned: 4096-point zones, 22 % of
            # them at 5e-3, the others at 1e-4
            if scale_spec == "isochoric":
                sc = torch.pow(10.0, torch.rand(n, dtype=torch.float64, device=device, generator=gen) * 1.7 - 4.0)
            else:
                zone = 4096
                pl = torch.rand((n + zone - 1) // zone, dtype=torch.float64, device=device, generator=gen) < 0.22
                sc = torch.where(pl, 5e-3, 1e-4).to(torch.float64).repeat_interleave(zone)[:n]
            gv = g.view(n, 9)
            gv.mul_(sc[:, None])
            tr = (gv[:, 0] + gv[:, 4] + gv[:, 8]) * (0.95 / 3.0)
            for c in (0, 4, 8):
                gv[:, c] -= tr
        else:
            g.mul_(float(scale_spec))
        return g

    stress = torch.zeros(6 * n, dtype=torch.float64, device=device)
    if kind == "von_mises_3d":
        hist = {"eps_n": torch.zeros(6 * n, dtype=torch.float64, device=device),
                "alpha": torch.rand(n, dtype=torch.float64, device=device, generator=gen) * 0.02}
    elif kind in ("spring_maxwell", "spring_kelvin"):
        hist = {"strain_visco": torch.zeros(6 * n, dtype=torch.float64, device=device),
                "strain": torch.zeros(6 * n, dtype=torch.float64, device=device)}
    elif kind == "comfe_drucker_prager":
        hist = {"history": torch.zeros(7 * n, dtype=torch.float64, device=device)}
        stress.view(n, 6)[:, :3] = -1000.0  # compressive prestress
    elif kind == "comfe_mises_plasticity":
        h = torch.zeros(7 * n, dtype=torch.float64, device=device)
        h.view(n, 7)[:, 0] = torch.rand(n, dtype=torch.float64, device=device, generator=gen) * 0.02
        hist = {"history": h}
    else:
        hist = None
        stress.normal_(generator=gen)  # cfg2: sigma_in ~ N(0,1) exercises the "+="
    return grad_array, stress, hist


def cpu_baseline(kind, params, grad, stress, hist, del_t, budget_s=12.0):
    """Time the C oracle ("port": serial per-point loop, 1 thread -- what the reference does per
    MPI rank) on a bounded sample of the same workload."""
    import numpy as np

    from oracle import c_oracle as CO

    ns = min(grad.numel() // 9, 2_000_000)
    g = grad[: 9 * ns].cpu().numpy()
    s0 = stress[: 6 * ns].cpu().numpy()
    dims = {"eps_n": 6, "alpha": 1, "strain_visco": 6, "strain": 6, "history": 7}
    h0 = None if hist is None else {k: v[: dims[k] * ns].cpu().numpy() for k, v in hist.items()}
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
        "sample": f"oracle/oracle.c serial loop, first {ns} points of the same workload x {reps} passes ({t_total:.1f} s)",
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
            extra["python_per_point_loop_port_Mpts_s"] = time_np(NO.von_mises_3d_loop, min(ns, 30_000))
            # BASELINE config 3 compares with the comfe-rs CPU path: our C restatement of the serial
            # evaluate_model loop around MisesPlasticity3D (interfaces.rs:354-456, mises_plasticity.rs:58-126;
            # mu, kappa, y_0 as above, h = 200 as in tests/models/test_plasticity.py:26-31) on the same
            # gradients and stresses, 1 thread
            hr = np.zeros(7 * ns)
            hr.reshape(-1, 7)[:, 0] = h0["alpha"]
            rs_p = {"mu": params["p_mu"], "kappa": params["p_ka"], "y_0": params["p_y0"], "h": 200.0}
            tt, rr = 0.0, 0
            while tt < 2.0 and rr < 100:
                s, hh = s0.copy(), {"history": hr.copy()}
                t0 = time.perf_counter()
                CO.MODELS["comfe_mises_plasticity"](rs_p, 0.0, del_t, g, s, tan, hh)
                tt += time.perf_counter() - t0
                rr += 1
            extra["comfe_rs_mises_c_port_1_thread_Mpts_s"] = round(ns * rr / tt / 1e6, 2)
        # the same C loop on all host cores (OpenMP over points), for scale only
        nthr = min(CO.max_threads(), os.cpu_count() or 1)
        CO.set_num_threads(nthr)
        one_pass()
        tt, rr = 0.0, 0
        while tt < 2.0 and rr < 200:
            tt += one_pass()
            rr += 1
        CO.set_num_threads(1)
        extra["c_port_all_cores_Mpts_s"] = round(ns * rr / tt / 1e6, 1)
        extra["c_port_all_cores_threads"] = nthr
        out["extra"] = extra
    except Exception as e:  # the extra figures are informational only
        out["extra"] = {"error": str(e)}
    return out


def time_allgather(args, dist, torch, device, rank, world, n, stress_t, tangent):
    """The optional exchange step of the single-assembler mode, timed separately on a
    bounded slice and never part of `value`: in-place all-gather of stress + tangent slices over RCCL,
    and (--gather-direct) the one-hop point-to-point variant."""
    from fenics_constitutive_amd.sharded import ShardPlan

    ng = min(args.gather_points, n)
    plan = ShardPlan.create(ng * world, world)
    per = plan.per_rank
    out_s = torch.empty(6 * per * world, dtype=torch.float64, device=device)
    out_t = torch.empty(36 * per * world, dtype=torch.float64, device=device)
    out_s[6 * per * rank : 6 * per * (rank + 1)].copy_(stress_t[: 6 * per])
    out_t[36 * per * rank : 36 * per * (rank + 1)].copy_(tangent[: 36 * per])

    def time_gather(fn):
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            dist.barrier()
            t_ = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t_
            best = dt_ if best is None else min(best, dt_)
        tt_ = torch.tensor([best], dtype=torch.float64, device=device)
        dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
        return float(tt_.item())

    def ring():
        dist.all_gather_into_tensor(out_s, out_s[6 * per * rank : 6 * per * (rank + 1)])
        dist.all_gather_into_tensor(out_t, out_t[36 * per * rank : 36 * per * (rank + 1)])

    def direct():  # one-hop point-to-point transfers, all peers at once (sharded.allgather_direct)
        for buf, dim in ((out_s, 6), (out_t, 36)):
            mine = buf[dim * per * rank : dim * per * (rank + 1)]
            ops = []
            for shift in range(1, world):
                dst, src = (rank + shift) % world, (rank - shift) % world
                ops.append(dist.P2POp(dist.isend, mine, dst))
                ops.append(dist.P2POp(dist.irecv, buf[dim * per * src : dim * per * (src + 1)], src))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()

    shard_bytes = 42 * 8 * per
    t_ring = time_gather(ring)
    gather = {"points_per_rank": per, "shard_GB": round(shard_bytes / 1e9, 3),
              "rccl_all_gather_ms": round(t_ring * 1e3, 3),
              "rccl_all_gather_recv_GBs_per_gpu": round(shard_bytes * (world - 1) / t_ring / 1e9, 1),
              "note": "in-place all_gather_into_tensor of stress+tangent slices, outside the timed steps"}
    if args.gather_direct:
        t_direct = time_gather(direct)
        gather["direct_p2p_ms"] = round(t_direct * 1e3, 3)
        gather["direct_p2p_recv_GBs_per_gpu"] = round(shard_bytes * (world - 1) / t_direct / 1e9, 1)
    del out_s, out_t
    return gather


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="von_mises_mixed", choices=sorted(WORKLOADS))
    ap.add_argument("--n", "--points", dest="n", type=int, default=100_000_000,
                    help="quadrature points per GPU (use --points under torch.distributed.run, whose parser claims --n)")
    ap.add_argument("--grid", type=int, default=0, help="override the launch grid (workgroups)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--history", choices=["sparse", "full"], default="sparse",
                    help="plasticity laws: how the trial history is written.  sparse (default) = the protocol of the "
                         "product's device-resident Newton loop (ResidentState, fcamd_evaluate_device_from_sparse): trial "
                         "== committed except at plastic / formerly plastic points, so elastic points cost no history "
                         "traffic; full = every launch rewrites the whole trial history (fcamd_evaluate_device_from)")
    ap.add_argument("--sparse-history", action="store_true", help="same as --history sparse (kept for old command lines)")
    ap.add_argument("--sparse-tangent", action="store_true",
                    help="with --history sparse: also the sparse-tangent protocol of ResidentState (FCAMD_EVAL_SPARSE_TANGENT: "
                         "rows of points that stay elastic are not rewritten).  Not the reference contract -- the reported "
                         "bytes stay the interface's 464/568 B/pt, so `frac` is an equivalent, not a traffic, figure")
    ap.add_argument("--placement-tries", type=int, default=6,
                    help="candidate allocations of the tangent array, timed with the real kernel before the run; the "
                         "fastest is kept (fenics_constitutive_amd.placement, DESIGN.md 6).  1 = take what the driver gives")
    ap.add_argument("--gather-direct", action="store_true",
                    help="N>1: also time the one-hop point-to-point gather (batched isend/irecv to all peers)")
    ap.add_argument("--gather-points", type=int, default=20_000_000,
                    help="points per rank of the separately timed stress/tangent all-gather (N>1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only to exercise the multi-rank control flow on a box "
                         "with fewer GPUs than ranks (ranks then share GPUs; the all-gather leg is skipped)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # convenience: started without the launcher -> start it as a child (nothing has touched the
            # GPU in this process yet) and pass its exit code on
            import subprocess

            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29517"),
                   os.path.abspath(__file__)] + [a if a != "--n" else "--points" for a in sys.argv[1:]]
            sys.exit(subprocess.run(cmd).returncode)
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    dev_index = local_rank % torch.cuda.device_count()  # identity on a node with one GPU per rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # launched by torch.distributed.run (RANK set): one rank per GPU over RCCL, also for world 1
    distributed = "RANK" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    kind, scale_spec, b_el, b_pl = WORKLOADS[args.workload]
    n = args.n
    del_t = 2.0
    law, params = make_law(kind)
    grad_array, stress_c, hist_c = synth_inputs(kind, scale_spec, n, seed=1234 + rank, device=device)

    # one in-place warm step from the initial state gives a committed state "from a previous step"
    tangent = torch.empty(36 * n, dtype=torch.float64, device=device)
    g_warm = grad_array()
    law.evaluate(0.0, del_t, g_warm, stress_c, tangent, hist_c)
    del g_warm
    # Two Newton iterates of one increment, evaluated alternately: between the iterations of the
    # reference's Newton loop only grad_del_u changes (solver/_solver.py:130-147), and with it the
    # plastic set at its margin -- so the sparse protocol sees new and stale points as it does in use.
    grads = [grad_array()]
    grads.append(grads[0] if os.environ.get("BENCH_SINGLE_ITERATE") == "1" else grads[0] * 1.03)  # knob: A/B only
    grad = grads[0]
    # trial-state arrays: every timed step reads the committed state and writes the trial state
    # (same traffic as in place, stationary workload)
    stress_t = torch.empty_like(stress_c)
    hist_t = None if hist_c is None else {k: torch.empty_like(v) for k, v in hist_c.items()}
    if args.grid:
        law._handle(dev_index).ctx.set_grid(args.grid)

    plasticity = kind in ("von_mises_3d", "comfe_mises_plasticity", "comfe_drucker_prager")
    sparse = plasticity and (args.history == "sparse" or args.sparse_history)
    hmask = None
    if sparse:
        for k in hist_c:
            hist_t[k].copy_(hist_c[k])  # contract: trial == committed where the mask is clear
        hmask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=device)

    # placement of the tangent (the dominant write stream): a few candidate allocations, the real
    # kernel timed on each, the fastest kept -- what a long-running simulation does once at start-up
    placement = None
    if args.placement_tries > 1:
        from fenics_constitutive_amd.placement import fastest_allocation

        tangent, placement = fastest_allocation(
            36 * n, lambda tan: law.evaluate_from(0.0, del_t, grads[0], stress_c, stress_t, tan, hist_c, hist_t,
                                                  history_mask=hmask),
            tries=args.placement_tries, device=device, first=tangent)

    sparse_tangent = bool(args.sparse_tangent and sparse)

    def step(i):
        law.evaluate_from(0.0, del_t, grads[i & 1], stress_c, stress_t, tangent, hist_c, hist_t, history_mask=hmask,
                          sparse_tangent=sparse_tangent)

    for i in range(args.warmup):
        step(i)
    # plastic counts of the two iterates (two more untimed launches)
    n_pl_ab, its_ab = [], []
    for i in (0, 1):
        step(i)
        torch.cuda.synchronize()
        st = law.device_stats(dev_index)
        n_pl_ab.append(int(st.n_plastic) if plasticity else 0)
        its_ab.append(int(st.n_newton_iters))
    # averaged over the timed steps (step i evaluates iterate i & 1)
    n_b = args.steps // 2
    n_a = args.steps - n_b
    n_pl = (n_a * n_pl_ab[0] + n_b * n_pl_ab[1]) / args.steps
    n_its = (n_a * its_ab[0] + n_b * its_ab[1]) / args.steps

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record()
        step(i)
        ev1[i].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = sorted(ev0[i].elapsed_time(ev1[i]) for i in range(args.steps))
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)
    per_rank_ms = None
    if distributed:  # every rank's own average kernel time: shows the balance behind the max-over-ranks wall time
        tk = torch.zeros(world, dtype=torch.float64, device=device)
        tk[rank] = kernel_avg_ms
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(x), 4) for x in tk.tolist()]

    # next to the headline: the same step without the sparse protocol (every launch rewrites the whole trial
    # history, fcamd_evaluate_device_from) -- five extra launches after the timed region
    full_ms = None
    if sparse:
        evf = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
        for i, (e0_, e1_) in enumerate(evf):
            e0_.record()
            law.evaluate_from(0.0, del_t, grads[i & 1], stress_c, stress_t, tangent, hist_c, hist_t)
            e1_.record()
        torch.cuda.synchronize()
        full_ms = sum(a_.elapsed_time(b_) for a_, b_ in evf[1:]) / (len(evf) - 1)

    # optional exchange step, timed separately (never part of `value`)
    gather = None
    if distributed and args.backend == "nccl":
        try:
            gather = time_allgather(args, dist, torch, device, rank, world, n, stress_t, tangent)
        except Exception as e:  # e.g. out of memory on every rank alike: the step timing above stands
            gather = {"error": f"{type(e).__name__}: {e}"[:300]}

    # "achievable" next to "peak" (SURVEY 8d): a plain device copy over half of the tangent array
    # (read + write counted), after everything that still needs the arrays
    copy_gbs = None
    try:
        half = (tangent.numel() // 2) & ~1
        cs, ce = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(4):
            cs.record()
            tangent[:half].copy_(tangent[half : 2 * half])
            ce.record()
            ce.synchronize()
            ms = cs.elapsed_time(ce)
            best = ms if best is None else min(best, ms)
        copy_gbs = 2 * 8 * half / (best * 1e-3) / 1e9
    except Exception:  # the probe is informational only
        copy_gbs = None

    if rank == 0:
        total_pts = n * world * args.steps
        value = total_pts / elapsed / 1e6
        alg_bytes = int(round((n - n_pl) * b_el + n_pl * b_pl))
        achieved = alg_bytes / (kernel_avg_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                with open(tf) as f:
                    tj = json.load(f)
                e = tj.get(args.workload + ("_full" if plasticity and not sparse else ""))
                if e and int(e.get("n", 0)) == n:
                    traffic = e.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": METRIC,
            "value": round(value, 1),
            "unit": "Mpts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {kind} FULL-3D, {n} quadrature points per GPU, device-resident AoS, "
                                   f"committed->trial evaluate of two alternating Newton iterates"
                                   f"{', sparse trial history (ResidentState protocol)' if sparse else (', full trial history' if plasticity else '')}"
                                   f"{', sparse tangent (rows of points that stay elastic are not rewritten)' if sparse_tangent else ''}",
                       "points_per_gpu": n, "plastic_fraction": round(n_pl / n, 4),
                       "mean_newton_iters": round(n_its / max(n_pl, 1), 3) if kind in ("von_mises_3d", "comfe_drucker_prager") else None,
                       "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms_avg": round(kernel_avg_ms, 4), "kernel_ms_min": round(kernel_ms[0], 4),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "traffic_GBs": None if traffic is None else round(traffic / (kernel_avg_ms * 1e-3) / 1e9, 1),
                         "device_copy_GBs": None if copy_gbs is None else round(copy_gbs, 1),
                         "bytes_per_point": {"elastic": b_el, "plastic": b_pl}},
        }
        if full_ms is not None:
            out["full_trial_history"] = {"kernel_ms_avg": round(full_ms, 4),
                                         "frac": round(alg_bytes / (full_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "note": "same step, whole trial history rewritten by every launch (--history full)"}
        if placement is not None:
            out["placement"] = {"tangent_" + k: v for k, v in placement.items()}
        if per_rank_ms is not None:
            out["per_rank_kernel_ms"] = per_rank_ms
        if gather:
            out["allgather"] = gather
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(kind, params, grad, stress_c, hist_c, del_t)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
