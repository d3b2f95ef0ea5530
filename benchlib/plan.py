"""bench.py --plan: the legs of a run, their memory per GPU and their time -- arithmetic only (no torch, no GPU).

The first real 8-GPU launch happens on the driver's node, once, without a rehearsal at that width (no multi-GPU node was ever
available to a builder round).  The plan states BEFORE that launch what every leg allocates and how long it should take, from
this round's single-GPU measurements and MI355X_MICROARCH.md's link rates, and tests/test_bench_plan.py holds the totals to the
limits (288 GB per GPU; the driver's 600 s per run; bench.py's own --wall-budget, after which optional legs are skipped)."""

from __future__ import annotations

HBM_GB = 288.0
DRIVER_LIMIT_S = 600.0
XGMI_LINK_GBS = 153.0  # per direction and link, 7 links per GPU (MI355X_MICROARCH.md)
PCIE_GBS = 56.0  # measured pinned copy rate per GPU
KERNEL_GBS = 6300.0  # what the evaluate kernels move in real bytes (0.79 of the 8 TB/s peak)
ALLOC_S_PER_GB = 0.008  # hipMalloc + first touch of large arrays as measured in the default run's legs (detail file: legs_s)
IMPORT_S = 4.0  # import torch + library load on a warm box (a cold image: up to 120 s, outside anything bench.py controls)

# bytes per point of the device-resident headline step (workloads.py): 2 gradients, committed + trial stress, tangent, committed + trial
# eps_n / alpha, protocol words
STATE_B = 8 * (2 * 9 + 2 * 6 + 36 + 2 * 6 + 2 * 1) + 1
ALG_B = 0.2243 * 568 + 0.7757 * 464  # algorithmic bytes per point of the 22 % mixture


def _leg(name, seconds, gb, note):
    return {"leg": name, "est_s": round(seconds, 1), "peak_GB_per_gpu": round(gb, 1), "note": note}


def bench_plan(args):
    world, n = int(args.gpus), int(args.n)
    steps, warmup, tries = int(args.steps), int(args.warmup), int(args.placement_tries)
    step_s = n * ALG_B / (KERNEL_GBS * 1e9)
    state_gb = STATE_B * n / 1e9
    tan_gb = 288 * n / 1e9
    legs = []
    # placement candidates are alive together while they are timed (max_tries_for_memory caps them for N > 1 with a 16 GiB reserve)
    if world > 1:
        tries = max(1, min(tries, int((HBM_GB - 17.2 - state_gb) // tan_gb) + 1))
    place_gb = state_gb + (tries - 1) * tan_gb
    vmm_gb = 2 * state_gb  # the VMM working set is built next to the hipMalloc arrays, the loser is released
    legs.append(_leg("setup", IMPORT_S + ALLOC_S_PER_GB * state_gb + 3.0, state_gb,
                     "import, process group (RCCL), synthetic state generated on the device, one warm in-place increment"))
    legs.append(_leg("placement", ALLOC_S_PER_GB * ((tries - 1) * tan_gb + state_gb) + (4 * tries + 8) * step_s + 2.0, max(place_gb, vmm_gb),
                     f"{tries} hipMalloc candidates of the tangent (4 launches each) + one interleaved VMM working set"))
    legs.append(_leg("timed_steps", (warmup + 2 + steps) * step_s + 0.5, state_gb, f"{warmup} warm-up + 2 counting + {steps} timed launches, barrier + synchronize brackets"))
    total_pts = n * world
    if world > 1:
        m = max(64, (n // world // 64) * 64)
        legs.append(_leg("strong_scaling_leg", (max(2, warmup) + steps + 2) * (m * ALG_B / (KERNEL_GBS * 1e9)) + 0.5, state_gb,
                         f"the first {m} points of every shard ({m * world} in total): same bracket"))
    legs.append(_leg("reference_layout_legs", 3.0 + 19 * step_s + ALLOC_S_PER_GB * 0.2 * state_gb, state_gb * 1.2,
                     "full trial history, sparse protocol on the reference layout" + ("" if world > 1 else ", in-place call")))
    if world > 1 and not args.no_gather:
        ng = ((min(args.gather_points, n) if args.gather_points > 0 else n) // 64) * 64
        shard_gb = 336 * ng / 1e9
        stress_all_gb = 48 * ng * world / 1e9
        free_gb = HBM_GB - 8.0 - (state_gb - 8 * (9 * 2 + 6 + 14) * n / 1e9) - stress_all_gb  # grads / trial history are released before the gather
        buf_gb = max(0.0, min(free_gb, 2 * 288 * ng * world / 1e9))
        chunks = max(1, -(-int(288 * ng * world / 1e9 * 2) // max(1, int(buf_gb))))
        direct_s = shard_gb / XGMI_LINK_GBS  # world-1 concurrent peer copies, one per link: every link carries one shard
        ring_s = shard_gb * (world - 1) / (XGMI_LINK_GBS * 2.0)  # RCCL ring over bidirectional links: (world-1) steps of one shard
        variants = ["rccl", "direct"] + (["p2p"] if args.gather_direct else [])
        per_variant = {"rccl": ring_s, "direct": direct_s, "p2p": ring_s}
        t = sum(2 * 1.6 * per_variant[v] + 6.0 for v in variants)  # 2 repetitions, 60 % margin on the link rate, IPC / buffer set-up
        legs.append(_leg("allgather", t, HBM_GB - 8.0,
                         f"shard {shard_gb:.1f} GB/GPU, stress gathered whole ({stress_all_gb:.1f} GB), tangent in ~{chunks} chunks through two buffers "
                         f"({buf_gb:.0f} GB); link-rate bounds: direct {direct_s:.2f} s, ring {ring_s:.2f} s; watchdog --gather-timeout {args.gather_timeout:.0f} s"))
    if world == 1 and args.workload is None and args.configs != "none":
        cfg_tries = 4 if args.full else 3
        legs.append(_leg("configs", 5 * (ALLOC_S_PER_GB * (state_gb + (cfg_tries - 1) * tan_gb) + 3.0 + (4 * cfg_tries + 12) * step_s * 1.2), state_gb + (cfg_tries - 1) * tan_gb,
                         f"five other BASELINE configurations, {cfg_tries} tangent candidates each"))
        if not args.no_frows:
            draws = 3 if args.full else 2
            legs.append(_leg("frows", 10 * draws * (ALLOC_S_PER_GB * 0.7 * state_gb + 1.2), draws * 0.8 * state_gb, f"ten SURVEY 8(f) rows, up to {draws} sets of allocations each"))
    if not args.no_host_path:
        if world > 1:
            per_dev = min(n, 2_500_000)
            pts = per_dev * world
            legs.append(_leg("host_path_multi", min(40.0, 2.0 + pts * 568 / 2.0e9 + 3 * 2 * pts * 568 / (PCIE_GBS * 1e9 * world)) + 8.0, 2.0,
                             f"rank 0 drives all {world} GPUs over their PCIe links on {pts} points of host arrays (NumPy generation dominates); the other ranks wait on the store"))
        elif args.workload is None:
            big = min(n, 10_000_000 if args.full else 4_000_000)
            legs.append(_leg("host_path", 2.0 + big * 568 / 2.0e9 + 18 * big * 400 / (PCIE_GBS * 1e9), 1.0, f"ndarray entries over PCIe at 1e6 and {big} points"))
    if world == 1:
        if not args.no_live_traffic:
            extra_rows = args.workload is None and args.configs != "none"
            items = 3 + (5 if extra_rows else 0) + (10 if (extra_rows and not args.no_frows) else 0)
            legs.append(_leg("live_traffic", 2 * (IMPORT_S + 4.0 + items * (ALLOC_S_PER_GB * 0.8 * state_gb + 1.5)), state_gb,
                             f"two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE), {items} items each; the CPU baseline (~20 s, alone on the host) runs before them"))
        if not args.no_cpu_baseline:
            legs.append(_leg("cpu_baseline", 0.0 if not args.no_live_traffic else 14.0, 0.0, "C port, 1 thread, 2e6 points x 8 s + side figures (overlapped with the PMC passes)"))
    # what bench.py checks before it enters an optional leg: seconds that must be left of --wall-budget (all ranks agree by all-reduce)
    needs = {"strong_scaling_leg": 60, "allgather": 90, "configs": 45, "frows": 60, "host_path_multi": 75, "host_path": 40, "live_traffic": 100}
    spent = 0.0
    for x in legs:
        x["starts_at_s"] = round(spent, 1)
        if x["leg"] in needs:
            x["skipped_if_budget_left_below_s"] = needs[x["leg"]]
            x["runs"] = bool(args.wall_budget - spent > needs[x["leg"]])
        spent += x["est_s"]
    total = sum(x["est_s"] for x in legs)
    peak = max(x["peak_GB_per_gpu"] for x in legs)
    return {"plan": True, "n_gpus": world, "points_per_gpu": n, "points_total": total_pts, "est_total_s": round(total, 1),
            "wall_budget_s": args.wall_budget, "driver_limit_s": DRIVER_LIMIT_S, "fits_driver_limit": total < 0.8 * DRIVER_LIMIT_S,
            "peak_GB_per_gpu": peak, "fits_memory": peak <= HBM_GB, "est_value_Mpts_s": round(total_pts / step_s / 1e6, 0),
            "legs": legs,
            "note": "estimates from single-GPU measurements of this round and the guide's link rates; legs after `timed_steps` are optional: each is skipped "
                    "when --wall-budget runs low (all ranks agree by all-reduce), and the line printed before it survives a leg that dies"}
