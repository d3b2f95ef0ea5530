#!/usr/bin/env python3
"""Benchmark of the hot path: quadrature-point stress updates per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--n POINTS]

One "step" = one ``evaluate`` pass of one constitutive law over n synthetic quadrature points
per GPU, device-resident (inputs already in HBM when the timed region starts), committed state in,
trial state out -- the call of the product's device-resident Newton loop (ResidentState.evaluate;
for the plasticity laws with the sparse trial-history protocol, --history full for the mask-less
form).  The steps alternate between two Newton iterates of the increment.  Headline workload:
BASELINE.json's target configuration, VonMises3D return mapping, 1e8 points, mixed elastic/plastic
(SURVEY.md 8d cfg3 "mixed").  Before the warm-up the placement of the tangent array is chosen out of a
few candidate allocations -- the line carries the chosen, the FIRST (what an untuned
caller gets) and the median candidate.

With no --workload (the driver's command) and N = 1 the same run then times every other single-GPU
configuration of BASELINE.json / SURVEY.md 8d with the same method -- LinearElasticity (cfg2),
VonMises3D all-plastic and all-elastic (cfg3 sub-cases), Maxwell (cfg4), Kelvin -- and reports them
under "configs".  For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank evaluates
its own contiguous shard of n points (weak scaling, no data-path collective: SURVEY.md 8e); the
stress/tangent all-gather of the single-assembler mode (config 5) is timed separately on the whole
shard -- stress in one piece, tangent in chunks that fit next to the working set -- and reported under
"allgather", never inside `value`.

Rank 0 prints ONE compact JSON line (< 4 KB, numbers only: benchlib/line.py; see DESIGN.md "Measurement") and writes the full
record -- workload sentences, launch logs, placement candidates, host-path tables, notes -- to the detail file the line names
(--detail, default gpurun_out/bench_detail.json).  `roofline.traffic` (HBM bytes per launch by the PMC counters) is measured at the
end of an N = 1 run by two child passes of this file under `rocprofv3 --pmc` (one pass per counter; every measured workload runs in
the same child: --pmc-child) while the CPU baseline runs on the host (--no-live-traffic: the stored figure of profiles/traffic.json
instead).  The default command is the lean one (about two minutes); --full adds the per-configuration CPU figures, the small-call
crossover table, more allocation draws per row and PMC passes for every configuration.  `--plan` prints what a run with these
arguments WOULD do -- per-leg memory and time estimates, no GPU work.
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

from benchlib.cpu import cpu_baseline, cpu_quick  # noqa: E402,F401
from benchlib.gather import time_allgather  # noqa: E402
from benchlib.hostpath import host_path_figures, main_host  # noqa: E402,F401
from benchlib.line import compact_line, dumps as dump_line, write_detail  # noqa: E402
from benchlib.traffic import library_hash, live_traffic_batch, read_traffic, read_traffic_split, under_profiler  # noqa: E402,F401
from benchlib.workloads import (BASELINE_CONFIG, EXTRA_CONFIGS, HBM_PEAK_GBS, HEADLINE, METRIC, WORKLOADS, Workload, placement_fracs,  # noqa: E402,F401
                                traffic_key)


class LineGuard:
    """The ONE JSON line survives the death of the process that measured it.

    The timed steps are over long before the optional legs (all-gather variants, extra configurations, host path, CPU
    baselines) are; a leg that takes the process down on hardware nobody could rehearse (a fault in a peer mapping, the
    launcher terminating rank 0 because another rank died) must not cost the measured line.  A small child process
    (started BEFORE anything touches the GPU, no GPU use of its own, a session of its own) reads lines from a pipe and
    prints the LAST one it received when the pipe closes: rank 0 sends the line as it stands before every optional leg
    (marked "incomplete": the leg it was about to enter) and the complete line at the end.  Exactly one line reaches
    stdout either way; the exit code of the job still says that a leg failed."""

    CHILD = ("import sys\nlast = None\nfor line in sys.stdin:\n    if line.strip():\n        last = line\n"
             "if last is not None:\n    sys.stdout.write(last if last.endswith('\\n') else last + '\\n')\n    sys.stdout.flush()\n")

    def __init__(self):
        import subprocess

        self.proc = None
        # a plain interpreter: under rocprofv3 the parent's environment preloads the profiler, which would initialise the GPU
        # (and claim counters) in the guard as well
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
        try:
            self.proc = subprocess.Popen([sys.executable, "-c", self.CHILD], stdin=subprocess.PIPE, text=True, start_new_session=True, env=env)
        except OSError:
            pass  # no guard: the line is printed directly at the end

    def _send(self, line):
        self.proc.stdin.write(line + "\n")
        self.proc.stdin.flush()

    def provisional(self, out, leg, detail_path=None):
        """the line as it stands (compact, like the final one), marked incomplete; the full record goes to the detail file"""
        if self.proc is None:
            return
        rec = dict(out, incomplete=f"the process ended in the optional leg '{leg}'; everything in this line was measured before it")
        written = write_detail(rec, detail_path) if detail_path else None
        try:
            self._send(dump_line(compact_line(rec, written)))
        except (OSError, ValueError):
            self.proc = None

    def final(self, line):
        """the complete line: through the guard when there is one -- and then NEVER printed directly as well (the guard prints
        the last line it received when the pipe closes, whatever happens to this process afterwards: two lines would break
        the one-line contract); directly only when no guard exists or the send itself failed"""
        if self.proc is not None:
            sent = False
            try:
                self._send(line)
                sent = True
                self.proc.stdin.close()
            except (OSError, ValueError):
                pass
            if sent:
                try:
                    self.proc.wait(timeout=30)
                except Exception:  # noqa: BLE001 -- a slow guard still prints the line on its own
                    pass
                return
            try:  # the guard never received the final line: silence it (it would print a stale provisional one), print here
                self.proc.kill()
            except Exception:  # noqa: BLE001
                pass
        print(line, flush=True)


def run_config(name, n, seed, device, dev_index, steps, warmup, tries, history="packed", placement="auto", cpu=True):
    """One extra configuration, same method as the headline: placement, warm up, count, >= 5 event-timed launches."""
    wl = Workload(name, n, seed, device, dev_index, history=history)
    try:
        wl.place(placement, tries)
        # an unlucky hand of allocations (every candidate a slow one: LinearElasticity 0.735 with three draws in one round-5 run, 0.78-0.82
        # otherwise) is recognisable from the byte rate alone: draw again, once, with the best so far as candidate 0
        first_round = wl.placement
        if placement == "tune" and first_round and tries > 1:
            bytes_est = n * (wl.b_pl if name == "von_mises_plastic" else wl.b_el)
            if bytes_est / (min(first_round["candidate_ms"]) * 1e-3) / 1e9 / HBM_PEAK_GBS < float(os.environ.get("BENCH_REDRAW_BELOW", "0.77")):  # (knob: tests force the second round)
                wl.tune_placement(tries + 1)
                wl.placement = {"candidate_ms": first_round["candidate_ms"] + wl.placement["candidate_ms"][1:],
                                "chosen": None, "second_round": True}
        wl.warmup(warmup)
        wl.count_plastic()
        ms = wl.timed_events(steps)
        n_pl, n_its = wl.mean_plastic(steps)
        alg = wl.alg_bytes(n_pl)
        avg = sum(ms) / len(ms)
        out = {"baseline_config": BASELINE_CONFIG.get(name), "workload": wl.config_text(), "launches": steps,
               "kernel_ms_avg": round(avg, 4), "kernel_ms_min": round(min(ms), 4),
               "Mpts_s": round(n / (avg * 1e-3) / 1e6, 1), "algorithmic_bytes_per_launch": alg,
               "achieved_GBs": round(alg / (avg * 1e-3) / 1e9, 1), "frac": round(alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "plastic_fraction": round(n_pl / n, 4),
               "mean_newton_iters": round(n_its / max(n_pl, 1), 3) if wl.kind in ("von_mises_3d", "comfe_drucker_prager") else None,
               "bytes_per_point": {"elastic": wl.b_el, "plastic": wl.b_pl}}
        out.update(placement_fracs(wl, wl.alg_bytes(wl.n_pl_ab[0])))
        if wl.placement:
            out["placement_candidate_ms"] = wl.placement["candidate_ms"]
        # PMC-measured HBM bytes of this configuration's launch (profiles/traffic.json: its own rocprofv3 passes, same kernels)
        key = traffic_key(wl)
        out["traffic"] = read_traffic(key, n)
        out["traffic_source"] = None if not out["traffic"] else f"profiles/traffic.json[{key}] (stored rocprofv3 --pmc measurement of these kernels)"
        out["traffic_over_algorithmic"] = None if not out["traffic"] else round(out["traffic"] / alg, 4)
        out["placement_mode"] = (wl.vmm_info or {}).get("mode", "hipmalloc_tuned" if wl.placement else "first")
        if wl.vmm_info and "vmm_ms" in wl.vmm_info:
            out["placement_vmm_ms"] = wl.vmm_info["vmm_ms"]
        out["launch_log"] = wl.launch_log
        if cpu:
            try:
                out["cpu"] = cpu_quick(wl)
            except Exception as e:  # informational
                out["cpu"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        return out
    finally:
        wl.free()


def all_agree(flag, dist, device):
    """the same yes / no on every rank (a leg with collectives must be entered by all ranks or by none): the minimum over the ranks"""
    import torch

    f = torch.tensor([1 if flag else 0], dtype=torch.int64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return bool(int(f.item()))


def main_frow(args):
    """`--frow NAME`: one SURVEY 8(f) row alone; prints one JSON line (with the launch_log the PMC slicing needs)"""
    import torch

    from benchlib import frows as bench_frows

    if args.frow not in bench_frows.FROWS:
        sys.exit(f"--frow: one of {sorted(bench_frows.FROWS)}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(device)
    out = bench_frows.run_frow(args.frow, args.n, device, launches=max(1, args.steps), warm=max(1, args.warmup), peak_gbs=HBM_PEAK_GBS, draws=1,
                               floor_launches=4)
    print(json.dumps({"metric": METRIC, "frow": args.frow, **out}), flush=True)
    return 0


def main_pmc_child(args):
    """`--pmc-child A,B,...`: the child of one `rocprofv3 --pmc` pass (benchlib.traffic.live_traffic_batch).  Every named workload
    (a WORKLOADS name, `+unpacked` / `+in_place` / `+full` for the reference-layout forms of the same step, or a SURVEY 8(f) row
    of benchlib.frows) is built in turn, on the allocator's arrays as they come (the bytes of a launch do not depend on where its
    arrays lie), warmed and launched 4 times; the ONE line printed is the ordered log [item, phase, evaluate launches] the parent
    slices the counter CSV with."""
    import torch

    from benchlib import frows as bench_frows

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(device)
    os.environ["FROW_PLACEMENT"] = "torch"
    history = "sparse" if args.sparse_history else args.history
    log = []
    t_child = time.perf_counter()
    for k, item in enumerate([x for x in args.pmc_child.split(",") if x]):
        name, _, form = item.partition("+")
        if args.pmc_budget > 0 and time.perf_counter() - t_child > args.pmc_budget:
            log.append([item, "error", "time budget of this pass"])  # the parent keeps what came before
            break
        try:
            if name in bench_frows.FROWS:
                out = bench_frows.run_frow(name, args.n, device, launches=4, warm=2, peak_gbs=HBM_PEAK_GBS, draws=1)
                log += [[item, ph, c] for ph, c in out["launch_log"]]
            else:
                wl = Workload(name, args.n, seed=1234 if name == (args.workload or HEADLINE) else 4321 + k, device=device,
                              dev_index=device.index or 0, history=history, sparse_tangent=args.sparse_tangent,
                              split_history=not args.no_split_history)
                try:
                    if form == "in_place":
                        wl.timed_in_place(4, phase="timed")
                    elif form in ("unpacked", "full"):
                        kw = {"unpacked": True} if form == "unpacked" else {"full_history": True}
                        wl.launch(0, sparse_tangent=False, **kw), wl.launch(1, sparse_tangent=False, **kw)
                        wl.launch_log.append(["warmup", 2])
                        wl.timed_events(4, phase="timed", sparse_tangent=False, **kw)
                    else:
                        wl.warmup(2)
                        wl.timed_events(4)
                    log += [[item, ph, c] for ph, c in wl.launch_log]
                finally:
                    wl.free()
        except Exception as e:  # the parent drops this item and everything after it (the dispatch count is unknown from here on)
            log.append([item, "error", f"{type(e).__name__}: {e}"[:200]])
            break
        torch.cuda.empty_cache()
    print(json.dumps({"metric": METRIC, "pmc_child": True, "log": log}), flush=True)
    return 0


def plan(args):
    """`--plan`: what a run with these arguments would do -- legs, memory per GPU, time estimates against the driver's limits --
    from arithmetic alone (no torch, no GPU): lets the first real 8-GPU launch be a one-shot (tests/test_bench_plan.py holds the
    totals to the limits).  Rates: this round's single-GPU measurements and MI355X_MICROARCH.md (xGMI 153 GB/s per link, 7 links,
    PCIe 56 GB/s per GPU)."""
    from benchlib.plan import bench_plan

    print(json.dumps(bench_plan(args), separators=(",", ":")), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help=f"time this workload only (default: {HEADLINE}, followed at N = 1 by the other BASELINE configurations)")
    ap.add_argument("--frow", default=None,
                    help="run ONE row of SURVEY 8(f) alone (benchlib.frows.FROWS: indexed evaluate, fused wrapper, low-dimensional kernels, "
                         "resident sparse-tangent iteration) and print its figures -- the child of the default run's PMC passes")
    ap.add_argument("--pmc-child", default=None, help="(internal) the child of a rocprofv3 --pmc pass: comma-separated workloads / 8(f) rows, see main_pmc_child")
    ap.add_argument("--pmc-budget", type=float, default=0.0, help="(internal) --pmc-child: seconds after which no further item is started")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the full record behind the compact stdout line is written (the line carries the path)")
    ap.add_argument("--full", action="store_true",
                    help="everything round 4's default ran: per-configuration CPU figures, small-call crossover table, three allocation draws per 8(f) row, "
                         "four placement candidates per configuration, PMC passes for the configurations too (about four minutes)")
    ap.add_argument("--plan", action="store_true", help="print the legs, memory and time estimates of a run with these arguments; no GPU work")
    ap.add_argument("--no-frows", action="store_true", help="N = 1 default run: skip the SURVEY 8(f) rows after the BASELINE configurations")
    ap.add_argument("--configs", choices=["auto", "all", "none"], default="auto",
                    help="the other single-GPU BASELINE configurations after the headline: auto = when no --workload is given and N = 1")
    ap.add_argument("--config-steps", type=int, default=6, help="timed launches per extra configuration (>= 5)")
    ap.add_argument("--n", "--points", dest="n", type=int, default=None,
                    help="quadrature points per GPU (use --points under torch.distributed.run, whose parser claims --n); default 1e8, "
                         "--mode host: 1e7 (the arrays are host memory: 568 B per point)")
    ap.add_argument("--grid", type=int, default=0, help="override the launch grid (workgroups)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--history", choices=["packed", "sparse", "full"], default="packed",
                    help="plasticity laws: how the trial history is written.  packed (default) = the protocol of the "
                         "product's device-resident Newton loop (ResidentState): trial == committed except at plastic / "
                         "formerly plastic points, so elastic points cost no history traffic, and both copies of the "
                         "plastic-strain array keep the rows of the ever-plastic points packed per tile (FCAMD_EVAL_PACKED_HISTORY; "
                         "the commit stays a pointer swap); sparse = the same protocol on the reference's array layout "
                         "(ResidentState(packed_history=False)); full = every launch rewrites the whole trial history")
    ap.add_argument("--sparse-history", action="store_true", help="same as --history sparse (kept for old command lines)")
    ap.add_argument("--no-split-history", action="store_true",
                    help="comfe-rs plasticity workloads: keep the reference's 7-double history rows in the state instead of "
                         "ResidentState's [scalar, eps_p rows] layout (FCAMD_EVAL_SPLIT_HISTORY)")
    ap.add_argument("--sparse-tangent", action="store_true",
                    help="with --history sparse: also the sparse-tangent protocol of ResidentState (FCAMD_EVAL_SPARSE_TANGENT: "
                         "rows of points that stay elastic are not rewritten).  Not the reference contract -- the reported "
                         "bytes stay the interface's 464/568 B/pt, so `frac` is an equivalent, not a traffic, figure")
    ap.add_argument("--placement", choices=["auto", "vmm", "tune", "first"], default="auto",
                    help="where the arrays of the timed steps live (DESIGN.md 6): vmm = one working set with 2 MiB physical "
                         "handles interleaved over the arrays; tune = the fastest of --placement-tries hipMalloc candidates of "
                         "the tangent; auto (default, = ResidentState's default) = the faster of the two; first = what the "
                         "allocator gives")
    ap.add_argument("--placement-tries", type=int, default=6,
                    help="hipMalloc candidate allocations of the tangent array, timed with the real kernel before the run "
                         "(fenics_constitutive_amd.placement): in tune mode the fastest is kept, in vmm mode they are "
                         "timed for the record only (frac_first_allocation / frac_median_candidate)")
    ap.add_argument("--gather-direct", action="store_true",
                    help="N>1: also time the batched isend/irecv gather (RCCL point-to-point) next to RCCL's all-gather "
                         "and the C ABI's peer copies")
    ap.add_argument("--gather-points", type=int, default=0,
                    help="points per rank of the separately timed stress/tangent all-gather (N>1); 0 = the whole shard")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather leg")
    ap.add_argument("--gather-timeout", type=float, default=240.0,
                    help="N>1: seconds after which an unfinished all-gather leg is given up (the line is printed without it)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only to exercise the multi-rank control flow on a box "
                         "with fewer GPUs than ranks (ranks then share GPUs; of the gather variants only the C ABI's "
                         "peer copies run)")
    ap.add_argument("--verbose", action="store_true", help="stage-by-stage progress lines on stderr (every rank)")
    ap.add_argument("--mode", choices=["device", "host"], default="device",
                    help="device (default): device-resident steps, one rank per GPU; host: the single-process multi-GPU HOST path "
                         "(fcamd_multi: one process, --gpus device contexts, every device evaluates its slice of the process's NumPy "
                         "arrays in place over its own PCIe link) -- PCIe-inclusive, its own line")
    ap.add_argument("--host-devices", default="", help="--mode host: explicit device ordinals, e.g. 0,0 (two contexts on one GPU)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = --points per GPU (default; a strong-scaling leg on the first points/N of every shard is reported "
                         "next to it under \"strong_scaling\"); strong = --points in total, cut into N contiguous shards")
    ap.add_argument("--no-host-path", action="store_true", help="N = 1 default run: skip the host_path block (PCIe-inclusive figures of the ndarray entries)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not measure roofline.traffic in this run (two child passes under rocprofv3 --pmc, about 40 s); the "
                         "stored figure of profiles/traffic.json is reported instead (labelled as stored)")
    ap.add_argument("--wall-budget", type=float, default=300.0,
                    help="seconds of wall clock after which the optional legs (strong-scaling leg, gather variants, extra configurations, host_path) "
                         "are skipped so that the line is printed inside the driver's limit")
    args = ap.parse_args()
    if args.n is None:
        args.n = 10_000_000 if args.mode == "host" else 100_000_000
    if args.mode == "host":
        sys.exit(main_host(args))
    if args.plan:
        sys.exit(plan(args))
    if args.pmc_child is not None:
        sys.exit(main_pmc_child(args))
    if args.frow is not None:
        sys.exit(main_frow(args))

    def stage(msg):
        if args.verbose:
            print(f"# [rank {os.environ.get('RANK', '0')} +{time.perf_counter() - t_prog:.1f}s] {msg}", file=sys.stderr, flush=True)

    t_prog = time.perf_counter()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL, fcamd_ipc_*)
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
    guard = LineGuard() if rank == 0 else None  # before the first call that initialises the GPU
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

    stage("process group up")
    t_start = time.perf_counter()
    name = args.workload or HEADLINE
    history = "sparse" if args.sparse_history else args.history
    n = args.n
    if args.scaling == "strong" and world > 1:
        # --points in total: rank r owns the contiguous, tile-aligned shard fcamd_shard_bounds(points, world, r)
        from fenics_constitutive_amd import _capi

        lo, hi = _capi.shard_bounds(args.n, world, rank)
        n = hi - lo
        if n <= 0:
            sys.exit(f"--scaling strong: rank {rank} owns no point of {args.n}")

    def budget_left():
        return args.wall_budget - (time.perf_counter() - t_start)

    def agree(flag):
        if not (distributed and world > 1):
            return bool(flag)
        return all_agree(flag, dist, device if args.backend == "nccl" else "cpu")
    wl = Workload(name, n, seed=1234 + rank, device=device, dev_index=dev_index, history=history,
                  sparse_tangent=args.sparse_tangent, grid=args.grid,
                  split_history=not args.no_split_history)
    tries = args.placement_tries
    if world > 1 and tries > 1:
        # the candidates are alive together while they are timed: never more than fit next to the working set
        from fenics_constitutive_amd.placement import max_tries_for_memory

        tries = max_tries_for_memory(36 * n, tries, device, reserve_bytes=16 << 30)
    stage(f"workload built, placement {args.placement} with {tries} hipMalloc candidates")
    wl.place(args.placement, tries)
    stage(f"placed: {wl.vmm_info or wl.placement}")
    wl.warmup(args.warmup)
    wl.count_plastic()
    n_pl, n_its = wl.mean_plastic(args.steps)
    stage("warm-up and plastic counts done")

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record()
        wl.launch(i)
        ev1[i].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    stage("timed steps done")
    wl.launch_log.append(["timed", args.steps])
    red_dev = device if args.backend == "nccl" else "cpu"
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = sorted(ev0[i].elapsed_time(ev1[i]) for i in range(args.steps))
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)
    per_rank_ms = None
    if distributed:  # every rank's own average kernel time: shows the balance behind the max-over-ranks wall time
        tk = torch.zeros(world, dtype=torch.float64, device=red_dev)
        tk[rank] = kernel_avg_ms
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(x), 4) for x in tk.tolist()]

    # N > 1, weak run: the strong-scaling figure next to it -- the same total as ONE GPU's shard (--points), i.e. every rank
    # evaluates the first points/world points (tile-aligned) of its arrays; same bracket (barrier + synchronize, max over ranks)
    strong = None
    if distributed and world > 1 and args.scaling == "weak" and agree(budget_left() > 60):
        m = max(64, (args.n // world // 64) * 64)
        for i in range(max(2, args.warmup)):
            wl.launch(i, m=m)
        sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        dist.barrier()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for i, (a_, b_) in enumerate(sev):
            a_.record()
            wl.launch(i, m=m)
            b_.record()
        torch.cuda.synchronize()
        dist.barrier()
        s_el = time.perf_counter() - ts0
        tt = torch.tensor([s_el], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        s_el = float(tt.item())
        tk = torch.zeros(world, dtype=torch.float64, device=red_dev)
        tk[rank] = sum(a_.elapsed_time(b_) for a_, b_ in sev) / len(sev)
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        wl.launch_log.append(["strong_scaling_leg", max(2, args.warmup) + args.steps])
        strong = {"value": round(m * world * args.steps / s_el / 1e6, 1), "unit": "Mpts/s", "points_total": m * world, "points_per_gpu": m,
                  "ms_per_step": round(s_el / args.steps * 1e3, 4), "per_rank_kernel_ms": [round(float(x), 4) for x in tk.tolist()],
                  "note": "strong scaling: the total of ONE GPU's weak shard cut over all ranks (every rank evaluates the first points/world points "
                          "of its arrays), same timing bracket as `value`; compare with the N = 1 line's value"}
        for i in (0, 1):  # the sparse protocol's masks and trial rows back in step with whole-array launches
            wl.launch(i)
        wl.launch_log.append(["strong_scaling_resync", 2])

    legs_s = {"import_and_setup": round(t_start - t_prog, 1), "headline": round(time.perf_counter() - t_start, 1)}  # wall seconds per leg (detail file)
    t_leg = time.perf_counter()

    def leg_done(name_):
        nonlocal t_leg
        legs_s[name_] = round(time.perf_counter() - t_leg, 1)
        t_leg = time.perf_counter()

    # next to the headline: the same step without the sparse protocol (every launch rewrites the whole trial
    # history, fcamd_evaluate_device_from) -- six extra launches after the timed region
    full_ms = None
    if wl.sparse:
        ms = wl.timed_events(6, phase="full_trial_history", full_history=True, sparse_tangent=False)
        full_ms = sum(ms[1:]) / (len(ms) - 1)

    # ... the sparse protocol on the reference's array layout (ResidentState(packed_history=False); the headline of rounds 1-3)
    unpacked_ms = None
    if wl.packed:
        wl.launch(0, unpacked=True), wl.launch(1, unpacked=True)
        wl.launch_log.append(["sparse_unpacked_history_warm", 2])
        ms = wl.timed_events(6, phase="sparse_unpacked_history", unpacked=True, sparse_tangent=False)
        unpacked_ms = sum(ms) / len(ms)

    # ... and the reference's own call: law.evaluate(...) IN PLACE on the interface's arrays (what a drop-in torch caller launches;
    # the committed state is copied into the call's arrays before every launch, outside the event bracket)
    in_place_ms = in_place_draws = None
    if wl.plasticity and not wl.split and world == 1:
        try:
            ms = wl.timed_in_place(5, sets=2)
            in_place_ms = sum(ms) / len(ms)
            in_place_draws = list(wl.in_place_draws_ms)  # (the workload is freed before the line is assembled)
        except Exception as e:  # informational (e.g. no room for the second copy of the state)
            stage(f"in-place leg skipped: {type(e).__name__}: {e}")
    # ... and what the memory system alone takes for the headline's request stream: the kernel's synthetic twin
    mem_floor_ms = in_place_lines = None
    if world == 1:
        try:
            mem_floor_ms = wl.mem_floor(4)
            if in_place_ms is not None:
                in_place_lines = wl.in_place_line_bytes()
        except Exception as e:  # informational
            stage(f"mem-floor leg skipped: {type(e).__name__}: {e}")
    leg_done("reference_layout_legs")

    # "achievable" next to "peak" (SURVEY 8d): a plain device copy over half of the tangent array
    # (read + write counted)
    copy_gbs = stream_copy_gbs = fill_gbs = read_gbs = None
    try:
        half = (wl.tangent.numel() // 2) & ~1
        cs, ce = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(4):
            cs.record()
            wl.tangent[:half].copy_(wl.tangent[half : 2 * half])
            ce.record()
            ce.synchronize()
            ms_ = cs.elapsed_time(ce)
            best = ms_ if best is None else min(best, ms_)
        copy_gbs = 2 * 8 * half / (best * 1e-3) / 1e9
        # ... and the library's own copy kernel (fcamd_copy_device: the evaluate kernels' access pattern -- 16 B per lane,
        # non-temporal -- with nothing but the copy): what this box's memory gives a kernel of this kind
        from fenics_constitutive_amd.hostio import copy_device

        best = None
        for _ in range(5):
            cs.record()
            copy_device(wl.tangent[:half], wl.tangent[half : 2 * half])
            ce.record()
            ce.synchronize()
            ms_ = cs.elapsed_time(ce)
            best = ms_ if best is None else min(best, ms_)
        stream_copy_gbs = 2 * 8 * half / (best * 1e-3) / 1e9
        # ... and the two directions on their own: a pure write (fill) and a pure read (sum) of the same memory
        rates = {}
        for key, fn in (("fill", lambda: wl.tangent[:half].fill_(1.0)), ("read", lambda: wl.tangent[half : 2 * half].sum())):
            fn()
            best = None
            for _ in range(4):
                cs.record()
                fn()
                ce.record()
                ce.synchronize()
                ms_ = cs.elapsed_time(ce)
                best = ms_ if best is None else min(best, ms_)
            rates[key] = 8 * half / (best * 1e-3) / 1e9
        fill_gbs, read_gbs = rates["fill"], rates["read"]
    except Exception:  # the probe is informational only
        copy_gbs = None

    cpu_args = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # sample of the headline arrays, taken before they are released for the other configurations
        from benchlib.cpu import host_sample

        ncpu = min(n, 10_000_000)  # n = 1e7
        cpu_args = host_sample(wl.kind, wl.params, wl.grads[0][: 9 * ncpu], wl.stress_c[: 6 * ncpu],
                               None if wl.hist_c is None else {k: v[: {"eps_n": 6, "alpha": 1, "strain_visco": 6, "strain": 6, "history": 7}[k] * ncpu]
                                                               for k, v in wl.reference_history().items()}, wl.del_t)
    wl.drop_plain_twin()
    ever_fraction = wl.ever_fraction()
    headline = {"placement": wl.placement, "vmm_info": wl.vmm_info, "launch_log": list(wl.launch_log), "config_text": wl.config_text(), "kind": wl.kind,
                "b_el": wl.b_el, "b_pl": wl.b_pl, "alg": wl.alg_bytes(n_pl), "alg0": wl.alg_bytes(wl.n_pl_ab[0]),
                "fracs": placement_fracs(wl, wl.alg_bytes(wl.n_pl_ab[0])), "sparse": wl.sparse, "plasticity": wl.plasticity,
                "tkey": traffic_key(wl), "packed": wl.packed}

    out = None
    if rank == 0:  # the line is complete up to here; what follows only adds to it
        total_pts = (args.n if (args.scaling == "strong" and world > 1) else n * world) * args.steps
        value = total_pts / elapsed / 1e6
        alg_bytes = headline["alg"]
        achieved = alg_bytes / (kernel_avg_ms * 1e-3) / 1e9
        tkey = headline["tkey"]
        traffic = read_traffic(tkey, n)
        # what this box's memory allows for THIS kernel's measured mix of reads and writes: read bytes at the pure-read rate
        # plus written bytes at the pure-write rate, both measured above on the same array
        streaming_model = None
        split = read_traffic_split(tkey, n)
        if split and fill_gbs and read_gbs:
            model_ms = (split[0] / read_gbs + split[1] / fill_gbs) / 1e6
            streaming_model = {"ms": round(model_ms, 3), "kernel_over_model": round(kernel_avg_ms / model_ms, 3),
                               "note": "PMC read bytes / read_GBs + PMC written bytes / fill_GBs: the time it takes to stream the kernel's bytes at "
                                       "the rates torch's fill_ and sum reach on the same (placed) tangent array of this box; "
                                       "kernel_over_model < 1: the evaluate kernel moves its bytes faster than that"}
        out = {
            "metric": METRIC,
            "value": round(value, 1),
            "unit": "Mpts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": headline["config_text"],
                       "workload_short": f"{name}: {headline['kind']} FULL-3D, {n} pts/GPU device-resident, committed->trial evaluate"
                                         + (", packed sparse history" if headline["packed"] else (", sparse history" if headline["sparse"] else "")),
                       "ever_fraction": ever_fraction,
                       "baseline_config": BASELINE_CONFIG.get(name),
                       "points_per_gpu": n, "points_total": args.n if (args.scaling == "strong" and world > 1) else n * world,
                       "plastic_fraction": round(n_pl / n, 4),
                       "mean_newton_iters": round(n_its / max(n_pl, 1), 3) if headline["kind"] in ("von_mises_3d", "comfe_drucker_prager") else None,
                       "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": None if traffic is None else f"profiles/traffic.json[{tkey}]@kernel_hash={str(library_hash(kernels_only=True))[:16]} "
                                                                        "(stored rocprofv3 --pmc measurement of these kernels, not measured in this run)",
                         "kernel_ms_avg": round(kernel_avg_ms, 4), "kernel_ms_min": round(kernel_ms[0], 4),
                         "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 4),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "traffic_GBs": None if traffic is None else round(traffic / (kernel_avg_ms * 1e-3) / 1e9, 1),
                         "device_copy_GBs": None if copy_gbs is None else round(copy_gbs, 1),
                         "stream_copy_GBs": None if stream_copy_gbs is None else round(stream_copy_gbs, 1),
                         "fill_GBs": None if fill_gbs is None else round(fill_gbs, 1),
                         "read_GBs": None if read_gbs is None else round(read_gbs, 1),
                         "streaming_model": streaming_model,
                         "copy_note": "device_copy_GBs: torch's device copy; fill_GBs / read_GBs: torch fill_ / sum over one half of the tangent array; stream_copy_GBs: fcamd_copy_device (16 B per lane, non-temporal, "
                                      "the evaluate kernels' access pattern with nothing but the copy) over half of the tangent array, read + "
                                      "write counted -- the achievable rate of this box next to the 8 TB/s peak (SURVEY 8d); traffic_GBs is the "
                                      "evaluate kernel's PMC-measured bytes over its time",
                         "bytes_per_point": {"elastic": headline["b_el"], "plastic": headline["b_pl"]},
                         "placement_note": "frac = the timed steps, arrays placed as `placement.mode` says (the product default); "
                                           "frac_first_allocation = hipMalloc candidate 0, what a caller gets who takes the allocator's "
                                           "arrays as they come, frac_median/worst/best_candidate = the other hipMalloc draws "
                                           "(min of 3 launches each, iterate 0)"},
        }
        out["roofline"].update(headline["fracs"])
        if full_ms is not None:
            out["full_trial_history"] = {"kernel_ms_avg": round(full_ms, 4),
                                         "frac": round(alg_bytes / (full_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "note": "same step, whole trial history rewritten by every launch (--history full)"}
        if unpacked_ms is not None:
            out["sparse_unpacked_history"] = {"kernel_ms_avg": round(unpacked_ms, 4),
                                              "frac": round(alg_bytes / (unpacked_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                              "note": "same step, same sparse protocol, plastic-strain arrays in the reference's layout "
                                                      "(--history sparse; isolated 48-byte rows instead of one run per tile)"}
        if in_place_ms is not None:
            out["in_place"] = {"kernel_ms_avg": round(in_place_ms, 4),
                               "frac": round(alg_bytes / (in_place_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "allocation_draws_ms": in_place_draws,
                               "line_bytes": in_place_lines,
                               "note": "same step as the reference's own call: law.evaluate(...) in place on the interface's arrays (reference layout, "
                                       "no protocol words) -- what a drop-in torch caller launches"}
        if mem_floor_ms is not None:
            out["roofline"]["mem_floor_ms"] = round(mem_floor_ms, 4)
            out["roofline"]["kernel_over_mem_floor"] = round(kernel_avg_ms / mem_floor_ms, 4)
            out["roofline"]["mem_floor_note"] = ("the dominant kernel's synthetic twin (evaluate_twin_kernel: the same loads and stores at the same addresses on "
                                                 "the same buffers, the tiles' plastic ballots read from a recording of the real step, no constitutive "
                                                 "arithmetic), 4 event-timed launches after the timed region: what the memory system alone takes")
        out["placement"] = dict(headline["vmm_info"] or {"mode": "hipmalloc_tuned" if headline["placement"] else "first"})
        if headline["placement"] is not None:
            out["placement"].update({"hipmalloc_tangent_" + k: v for k, v in headline["placement"].items()})
            out["placement"]["tries"] = tries
        if per_rank_ms is not None:
            out["per_rank_kernel_ms"] = per_rank_ms
        if strong is not None:
            out["strong_scaling"] = strong
        out["launch_log"] = headline["launch_log"]
        out["library"] = {"srchash": library_hash(), "kernel_hash": library_hash(kernels_only=True)}

    def emit():
        out["wall_s"] = round(time.perf_counter() - t_prog, 1)
        out["legs_s"] = legs_s
        guard.final(dump_line(compact_line(out, write_detail(out, args.detail))))

    def checkpoint(leg):  # the line as it stands, in case the process does not survive `leg`
        if rank == 0:
            guard.provisional(dict(out, wall_s=round(time.perf_counter() - t_prog, 1), legs_s=legs_s), leg, args.detail)
            if os.environ.get("BENCH_DIE_IN") == leg.split(":")[0]:  # knob: fault injection (tests/test_gpu_bench_cli.py)
                os.kill(os.getpid(), 9)

    # the exchange step of config 5, timed separately (never part of `value`)
    do_gather = distributed and world > 1 and not args.no_gather
    if do_gather and not agree(budget_left() > 90):
        do_gather = False
        if rank == 0:
            out["allgather"] = {"skipped": f"wall budget: {budget_left():.0f} s left of --wall-budget {args.wall_budget:.0f}"}
    if do_gather:
        checkpoint("allgather")
        # An exchange between 8 processes can hang in ways a single GPU cannot rehearse (a peer mapping that never
        # returns, ranks leaving a collective in different places): the measured line must survive that.  If the leg
        # has not finished after --gather-timeout seconds, rank 0 prints the line with an error entry and every rank
        # leaves the process.
        import threading

        def give_up():
            if rank == 0:
                out["allgather"] = {"error": f"the all-gather leg did not finish within {args.gather_timeout} s; step timing above is complete"}
                emit()
            os._exit(3)  # the line is out, but a hung exchange is a FAILED leg: the driver must see it

        watchdog = threading.Timer(min(args.gather_timeout, max(30.0, budget_left() - 30.0)), give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            stage("all-gather leg")
            wl.launch(0, sparse_tangent=False)  # a complete trial stress / tangent for the gather to move
            keep_s, keep_t = wl.stress_t, wl.tangent
            wl.grads = wl.hist_t = wl.hmask = None  # the gather needs the room, the step timing is done
            torch.cuda.empty_cache()
            gather = time_allgather(args, dist, torch, device, rank, world, n, keep_s, keep_t, agree=agree, budget_left=budget_left)
        except Exception as e:  # e.g. no room on every rank alike: the step timing above stands
            gather = {"error": f"{type(e).__name__}: {e}"[:400]}
        watchdog.cancel()
        if rank == 0:
            out["allgather"] = gather
    wl.free()

    if distributed and world > 1:
        leg_done("multi_gpu_legs")
    # every other single-GPU configuration of BASELINE.json, same method, >= 5 event-timed launches each
    do_configs = args.configs == "all" or (args.configs == "auto" and args.workload is None and world == 1)
    if do_configs and rank == 0:
        configs = {}
        for k, cname in enumerate(EXTRA_CONFIGS):
            out["configs"] = configs
            checkpoint(f"configs: {cname}")
            if budget_left() < 45:
                configs[cname] = {"skipped": "wall budget"}
                continue
            try:
                # lean default: three hipMalloc candidates of the tangent, no VMM set; --full: four + the VMM set
                # (under rocprofv3 released VMM memory stays alive: five more working sets would not fit -- hipMalloc candidates only)
                lean_placement = "tune" if (args.placement == "auto" and (under_profiler() or not args.full)) else args.placement
                configs[cname] = run_config(cname, n, 4321 + k, device, dev_index, max(5, args.config_steps), 2,
                                            min(tries, 4 if args.full else 3), history=history, placement=lean_placement,
                                            cpu=args.full and not args.no_cpu_baseline)
            except Exception as e:  # one configuration failing must not lose the line
                configs[cname] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
            # the same streaming model as for the headline, with the rates measured on this box
            split = read_traffic_split(cname, n)
            if split and fill_gbs and read_gbs and "kernel_ms_avg" in configs[cname]:
                model_ms = (split[0] / read_gbs + split[1] / fill_gbs) / 1e6
                configs[cname]["streaming_model"] = {"ms": round(model_ms, 3),
                                                     "kernel_over_model": round(configs[cname]["kernel_ms_avg"] / model_ms, 3)}
        out["configs"] = configs
        leg_done("configs")

    # the "next" rows of SURVEY 8(f) on the roofline: indexed evaluate (f2), fused wrapper and low-dimensional kernels (f3), the
    # resident state's sparse-tangent Newton iteration (f1) -- same method as the configurations, appended to `configs`
    frow_names = []
    if do_configs and rank == 0 and not args.no_frows:
        from benchlib import frows as bench_frows

        for fname in bench_frows.FROWS:
            checkpoint(f"frows: {fname}")
            if budget_left() < 60:
                out["configs"][fname] = {"skipped": "wall budget"}
                continue
            try:
                out["configs"][fname] = bench_frows.run_frow(fname, n, device, launches=max(5, args.config_steps), peak_gbs=HBM_PEAK_GBS,
                                                             draws=3 if args.full else 2, floor_launches=4)
                frow_names.append(fname)
            except Exception as e:  # one row failing must not lose the line
                out["configs"][fname] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
        leg_done("frows")

    # N > 1: the single-process form of the host path on THIS node's GPUs (fcamd_multi) -- rank 0 alone drives all
    # of them over their own PCIe links while the other ranks wait on the CPU (a key in the process group's store, not a
    # collective: an RCCL barrier would keep their GPUs busy with a spinning kernel)
    if distributed and world > 1 and not args.no_host_path:
        import datetime

        store = None
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            pass
        if rank == 0:
            import threading

            checkpoint("host_path_multi")

            def give_up_host():  # as for the gather leg: a leg that hangs on hardware nobody could rehearse must not cost the line
                out["host_path_multi"] = {"error": "the single-process multi-GPU host leg did not finish within its time limit; everything above is complete"}
                emit()
                os._exit(3)

            host_watchdog = threading.Timer(max(45.0, min(150.0, budget_left() - 20.0)), give_up_host)
            host_watchdog.daemon = True
            host_watchdog.start()
            try:
                if budget_left() > 75 and store is not None:
                    have = torch.cuda.device_count()
                    devs = [k % have for k in range(world)]  # fewer GPUs than ranks (gloo rehearsal): contexts share devices
                    torch.cuda.empty_cache()
                    per_dev = min(n, 2_500_000)
                    out["host_path_multi"] = host_path_figures(devices=devs, sizes=(min(1_000_000, per_dev * world), per_dev * world),
                                                               latency_sizes=(), reps=2, budget_s=40.0)
                else:
                    out["host_path_multi"] = {"skipped": "wall budget" if store is not None else "no process-group store to wait on"}
            except Exception as e:  # informational: must not lose the line
                out["host_path_multi"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            host_watchdog.cancel()
            if store is not None:
                store.set("fcamd_host_multi_done", "1")
        elif store is not None:
            try:
                store.wait(["fcamd_host_multi_done"], datetime.timedelta(seconds=max(60.0, budget_left())))
            except Exception:
                pass  # rank 0 is late or gone: the closing barrier below decides

    if rank == 0:
        if world == 1 and not args.no_host_path and args.workload is None and budget_left() > 40:
            # the number a dolfinx user sees: the ndarray entries over PCIe ("timed separately and labelled as such")
            checkpoint("host_path")
            try:
                torch.cuda.empty_cache()
                big = min(n, 10_000_000 if args.full else 4_000_000)
                out["host_path"] = host_path_figures(devices=None, sizes=(1_000_000, big) if n > 1_000_000 else (n,))
            except Exception as e:  # informational: must not lose the line
                out["host_path"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            leg_done("host_path")
        # The CPU baseline has the host to itself: it runs BEFORE the PMC child passes start (round 5 ran it on a thread beside them --
        # two rocprofv3 children starting Python and parsing CSVs on the same cores can only slow the figure the GPU is compared with,
        # and the all-cores leg, 256 threads under a cgroup quota of 16 CPUs, came out at the one-thread rate: ADVICE r5)
        cpu_thread = None
        if world == 1:
            checkpoint("cpu_baseline")
            if cpu_args is not None:
                try:
                    out["cpu_baseline"] = cpu_baseline(*cpu_args, extras="full" if args.full else "lean")
                except Exception as e:  # noqa: BLE001 -- reported, never fatal
                    out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                leg_done("cpu_baseline")
            else:
                out["cpu_baseline"] = None
        if world == 1 and not args.no_live_traffic and budget_left() > 100:
            # roofline.traffic measured in THIS run (the kernels' memory is free by now); the stored figure stays next to it
            checkpoint("live_traffic")
            extra = (["--no-split-history"] if args.no_split_history else []) \
                + (["--sparse-tangent"] if args.sparse_tangent else [])
            torch.cuda.empty_cache()
            # ONE child per counter runs every measured item: the headline, its reference-layout forms, the configurations, the 8(f) rows
            items = [name] + ([name + "+unpacked"] if headline["packed"] else []) + ([name + "+in_place"] if in_place_ms is not None else [])
            measured = [c for c, v in (out.get("configs") or {}).items() if isinstance(v, dict) and "kernel_ms_avg" in v]
            items += [c for c in measured if c in frow_names] + [c for c in measured if c not in frow_names]
            # the default command stays under two minutes: what is left of 108 s goes to the two passes (each stops starting new
            # items when its half is used up: headline first, then its forms, the 8(f) rows, the configurations)
            # (counted from the end of the imports: a cold image pages torch in for a minute or two, which is nobody's to spend)
            target = 240.0 if args.full else 150.0  # (the CPU baseline, ~14 s, now runs before the passes instead of beside them)
            pass_s = max(12.0, (target - (time.perf_counter() - t_start)) / 2.0 - 3.0)
            lt_all = live_traffic_batch(items, n, history, extra, min(150.0, budget_left() - 30.0), headline=name, pass_budget_s=pass_s) or {}
            lt = lt_all.get(name)
            if lt is not None:
                rf = out["roofline"]
                rf["traffic_stored"] = rf["traffic"]
                rf["traffic"] = lt["hbm_bytes_per_launch"]
                rf["traffic_read_write"] = [lt["read_bytes"], lt["write_bytes"]]
                rf["traffic_GBs"] = round(lt["hbm_bytes_per_launch"] / (kernel_avg_ms * 1e-3) / 1e9, 1)
                rf["traffic_source"] = ("measured in this run: two child passes under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                                        "(4 timed launches each; KiB x 1024, FETCH_SIZE x 2 on gfx950); traffic_stored = profiles/traffic.json")
            for form, key in (("+unpacked", "sparse_unpacked_history"), ("+in_place", "in_place")):
                lf = lt_all.get(name + form)
                if lf is not None and key in out:
                    out[key]["traffic"] = lf["hbm_bytes_per_launch"]
                    out[key]["traffic_over_algorithmic"] = round(lf["hbm_bytes_per_launch"] / out["roofline"]["algorithmic_bytes_per_launch"], 4)
                    lb = out[key].get("line_bytes")
                    if lb:  # measured bytes over what the call must move at line / granule granularity (1.0: every line fetched is needed)
                        out[key]["traffic_read_write"] = [lf["read_bytes"], lf["write_bytes"]]
                        out[key]["read_over_needed_lines"] = round(lf["read_bytes"] / lb["read_bytes"], 4)
                        out[key]["write_over_needed_granules"] = round(lf["write_bytes"] / lb["write_bytes"], 4)
            for cname, c in (out.get("configs") or {}).items():
                lc = lt_all.get(cname)
                if lc is not None and isinstance(c, dict) and c.get("algorithmic_bytes_per_launch"):
                    c["traffic_stored"], c["traffic"] = c.get("traffic"), lc["hbm_bytes_per_launch"]
                    c["traffic_source"] = "measured in this run (rocprofv3 --pmc child passes, as roofline.traffic)"
                    c["traffic_over_algorithmic"] = round(lc["hbm_bytes_per_launch"] / c["algorithmic_bytes_per_launch"], 4)
            leg_done("live_traffic")
        if cpu_thread is not None:
            cpu_thread.join()
            leg_done("cpu_baseline_after_traffic")
        emit()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
