#!/usr/bin/env python3
"""A/B two builds of libfcamd.so in ONE process on identical buffers (interleaved rounds):
    python tools/ab_lib.py libA.so libB.so [n ...]
VonMises3D mixed workload, committed->trial evaluate.  AB_SPARSE=1: sparse trial-history protocol
(fcamd_evaluate_device_ex with history_mask, VonMises3D only); AB_ZONED=1: plastic points in contiguous
4096-point zones instead of a random mixture; AB_SCALE=1e-2 / 1e-4: uniform strain scale (all plastic / all elastic);
AB_CONSTRAINT=1..4 (le / maxwell): a low-dimensional constraint on the first entries of the same arrays."""
import ctypes as C
import sys

import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenics_constitutive_amd._capi import EvalArgs  # noqa: E402  (the struct only; the libraries are loaded by path below)
libs = [a for a in sys.argv[1:] if '.so' in a]  # path.so or path.so@GRID (context option "grid": workgroups of the launch)
sizes = [int(float(x)) for x in sys.argv[1:] if '.so' not in x] or [1_000_000, 10_000_000, 100_000_000]
LAW = os.environ.get("AB_LAW", "vm")  # vm | le | maxwell
SPARSE = os.environ.get("AB_SPARSE", "0") == "1"
INPLACE = os.environ.get("AB_INPLACE", "0") == "1"  # stress_prev == stress, history_prev == history (state restored before every launch)
FLAGS = int(os.environ.get("AB_FLAGS", "0"))  # fcamd_eval_args.flags of the sparse protocol's launches (1 = sparse tangent)
ZONED = os.environ.get("AB_ZONED", "0") == "1"
CONSTRAINT = int(os.environ.get("AB_CONSTRAINT", "5"))
MODEL = {"vm": (2, [175000.0, 80769.0, 1200.0, 2500.0, 200.0], 2), "le": (1, [42.0, 0.3], 0), "maxwell": (3, [42.0, 10.0, 10.0, 0.2], 2),
         "dp": (7, [80769.0, 175000.0, 100.0, 0.05, 0.02], 1)}[LAW]  # dp: Drucker-Prager (classic), bench.py's drucker_prager_mixed inputs
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = (C.c_double * len(MODEL[1]))(*MODEL[1])


class Lib:
    def __init__(self, path):
        path, _, grid = path.partition("@")
        self.l = C.CDLL(path)
        self.ctx, self.m = C.c_void_p(), C.c_void_p()
        assert self.l.fcamd_context_create(0, stream, C.byref(self.ctx)) == 0
        # torch's default stream has handle 0 = "own a private stream" for create(); bind it explicitly
        assert self.l.fcamd_context_set_stream(self.ctx, stream) == 0
        if grid:
            self.l.fcamd_context_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
            assert self.l.fcamd_context_set_option(self.ctx, b"grid", int(grid)) == 0
        assert self.l.fcamd_model_create(self.ctx, MODEL[0], CONSTRAINT, P, len(MODEL[1]), C.byref(self.m)) == 0
        self.l.fcamd_evaluate_device_ex.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int64, C.POINTER(EvalArgs)]
        self.mask = None
        self.warm = False

    def run(self, n, g, s0, s1, t, h0, h1, ev=None):
        a0 = (C.c_void_p * 2)(h0[0].data_ptr(), h0[-1].data_ptr())
        a1 = (C.c_void_p * 2)(h1[0].data_ptr(), h1[-1].data_ptr())
        if SPARSE:
            if self.mask is None or self.mask.numel() != (n + 63) // 64:
                # protocol: trial == committed wherever the mask is clear
                h1[0].copy_(h0[0]), h1[-1].copy_(h0[-1])
                self.mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
                self.warm = False
            flags = FLAGS if self.warm else 0  # the first launch of a size writes the whole tangent (the sparse-tangent protocol's premise)
            self.warm = True
            x = EvalArgs(g.data_ptr(), s0.data_ptr(), s1.data_ptr(), t.data_ptr(), a0, a1, MODEL[2], None, self.mask.data_ptr(), flags,
                         None, None, None, None, 0, None)
            rc = self.l.fcamd_evaluate_device_ex(self.m, 0.0, 1.0, n, C.byref(x))
            assert rc == 0, rc
            return
        if INPLACE:  # the reference's own call: in place on the interface's arrays (the committed state is copied in first, outside the caller's events)
            s1.copy_(s0), h1[0].copy_(h0[0]), h1[-1].copy_(h0[-1])
            if ev is not None:
                ev[0].record()
            x = EvalArgs(g.data_ptr(), s1.data_ptr(), s1.data_ptr(), t.data_ptr(), a1, a1, MODEL[2], None, None, 0, None, None, None, None, 0, None)
        else:
            x = EvalArgs(g.data_ptr(), s0.data_ptr(), s1.data_ptr(), t.data_ptr(), a0, a1, MODEL[2], None, None, 0, None, None, None, None, 0, None)
        rc = self.l.fcamd_evaluate_device_ex(self.m, 0.0, 1.0, n, C.byref(x))
        assert rc == 0, rc


L = [Lib(p) for p in libs]
f = dict(dtype=torch.float64, device=dev)
for n in sizes:
    gen = torch.Generator(device=dev).manual_seed(1)
    g = torch.randn(9 * n, generator=gen, **f)
    if ZONED:
        pl = torch.rand((n + 4095) // 4096, generator=gen, **f) < 0.22
        g.view(n, 9).mul_(torch.where(pl, 1e-2, 1e-4).to(torch.float64).repeat_interleave(4096)[:n][:, None])
    elif os.environ.get("AB_SCALE"):  # uniform strain scale: 1e-2 all plastic, 1e-4 all elastic
        g.mul_(float(os.environ["AB_SCALE"]))
    else:
        g.view(n, 9).mul_(torch.pow(10.0, torch.rand(n, generator=gen, **f) * 2 - 4)[:, None])
    s0, s1 = torch.zeros(6 * n, **f), torch.empty(6 * n, **f)
    if LAW == "dp":  # mostly isochoric increments, scale log-uniform in [1e-4, 5e-3], compressive prestress (benchlib/workloads.py)
        g = torch.randn(9 * n, generator=gen, **f)
        gv = g.view(n, 9)
        gv.mul_(torch.pow(10.0, torch.rand(n, generator=gen, **f) * 1.7 - 4.0)[:, None])
        tr = (gv[:, 0] + gv[:, 4] + gv[:, 8]) * (0.95 / 3.0)
        for c_ in (0, 4, 8):
            gv[:, c_] -= tr
        s0.view(n, 6)[:, :3] = -1000.0
        h0 = [torch.zeros(7 * n, **f)]
        h1 = [torch.empty(7 * n, **f)]
    elif LAW == "maxwell":
        h0 = [torch.zeros(6 * n, **f), torch.zeros(6 * n, **f)]
        h1 = [torch.empty(6 * n, **f), torch.empty(6 * n, **f)]
    else:
        h0 = [torch.zeros(6 * n, **f), torch.rand(n, generator=gen, **f) * 0.02]
        h1 = [torch.empty(6 * n, **f), torch.empty(n, **f)]
    t = torch.empty(36 * n, **f)
    ncand = int(os.environ.get("AB_TANGENT_CANDS", "0"))
    if ncand:  # placement study: every build on the same candidate allocations of the tangent
        cands = [t] + [torch.empty(36 * n, **f) for _ in range(ncand - 1)]
        for ci, tc in enumerate(cands):
            row = []
            for lib in L:
                lib.run(n, g, s0, s1, tc, h0, h1)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
                for a, b in ev:
                    a.record()
                    lib.run(n, g, s0, s1, tc, h0, h1)
                    b.record()
                torch.cuda.synchronize()
                row.append(min(a.elapsed_time(b) for a, b in ev))
            print(f"n={n} cand {ci:2d}: " + "  ".join(f"{libs[i].split('/')[-1]} {x:7.3f}" for i, x in enumerate(row)), flush=True)
        del cands
        continue
    outs = []
    res = [[] for _ in L]
    reps = 5 if n >= 10**8 else 20
    for rnd in range(5):
        for i, lib in enumerate(L):
            for _ in range(2):
                lib.run(n, g, s0, s1, t, h0, h1)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in ev:
                if not INPLACE:
                    a.record()
                lib.run(n, g, s0, s1, t, h0, h1, ev=(a, b))
                b.record()
            torch.cuda.synchronize()
            res[i].append(sum(a.elapsed_time(b) for a, b in ev) / reps)
            if rnd == 0:
                outs.append((s1.clone(), t.clone() if n <= 10**7 else None, h1[0].clone()))
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][2], o[2]) and (o[1] is None or torch.equal(outs[0][1], o[1])) for o in outs[1:])
    maxdiff = max(float((outs[0][0] - o[0]).abs().max()) for o in outs[1:]) if len(outs) > 1 else 0.0
    print(f"n={n:>10}: " + "  ".join(f"{libs[i].split('/')[-1]} {sorted(r)[len(r)//2]*1e3:9.1f} us" for i, r in enumerate(res)) + f"   identical={same} max|dstress|={maxdiff:.3e}", flush=True)
    del g, s0, s1, h0, h1, t
    torch.cuda.empty_cache()
