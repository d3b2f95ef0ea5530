#!/usr/bin/env python3
"""Full-size soak of the resident protocols (1e8 points, one MI355X): several load increments of several Newton iterates each,
plastic sets shrinking and growing between iterates, a commit after every increment -- the default state (sparse trial history,
sparse tangent, packed / split plastic-strain history) against a state with every shortcut off, bit for bit after EVERY iterate.
    python tests/fullsize_soak.py [points] [increments] [seed]        one line per law, exit code 1 on the first difference
tests/test_gpu_fullsize.py::test_resident_protocols_1e8 is the three-iterate version of this inside the suite."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
INCREMENTS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
f = dict(dtype=torch.float64, device=dev)
LAWS = {
    "VonMises3D": lambda: fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}),
    "MisesPlasticityLinearHardening3D": lambda: fc.MisesPlasticityLinearHardening3D(
        {k: np.array([v]) for k, v in {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0}.items()}),
    "DruckerPragerHyperbolic3D": lambda: fc.DruckerPragerHyperbolic3D(
        {k: np.array([v]) for k, v in {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02, "d": 40.0}.items()}),
    # every point has yielded once (no +0.0 row: runs of 64 rows) and few yield now: the rows-inside-the-run path of the packed layout
    "VonMises3D, every point yielded once": lambda: fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}),
}


def run(name):
    try:
        law = LAWS[name]()
    except Exception as e:  # noqa: BLE001 -- parameter names differ per law: say so instead of guessing
        print(name, "not constructed:", e, flush=True)
        return True
    gen = torch.Generator(device=dev).manual_seed(SEED)
    base = torch.randn(9 * N, generator=gen, **f)
    if name.startswith("DruckerPrager"):  # bench.py's recipe: mostly isochoric increments, scale log-uniform in [1e-4, 5e-3]
        gv = base.view(N, 9)
        gv.mul_(torch.pow(10.0, torch.rand(N, generator=gen, **f) * 1.7 - 4.0)[:, None])
        tr = (gv[:, 0] + gv[:, 4] + gv[:, 8]) * (0.95 / 3.0)
        for c in (0, 4, 8):
            gv[:, c] -= tr
        del tr, gv
    else:
        base.view(N, 9).mul_(torch.pow(10.0, torch.rand(N, generator=gen, **f) * 2 - 4)[:, None])
    yielded = "yielded once" in name
    h0 = None
    if yielded:
        h0 = {"eps_n": torch.randn(6 * N, generator=gen, **f) * 1e-4, "alpha": torch.rand(N, generator=gen, **f) * 0.01}
    sp = ResidentState(law, N, history0=h0)                              # the product default
    fu = ResidentState(law, N, history0=h0, sparse_history=False, sparse_tangent=False, packed_history=False,
                       **({"split_history": False} if not name.startswith("VonMises3D") else {}))
    del h0
    t0, fr, iterates = time.time(), [], 0
    for inc in range(INCREMENTS):
        for it in range(3):
            # iterate 0 overshoots, 1 falls back, 2 settles in between: the plastic set of the trial state shrinks and grows
            scale = (0.6 + 0.1 * inc) * (1.5, 0.4, 1.0)[it] * (0.22 if yielded else 1.0)
            g = base * scale
            sp.evaluate(float(inc), 1.0, g)
            fu.evaluate(float(inc), 1.0, g)
            del g
            iterates += 1
            fr.append(round(fu.check().n_plastic / N, 3))
            ok = torch.equal(sp.stress, fu.stress) and torch.equal(sp.tangent, fu.tangent)
            hs, hf = sp.history, fu.history
            ok = ok and all(torch.equal(hs[k], hf[k]) for k in hf)
            del hs, hf
            if not ok:
                print(f"{name}: DIFFERENCE at increment {inc} iterate {it} (plastic fractions so far {fr})", flush=True)
                return False
        sp.update()
        fu.update()
        hs, hf = sp.history_committed, fu.history_committed
        if not all(torch.equal(hs[k], hf[k]) for k in hf):
            print(f"{name}: committed history differs after increment {inc}", flush=True)
            return False
        del hs, hf
    print(f"{name}: {N} points, {INCREMENTS} increments x 3 iterates = {iterates} evaluates per state, all arrays bit-identical after every "
          f"iterate and every commit; plastic fraction per iterate {fr}; {time.time() - t0:.0f} s", flush=True)
    del sp, fu, base
    torch.cuda.empty_cache()
    return True


good = all([run(name) for name in LAWS])
sys.exit(0 if good else 1)
