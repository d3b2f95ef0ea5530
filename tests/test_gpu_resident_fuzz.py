"""Randomised call sequences on ResidentState: every shortcut on (sparse trial history, sparse tangent,
constant tangent written / downloaded once, zero-copy host arrays, placement tuning) against a state with
every shortcut off.  After every call the arrays a caller can see must be identical."""

import mmap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from fenics_constitutive_amd import _capi  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402
from test_gpu_parity import make_law, random_case  # noqa: E402


def own(k):
    return np.frombuffer(mmap.mmap(-1, max(8 * k, 8)), dtype=np.float64, count=k)


def drucker_prager_case(kind, n, seed):
    """(law, base gradient, stress0, history0) in the regime in which the reference's Newton iteration converges for every
    scale the sequences apply (<= 1.9 x the base): compressive prestress, mostly isochoric increments up to 3e-3."""
    import fenics_constitutive_amd as fc

    rng = np.random.default_rng(seed)
    p = {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02}
    cls = fc.DruckerPrager3D
    if kind.endswith("hyperbolic"):
        p = {"mu": p["mu"], "kappa": p["kappa"], "a": p["a"], "b": p["b"], "d": 40.0, "b_flow": p["b_flow"]}
        cls = fc.DruckerPragerHyperbolic3D
    law = cls({k: np.array([v]) for k, v in p.items()})
    g = (rng.normal(size=9 * n) * np.repeat(10 ** rng.uniform(-5.0, -2.52, size=n), 9)).reshape(-1, 9)
    g[:, [0, 4, 8]] -= (0.95 * g[:, [0, 4, 8]].sum(axis=1) / 3.0)[:, None]
    s = rng.normal(scale=30.0, size=6 * n)
    s.reshape(-1, 6)[:, :3] -= 1000.0
    h = rng.normal(scale=1e-3, size=7 * n)
    h.reshape(-1, 7)[:, 0] = 0.0
    return law, g.reshape(-1).copy(), s, {"history": h}


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("kind", ["von_mises_3d", "von_mises_3d+unpacked", "von_mises_3d+dense_rows", "comfe_mises_plasticity",
                                  "comfe_mises_plasticity+unpacked", "comfe_mises_plasticity+rows7", "drucker_prager",
                                  "drucker_prager+unpacked", "drucker_prager_hyperbolic", "drucker_prager_hyperbolic+unpacked",
                                  "linear_elasticity", "spring_maxwell"])
def test_random_call_sequences(kind, seed):
    n = 64 * 90 + 17
    kind, _, option = kind.partition("+")
    if kind.startswith("drucker_prager"):
        law, g0, s, h = drucker_prager_case(kind, n, seed)
    else:
        p, g0, s, h = random_case(kind, n, seed=seed)
        law = make_law(kind, p)
    plasticity = kind in ("von_mises_3d", "comfe_mises_plasticity") or kind.startswith("drucker_prager")
    if plasticity and option != "dense_rows":
        # most points have never been plastic: their plastic-strain rows are +0.0 (what the packed layout leaves out)
        key, w = ("eps_n", 6) if kind == "von_mises_3d" else ("history", 7)
        virgin = np.random.default_rng(7 + seed).random(n) < 0.6
        h[key].reshape(-1, w)[virgin, w - 6:] = 0.0
    opt = ResidentState(law, n, stress0=s, history0=h, split_history=option != "rows7", packed_history=option != "unpacked")
    # the packed plastic-strain layout is the default wherever a law has a plastic-strain array of its own
    assert opt._packed == (plasticity and option in ("", "dense_rows"))
    assert opt._split == (option != "rows7" and (kind == "comfe_mises_plasticity" or kind.startswith("drucker_prager")))
    ref = ResidentState(law, n, stress0=s, history0=h, sparse_history=False, sparse_tangent=False,
                        reuse_constant_tangent=False)
    rng = np.random.default_rng(100 + seed)
    ctx = law._handle(_capi.default_device()).ctx
    # host array sets of the optimised state: pageable, and page-locked (zero copy)
    pageable = {"g": np.empty(9 * n), "s": np.empty(6 * n), "t": np.full(36 * n, np.nan)}
    pinned = {"g": own(9 * n), "s": own(6 * n), "t": own(36 * n)}
    pinned["t"][:] = np.nan
    for a in pinned.values():
        ctx.register_host_buffer(a)
    s_ref, t_ref = np.empty(6 * n), np.empty(36 * n)
    del_t, evaluated = 1.0, False
    try:
        for step in range(40):
            op = rng.choice(["dev", "host_pageable", "host_pinned", "host_pinned", "update", "del_t", "tune"],
                            p=[0.2, 0.15, 0.2, 0.15, 0.15, 0.1, 0.05])
            scale = rng.choice([0.0, 0.02, 0.5, 1.0, 1.7])
            g = g0 * scale * (1.0 + 0.1 * rng.standard_normal())
            if op == "dev":
                opt.evaluate(0.0, del_t, g)
                ref.evaluate(0.0, del_t, g)
                torch.cuda.synchronize()
                assert torch.equal(opt.tangent, ref.tangent), (step, op)
                evaluated = True
            elif op in ("host_pageable", "host_pinned"):
                arrs = pageable if op == "host_pageable" else pinned
                arrs["g"][:] = g
                with_tangent = rng.random() < 0.8
                opt.evaluate_into(0.0, del_t, arrs["g"], arrs["s"], arrs["t"] if with_tangent else None)
                ref.evaluate_into(0.0, del_t, g, s_ref, t_ref)
                assert np.array_equal(arrs["s"], s_ref), (step, op)
                if with_tangent:
                    assert np.array_equal(arrs["t"], t_ref), (step, op)
                evaluated = True
            elif op == "update" and evaluated:
                opt.update()
                ref.update()
                evaluated = False
            elif op == "del_t":
                del_t = float(rng.choice([0.5, 1.0, 2.0]))
                continue
            elif op == "tune":
                opt.tune_placement(0.0, del_t, g, tries=2)
                ref.evaluate(0.0, del_t, g)
                torch.cuda.synchronize()
                assert torch.equal(opt.tangent, ref.tangent), (step, op)
                evaluated = True
            else:
                continue
            torch.cuda.synchronize()
            assert torch.equal(opt.stress, ref.stress), (step, op)
            assert torch.equal(opt.stress_committed, ref.stress_committed), (step, op)
            if h is not None:
                for k in h:
                    assert torch.equal(opt.history[k], ref.history[k]) or not evaluated, (step, op, k)
                    assert torch.equal(opt.history_committed[k], ref.history_committed[k]), (step, op, k)
    finally:
        for a in pinned.values():
            ctx.unregister_host_buffer(a)
