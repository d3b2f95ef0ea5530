"""Randomised call sequences on MultiDeviceResidentState (ONE process, several device contexts; here all on GPU 0): every
shortcut on -- sparse trial history per device, sparse tangent rows into the host array that holds the previous tangent,
constant tangent written once per del_t, split history, page-locked and pageable caller arrays, two tangent arrays taking
turns -- against a single-device ResidentState with every shortcut off.  After every call the arrays a caller can see
must be identical."""

import mmap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from fenics_constitutive_amd.multidevice import MultiDeviceResidentState  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402
from test_gpu_parity import make_law, random_case  # noqa: E402
from test_gpu_resident_fuzz import drucker_prager_case  # noqa: E402


def own(k):
    return np.frombuffer(mmap.mmap(-1, max(8 * k, 8)), dtype=np.float64, count=k)


def host(t):
    return {k: v.cpu().numpy() for k, v in t.items()} if isinstance(t, dict) else t.cpu().numpy()


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("kind,devices", [("von_mises_3d", [0, 0]), ("comfe_mises_plasticity", [0, 0, 0]), ("comfe_mises_plasticity+rows7", [0, 0]),
                                          ("drucker_prager", [0, 0, 0]), ("drucker_prager_hyperbolic", [0, 0]), ("linear_elasticity", [0, 0, 0]),
                                          ("spring_maxwell", [0, 0]), ("spring_kelvin", [0, 0, 0, 0])])
def test_random_call_sequences(kind, devices, seed):
    world = len(devices)
    n = 64 * (30 * world + 1) + 17  # ragged last slice, every slice above the one-device threshold set below
    kind, _, option = kind.partition("+")
    if kind.startswith("drucker_prager"):
        law, g0, s, h = drucker_prager_case(kind, n, seed)
    else:
        p, g0, s, h = random_case(kind, n, seed=seed)
        law = make_law(kind, p)
    md = MultiDeviceResidentState(law, n, devices=devices, stress0=s, history0=h, split_history=option != "rows7")
    assert [hi - lo for lo, hi in md.slices()].count(0) == 0
    ref = ResidentState(law, n, stress0=s, history0=h, sparse_history=False, sparse_tangent=False, reuse_constant_tangent=False,
                        placement="torch")
    rng = np.random.default_rng(500 + seed)
    pageable = {"g": np.empty(9 * n), "s": np.empty(6 * n), "t": np.full(36 * n, np.nan), "t2": np.full(36 * n, np.nan)}
    pinned = {"g": own(9 * n), "s": own(6 * n), "t": own(36 * n), "t2": own(36 * n)}
    pinned["t"][:] = np.nan
    pinned["t2"][:] = np.nan
    md.pin_host_arrays(*pinned.values())
    s_ref, t_ref = np.empty(6 * n), np.empty(36 * n)
    del_t, evaluated = 1.0, False
    try:
        for step in range(40):
            op = rng.choice(["host_pageable", "host_pinned", "update", "del_t", "set_state"], p=[0.3, 0.4, 0.17, 0.08, 0.05])
            scale = rng.choice([0.0, 0.02, 0.5, 1.0, 1.7])
            g = g0 * scale * (1.0 + 0.1 * rng.standard_normal())
            if op in ("host_pageable", "host_pinned"):
                arrs = pageable if op == "host_pageable" else pinned
                arrs["g"][:] = g
                which = rng.choice(["t", "t", "t", "t2", None], p=[0.3, 0.2, 0.2, 0.15, 0.15])
                with_stress = rng.random() < 0.85
                st = md.evaluate_into(0.0, del_t, arrs["g"], arrs["s"] if with_stress else None, None if which is None else arrs[which])
                law.last_stats = None
                ref.evaluate_into(0.0, del_t, g, s_ref, t_ref)
                if with_stress:
                    assert np.array_equal(arrs["s"], s_ref), (step, op)
                if which is not None:
                    assert np.array_equal(arrs[which], t_ref), (step, op, which)
                assert law.last_stats is None or (st.n_plastic, st.n_newton_iters) == (law.last_stats.n_plastic, law.last_stats.n_newton_iters)
                evaluated = True
            elif op == "update" and evaluated:
                md.update()
                ref.update()
                evaluated = False
            elif op == "del_t":
                del_t = float(rng.choice([0.5, 1.0, 2.0]))
                continue
            elif op == "set_state":
                s_new = s * float(rng.uniform(0.5, 1.5))
                md.set_state(s_new, h)
                ref.set_state(s_new, h)
                evaluated = False
            else:
                continue
            if evaluated:  # (before the first evaluate of an increment the multi-device state reports the committed state as trial)
                assert np.array_equal(md.stress, host(ref.stress)), (step, op)
            else:
                assert np.array_equal(md.stress, md.stress_committed), (step, op)
            assert np.array_equal(md.stress_committed, host(ref.stress_committed)), (step, op)
            if h is not None:
                got, got_c, want, want_c = md.history, md.history_committed, host(ref.history), host(ref.history_committed)
                for k in h:
                    assert np.array_equal(got[k], want[k]) or not evaluated, (step, op, k)
                    assert np.array_equal(got_c[k], want_c[k]), (step, op, k)
    finally:
        md.close()
