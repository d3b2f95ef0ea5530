"""Placement of the arrays a resident state streams: "tune" (hipMalloc candidates of the
tangent), "vmm" (one interleaved VMM working set) and "auto" (the faster of the two) must leave every number
exactly as the plain torch-allocated state computes it -- over Newton iterations, commits, host and device
gradients.  The size threshold of the placement step is lowered so that small states take it."""

import numpy as np
import pytest
import torch

import fenics_constitutive_amd as fc
from fenics_constitutive_amd.problem import ResidentProblemState
from fenics_constitutive_amd.resident import ResidentState
from test_gpu_parity import make_law, random_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["tune", "vmm", "auto"])
@pytest.mark.parametrize("kind", ["von_mises_3d", "linear_elasticity", "spring_kelvin", "comfe_mises_plasticity"])
def test_resident_state_placement_modes_keep_results(kind, mode, monkeypatch):
    monkeypatch.setattr(ResidentState, "AUTO_TUNE_MIN_BYTES", 0)
    n = 64 * 150 + 7
    p, g, s, h = random_case(kind, n, seed=5)
    law = make_law(kind, p)
    a = ResidentState(law, n, stress0=s, history0=h, placement="torch")
    b = ResidentState(law, n, stress0=s, history0=h, placement=mode)
    gd = torch.from_numpy(g).cuda()
    for inc in range(3):
        for it, scale in enumerate((1.0, 0.4)):
            a.evaluate(0.0, 1.0, gd * scale)
            b.evaluate(0.0, 1.0, (g * scale) if (inc + it) % 2 else gd * scale)  # NumPy and device gradients alternate
            assert torch.equal(a.stress, b.stress) and torch.equal(a.tangent, b.tangent), (inc, it)
            for k in (h or {}):
                assert torch.equal(a.history[k], b.history[k]), (inc, it, k)
        a.update()
        b.update()
        assert torch.equal(a.stress_committed, b.stress_committed)
    const = kind in ("linear_elasticity", "spring_kelvin")
    assert a.placement is None
    if mode == "vmm":
        assert b.placement["mode"] == "vmm_interleaved" and b._vmm is not None
    elif mode == "tune":  # a tangent that is written once per del_t has nothing to tune
        assert (b.placement is None) if const else (b.placement["mode"] == "hipmalloc_tuned")
    else:  # auto: one of the two, and the record says which timings decided
        assert b.placement["mode"] in ("vmm_interleaved", "hipmalloc_tuned")
        assert const or {"vmm_ms", "hipmalloc_best_ms", "candidate_ms"} <= set(b.placement)
        assert (b._vmm is not None) == (b.placement["mode"] == "vmm_interleaved")


@pytest.mark.parametrize("mode", ["tune", "vmm", "auto"])
def test_problem_state_placement_modes_keep_results(mode, monkeypatch):
    monkeypatch.setattr(ResidentProblemState, "AUTO_TUNE_MIN_BYTES", 0)
    n = 6000
    rng = np.random.default_rng(3)
    perm = rng.permutation(n)
    rows = [np.sort(perm[:2500]).astype(np.int32), np.sort(perm[2500:5200]).astype(np.int32)]  # 800 points unowned
    kinds = ["linear_elasticity", "von_mises_3d"]
    cases = [random_case(k, r.size, seed=11 + i) for i, (k, r) in enumerate(zip(kinds, rows))]
    laws = [make_law(k, c[0]) for k, c in zip(kinds, cases)]
    grads = [c[1] for c in cases]
    stress, hist = rng.normal(size=6 * n), [c[3] for c in cases]
    a = ResidentProblemState(list(zip(laws, rows)), n, del_t=1.0, placement="torch")
    b = ResidentProblemState(list(zip(laws, rows)), n, del_t=1.0, placement=mode)
    for st in (a, b):
        st.set_state(stress, hist)
    for inc in range(3):
        for scale in (1.0, 0.5):
            gs = [g * scale for g in grads]
            a.evaluate(gs)
            b.evaluate(gs)
            assert torch.equal(a.stress_1, b.stress_1) and torch.equal(a.tangent, b.tangent), inc
            for ha, hb in zip(a._history_1, b._history_1):
                for k in (ha or {}):
                    assert torch.equal(ha[k], hb[k])
        a.update()
        b.update()
    assert b.placement["mode"] in ("vmm_interleaved", "hipmalloc_tuned")
    assert (b._vmm is not None) == (b.placement["mode"] == "vmm_interleaved")


def test_empty_tangent_places_without_touching_the_callers_state():
    """DeviceLaw.empty_tangent: the candidate search for a caller who keeps the reference's in-place protocol on tensors of their own"""
    p, g, s, h = random_case("von_mises_3d", 64 * 500 + 9, seed=4)
    law = make_law("von_mises_3d", p)
    gd, sd = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()
    hd = {k: torch.from_numpy(v).cuda() for k, v in h.items()}
    s0, h0 = sd.clone(), {k: v.clone() for k, v in hd.items()}
    tangent, info = law.empty_tangent(gd, sd, hd, tries=3)
    assert tangent.numel() == 36 * (s.size // 6) and tangent.is_cuda and tangent.dtype == torch.float64
    assert len(info["candidate_ms"]) == 3 and 0 <= info["chosen"] < 3 and all(ms > 0 for ms in info["candidate_ms"])
    assert torch.equal(sd, s0) and all(torch.equal(hd[k], h0[k]) for k in hd)  # the probes wrote scratch arrays only
    # ... and the in-place call on the placed array gives what it gives on any other
    ref_t = torch.empty_like(tangent)
    s1, h1 = sd.clone(), {k: v.clone() for k, v in hd.items()}
    law.evaluate(0.0, 1.0, gd, s1, ref_t, h1)
    law.evaluate(0.0, 1.0, gd, sd, tangent, hd)
    assert torch.equal(tangent, ref_t) and torch.equal(sd, s1)
