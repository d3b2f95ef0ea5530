"""Randomised call sequences on ResidentProblemState (three laws on interleaved rows): device evaluates and
host-assembler passes (``evaluate_law_into``: kernels write the page-locked parent arrays) with every
shortcut on, against a state with every shortcut off."""

import mmap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from fenics_constitutive_amd import _capi  # noqa: E402
from fenics_constitutive_amd.problem import ResidentProblemState  # noqa: E402
from test_gpu_parity import make_law, random_case  # noqa: E402


def own(k):
    return np.frombuffer(mmap.mmap(-1, max(8 * k, 8)), dtype=np.float64, count=k)


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_call_sequences(seed, packed):
    n = 9000
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n)
    rows = [np.sort(perm[:3500]), np.sort(perm[3500:6000]), np.sort(perm[6000:8200])]   # 800 points belong to no law
    kinds = ["von_mises_3d", "linear_elasticity", "spring_maxwell"]
    cases = [random_case(k, r.size, seed=seed + i) for i, (k, r) in enumerate(zip(kinds, rows))]
    laws = [make_law(k, c[0]) for k, c in zip(kinds, cases)]
    stress0 = rng.normal(size=6 * n)
    hist0 = [c[3] for c in cases]
    # most points of the VonMises3D law have never been plastic: +0.0 plastic-strain rows (what the packed layout leaves out)
    hist0[0]["eps_n"].reshape(-1, 6)[np.random.default_rng(50 + seed).random(rows[0].size) < 0.6] = 0.0
    opt = ResidentProblemState(list(zip(laws, rows)), n, del_t=1.0, packed_history=packed)
    assert opt._laws[0].packed == packed and not opt._laws[1].packed
    ref = ResidentProblemState(list(zip(laws, rows)), n, del_t=1.0, sparse_history=False, sparse_tangent=False,
                               reuse_constant_tangent=False)
    for st in (opt, ref):
        st.set_state(stress0, hist0)
    S, T = own(6 * n), own(36 * n)
    G = [own(9 * r.size) for r in rows]
    ctx = laws[0]._handle(_capi.default_device()).ctx
    for a in [S, T] + G:
        ctx.register_host_buffer(a)
    s_ref, t_ref = np.empty(6 * n), np.empty(36 * n)
    owned = np.concatenate(rows)
    free = np.setdiff1d(np.arange(n), owned)
    evaluated = False
    try:
        for step in range(30):
            op = rng.choice(["dev", "host", "host", "update", "del_t"], p=[0.25, 0.3, 0.25, 0.12, 0.08])
            if op == "update":
                if evaluated:
                    opt.update(), ref.update()
                    evaluated = False
                continue
            if op == "del_t":
                dt = float(rng.choice([0.5, 1.0, 2.0]))
                opt._del_t = ref._del_t = dt
                continue
            scale = rng.choice([0.0, 0.05, 0.6, 1.0, 1.8])
            grads = [c[1] * scale * (1.0 + 0.1 * rng.standard_normal()) for c in cases]
            ref.evaluate(grads)
            if op == "dev":
                opt.evaluate(grads)
                torch.cuda.synchronize()
                assert torch.equal(opt.tangent, ref.tangent), step
            else:
                for k in range(3):
                    G[k][:] = grads[k]
                    opt.evaluate_law_into(k, G[k], S, T, sync=(k == 2))
                ref.download(s_ref, t_ref)   # rows that belong to no law are nobody's business: compare the owned ones
                assert np.array_equal(S.reshape(-1, 6)[owned], s_ref.reshape(-1, 6)[owned]), step
                assert np.array_equal(T.reshape(-1, 36)[owned], t_ref.reshape(-1, 36)[owned]), step
                assert not S.reshape(-1, 6)[free].any() and not T.reshape(-1, 36)[free].any()  # ... and never written
            evaluated = True
            torch.cuda.synchronize()
            assert torch.equal(opt.stress_1, ref.stress_1) and torch.equal(opt.stress_0, ref.stress_0), step
            for ho, hr in zip(opt._history_1, ref._history_1):
                if ho is not None:
                    for key in ho:
                        assert torch.equal(ho[key], hr[key]), (step, key)
    finally:
        for a in [S, T] + G:
            ctx.unregister_host_buffer(a)
