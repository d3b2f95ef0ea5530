"""Device-resident increment state replays the reference protocol -- several
Newton re-evaluations per increment, commit, next increment -- and must reproduce the golden
multi-step sequences captured from the reference."""

import numpy as np
import pytest
from golden_util import load_calls, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402

FULL = fc.StressStrainConstraint.FULL


def test_von_mises_mixed_sequence():
    calls = {c.name: c for c in load_calls("von_mises_3d.npz")}
    c0 = calls["mixed_step0_iter0"]
    law = fc.VonMises3D(c0.params)
    st = ResidentState(law, c0.n, stress0=c0.stress_in, history0=c0.hist_in)
    for k in range(4):
        for it in (0, 1):
            c = calls[f"mixed_step{k}_iter{it}"]
            st.evaluate(0.0, c.del_t, c.grad)  # NumPy grad: the only PCIe upload
            s, t = np.empty(6 * c.n), np.empty(36 * c.n)
            h = {"eps_n": np.empty(6 * c.n), "alpha": np.empty(c.n)}
            st.download(s, t, h)
            assert rel_err(s, c.stress_out) <= 1e-6 and rel_err(t, c.tangent_out) <= 1e-6
            assert rel_err(s, c.stress_out) <= 1e-11
            for key in h:
                assert rel_err(h[key], c.hist_out[key]) <= 1e-6
            # the committed copy is untouched by trial evaluations
            assert np.array_equal(st.stress_committed.cpu().numpy(), c.stress_in)
        st.check()
        st.update()
    assert np.array_equal(st.stress_committed.cpu().numpy(), s)


@pytest.mark.parametrize("fname,cls", [("spring_maxwell.npz", "SpringMaxwellModel"), ("spring_kelvin.npz", "SpringKelvinModel")])
def test_sls_sequence(fname, cls):
    calls = {c.name: c for c in load_calls(fname)}
    c0 = calls["step0_iter0"]
    law = getattr(fc, cls)(c0.params, FULL)
    st = ResidentState(law, c0.n, stress0=c0.stress_in, history0=c0.hist_in)
    for k in range(5):
        for it in (0, 1):
            c = calls[f"step{k}_iter{it}"]
            st.evaluate(0.0, c.del_t, torch.from_numpy(c.grad).cuda())  # device grad: zero copies
            assert rel_err(st.stress.cpu().numpy(), c.stress_out) <= 1e-10
            assert rel_err(st.tangent.cpu().numpy(), c.tangent_out) <= 1e-10
            for key in c.hist_out:
                assert rel_err(st.history[key].cpu().numpy(), c.hist_out[key]) <= 1e-10
        st.update()


def test_update_without_evaluate_raises():
    st = ResidentState(fc.LinearElasticityModel({"E": 1.0, "nu": 0.3}, FULL), 10)
    with pytest.raises(RuntimeError):
        st.update()
    st.evaluate(0, 1, np.zeros(90))
    st.update()
