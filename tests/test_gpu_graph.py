"""The device path is capturable in a HIP graph (no allocation, no synchronisation, launches on the
caller's stream): a small-n Newton loop can replay evaluate() without per-call launch overhead."""

import numpy as np
import pytest
from test_gpu_parity import make_law, oracle_run, random_case
from golden_util import rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("kind", ["von_mises_3d", "spring_maxwell"])
def test_evaluate_in_hip_graph(kind):
    n = 64 * 50 + 11
    p, g, s, h = random_case(kind, n, seed=12)
    law = make_law(kind, p)
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    gd, s0, s1, t = d(g), d(s), torch.zeros(6 * n, dtype=torch.float64, device="cuda"), torch.zeros(36 * n, dtype=torch.float64, device="cuda")
    h0 = {k: d(v) for k, v in h.items()}
    h1 = {k: torch.zeros_like(v) for k, v in h0.items()}
    law.evaluate_from(0.0, 0.5, gd, s0, s1, t, h0, h1)  # warm-up outside capture (creates handles)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        law.evaluate_from(0.0, 0.5, gd, s0, s1, t, h0, h1)
    for trial in range(3):
        g2 = g * (1.0 + 0.1 * trial)
        gd.copy_(d(g2))                # new Newton iterate, same buffers
        s1.zero_(), t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        ref = oracle_run(kind, p, 0.5, g2, s, h)
        assert rel_err(s1.cpu().numpy(), ref[0]) <= 1e-6
        assert rel_err(t.cpu().numpy(), ref[1]) <= 1e-6
        for k in h:
            assert rel_err(h1[k].cpu().numpy(), ref[2][k]) <= 1e-6
