#!/usr/bin/env python3
"""One Newton iteration of a multi-material problem on the device: ResidentProblemState.evaluate with the laws of the form() as ONE
fcamd_evaluate_batch against one fcamd_evaluate_device_ex per law (round 4's sequence).
    python tools/batch_bench.py            -> JSON: {materials: {points per law: [batched us, sequential us, ratio]}}
Every law is VonMises3D on its own rows (cells of 4 points dealt to the laws at random), gradients device-resident, wall time per
evaluate() incl. Python and a final stream synchronise, median of 30."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.problem import ResidentProblemState, rows_of_cells  # noqa: E402

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
os.environ.setdefault("FCAMD_SMALL_CALL_WARNING", "0")
out = {}
for n_laws in [int(x) for x in os.environ.get('BATCH_BENCH_LAWS', '2,8').split(',')]:
    out[str(n_laws)] = {}
    for per_law in [int(x) for x in os.environ.get('BATCH_BENCH_SIZES', '10000,100000,1000000').split(',')]:
        q = 4
        n_cells = n_laws * per_law // q
        rng = np.random.default_rng(1)
        owner = rng.permutation(np.arange(n_cells) % n_laws)
        rows = [rows_of_cells(np.flatnonzero(owner == k), q) for k in range(n_laws)]
        res = []
        for batch in (True, False):
            laws = [fc.VonMises3D(VM_P) for _ in range(n_laws)]
            st = ResidentProblemState(list(zip(laws, rows)), q * n_cells, del_t=1.0, batch_launches=batch, placement="torch")
            gen = torch.Generator(device="cuda").manual_seed(3)
            grads = [torch.randn(9 * r.size, dtype=torch.float64, device="cuda", generator=gen) * 2e-3 for r in rows]
            for _ in range(5):
                st.evaluate(grads)
            torch.cuda.synchronize()
            ts = []
            for _ in range(30):
                t0 = time.perf_counter()
                st.evaluate(grads)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res.append(sorted(ts)[len(ts) // 2] * 1e6)
            del st
        out[str(n_laws)][str(per_law)] = [round(res[0], 1), round(res[1], 1), round(res[1] / res[0], 2)]
print(json.dumps({"columns": ["batched_us", "law_by_law_us", "ratio"], "materials": out}))
