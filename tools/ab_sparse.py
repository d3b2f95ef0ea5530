#!/usr/bin/env python3
"""In-process A/B: full vs sparse trial history for the three VonMises3D workloads."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
n = 100_000_000
for wl in ("von_mises_elastic", "von_mises_mixed", "von_mises_plastic"):
    kind, scale, _, _ = bench.WORKLOADS[wl]
    law, _ = bench.make_law(kind)
    grad_array, s0, h0 = bench.synth_inputs(kind, scale, n, 1, dev)
    t = torch.empty(36 * n, dtype=torch.float64, device=dev)
    gw = grad_array()
    law.evaluate(0, 2.0, gw, s0, t, h0)
    del gw
    g = grad_array()
    s1 = torch.empty_like(s0)
    h1 = {k: v.clone() for k, v in h0.items()}
    mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
    res = {"full": [], "sparse": []}
    for rnd in range(4):
        for mode in ("full", "sparse"):
            hm = mask if mode == "sparse" else None
            for _ in range(2):
                law.evaluate_from(0, 2.0, g, s0, s1, t, h0, h1, history_mask=hm)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in ev:
                a.record(); law.evaluate_from(0, 2.0, g, s0, s1, t, h0, h1, history_mask=hm); b.record()
            torch.cuda.synchronize()
            res[mode].append(sum(a.elapsed_time(b) for a, b in ev) / len(ev))
    st = law.device_stats()
    print(wl, f"plastic {st.n_plastic / n:.3f}", "full %.3f ms" % sorted(res["full"])[2], "sparse %.3f ms" % sorted(res["sparse"])[2], flush=True)
    del g, t, s0, s1, h0, h1, mask, law
    torch.cuda.empty_cache()
