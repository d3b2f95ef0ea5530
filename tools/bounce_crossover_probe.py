import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import fenics_constitutive_amd as fc
VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
law = fc.VonMises3D(VM_P); ctx = law._handle(0).ctx
for n in (250, 500, 1000, 2000, 4000, 8000):
    rng = np.random.default_rng(0)
    g, s, t = rng.normal(size=9*n)*3e-3, np.zeros(6*n), np.zeros(36*n)
    h = {"eps_n": np.zeros(6*n), "alpha": np.zeros(n)}
    row = {"n": n, "KB": round((9+6+36+6+1)*8*n/1024)}
    for name, bm in (("bounce", 1 << 30), ("lock", 0)):
        ctx.set_option("bounce_max", bm)
        best = 1e9
        for _ in range(200):
            t0 = time.perf_counter(); law.evaluate(0.0, 1.0, g, s, t, h); best = min(best, time.perf_counter() - t0)
        row[name + "_us"] = round(best*1e6, 1); row[name + "_mode"] = ctx.last_host_mode()
    print(row, flush=True)
