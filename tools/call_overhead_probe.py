import sys, os, time, mmap, numpy as np
sys.path.insert(0, os.getcwd())
import fenics_constitutive_amd as fc
VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
def own(a):
    out = np.frombuffer(mmap.mmap(-1, max(a.nbytes, 8)), dtype=np.float64, count=a.size); out[:] = a; return out
for name, law, hist in (("vm", fc.VonMises3D(VM_P), True), ("le", fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.FULL), False)):
    n = 1000
    rng = np.random.default_rng(0)
    g, s, t = own(rng.normal(size=9*n)*3e-3), own(np.zeros(6*n)), own(np.zeros(36*n))
    h = {"eps_n": own(np.zeros(6*n)), "alpha": own(np.zeros(n))} if hist else None
    m = law._handle(0); ctx = m.ctx
    for a in [g, s, t] + (list(h.values()) if h else []): ctx.register_host_buffer(a)
    ctx.set_timing(True)
    best_py, best_c = 1e9, 1e9
    for _ in range(300):
        t0 = time.perf_counter(); law.evaluate(0.0, 1.0, g, s, t, h); dt = time.perf_counter() - t0
        best_py = min(best_py, dt); best_c = min(best_c, m.last_kernel_ms()*1e-3)
    print(name, "python total us", round(best_py*1e6,1), "C call us", round(best_c*1e6,1), "mode", ctx.last_host_mode())
