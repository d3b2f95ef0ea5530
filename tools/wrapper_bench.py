import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
dev = torch.device("cuda", 0)
import numpy as np
LAWS = {"VonMises3D": lambda: fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}),
        "MisesPlasticityLinearHardening3D": lambda: fc.MisesPlasticityLinearHardening3D(
            {k: np.array([v]) for k, v in {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0}.items()})}
lname = sys.argv[2] if len(sys.argv) > 2 else "VonMises3D"
fused = not (len(sys.argv) > 3 and sys.argv[3] == "generic")
vm = LAWS[lname]()
for name, W, gd2, sd in (("plane_strain", fc.PlaneStrainFrom3D, 4, 4), ("uniaxial_strain", fc.UniaxialStrainFrom3D, 1, 1)):
    w = W(vm)
    w.fused = fused
    gen = torch.Generator(device=dev).manual_seed(1)
    g = torch.randn(gd2 * n, dtype=torch.float64, device=dev, generator=gen)
    g.view(n, gd2).mul_(torch.pow(10.0, torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 4)[:, None])
    s0 = torch.zeros(sd * n, dtype=torch.float64, device=dev)
    s = torch.zeros_like(s0)
    t = torch.zeros(sd * sd * n, dtype=torch.float64, device=dev)
    if lname == "VonMises3D":
        h0 = {"eps_n": torch.zeros(6 * n, dtype=torch.float64, device=dev), "alpha": torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 0.02}
    else:
        h0 = {"history": torch.zeros(7 * n, dtype=torch.float64, device=dev)}
    h = {k: v.clone() for k, v in h0.items()}
    def step():
        s.copy_(s0)
        for k in h: h[k].copy_(h0[k])
        w.evaluate(0.0, 1.0, g, s, t, h)
    for _ in range(3): step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
    for a, b in ev:
        s.copy_(s0)
        for k in h: h[k].copy_(h0[k])
        a.record(); w.evaluate(0.0, 1.0, g, s, t, h); b.record()
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    print(json.dumps({"law": lname, "fused": fused, "wrapper": name, "n": n, "ms": round(ms, 3), "Gpts_s": round(n / ms / 1e6, 2), "plastic": int(vm.device_stats().n_plastic)}), flush=True)
    del w, g, s, s0, t, h, h0
    torch.cuda.empty_cache()
