import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import fenics_constitutive_amd as fc
from fenics_constitutive_amd.problem import ResidentProblemState, rows_of_cells
VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
os.environ["FCAMD_SMALL_CALL_WARNING"] = "0"
n_laws = int(sys.argv[1]); per_law = 10000; q = 4
use_stream = len(sys.argv) > 2
n_cells = n_laws * per_law // q
rng = np.random.default_rng(1)
owner = rng.permutation(np.arange(n_cells) % n_laws)
rows = [rows_of_cells(np.flatnonzero(owner == k), q) for k in range(n_laws)]
laws = [fc.VonMises3D(VM_P) for _ in range(n_laws)]
st = ResidentProblemState(list(zip(laws, rows)), q * n_cells, del_t=1.0, batch_launches=True, placement="torch")
gen = torch.Generator(device="cuda").manual_seed(3)
grads = [torch.randn(9 * r.size, dtype=torch.float64, device="cuda", generator=gen) * 2e-3 for r in rows]
s = torch.cuda.Stream() if use_stream else torch.cuda.current_stream()
with torch.cuda.stream(s):
    for i in range(6):
        print("evaluate", i, flush=True)
        st.evaluate(grads)
        torch.cuda.synchronize()
        print("done", i, flush=True)
        if i == 3:
            st.update(); print("updated", flush=True)
print("ok")
