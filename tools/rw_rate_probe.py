import json, sys, os, torch
sys.path.insert(0, os.getcwd())
from fenics_constitutive_amd import hostio
n = (12 << 30) // 8
a = torch.ones(n, dtype=torch.float64, device="cuda"); b = torch.empty(n, dtype=torch.float64, device="cuda")
def t(fn, reps=5):
    ev=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    fn()
    for x,y in ev:
        x.record(); fn(); y.record()
    torch.cuda.synchronize()
    return min(x.elapsed_time(y) for x,y in ev)*1e-3
print(json.dumps({"fill_TBs": round(8*n/t(lambda: b.fill_(2.0))/1e12,2), "zero_TBs": round(8*n/t(lambda: b.zero_())/1e12,2),
  "sum_read_TBs": round(8*n/t(lambda: a.sum())/1e12,2), "copy_TBs(r+w)": round(16*n/t(lambda: hostio.copy_device(b,a))/1e12,2),
  "torch_copy_TBs": round(16*n/t(lambda: b.copy_(a))/1e12,2)}))
