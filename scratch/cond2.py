import torch, sys, copy, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenario
from test_oracle_golden import build_oracle_step, load_golden, set_alpha
torch.set_num_threads(8)
expt, size = sys.argv[1], sys.argv[2]
inputs, golden = load_golden(expt, size)
s32 = build_oracle_step(expt, size)
o32 = scenario.run_scenario(s32, inputs, 'cpu', full=True, set_alpha=set_alpha)
s64 = build_oracle_step(expt, size).double()
in64 = {k: v.double() for k, v in inputs.items()}
torch.set_default_dtype(torch.float64)
o64 = scenario.run_scenario(s64, in64, 'cpu', full=True, set_alpha=set_alpha)
rows=[]
for k in o32:
    a, b = np.asarray(o32[k], dtype=np.float64), np.asarray(o64[k], dtype=np.float64)
    if a.ndim == 0:
        rows.append((abs(a-b)/max(abs(b),1e-30), 0.0, k)); continue
    mx = np.abs(a-b).max()/max(np.abs(b).max(),1e-30)
    l2 = np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30)
    rows.append((mx, l2, k))
rows.sort(reverse=True)
for r in rows[:25]: print('%.2e  %.2e  %s' % r)
