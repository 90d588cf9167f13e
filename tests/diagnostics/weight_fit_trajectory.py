"""How the weight-scale fixed point behaves ACROSS the ADMM iterations of a layer (oracle run on the CPU): per call the
start alpha0 = mean|v|, the fixed point alpha*, the iteration count, the share of weights whose level differs between the
two ends of the trajectory, and how far alpha0 / alpha* moved since the previous ADMM iteration.  Basis for predicting a
bracket for the next call from the last one (design aid; not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import effq_oracle as O

c1 = int(sys.argv[1]) if len(sys.argv) > 1 else 32
c2 = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 12
L = int(sys.argv[4]) if len(sys.argv) > 4 else 4
gen = torch.Generator().manual_seed(5)
w = torch.randn(c2, c1, 3, 3, 3, generator=gen) * (2.0 / (27 * c1)) ** 0.5
b = torch.randn(c2, generator=gen) * 0.05
x_fp = torch.relu(torch.randn(2, c1, S, S, S, generator=gen))
y = F.conv3d(x_fp, w, b, 1, 1)
x = torch.relu(x_fp + 0.05 * torch.randn(x_fp.shape, generator=gen))

rec = []
orig = O.fit_scale


def spy(v, levels, lo=-1.0, hi=1.0, *a, **k):
    fit = orig(v, levels, lo, hi, *a, **k)
    if lo == -1.0:
        vv = v.double().flatten()
        a0 = vv.abs().mean().item()
        a1 = fit.alpha
        lo_a, hi_a = min(a0, a1), max(a0, a1)
        l0 = O.quant_index(vv / lo_a, levels, lo, hi)
        l1 = O.quant_index(vv / hi_a, levels, lo, hi)
        rec.append((a0, a1, fit.iters, (l0 != l1).double().mean().item()))
    return fit


O.fit_scale = spy
res = O.calibrate_layer(x, y, w, b, 1, 1, qlvl_w=L, qlvl_act=4)
print(f"{c1}->{c2} 3^3, {w.numel()} weights, L={L}: best iterate {res.best_iter}")
pa0 = pa1 = None
for i, (a0, a1, it, frac) in enumerate(rec):
    d0 = abs(a0 - pa0) / pa0 if pa0 else float("nan")
    d1 = abs(a1 - pa1) / pa1 if pa1 else float("nan")
    if i < 30 or i % 10 == 0:
        print(f"admm {i:3d}  alpha0 {a0:.6f} ({d0:.1e})  alpha* {a1:.6f} ({d1:.1e})  its {it:3d}  undecided over [a0, a*]: {100 * frac:5.2f} %")
    pa0, pa1 = a0, a1
