"""Offline evaluation of margin policies for effq_fixed_point_traj: the oracle's weight fixed points over the ADMM
iterations of a layer (all iterates recorded), replayed against 'bracket = last call's iterate +- eps' with several rules
for eps.  Prints miss rate (iterates outside their bracket) and mean eps.  Design aid, not a test."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import effq_oracle as O

c1 = int(sys.argv[1]) if len(sys.argv) > 1 else 32
c2 = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L = 4
SLOTS = 8
gen = torch.Generator().manual_seed(5)
w = torch.randn(c2, c1, 3, 3, 3, generator=gen) * (2.0 / (27 * c1)) ** 0.5
b = torch.randn(c2, generator=gen) * 0.05
x_fp = torch.relu(torch.randn(2, c1, S, S, S, generator=gen))
y = F.conv3d(x_fp, w, b, 1, 1)
x = torch.relu(x_fp + 0.05 * torch.randn(x_fp.shape, generator=gen))

calls = []


def traj(v, levels, lo, hi, tol=1e-5):
    vv = v.double().flatten()
    a = vv.abs().mean().item()
    its = [a]
    d = (hi - lo) / (levels - 1)
    for _ in range(100 * levels):
        bq = torch.round((torch.clamp(vv / a, lo, hi) - lo) / d) * d + lo
        a_new = ((bq * vv).sum() / (bq * bq).sum()).item()
        its.append(a_new)
        if abs(a_new - a) <= tol:
            break
        a = a_new
    return its


orig = O.fit_scale


def spy(v, levels, lo=-1.0, hi=1.0, *a, **k):
    fit = orig(v, levels, lo, hi, *a, **k)
    if lo == -1.0:
        t = traj(v, levels, lo, hi)
        assert len(t) - 1 == fit.iters and abs(t[-1] - fit.alpha) <= 1e-9 * fit.alpha, (len(t) - 1, fit.iters)
        calls.append(t)
    return fit


O.fit_scale = spy
O.calibrate_layer(x, y, w, b, 1, 1, qlvl_w=L, qlvl_act=4)
print(f"{len(calls)} calls, {sum(len(t) - 1 for t in calls)} iterates")


def slots_of(t):
    """classification scales t[0..iters-1] + final t[-1] -> [(lo, hi)] per slot"""
    iters = len(t) - 1
    K = min(iters, SLOTS)
    out = []
    for j in range(K):
        if j < K - 1:
            out.append((t[j], t[j]))
        else:
            tail = t[j:]
            out.append((min(tail), max(tail)))
    return out


def simulate(rule, name):
    pred, eps = None, None
    miss = tot = 0
    eps_sum = eps_n = 0
    miss_by_call = []
    for k, t in enumerate(calls):
        iters = len(t) - 1
        m = 0
        if pred is not None:
            K = len(pred)
            for i in range(iters):
                j = min(i, K - 1)
                lo_, hi_ = pred[j][0] * (1 - eps[j]), pred[j][1] * (1 + eps[j])
                tot += 1
                if not (lo_ <= t[i] <= hi_):
                    m += 1
            eps_sum += sum(eps)
            eps_n += len(eps)
        miss += m
        miss_by_call.append(m)
        new = slots_of(t)
        new_eps = []
        for j, (l, h) in enumerate(new):
            if pred is not None and j < len(pred):
                drift = max(abs(l - pred[j][0]), abs(h - pred[j][1])) / abs(h)
                new_eps.append(rule(drift, eps[j], k))
            else:
                new_eps.append(0.01)
        pred, eps = new, new_eps
    worst = [i for i, m in enumerate(miss_by_call) if m > 0]
    print(f"{name:34s} miss {miss:4d} / {tot} = {100 * miss / max(tot, 1):5.2f} %   mean eps {eps_sum / max(eps_n, 1):.2e}   calls with a miss: {len(worst)} {worst[:24]}")


clamp = lambda e: min(max(e, 1e-4), 0.03)
simulate(lambda d, e, k: clamp(3 * d), "3 x drift")
simulate(lambda d, e, k: clamp(max(3 * d, 0.7 * e)), "max(3 drift, 0.7 eps)")
simulate(lambda d, e, k: clamp(max(4 * d, 0.85 * e)), "max(4 drift, 0.85 eps)")
simulate(lambda d, e, k: clamp(max(6 * d, 0.9 * e)), "max(6 drift, 0.9 eps)")
simulate(lambda d, e, k: clamp(max(8 * d, 0.93 * e)), "max(8 drift, 0.93 eps)")
simulate(lambda d, e, k: min(max(max(4 * d, 0.85 * e), 1e-3), 0.03), "max(4 drift, 0.85 eps), floor 1e-3")
simulate(lambda d, e, k: min(max(max(6 * d, 0.9 * e), 2e-3), 0.05), "max(6 drift, 0.9 eps), floor 2e-3")
