"""Activation scale fit: per-iteration passes (k_fp_iter) against the bracketed fixed point (effq_fp_bracket_*)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import efficientq_amd.hip_ops as H
from efficientq_amd.hip_ops import get_ops

ops = get_ops("cuda:0")
gen = torch.Generator(device="cuda:0").manual_seed(3)


def timed(x, L, bracket, reps=3):
    H.FP_BRACKET_MIN = 1 if bracket else 1 << 62
    ops.fit_scale(x, L, 0.0, 1.0, guess_iters=12 * L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        a, it, _ = ops.fit_scale(x, L, 0.0, 1.0, guess_iters=12 * L)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, a, it


for n in (1 << 27, 1 << 24, 1 << 21):
    for kind in ("relu", "bg"):
        x = torch.randn(n, device="cuda:0", generator=gen)
        if kind == "relu":
            x = torch.relu(x + 0.1)
        else:                                   # zero background: 70 % zeros, smooth positive rest
            x = torch.relu(x - 0.5) * 1.3
        for L in (4, 16):
            t_old, a0, it0 = timed(x, L, False)
            t_new, a1, it1 = timed(x, L, True)
            dg = ops.fp_bracket_diagnostics()
            print(f"n=2^{n.bit_length() - 1} {kind:5s} L={L:3d}  passes {t_old:8.3f} ms   bracket {t_new:8.3f} ms   its {it0}/{it1}  "
                  f"rel diff {abs(a1 - a0) / a0:.1e}  read {dg['visited'] / n:6.2f} n  narrowings {dg['narrowings']} escapes {dg['escapes']}",
                  flush=True)
        del x
