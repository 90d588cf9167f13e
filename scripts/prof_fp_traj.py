"""Weight projection inside the ADMM loop: the older fixed points against effq_fixed_point_traj on a slowly drifting,
level-clustered tensor (microseconds per call, warm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops

ops = get_ops("cuda:0")
gen = torch.Generator().manual_seed(1)


def clustered(n, alpha=0.07, spread=0.15):
    lv = torch.tensor([-1.0, -1 / 3, 1 / 3, 1.0])[torch.randint(0, 4, (n,), generator=gen)]
    return (alpha * (lv + spread * torch.randn(n, generator=gen))).float().cuda()


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (27648, 110592, 442368, 1769472):
    for L in (4, 16):
        w = clustered(n)
        du = torch.randn(n, generator=gen).cuda() * 0.002
        v, st, pred = torch.empty_like(w), ops.new_fp_state(), ops.new_fp_pred()

        def old():
            if n <= (1 << 19):
                ops.fixed_point_bucket(w, du, v, L, st)
            else:
                ops.weight_fixed_point(w, du, v, L, st)

        def new():
            ops.fixed_point_traj(w, du, v, L, st, pred)

        t_old = timed(old)
        it_old = ops.read_fp_state(st)[1]
        t_new = timed(new)
        p = ops.read_fp_pred(pred)
        print(f"n={n:8d} L={L:3d}  old {t_old:7.1f} us   traj {t_new:7.1f} us   its {it_old}/{ops.read_fp_state(st)[1]}   "
              f"list {p['listed'] / max(p['calls'], 1):9.0f} per call, full-pass iterates {p['full_iters']}  last wg: {p['trace_us']}", flush=True)
