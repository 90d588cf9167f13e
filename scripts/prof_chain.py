"""Time the pieces of one ADMM chain step (prox, scale fixed point, projection/dual, keep-best) per layer width."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops
dev = "cuda:0"; ops = get_ops(dev)
def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for c in (32, 64, 128, 256):
    g = torch.Generator().manual_seed(c)
    n = 27 * c + 1
    W = (torch.randn(c, 27 * c, generator=g) * (2.0 / (27 * c)) ** 0.5).to(dev)
    dual = (torch.randn(c, 27 * c, generator=g) * 0.01).to(dev)
    v = torch.empty_like(W); G = torch.empty_like(W); Gq = torch.empty(W.shape, dtype=torch.int8, device=dev)
    st = ops.new_fp_state()
    bst = torch.zeros(c, device=dev); bG = torch.empty_like(W); bb = torch.empty(c, device=dev)
    sq = torch.ones(2, dtype=torch.float64, device=dev); best = torch.zeros(4, dtype=torch.float64, device=dev)
    t_fp = timeit(lambda: ops.weight_fixed_point(W, dual, v, 4, st, 16))
    its = ops.read_fp_state(st)[1]
    d2 = dual.clone()
    t_pd = timeit(lambda: ops.admm_project_dual(v, W, st, 4, G, d2, 1.0, Gq))
    t_kb = timeit(lambda: ops.admm_keep_best(sq, best, 0, G, bst, bG, bb))
    print(f"c={c:3d} numel={W.numel():8d}: fixed point {t_fp:7.1f} us ({its} its, {t_fp / max(its, 1):.2f} us/it)  "
          f"project_dual {t_pd:6.1f} us  keep_best(copy) {t_kb:6.1f} us")
