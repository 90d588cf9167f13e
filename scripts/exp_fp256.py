import os, sys
sys.path.insert(0, "/root/repo")
import torch
from efficientq_amd.hip_ops import get_ops
dev = "cuda:0"; ops = get_ops(dev)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, L in ((3456, 256), (96, 256), (864, 256), (6912, 256), (3456, 64), (3456, 4), (8192, 4)):
    g = torch.Generator().manual_seed(n)
    W = (torch.randn(n, generator=g) * 0.05).to(dev); dual = torch.zeros_like(W); v = torch.empty_like(W)
    st = ops.new_fp_state()
    t = timeit(lambda: ops.weight_fixed_point(W, dual, v, L, st, 16))
    its = ops.read_fp_state(st)[1]
    print(f"n={n} L={L}: {t:.1f} us, {its} its, {t/its:.2f} us/it")
