"""Weight-scale fixed point: bucketed single-workgroup kernel vs the all-values kernels, per layer size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops

ops = get_ops("cuda:0")
gen = torch.Generator().manual_seed(0)
CASES = [tuple(int(x) for x in c.split(':')) for c in sys.argv[1:]]
for n, L in CASES or [(96, 256), (3456, 256), (2048, 4), (8192, 4), (16384, 4), (27648, 4), (32768, 4), (27648, 16), (110592, 4), (442368, 4), (1769472, 4), (7077888, 4)]:
    w = (torch.randn(n, generator=gen) * 0.05).cuda()
    du = (torch.randn(n, generator=gen) * 0.005).cuda()
    v = torch.empty(n, device="cuda:0")
    st = ops.new_fp_state()
    res = {}
    for name, fn in (("bucket", lambda: ops.fixed_point_bucket(w, du, v, L, st)),
                     ("old", lambda: ops.weight_fixed_point(w, du, v, L, st))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 50 * 1e3, ops.read_fp_state(st))
    print(f"n={n:7d} L={L:3d}  bucket {res['bucket'][0]:7.1f} us  old {res['old'][0]:7.1f} us   iters {res['bucket'][1][1]} / {res['old'][1][1]}")
