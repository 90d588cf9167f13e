"""Phase timestamps of the bucketed fixed point (library built with -DEFFQ_TRACE)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops
ops = get_ops("cuda:0")
ops.lib.effq_fpb_trace_read.argtypes = [C.c_void_p, C.c_int]
gen = torch.Generator().manual_seed(0)
for n, L in [(27648, 4), (2048, 4), (3456, 256), (27648, 16)]:
    w = (torch.randn(n, generator=gen) * 0.05).cuda(); du = (torch.randn(n, generator=gen) * 0.005).cuda()
    v = torch.empty(n, device="cuda:0"); st = ops.new_fp_state()
    for _ in range(3):
        ops.fixed_point_bucket(w, du, v, L, st)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 1024)()
    ops.lib.effq_fpb_trace_read(buf, 1024)
    t = list(buf)
    _, iters, _ = ops.read_fp_state(st)
    names = ["pass0", "pass1", "scan", "pass2", "pass3+scan"]
    print(f"n={n} L={L} iters={iters}: " + "  ".join(f"{nm} {t[i+1]-t[i]}" for i, nm in enumerate(names)) + f"  build total {t[5]-t[0]} cycles")
    for it in range(min(iters, 6)):
        b = 16 + 4 * it
        print(f"   it {it}: boundaries {t[b+1]-t[b]}  barrier {t[b+2]-t[b+1]}  sums {t[b+3]-t[b+2]}  div+loop {t[b+4]-t[b+3] if it+1<iters else 0}")
    b0, b1 = 16, 16 + 4 * (iters - 1)
    print(f"   avg per iteration {(t[b1]-t[b0])/max(iters-1,1):.0f} cycles")
