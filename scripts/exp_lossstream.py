"""What do the small launches around the loss conv cost on its stream? conv alone vs conv + keep_best per iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops, make_geom
dev = "cuda:0"; ops = get_ops(dev)
for c, S in ((32, 64), (64, 32)):
    g = torch.Generator().manual_seed(0); N = 16
    geom = make_geom((N, c, S, S, S), c, 3, 1, 1)
    xidx = torch.randint(0, 4, (N, S, S, S, c), generator=g).to(torch.uint8).to(dev)
    y = torch.randn(N, S, S, S, c, generator=g).to(dev)
    Gq = torch.randint(-3, 4, (c, c, 3, 3, 3), generator=g).to(torch.int8).to(dev)
    G = Gq.float(); bG = torch.empty_like(G); b = torch.zeros(c, device=dev); bb = torch.empty_like(b)
    st = ops.new_fp_state(); st[0] = 0.05
    sq = torch.zeros(2, dtype=torch.float64, device=dev); best = torch.zeros(4, dtype=torch.float64, device=dev)
    alpha = torch.tensor(0.8, device=dev)
    def timeit(fn, reps=200):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps): fn(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t_conv = timeit(lambda i=0: ops.conv_step_i8(xidx, Gq, b, geom, y, alpha, 4, st, 4, sq))
    def both(i=1):
        ops.conv_step_i8(xidx, Gq, b, geom, y, alpha, 4, st, 4, sq)
        ops.admm_keep_best(sq, best, i + 1, G, b, bG, bb)
    t_both = timeit(both)
    print(f"c={c}: conv call {t_conv:.1f} us, conv + keep_best {t_both:.1f} us per iteration")
