"""Time effq_gram_accum_i8 (and the fp32 Gram beside it) on the BraTS layer shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops, make_geom
dev = "cuda:0"; ops = get_ops(dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for C, S in ((32, 64), (64, 32), (128, 16), (256, 8)):
    g = torch.Generator().manual_seed(C)
    idx = torch.randint(0, 4, (N, S, S, S, C), generator=g).to(torch.uint8).to(dev)
    alpha = torch.tensor(0.8, device=dev)
    xhat = (idx.float() * (alpha / 3)).contiguous()
    y = torch.randn(N, S, S, S, C, generator=g).to(dev)
    att = torch.randint(1, 3, (N, S, S, S), generator=g).float().to(dev)
    geom = make_geom((N, C, S, S, S), C, 3, 1, 1)
    t = time.time(); cls = ops.att_classes(att); torch.cuda.synchronize(); tcls = time.time() - t
    n = C * 27 + 1; V = N * S ** 3
    ops_n = 2.0 * n * n * V + 2.0 * C * n * V
    for name, fn in (("i8", lambda: ops.gram_i8(idx, cls, y, geom, True, alpha, 4)),
                     ("f32", lambda: ops.gram(xhat, att, y, geom, True))):
        A0, B0 = fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"gram {name:3s} C={C} n={n} V={V}: {ms:8.3f} ms  {ops_n / ms / 1e9:8.1f} T(FL)OP/s algorithmic"
              + (f"  (class list build {tcls * 1e3:.1f} ms)" if name == "i8" else ""), flush=True)
