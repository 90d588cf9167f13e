"""Time effq_gram_f64 on the first conv / classifier shapes of the BraTS net (16 volumes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops, make_geom
dev = "cuda:0"; ops = get_ops(dev)
for (c1, c2, k, s, p, S) in ((4, 32, 3, 2, 1, 128), (32, 3, 1, 1, 0, 64)):
    N = 16
    x = torch.randn(N, S, S, S, c1, device=dev)
    geom = make_geom((N, c1, S, S, S), c2, k, s, p)
    od, oh, ow = geom.out_dims()
    y = torch.randn(N, od, oh, ow, c2, device=dev)
    ops.gram_f64(x, y, geom, True); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.gram_f64(x, y, geom, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    n = c1 * k ** 3 + 1
    V = N * od * oh * ow
    print(f"gram_f64 {c1}->{c2} k{k}: {ms:.2f} ms  ({(n * n + 2.0 * c2 * n) * V / ms / 1e9:.2f} TFLOP/s fp64 algorithmic, upper triangle)")
