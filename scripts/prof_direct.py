"""Time the loss-only conv of the first conv (4->32, 3^3/s2) and the classifier (32->3, 1^3) at BraTS size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops, make_geom
dev = "cuda:0"; ops = get_ops(dev); N = 16
for (c1, c2, k, s, p, S) in ((4, 32, 3, 2, 1, 128), (32, 3, 1, 1, 0, 64)):
    g = torch.Generator().manual_seed(c1)
    x = torch.randn(N, S, S, S, c1, generator=g).to(dev)
    geom = make_geom((N, c1, S, S, S), c2, k, s, p)
    od, oh, ow = geom.out_dims()
    y = torch.randn(N, od, oh, ow, c2, generator=g).to(dev)
    w = (torch.randn(c2, c1, k, k, k, generator=g) * 0.1).to(dev); b = torch.zeros(c2, device=dev)
    sq = torch.zeros(2, dtype=torch.float64, device=dev)
    ops.conv_step(x, w, b, geom, y, None, sqerr=sq); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_step(x, w, b, geom, y, None, sqerr=sq)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    byts = 4.0 * (x.numel() + y.numel()); fl = 2.0 * c1 * c2 * k ** 3 * y.numel() / c2
    print(f"conv {c1}->{c2} k{k} s{s}: {ms * 1e3:.1f} us  {byts / ms / 1e6:.0f} GB/s  {fl / ms / 1e9:.1f} TFLOP/s  loss {sq[0].item():.6e}")
