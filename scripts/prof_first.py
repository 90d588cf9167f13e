"""Time the first-layer conv step (4->32, 3^3 stride 2, 16x128^3 -> 64^3) and the classifier (32->3, 1^3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops, make_geom
dev = "cuda:0"; ops = get_ops(dev); g = torch.Generator().manual_seed(0)
def run(tag, N, c1, c2, S, k, s, p):
    So = (S + 2 * p - k) // s + 1
    x = torch.randn(N, S, S, S, c1, generator=g).to(dev)
    y = torch.randn(N, So, So, So, c2, generator=g).to(dev)
    w = (torch.randn(c2, c1, k, k, k, generator=g) * 0.1).to(dev)
    geom = make_geom((N, c1, S, S, S), c2, k, s, p)
    sq = torch.zeros(2, dtype=torch.float64, device=dev)
    ops.conv_step(x, w, None, geom, y, None, sqerr=sq); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv_step(x, w, None, geom, y, None, sqerr=sq)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = 4.0 * (x.numel() + y.numel())
    print(f"{tag}: {ms:.3f} ms  {by / ms / 1e6:.0f} GB/s algorithmic")
run("first 4->32 s2", 16, 4, 32, 128, 3, 2, 1)
run("cls 32->3 1^3", 16, 32, 3, 64, 1, 1, 0)
run("down 32->64 1^3 @32^3", 16, 32, 64, 32, 1, 1, 0)
