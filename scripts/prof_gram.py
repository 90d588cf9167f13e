"""Time effq_gram_accum on the dominant shape (32 ch, 3^3, 16 x 64^3 voxels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops, make_geom
N, C, S = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 32, 64
dev = "cuda:0"; ops = get_ops(dev); g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(N, S, S, S, C, generator=g)).to(dev)
y = torch.randn(N, S, S, S, C, generator=g).to(dev)
att = torch.randint(1, 3, (N, S, S, S), generator=g).float().to(dev)
geom = make_geom((N, C, S, S, S), C, 3, 1, 1)
A0, B0 = ops.gram(x, att, y, geom, True); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): ops.gram(x, att, y, geom, True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
n = C * 27 + 1; V = N * S ** 3
fl = 2.0 * n * n * V + 2.0 * C * n * V
print(f"gram n={n} V={V}: {ms:.2f} ms  {fl / ms / 1e9:.1f} TFLOP/s algorithmic ({fl / ms / 1e9 / 157.3 * 100:.1f}% of f32 MFMA peak)")
