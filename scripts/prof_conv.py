"""Launch the dominant conv step (32->32, 3^3, 16 x 64^3 voxels, loss-only) a few times; used under
rocprofv3 (--kernel-trace / --pmc) so that the traces stay small."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops, make_geom
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N, C, S = 16, 32, 64
dev = "cuda:0"
ops = get_ops(dev)
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(N, S, S, S, C, generator=g)).to(dev)        # NDHWC
x = (torch.round(x.clamp(0, 1) * 3) / 3 * 0.9).contiguous()            # 4-level activations like the real path
y = torch.randn(N, S, S, S, C, generator=g).to(dev)
w = (torch.randn(C, C, 3, 3, 3, generator=g) * 0.03).to(dev)
b = torch.zeros(C, device=dev)
geom = make_geom((N, C, S, S, S), C, 3, 1, 1)
sq = torch.zeros(2, dtype=torch.float64, device=dev)
ops.conv_step(x, w, b, geom, y, None, sqerr=sq)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.conv_step(x, w, b, geom, y, None, sqerr=sq)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * C * C * 27 * N * S ** 3
print(f"sqerr {sq.tolist()}  avg {ms:.4f} ms  {fl / ms / 1e9:.2f} TFLOP/s  {fl / ms / 1e9 / 157.3 * 100:.1f}% of f32 MFMA peak")
