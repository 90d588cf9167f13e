"""Time effq_spd_inverse for a few system sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops
dev = "cuda:0"; ops = get_ops(dev)
for n in ((865, 3457, 6913) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1:])):
    g = torch.Generator().manual_seed(n)
    X = torch.randn(n, 2 * n + 7, generator=g).to(dev)
    A0 = (2 * X @ X.T).contiguous()
    out = ops.spd_inverse(A0, True, 30.0, 3.0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.spd_inverse(A0, True, 30.0, 3.0, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"spd_inverse n={n}: {ms:.2f} ms  {1.0 * n ** 3 / ms / 1e9:.2f} TFLOP/s fp64, n^3 flop: symmetric sweep ({1.0 * n ** 3 / ms / 1e9 / 78.6 * 100:.1f}% of f64 MFMA peak)")
