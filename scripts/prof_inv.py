"""Time effq_spd_inverse for a few system sizes and check it against fp64 (||A X - I||_max)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops
dev = "cuda:0"; ops = get_ops(dev)
print("EFFQ_GJ_WIDE", os.environ.get("EFFQ_GJ_WIDE"), "EFFQ_GJ_OVERLAP", os.environ.get("EFFQ_GJ_OVERLAP"))
for n in ((865, 1729, 3457, 6913) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1:])):
    g = torch.Generator().manual_seed(n)
    X = torch.randn(n, 2 * n + 7, generator=g).to(dev)
    A0 = (2 * X @ X.T).contiguous()
    del X
    out = ops.spd_inverse(A0, True, 30.0, 3.0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.spd_inverse(A0, True, 30.0, 3.0, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    A = A0.double()
    d = torch.full((n,), 33.0, dtype=torch.float64, device=dev); d[-1] = 3.0
    A += torch.diag(d)
    R = A @ out[:, :n].double() - torch.eye(n, dtype=torch.float64, device=dev)
    print(f"spd_inverse n={n}: {ms:.2f} ms  {1.0 * n ** 3 / ms / 1e9:.2f} TFLOP/s fp64, n^3 flop: symmetric sweep "
          f"({1.0 * n ** 3 / ms / 1e9 / 78.6 * 100:.1f}% of f64 MFMA peak); max|A X - I| = {R.abs().max().item():.2e}", flush=True)
    del A, R, A0, out
