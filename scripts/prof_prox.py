"""Time effq_prox_solve for the BraTS system sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops
dev = "cuda:0"; ops = get_ops(dev)
for c2, n in ((32, 865), (64, 1729), (128, 3457), (256, 6913)) + (((512, 13825),) if os.environ.get('PROX_LITS') else ()):
    g = torch.Generator().manual_seed(n)
    lda = ops.lib.effq_ainv_ld(n)
    Ainv = torch.zeros(n, lda, device=dev); Ainv[:, :n] = torch.randn(n, n, generator=g).to(dev) * 1e-3
    Ainv[:, :n] = 0.5 * (Ainv[:, :n] + Ainv[:, :n].T)
    B0 = torch.randn(c2, n, generator=g).to(dev); W0 = torch.randn(c2, n - 1, generator=g).to(dev)
    b0 = torch.randn(c2, generator=g).to(dev); G = W0.clone(); dual = torch.zeros_like(W0)
    ws, bs = torch.empty_like(W0), torch.empty(c2, device=dev)
    ops.prox_solve(B0, Ainv, W0, b0, G, dual, 10.0, 1.0, ws, bs); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.prox_solve(B0, Ainv, W0, b0, G, dual, 10.0, 1.0, ws, bs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"prox c2={c2} n={n}: {ms * 1e3:.1f} us  {2.0 * c2 * n * n / ms / 1e9:.2f} TFLOP/s ({2.0 * c2 * n * n / ms / 1e9 / 157.3 * 100:.1f}% of f32 MFMA peak)")
