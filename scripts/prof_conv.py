"""Launch one conv-step shape a few times (used under rocprofv3 --kernel-trace / --pmc so traces stay small).
usage: prof_conv.py MODE [n]   MODE: f32_32 (32->32 3^3, 16x64^3, f32 MFMA) | i8_32 (same shape, exact-int)
                               | f32_128 (128->128 3^3, 16x16^3) | i8_64 | i8_128 | i8_256 (exact-int, 16x32^3 / 16^3 / 8^3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops, make_geom
mode = sys.argv[1] if len(sys.argv) > 1 else "f32_32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, C, S = {"f32_128": (16, 128, 16), "i8_64": (16, 64, 32), "i8_128": (16, 128, 16), "i8_256": (16, 256, 8)}.get(mode, (16, 32, 64))
dev = "cuda:0"
ops = get_ops(dev)
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(N, S, S, S, C, generator=g)).to(dev)                 # NDHWC
y = torch.randn(N, S, S, S, C, generator=g).to(dev)
w = (torch.randn(C, C, 3, 3, 3, generator=g) * 0.03).to(dev)
b = torch.zeros(C, device=dev)
geom = make_geom((N, C, S, S, S), C, 3, 1, 1)
sq = torch.zeros(2, dtype=torch.float64, device=dev)
a_act, _, st_a = ops.fit_scale(x, 4, 0.0, 1.0)
xq, _, xidx = ops.quant_dequant_f64path(x, st_a, 4, 0.0, 1.0, want_idx=True)
alpha = torch.tensor(a_act, dtype=torch.float32, device=dev)
dual, v, G = torch.zeros_like(w), torch.empty_like(w), torch.empty_like(w)
Gq = torch.empty(w.shape, dtype=torch.int8, device=dev)
st_w = ops.new_fp_state()
ops.weight_fixed_point(w, dual, v, 4, st_w)
ops.admm_project_dual(v, w, st_w, 4, G, dual, 1.0, Gq)
def step():
    if mode.startswith("i8"):
        ops.conv_step_i8(xidx, Gq, b, geom, y, alpha, 4, st_w, 4, sq)
    else:
        ops.conv_step(xq, G, b, geom, y, None, sqerr=sq)
step(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * C * C * 27 * N * S ** 3
by = (4.0 * C + (1.0 if mode.startswith("i8") else 4.0) * C) * N * S ** 3 * (2 if mode == "i8_32p" else 1)
fl *= (2 if mode == "i8_32p" else 1)
print(f"{mode}: sqerr {sq.tolist()[0]:.6e}  avg {ms:.4f} ms  {fl / ms / 1e9:.2f} TFLOP/s ({fl / ms / 1e9 / 157.3 * 100:.1f}% f32 MFMA)  "
      f"{by / ms / 1e6:.0f} GB/s algorithmic ({by / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s)")
