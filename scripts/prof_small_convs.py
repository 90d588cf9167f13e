"""Stand-alone timing of the loss evaluations outside the tiled i8 kernels: the 1^3 convs (exact-integer short-K kernel and
f32 path), the first conv (4->32 3^3/s2) and the classifier (32->3 1^3) at the BraTS shapes of configs[1]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops, make_geom
dev = "cuda:0"; ops = get_ops(dev); N = 16
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP

for (c1, c2, S) in ((32, 64, 32), (64, 32, 32), (64, 128, 16), (128, 64, 16), (128, 256, 8), (256, 128, 8)):
    g = torch.Generator().manual_seed(c1)
    x = torch.relu(torch.randn(N, S, S, S, c1, generator=g)).to(dev)
    y = torch.randn(N, S, S, S, c2, generator=g).to(dev)
    w = (torch.randn(c2, c1, 1, 1, 1, generator=g) * 0.05).to(dev); b = torch.zeros(c2, device=dev)
    geom = make_geom((N, c1, S, S, S), c2, 1, 1, 0)
    sq = torch.zeros(2, dtype=torch.float64, device=dev)
    a_act, _, st_a = ops.fit_scale(x, 4, 0.0, 1.0)
    xq, _, xidx = ops.quant_dequant_f64path(x, st_a, 4, 0.0, 1.0, want_idx=True)
    alpha = torch.tensor(a_act, dtype=torch.float32, device=dev)
    dual, v, G = torch.zeros_like(w), torch.empty_like(w), torch.empty_like(w)
    Gq = torch.empty(w.shape, dtype=torch.int8, device=dev)
    st_w = ops.new_fp_state()
    ops.weight_fixed_point(w, dual, v, 4, st_w)
    ops.admm_project_dual(v, w, st_w, 4, G, dual, 1.0, Gq)
    vox = N * S ** 3
    if ops.conv_i8s_supported(geom, 4, 4):
        ops.conv_step_i8s(xidx, Gq, b, geom, y, alpha, 4, st_w, 4, sq, True)
        ms = timeit(lambda: ops.conv_step_i8s(xidx, Gq, b, geom, y, alpha, 4, st_w, 4, sq, False))
        byts = vox * (c1 + 4.0 * c2)
        print(f"i8s {c1}->{c2} 1^3 @{S}^3: {ms * 1e3:7.1f} us  {byts / ms / 1e6:6.0f} GB/s ({byts / ms / 1e6 / 80:.1f}% of 8 TB/s)  loss {sq[0].item():.6e}")
    ms = timeit(lambda: ops.conv_step(xq, G, b, geom, y, None, sqerr=sq))
    byts = vox * 4.0 * (c1 + c2)
    print(f"f32 {c1}->{c2} 1^3 @{S}^3: {ms * 1e3:7.1f} us  {byts / ms / 1e6:6.0f} GB/s ({byts / ms / 1e6 / 80:.1f}% of 8 TB/s)  loss {sq[0].item():.6e}")

for (c1, c2, k, s, p, S) in ((4, 32, 3, 2, 1, 128), (32, 3, 1, 1, 0, 64)):
    g = torch.Generator().manual_seed(c1)
    x = torch.randn(N, S, S, S, c1, generator=g).to(dev)
    geom = make_geom((N, c1, S, S, S), c2, k, s, p)
    od, oh, ow = geom.out_dims()
    y = torch.randn(N, od, oh, ow, c2, generator=g).to(dev)
    w = (torch.randn(c2, c1, k, k, k, generator=g) * 0.1).to(dev); b = torch.zeros(c2, device=dev)
    sq = torch.zeros(2, dtype=torch.float64, device=dev)
    ms = timeit(lambda: ops.conv_step(x, w, b, geom, y, None, sqerr=sq))
    byts = 4.0 * (x.numel() + y.numel())
    print(f"f32 {c1}->{c2} k{k} s{s} @{S}^3: {ms * 1e3:7.1f} us  {byts / ms / 1e6:6.0f} GB/s ({byts / ms / 1e6 / 80:.1f}% of 8 TB/s)  loss {sq[0].item():.6e}")
