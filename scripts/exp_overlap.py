"""Experiment: does the per-iteration loss conv overlap with the prox/fixed-point chain on a second stream?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops, make_geom
dev = "cuda:0"; ops = get_ops(dev)
c, S, N = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 64, 16
mask_chain = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # every k-th CU for the chain stream (0 = no masks)
g = torch.Generator().manual_seed(0)
geom = make_geom((N, c, S, S, S), c, 3, 1, 1)
xidx = torch.randint(0, 4, (N, S, S, S, c), generator=g).to(torch.uint8).to(dev)
y = torch.randn(N, S, S, S, c, generator=g).to(dev)
n = 27 * c + 1
W0 = (torch.randn(c, 27 * c, generator=g) * 0.05).to(dev)
b0 = torch.zeros(c, device=dev)
B0 = torch.randn(c, n, generator=g).to(dev)
lda = ops.lib.effq_ainv_ld(n)
Ainv = torch.zeros(n, lda, device=dev); Ainv[:, :n] = torch.eye(n, device=dev) * 1e-3
G = W0.clone(); dual = torch.zeros_like(W0); wstar = torch.empty_like(W0); v = torch.empty_like(W0)
bstar = torch.zeros(c, device=dev); st = ops.new_fp_state(); Gq = torch.zeros(W0.shape, dtype=torch.int8, device=dev)
sq = torch.zeros(2, dtype=torch.float64, device=dev); alpha = torch.tensor(0.8, device=dev)

def chain():
    ops.prox_solve(B0, Ainv, W0, b0, G, dual, 10.0, 1.0, wstar, bstar)
    ops.weight_fixed_point(wstar, dual, v, 4, st, 16)
    ops.admm_project_dual(v, wstar, st, 4, G, dual, 1.0, Gq)

def loss():
    ops.conv_step_i8(xidx, Gq, bstar, geom, y, alpha, 4, st, 4, sq)

def make_stream(sel):
    if mask_chain < 0:      # stream priorities: chain high, loss low
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        return torch.cuda.Stream(dev, priority=(hi if sel == "chain" else lo))
    if not mask_chain:
        return torch.cuda.Stream(dev)
    hip = C.CDLL("libamdhip64.so")
    words = (C.c_uint32 * 8)()
    for cu in range(256):
        on = (cu % mask_chain == 0) if sel == "chain" else (cu % mask_chain != 0)
        if on: words[cu // 32] |= (1 << (cu % 32))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)

sA, sB = make_stream("loss"), make_stream("chain")
def timeit(fa, fb, reps=200):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
    for _ in range(reps):
        if fa:
            with torch.cuda.stream(sA): fa()
        if fb:
            with torch.cuda.stream(sB): fb()
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
chain(); loss(); torch.cuda.synchronize()
for _ in range(2):
    ta, tb, tab = timeit(loss, None), timeit(None, chain), timeit(loss, chain)
print(f"c={c} S={S} mask={mask_chain}: loss conv alone {ta:.1f} us, chain alone {tb:.1f} us, both streams {tab:.1f} us per iteration (serial would be {ta + tb:.1f})")
