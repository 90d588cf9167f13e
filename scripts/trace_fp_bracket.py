"""Per-iteration trace of the bracketed fixed point: alpha, bracket, list length (tuning aid)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from efficientq_amd.hip_ops import get_ops, ADMM_TOL, _ptr
from efficientq_amd._lib import check

L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 22)
kind = sys.argv[3] if len(sys.argv) > 3 else "relu"
ops = get_ops("cuda:0")
gen = torch.Generator(device="cuda:0").manual_seed(3)
x = torch.randn(n, device="cuda:0", generator=gen)
x = torch.relu(x + 0.1) if kind == "relu" else torch.relu(x - 0.5) * 1.3
lib, stream = ops.lib, ops.stream
s0 = ops.abs_sum(x)
st = ops.new_fp_state()
ws = torch.zeros(lib.effq_fp_bracket_ws_bytes(n), dtype=torch.uint8, device="cuda:0")
check(lib.effq_fp_bracket_init(_ptr(st), _ptr(s0), n, L, 1, _ptr(ws), ws.numel(), stream), "init")
prev = None
for i in range(100 * L):
    check(lib.effq_fp_bracket_run(_ptr(x), n, L, 0.0, 1.0, ADMM_TOL, 100 * L, 1, _ptr(st), _ptr(ws), stream), "run")
    a, it, done = ops.read_fp_state(st)
    f = ws[:128].view(torch.float64).cpu()
    q = ws[:128].view(torch.int64).cpu()
    step = (a - prev) if prev is not None else float("nan")
    print(f"it {it:3d} alpha {a:.9f} step {step: .3e}  src {int(q[7])} list {int(q[15]) / n:7.4f} n  bracket [{float(f[0]):.6f}, {float(f[1]):.6f}]"
          f"  plan narrow {int(q[9])} [{float(f[2]):.6f}, {float(f[3]):.6f}]  esc {int(q[12])} read {int(q[14]) / n:.2f} n")
    prev = a
    if done:
        break
