"""One bracketed fit of a 2^27-value post-ReLU tensor (for rocprofv3 --kernel-trace)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import efficientq_amd.hip_ops as H
from efficientq_amd.hip_ops import get_ops

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ops = get_ops("cuda:0")
gen = torch.Generator(device="cuda:0").manual_seed(3)
x = torch.relu(torch.randn(1 << 27, device="cuda:0", generator=gen) + 0.1)
H.FP_BRACKET_MIN = 1
for _ in range(3):
    a, it, _ = ops.fit_scale(x, L, 0.0, 1.0, guess_iters=12 * L)
torch.cuda.synchronize()
print(a, it)
