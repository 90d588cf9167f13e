"""cProfile of the host side of one small layer's ADMM loop (where do the ~90 us per iteration go?)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.qconv import EfficientQConvHIP
dev = "cuda:0"
gen = torch.Generator().manual_seed(1)
c1, c2, S, N = 64, 128, 16, 16
conv = EfficientQConvHIP(c1, c2, 1, 1, 0, 1, 1, True, q_weight=True, qlvl=4, q_act=True, qlvl_act=4)
with torch.no_grad():
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * 0.1)
    conv.bias.copy_(torch.randn(c2, generator=gen) * 0.1)
x = torch.relu(torch.randn(N, c1, S, S, S, generator=gen))
y = torch.nn.functional.conv3d(x, conv.weight.data, conv.bias.data)
conv.output_fp, conv.name, conv.layer_loss = y.to(dev), "l", []
conv.to(dev); conv.set_quantizing()
xd = x.to(dev)
with torch.no_grad():
    conv(xd); conv._act_inited = False
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    conv.set_quantizing(); conv(xd); torch.cuda.synchronize()
    pr.disable()
print(conv.last_trace["host_enqueue_s"], conv.last_trace["admm_loop_s"])
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
