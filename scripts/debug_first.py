"""Diagnostic (GPU): first conv of the BraTS net (4->32, 3^3 stride 2, FP input, 256 weight levels) vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import effq_oracle as O
from efficientq_amd import synth
from efficientq_amd.qconv import EfficientQConvHIP
N, S = int(sys.argv[1]), int(sys.argv[2])
use_mask = len(sys.argv) > 3 and sys.argv[3] == "mask"
vols = synth.calib_batch("brats", range(N), S)
g = torch.Generator().manual_seed(0)
w = torch.randn(32, 4, 3, 3, 3, generator=g) * (2.0 / 108) ** 0.5
b = torch.randn(32, generator=g) * 0.05
y = torch.nn.functional.conv3d(vols, w, b, 2, 1)
att = torch.randint(1, 4, (N, S // 2, S // 2, S // 2), generator=g).float() if use_mask else None
want = O.calibrate_layer(vols, y, w, b, 2, 1, qlvl_w=256, qlvl_act=-1, q_act=False, mask_pyramid=[att] if use_mask else None)
conv = EfficientQConvHIP(4, 32, 3, 2, 1, 1, 1, True, q_weight=True, qlvl=256, q_act=False, qlvl_act=-1)
conv.weight.data, conv.bias.data = w.clone(), b.clone()
conv.output_fp, conv.name, conv.layer_loss = y, "first", []
if use_mask:
    conv.mask_pyramid = [att]
conv.to("cuda:0"); conv.output_fp = conv.output_fp.to("cuda:0")
if use_mask:
    conv.mask_pyramid = [att.to("cuda:0")]
conv.set_quantizing()
with torch.no_grad():
    conv(vols.to("cuda:0"))
tr = conv.last_trace
h, rh = np.array(tr["loss_history"]), np.array(want.loss_history)
print("N", N, "S", S, "mask", use_mask, "layer_loss hip", float(conv.layer_loss[0].split(":")[1]), "oracle", want.layer_loss)
print("rho_scale hip", tr["rho_scale"], "oracle", want.rho_scale)
print("hist hip  ", h[:4], h[50:52], h[-2:], "best", tr["best_iter"], h.min())
print("hist orcl ", rh[:4], rh[50:52], rh[-2:], "best", want.best_iter, rh.min())
print("w_iters", tr["w_iters"][:4], "alpha_w", tr["alpha_w"], want.alpha_w)
