"""Diagnostic (GPU): conv0 of the BraTS net in the network context vs the oracle on the very same tensors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import effq_oracle as O
from efficientq_amd import calibrate as K, config as Cf, synth
from efficientq_amd.qconv import PTQConv
N, S = int(sys.argv[1]), int(sys.argv[2])
args = Cf.make_args(Cf.BRATS_NET, 4, 4)
QConv, _, kwQ = Cf.get_conv_class(args)
model = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
synth.randomise_network(model, 0)
model.eval(); K.search_fold_and_remove_bn(model); K.set_name(model)
name0, c0 = [(n, m) for n, m in model.named_modules() if isinstance(m, PTQConv)][0]
w0, b0 = c0.weight.data.clone(), c0.bias.data.clone()
print("first layer", name0, tuple(w0.shape), "stride", c0.stride, "pad", c0.padding, "qlvl", c0.qlvl_w, c0.qlvl_act, "w std", w0.std().item(), "b", b0.abs().max().item())
vols = synth.calib_batch("brats", range(N), S)
model.to("cuda:0")
c0.lwq_trace = True
res = K.calibrate_model(model, vols.to("cuda:0"), "brats", args.init_stride)
tr = c0.last_trace
y = torch.nn.functional.conv3d(vols, w0, b0, c0.stride, c0.padding)
pyr = [m.cpu() for m in res["pyramid"]]
print("pyramid shapes", [tuple(m.shape) for m in pyr], "att values", [torch.unique(m).tolist()[:6] for m in pyr])
want = O.calibrate_layer(vols, y, w0, b0, c0.stride, c0.padding, qlvl_w=256, qlvl_act=-1, q_act=False, mask_pyramid=pyr)
h, rh = np.array(tr["loss_history"]), np.array(want.loss_history)
print("layer_loss hip", tr["layer_loss"], "oracle", want.layer_loss, "y var", y.var().item())
print("rho_scale hip", tr["rho_scale"], "oracle", want.rho_scale)
print("hist hip  ", h[:4], h[-2:], "best", tr["best_iter"], h.min(), "final_mse", tr["final_mse"])
print("hist orcl ", rh[:4], rh[-2:], "best", want.best_iter, rh.min())
