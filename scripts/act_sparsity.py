"""Fraction of exact zeros in the input of every quantised conv of the BraTS net (diagnostic, GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd import calibrate as K, config as Cf, synth
from efficientq_amd.qconv import PTQConv
dev = "cuda:0"
args = Cf.make_args(Cf.BRATS_NET, 4, 4)
QConv, _, kwQ = Cf.get_conv_class(args)
model = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
synth.randomise_network(model, 0)
model.eval(); K.search_fold_and_remove_bn(model); model.to(dev); K.set_name(model)
vols = synth.calib_batch("brats", range(4), 128).to(dev)
for name, q in model.named_modules():
    if isinstance(q, PTQConv):
        q.register_forward_pre_hook(lambda m, i, n=name: print(f"{n:45s} zeros {float((i[0] == 0).float().mean()):.3f}  numel {i[0].numel()}"))
K.set_fp(model)
with torch.no_grad():
    model(vols)
