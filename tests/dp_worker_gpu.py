"""World-size-2 worker on ONE MI355X (gloo rendezvous, host-staged reductions): each rank calibrates its
shard of the volumes through the real HIP path; rank 0 also runs the unsharded calibration."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientq_amd import calibrate as K, config as Cf, synth  # noqa: E402

DEV = "cuda:0"


def run(vols):
    net = dict(Cf.TINY_NET, width="32,32,32")       # 32-channel layers: exercises the 3^3 fast path and the i8 path
    args = Cf.make_args(net, 4, 4)
    QConv, _, kwQ = Cf.get_conv_class(args)
    model = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
    synth.randomise_network(model, 3)
    model.eval()
    K.search_fold_and_remove_bn(model)
    model.to(DEV)
    K.set_name(model)
    res = K.calibrate_model(model, vols.to(DEV), "lits", args.init_stride)
    loss = [float(l.split(":")[1]) for l in res["layer_loss"]]
    return dict(sd={k: v.cpu().clone() for k, v in model.state_dict().items()}, loss=loss, nums=res["nums"])


def main():
    out = sys.argv[1]
    vols = torch.randn(2, 1, 16, 16, 16, generator=torch.Generator().manual_seed(5))
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    r = run(vols[rank:rank + 1])
    torch.save(r, f"{out}_rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        torch.save(run(vols), f"{out}_single.pt")


if __name__ == "__main__":
    main()
