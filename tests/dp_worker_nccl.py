"""ONE rank on RCCL (backend "nccl", world size 1): the same calibration once with every data-parallel collective issued
(EFFQ_DP_FORCE=1: statistics, activation fixed points, packed Gram system, loss history - on the streams the product
uses them on) and once without any.  An all-reduce over one rank is the identity, so the two must agree bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientq_amd import calibrate as K, config as Cf, synth  # noqa: E402

DEV = "cuda:0"


def run(vols):
    net = dict(Cf.TINY_NET, width="32,32,32")
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
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    from efficientq_amd.qconv import SumReducer
    from efficientq_amd import rccl
    os.environ["EFFQ_DP_FORCE"] = "1"
    SumReducer.calls = 0
    forced = run(vols)                                  # RCCL called directly on the kernels' stream (rccl.py)
    forced["collectives"] = SumReducer.calls
    comm = rccl.get_comm(None)
    forced["direct_calls"] = comm.calls if comm is not None else -1
    os.environ["EFFQ_RCCL_DIRECT"] = "0"
    SumReducer.calls = 0
    via_torch = run(vols)                               # the same collectives through torch.distributed
    via_torch["collectives"] = SumReducer.calls
    os.environ["EFFQ_RCCL_DIRECT"] = "1"
    os.environ["EFFQ_DP_FORCE"] = "0"
    plain = run(vols)
    rccl.close_all()
    dist.destroy_process_group()
    torch.save(dict(forced=forced, via_torch=via_torch, plain=plain), out)


if __name__ == "__main__":
    main()
