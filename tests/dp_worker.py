"""World-size-2 gloo worker for test_host_cpu.py: each rank calibrates on ITS shard of the
calibration volumes with the product's host logic (oracle CPU backend); rank 0 also runs the
unsharded calibration for comparison."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import cpu_backend  # noqa: E402
import efficientq_amd.qconv as Q  # noqa: E402
from efficientq_amd import calibrate as K, config as Cf, synth  # noqa: E402


def build():
    net = dict(Cf.TINY_NET, width="4,8,4")
    args = Cf.make_args(net, 4, 4)
    QConv, _, kwQ = Cf.get_conv_class(args)
    model = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
    synth.randomise_network(model, 3)
    model.eval()
    K.search_fold_and_remove_bn(model)
    K.set_name(model)
    return args, model


def run(vols):
    args, model = build()
    res = K.calibrate_model(model, vols, "lits", args.init_stride)
    loss = [float(l.split(":")[1]) for l in res["layer_loss"]]
    return dict(sd={k: v.clone() for k, v in model.state_dict().items()}, loss=loss, nums=res["nums"])


def main():
    out = sys.argv[1]
    torch.set_num_threads(2)
    ops = cpu_backend.OracleOps()
    Q.get_ops = lambda device: ops
    vols = torch.randn(2, 1, 16, 16, 16, generator=torch.Generator().manual_seed(5))
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    assert dist.get_world_size() == 2
    # the RCCL unique id travels through the rendezvous store (efficientq_amd/rccl.py): 128 bytes with NULs inside, twice
    from efficientq_amd import rccl
    for rep in range(2):
        want = bytes((7 * i + rep) % 256 if i % 5 else 0 for i in range(rccl.NCCL_UNIQUE_ID_BYTES))
        got = rccl.exchange_unique_id(want if rank == 0 else None, rank, 2, None)
        assert got == want, (rank, rep)
    sub = dist.new_group([0, 1])
    got = rccl.exchange_unique_id(want if rank == 0 else None, rank, 2, sub)
    assert got == want
    assert rccl._all_ranks_ok(True, None) is True
    assert rccl._all_ranks_ok(rank == 0, None) is False           # one rank failed: every rank learns it
    r = run(vols[rank:rank + 1])          # shard: one volume per rank
    torch.save(r, f"{out}_rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        torch.save(run(vols), f"{out}_single.pt")   # no process group => no reduction


if __name__ == "__main__":
    main()
