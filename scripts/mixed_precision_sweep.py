"""BASELINE config 5: per-layer qlvl_w in {4,8,16} search over the BraTS net.  Replicas only: with
torchrun --nproc-per-node N each rank takes the budgets b[rank::N] (no collectives).
usage: mixed_precision_sweep.py [N_VOLS] [SIZE] [budgets, e.g. 2,2.5,3,3.5,4] [act_follows 0/1]
One JSON line per budget (and per uniform reference point) on stdout."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd import calibrate as K, config as Cf, mixed, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
budgets = [float(b) for b in (sys.argv[3] if len(sys.argv) > 3 else "2,2.5,3,3.5,4").split(",")]
act_follows = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
args = Cf.make_args(Cf.BRATS_NET, 4, 4)
QConv, _, kwQ = Cf.get_conv_class(args)


def build():
    m = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
    synth.randomise_network(m, 0)
    m.eval(); K.search_fold_and_remove_bn(m); m.to(dev); K.set_name(m)
    return m


vols = synth.calib_batch("brats", range(N), size).to(dev)
if rank == 0:
    for L in (4, 8, 16):
        u = mixed.uniform(build, vols, "brats", args.init_stride, L, act_follows)
        print(json.dumps(dict(kind="uniform", qlvl_w=L, avg_bits=u["avg_bits"], fp_vs_q_agreement=round(u["agreement"], 5),
                              output_rel_mse=u["output_error"], seconds=round(u["seconds"], 2),
                              act_follows=act_follows)), flush=True)
for r in mixed.search(build, vols, "brats", args.init_stride, budgets[rank::world], act_follows=act_follows):
    hist = {}
    for v in r["levels"].values():
        hist[v] = hist.get(v, 0) + 1
    print(json.dumps(dict(kind="searched", rank=rank, budget_bits=r["budget_bits"], avg_bits=round(r["avg_bits"], 3),
                          fp_vs_q_agreement=round(r["agreement"], 5), output_rel_mse=r["output_error"],
                          sum_layer_loss=round(r["sum_layer_loss"], 3), layers_per_level=hist,
                          seconds=round(r["seconds"], 2), act_follows=act_follows)), flush=True)
