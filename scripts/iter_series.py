"""Time of the chain ops of the widest layers against the ADMM iteration (diagnostic, GPU): shows what the side-stream
inverses and the loss stream cost the chain kernels that run beside them.
    python scripts/iter_series.py [volumes] [every]      (EFFQ_SIDE=0 / EFFQ_OVERLAP_LOSS=0 for the comparison runs)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd import _lib, calibrate as K, config as Cf, synth, hip_ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = "cuda:0"
if os.environ.get("EFFQ_DP_FORCE", "0") == "1":        # every data-parallel collective on a 1-rank RCCL group
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29678")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
args = Cf.make_args(Cf.BRATS_NET, 4, 4)
QConv, _, kwQ = Cf.get_conv_class(args)
model = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
synth.randomise_network(model, 0)
model.eval(); K.search_fold_and_remove_bn(model); model.to(dev); K.set_name(model)
vols = synth.calib_batch("brats", range(N), 128).to(dev)
pristine = {k: v.clone() for k, v in model.state_dict().items()}
K.calibrate_model(model, vols, "brats", args.init_stride)          # warm-up
model.load_state_dict(pristine)
lib = hip_ops.get_ops(torch.device(dev)).lib if hasattr(hip_ops, "get_ops") else _lib.load()
lib.effq_prof_enable(EVERY)
torch.cuda.synchronize()
import time
t0 = time.time()
K.calibrate_model(model, vols, "brats", args.init_stride)
torch.cuda.synchronize()
print(f"calibration (with sampling every {EVERY}): {(time.time() - t0) * 1e3:.1f} ms")
r = _lib.ProfRecord()
KIND = {1: "prox", 2: "fp", 3: "proj", 4: "loss", 5: "inv", 6: "wait"}
layers, cur, last_key = [], None, None
for i in range(lib.effq_prof_count()):
    _lib.check(lib.effq_prof_read(i, C.byref(r)), "effq_prof_read")
    key = (r.c2, r.n)
    if r.kind == 5 and r.iter == -1:                 # the first inverse of a layer opens it
        cur = dict(key=key, rows={}, inv=[], wait=0.0, waits=[])
        layers.append(cur)
    if cur is None:
        continue
    k = KIND[r.kind]
    if k == "inv":
        cur["inv"].append(round(r.ms, 2))
    elif k == "wait":
        cur["wait"] += r.ms
        cur["waits"].append((r.iter, round(r.ms, 3)))
    else:
        cur["rows"].setdefault(r.iter, {})[k] = r.ms
lib.effq_prof_enable(0)
for L in layers:
    c2, n = L["key"]
    if n < 3000:
        if L["wait"] > 0.3:
            print(f"layer c2={c2} n={n}: inverses {L['inv']} ms, waits {L['wait']:.2f} ms  at (iteration, ms): {L['waits']}")
        continue
    print(f"layer c2={c2} n={n}: inverses {L['inv']} ms, waits {L['wait']:.2f} ms  at (iteration, ms): {L['waits']}")
    for it in sorted(L["rows"]):
        row = L["rows"][it]
        print(f"   it {it:3d}  " + "  ".join(f"{k} {row[k] * 1e3:7.1f} us" for k in ("prox", "fp", "proj", "loss") if k in row))
