"""Per-layer timing of one full BraTS calibration (diagnostic, GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd import calibrate as K, config as Cf, synth
from efficientq_amd.qconv import PTQConv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = "cuda:0"
if os.environ.get("EFFQ_DP_FORCE", "0") == "1" or os.environ.get("EFFQ_INIT_PG"):        # every data-parallel collective on a 1-rank RCCL group
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29677")
    torch.cuda.set_device(0)
    if os.environ.get("EFFQ_INIT_PG", "nccl") == "gloo":
        dist.init_process_group("gloo", rank=0, world_size=1)
    else:
        dist.init_process_group("nccl", rank=0, world_size=1)
args = Cf.make_args(Cf.BRATS_NET, L, L)
QConv, _, kwQ = Cf.get_conv_class(args)
model = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
synth.randomise_network(model, 0)
model.eval(); K.search_fold_and_remove_bn(model); model.to(dev); K.set_name(model)
t = time.time(); vols = synth.calib_batch("brats", range(N), size).to(dev); print(f"synth {time.time()-t:.1f}s", flush=True)
# wrap ptq for per-layer timing
times = {}
for name, q in model.named_modules():
    if isinstance(q, PTQConv):
        orig = q.ptq
        def timed(x, _o=orig, _n=name):
            torch.cuda.synchronize(); t0 = time.time(); _o(x); torch.cuda.synchronize(); times[_n] = time.time() - t0
            tr = _o.__self__.last_trace
            print(f"  {_n:45s} {times[_n]:7.3f}s  loop {tr['admm_loop_s']:6.3f}s host-enqueue {tr['host_enqueue_s']:6.3f}s  in={tuple(x.shape)}", flush=True)
        q.ptq = timed
pristine = {k: v.clone() for k, v in model.state_dict().items()}
for rep in range(int(os.environ.get("REPS", "1"))):
    print(f"--- pass {rep}", flush=True)
    model.load_state_dict(pristine)
    res = K.calibrate_model(model, vols, "brats", args.init_stride)
print(f"FP pass {res['t1']-res['t0']:.3f}s  PTQ pass {res['t2']-res['t1']:.3f}s  total {res['t2']-res['t0']:.3f}s  vols/s {N/(res['t2']-res['t0']):.4f}")
print("\n".join(res["layer_loss"]))
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
