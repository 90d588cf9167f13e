#!/usr/bin/env python3
"""Headline benchmark: full layer-wise PTQ calibration of the BraTS 3D-UNet (BASELINE.json config 2:
fp32 -> 2-bit, qlvl_w=4 qlvl_a=4, 16 synthetic 4x128^3 volumes per GPU).

A "step" is ONE complete calibration (the reference's t2-t0 window, ptqer.py:333-363: FP pass with
target capture + mask pyramid + quantising pass over all 22 quantised convs) of this rank's 16
volumes, inputs resident in HBM.  With N GPUs the calibration set is 16*N volumes sharded N ways
(data-parallel, Gram / statistics all-reduced over RCCL) => weak scaling; value = total volumes / s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--vols V] [--size S] [--levels L]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (progress goes to stderr).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# conv flops of the dominant layer shape (BraTS level-1, 32->32, 3^3, 64^3 voxels) per volume per forward:
# 2*c2*c1*k^3*V = 2*32*32*27*64^3  (14.5 GFLOP/volume/forward)
DOMINANT = dict(c1=32, c2=32, k=3, vox=64 ** 3)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(levels, device):
    from efficientq_amd import calibrate as K, config as Cf, synth
    args = Cf.make_args(Cf.BRATS_NET, levels, levels)
    QConv, _, kwQ = Cf.get_conv_class(args)
    model = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
    synth.randomise_network(model, 0)          # same "pretrained" net on every rank
    model.eval()
    K.search_fold_and_remove_bn(model)
    model.to(device)
    K.set_name(model)
    return args, model


class ConvTimer:
    """HIP-event pairs around the launches of the dominant conv shape inside the timed steps."""

    def __init__(self):
        self.pairs = []
        self.flops_per_launch = None

    def wrap(self, ops):
        inner = ops.conv_step
        timer = self

        def conv_step(x, w, b, geom, y=None, att=None, **kw):
            dom = (geom.C1 == DOMINANT["c1"] and geom.C2 == DOMINANT["c2"] and geom.KD == 3 and y is not None
                   and geom.D * geom.H * geom.W == DOMINANT["vox"] and att is None)
            if not dom:
                return inner(x, w, b, geom, y, att, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = inner(x, w, b, geom, y, att, **kw)
            e1.record()
            timer.pairs.append((e0, e1))
            timer.flops_per_launch = 2.0 * geom.C2 * geom.C1 * 27 * geom.N * geom.D * geom.H * geom.W
            return r
        ops.conv_step = conv_step
        return lambda: setattr(ops, "conv_step", inner)

    def summary(self):
        if not self.pairs:
            return None
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        avg = sum(ms) / len(ms)
        ach = self.flops_per_launch / (avg * 1e-3) / 1e12
        return dict(bound="mfma", achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), traffic=None,
                    kernel="k_conv3d<1> (32->32, 3x3x3, 64^3 x %d vol)" % (self.flops_per_launch / (2 * 32 * 32 * 27 * 64 ** 3)),
                    launches=len(ms), avg_ms=round(avg, 4), flops_per_launch=self.flops_per_launch)


def cpu_baseline(levels):
    """Bounded CPU sample: the oracle (validated bit-exact against the reference) calibrating the
    dominant layer shape on a reduced volume, scaled to whole-net volumes/s by algorithmic work."""
    from oracle import effq_oracle as O
    S = 32                                      # 32^3 = 1/8 of the 64^3 voxels of one volume
    gen = torch.Generator().manual_seed(0)
    c = 32
    w = torch.randn(c, c, 3, 3, 3, generator=gen) * (2.0 / (c * 27)) ** 0.5
    b = torch.randn(c, generator=gen) * 0.1
    xf = torch.relu(torch.randn(1, c, S, S, S, generator=gen))
    y = torch.nn.functional.conv3d(xf, w, b, 1, 1)
    x = torch.relu(xf + 0.05 * torch.randn(xf.shape, generator=gen))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = min(torch.get_num_threads(), avail)
    torch.set_num_threads(cores)
    t = time.time()
    O.calibrate_layer(x, y, w, b, 1, 1, qlvl_w=levels, qlvl_act=levels)
    dt = time.time() - t
    # volume-dependent work of the whole net per volume (conv 21.59 + Gram 2.96 TFLOP) over the
    # same quantity for the sample (201 convs + Gram on S^3 voxels)
    v = S ** 3
    sample_tflop = (201 * 2 * c * c * 27 * v + 2 * (c * 27 + 1) ** 2 * v + 2 * c * (c * 27 + 1) * v) / 1e12
    net_tflop_per_vol = 24.55
    est_s_per_vol = dt * net_tflop_per_vol / sample_tflop
    return dict(value=round(1.0 / est_s_per_vol, 6), unit="calib-vols/s", cores=cores, kind="port",
                sample=f"oracle.calibrate_layer (200 ADMM its, per-iteration LU like the reference) on one 32->32 3^3 "
                       f"layer, 1 volume of {S}^3 voxels: {dt:.1f} s for {sample_tflop:.3f} TFLOP; scaled by "
                       f"conv+Gram work to the whole net ({net_tflop_per_vol} TFLOP/volume), solves of the wide "
                       f"layers not included (favours the CPU)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--vols", type=int, default=16, help="calibration volumes per GPU")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from efficientq_amd import calibrate as K, synth
    from efficientq_amd.hip_ops import get_ops
    args, model = build_model(a.levels, device)
    pristine = {k: v.clone() for k, v in model.state_dict().items()}
    t = time.time()
    ids = range(rank * a.vols, (rank + 1) * a.vols)           # rank r holds its own shard of the volumes
    vols = synth.calib_batch("brats", ids, a.size).to(device)
    log(f"[rank {rank}] {a.vols} synthetic volumes 4x{a.size}^3 in HBM ({time.time() - t:.1f}s)")

    ops = get_ops(device)
    timer = ConvTimer()

    def one_step():
        model.load_state_dict(pristine, strict=True)
        return K.calibrate_model(model, vols, "brats", args.init_stride)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(a.warmup):
        t = time.time()
        one_step()
        torch.cuda.synchronize(device)
        log(f"[rank {rank}] warmup {i}: {time.time() - t:.2f}s")
    unwrap = timer.wrap(ops)
    fence()
    t0 = time.time()
    res = None
    for i in range(a.steps):
        res = one_step()
    fence()
    dt = time.time() - t0
    unwrap()
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    total_vols = a.vols * world * a.steps
    roof = timer.summary()
    out = {
        "metric": "ptq_calibration_throughput", "value": round(total_vols / dt, 5), "unit": "calib-vols/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
        "wall_clock_s_per_calibration": round(dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BraTS 3D-UNet fp32->{a.levels}-level PTQ (qlvl_w={a.levels} qlvl_a={a.levels}, "
                               f"q_first=q_last=256,-1), 22 quantised convs, 200 ADMM its/layer, "
                               f"{a.vols} synthetic 4x{a.size}^3 volumes per GPU (BASELINE.json configs[1])",
                   "vols_per_gpu": a.vols, "volume": f"4x{a.size}^3", "parallelism": f"dp{world}",
                   "fp_pass_s": round(res["t1"] - res["t0"], 3), "ptq_pass_s": round(res["t2"] - res["t1"], 3)},
        "roofline": roof,
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log("[rank 0] timing the CPU baseline sample ...")
        out["cpu_baseline"] = cpu_baseline(a.levels)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
