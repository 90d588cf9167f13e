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


PEAK_I8_MFMA_TOPS = 5033.2         # MI355X dense int8 matrix peak (MI355X_MICROARCH.md: = fp8 dense)
PEAK_F64_MFMA_TFLOPS = 78.6       # MI355X fp64 matrix peak (vendor)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


SAMPLE = 4      # one launch in SAMPLE is timed with HIP events


class OpTimer:
    """HIP-event pairs (on the launch stream) around every call of the heavy library ops inside the timed
    steps, aggregated per (op, shape).  The op class with the largest total is the dominant kernel; its
    roofline uses the ALGORITHMIC work of one launch over the average duration."""

    def __init__(self):
        self.rec = {}
        self.count = {}

    def _wrap(self, ops, name, keyfn):
        inner = getattr(ops, name)
        rec = self.rec

        count = self.count

        def call(*a, **kw):
            key = keyfn(*a, **kw)
            if key is None:
                return inner(*a, **kw)
            pinned = getattr(ops, "_pinned_stream", None)
            handle = (pinned.value or 0) if pinned is not None else torch.cuda.current_stream().cuda_stream
            # work issued on the inverse side stream overlaps the calibration stream: timed, but kept out of the
            # ranking (the loss stream carries the critical-path loss convs and counts as calibration work)
            side = getattr(ops, "_side", None) is not None and handle == ops._side.cuda_stream
            ck = (name + ("@side" if side else ""),) + key
            # every launch is counted, every SAMPLE-th one is bracketed by events: two event records per op cost
            # the host ~10 us, which the host-bound small layers would pay in the measured wall clock
            c = count.get(ck, 0)
            count[ck] = c + 1
            if c % SAMPLE:
                return inner(*a, **kw)
            st = torch.cuda.current_stream() if pinned is None else (
                torch.cuda.default_stream(ops.device) if handle == 0 else torch.cuda.ExternalStream(handle, device=ops.device))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)                    # records on the stream the kernels go to
            r = inner(*a, **kw)
            e1.record(st)
            rec.setdefault(ck, []).append((e0, e1))
            return r
        setattr(ops, name, call)
        return lambda: setattr(ops, name, inner)

    def wrap(self, ops):
        def gkey(g):
            return (g.N, g.C1, g.C2, g.D, g.H, g.W, g.KD, g.SD)
        undo = [
            self._wrap(ops, "conv_step", lambda x, w, b, geom, y=None, att=None, **kw: gkey(geom)),
            self._wrap(ops, "conv_step_i8", lambda xi, gq, b, geom, *a, **kw: gkey(geom)),
            self._wrap(ops, "conv_step_i8s", lambda xi, gq, b, geom, *a, **kw: gkey(geom)),
            self._wrap(ops, "gram", lambda x, att, y, geom, hb, *a, **kw: gkey(geom)),
            self._wrap(ops, "gram_i8", lambda xi, cls, y, geom, *a, **kw: gkey(geom)),
            self._wrap(ops, "spd_inverse", lambda A0, *a, **kw: (int(A0.shape[0]),)),
            self._wrap(ops, "prox_solve", lambda B0, *a, **kw: (int(B0.shape[0]), int(B0.shape[1]))),
            # the fused chain step (prox GEMM + scale fixed point + projection; iteration 0's shifted solve is left out)
            self._wrap(ops, "chain_step", lambda a, *r, **kw: None if (len(r) > 9 and r[9]) or kw.get("shift_terms")
                       else (int(a.c2), int(a.n))),
        ]
        return lambda: [u() for u in undo]

    @staticmethod
    def _work(key):
        op = key[0].replace("@side", "")
        if op in ("conv_step", "conv_step_i8", "conv_step_i8s", "gram", "gram_i8"):
            N, c1, c2, D, H, W, k, s = key[1:]
            od, oh, ow = (D + 2 * (k // 2) - k) // s + 1, (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
            V, Vin = N * od * oh * ow, N * D * H * W
            if op == "conv_step":
                return ("mfma", 2.0 * c2 * c1 * k ** 3 * V, "f32 MFMA", PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                        f"k_conv3d* ({c1}->{c2}, {k}^3/s{s}, {N}x{od}x{oh}x{ow} voxels, f32 MFMA)")
            if op == "conv_step_i8s":
                return ("hbm", 4.0 * c2 * V + 1.0 * c1 * Vin, "HBM", PEAK_HBM_GBS, "GB/s",
                        f"k_conv3d_i8s ({c1}->{c2}, {k}^3/s{s}, {N}x{od}x{oh}x{ow} voxels: 4*c2 B of target + c1 B of "
                        f"level ids per input voxel, i8 MFMA exact)")
            if op == "conv_step_i8":
                return ("hbm", 4.0 * c2 * V + 1.0 * c1 * Vin, "HBM", PEAK_HBM_GBS, "GB/s",
                        f"k_conv3d_i8 ({c1}->{c2}, 3^3, {N}x{od}x{oh}x{ow} voxels: 4*c2 B of target + c1 B of level "
                        f"ids per voxel, i8 MFMA exact)")
            n = c1 * k ** 3 + 1
            if op == "gram_i8":
                # integer ops of the products that are needed: x-x upper triangle + 4 y digit rows per channel
                return ("mfma", 1.0 * n * n * V + 2.0 * 4 * c2 * n * V, "i8 MFMA", PEAK_I8_MFMA_TOPS, "TOP/s",
                        f"k_gram_i8 (n={n}, {V} voxels; n^2 V (upper triangle) + 8 c2 n V int8 op, exact)")
            return ("mfma", 2.0 * n * n * V + 2.0 * c2 * n * V, "f32 MFMA", PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                    f"k_gram (n={n}, {V} voxels; 2n^2V+2c2nV flop, upper triangle computed)")
        if op == "spd_inverse":
            n = key[1]
            return ("mfma", 2.0 * n ** 3, "f64 MFMA", PEAK_F64_MFMA_TFLOPS, "TFLOP/s", f"k_gj_* (n={n}, 2n^3 fp64 flop)")
        c2, n = key[1:]
        if op == "chain_step":
            return ("mfma", 2.0 * c2 * n * n, "f32 MFMA", PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                    f"ADMM chain step (c2={c2}, n={n}): k_build_b + k_prox_gemm (2 c2 n^2 flop, counted) + scale fixed "
                    f"point + projection, one binding call")
        return ("mfma", 2.0 * c2 * n * n, "f32 MFMA", PEAK_F32_MFMA_TFLOPS, "TFLOP/s", f"k_prox_gemm (c2={c2}, n={n})")

    def summary(self):
        rows = []
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        except (OSError, ValueError):
            traffic = {}
        for key, pairs in self.rec.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            bound, work, pname, peak, unit, label = self._work(key)
            avg = sum(ms) / len(ms)
            ach = work / (avg * 1e-3) / (1e12 if unit in ("TFLOP/s", "TOP/s") else 1e9)
            side = key[0].endswith("@side")
            tkey = key[0].replace("@side", "") + "|" + ",".join(str(v) for v in key[1:])
            tr = traffic.get(tkey)
            rows.append(dict(kernel=label + (" [side stream, overlapped with the ADMM iterations]" if side else ""),
                             bound=bound, achieved=round(ach, 2), peak=peak, unit=unit,
                             frac=round(ach / peak, 4),
                             traffic=(tr["bytes"] if tr else None),      # HBM-side bytes per launch (PMC passes)
                             traffic_algorithmic=(tr["algorithmic_bytes"] if tr else None),
                             traffic_source=(tr["source"] if tr else None),
                             launches=self.count.get(key, len(ms)), timed_launches=len(ms), avg_ms=round(avg, 4),
                             total_ms=round(avg * self.count.get(key, len(ms)), 1), work_per_launch=work,
                             overlapped=side))
        # A loss conv runs on the loss stream under the chain step of the NEXT iteration of its layer: where the chain
        # step is the longer of the two, the conv is hidden (its event time is mostly queueing behind the chain's
        # kernels) and does not belong in the ranking of critical-path ops.
        chain_avg = {}
        for key, pairs in self.rec.items():
            if key[0] == "chain_step":
                ms = [a.elapsed_time(b) for a, b in pairs]
                chain_avg[key[1:]] = sum(ms) / len(ms)
        for key, row in zip(self.rec.keys(), rows):
            if key[0] in ("conv_step", "conv_step_i8", "conv_step_i8s") and not row["overlapped"]:
                N, c1, c2, D, H, W, k, st = key[1:]
                ca = chain_avg.get((c2, c1 * k ** 3 + 1))
                if ca is not None and row["launches"] >= 100 and row["avg_ms"] < 0.8 * ca:
                    row["overlapped"] = True
                    row["kernel"] += f" [loss stream, hidden under the {ca:.3f} ms chain step of its layer]"
        rows.sort(key=lambda r: (r["overlapped"], -r["total_ms"]))
        if not rows:
            return None, []
        # `roofline` is the largest SINGLE-KERNEL op of the critical path (its duration can be checked against the
        # rocprofv3 kernel summary); the chain steps are several kernels behind one binding call and are listed with
        # the other ops (composite = true)
        for r in rows:
            r["composite"] = r["kernel"].startswith("ADMM chain step")
        first = next((i for i, r in enumerate(rows) if not r["composite"] and not r["overlapped"]), 0)
        rows.insert(0, rows.pop(first))
        for r in rows:
            log(f"[ops] {r['total_ms']:9.1f} ms {r['launches']:6d} x {r['avg_ms']:9.4f} ms  {r['frac']:.3f} of {r['bound']} "
                f"peak  {r['kernel'][:90]}")
        return rows[0], rows[1:7]


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
    ap.add_argument("--f32-only", action="store_true",
                    help="evaluate every per-iteration loss on the f32 matrix cores (no exact-integer path)")
    a = ap.parse_args()

    if a.f32_only:
        os.environ["EFFQ_EXACT_INT"] = "0"
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
    timer = OpTimer()

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
    roof, others = timer.summary()
    from efficientq_amd import qconv as _Q
    exact = bool(_Q.EXACT_INT_DEFAULT)
    out = {
        "metric": "ptq_calibration_throughput", "value": round(total_vols / dt, 5), "unit": "calib-vols/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
        "wall_clock_s_per_calibration": round(dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i8xi8->i32 (exact) for the Gram systems and per-iteration loss convs of the layers with quantised input, f32/f64 elsewhere"
                 if exact else "f32", "data": "synthetic",
        "config": {"workload": f"BraTS 3D-UNet fp32->{a.levels}-level PTQ (qlvl_w={a.levels} qlvl_a={a.levels}, "
                               f"q_first=q_last=256,-1), 22 quantised convs, 200 ADMM its/layer, "
                               f"{a.vols} synthetic 4x{a.size}^3 volumes per GPU (BASELINE.json configs[1])",
                   "vols_per_gpu": a.vols, "volume": f"4x{a.size}^3", "parallelism": f"dp{world}",
                   "fp_pass_s": round(res["t1"] - res["t0"], 3), "ptq_pass_s": round(res["t2"] - res["t1"], 3)},
        "roofline": roof,
        "other_kernels": others,
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
