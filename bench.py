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
import ctypes as C
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
PEAK_I8_MFMA_TOPS = 5033.2        # MI355X dense int8 matrix peak (MI355X_MICROARCH.md: = fp8 dense)
PEAK_F64_MFMA_TFLOPS = 78.6       # MI355X fp64 matrix peak (vendor)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
NET_TFLOP_PER_VOLUME = 24.55      # conv 21.59 + Gram 2.96 TFLOP per 4x128^3 volume (fp32 algorithmic work)
SAMPLE = 4                        # one launch / iteration in SAMPLE is bracketed by HIP events


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(levels, device, net="brats"):
    from efficientq_amd import calibrate as K, config as Cf, synth
    args = Cf.make_args(Cf.BRATS_NET if net == "brats" else Cf.LITS_NET, levels, levels)
    QConv, _, kwQ = Cf.get_conv_class(args)
    model = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
    synth.randomise_network(model, 0)          # same "pretrained" net on every rank
    model.eval()
    K.search_fold_and_remove_bn(model)
    model.to(device)
    K.set_name(model)
    return args, model


class OpTimer:
    """HIP-event pairs (on the launch stream) around the heavy library ops inside the timed steps, aggregated per
    (op, shape).  Ops issued from Python (Gram systems, forwards, final losses) are wrapped here; the ops of the ADMM
    loop, which one library call enqueues, are sampled by the library's own profiler (effq_prof_*: every SAMPLE-th
    iteration).  The largest single-kernel op of the critical path is the dominant kernel; its roofline uses the
    ALGORITHMIC work of one launch over the average duration."""

    KIND = {1: "prox", 2: "fixed_point", 3: "project", 4: "loss", 5: "inverse"}

    def __init__(self):
        self.rec = {}
        self.count = {}

    def _wrap(self, ops, name, keyfn):
        inner = getattr(ops, name)
        rec, count = self.rec, self.count

        def call(*a, **kw):
            key = keyfn(*a, **kw)
            if key is None:
                return inner(*a, **kw)
            ck = (name,) + key
            c = count.get(ck, 0)
            count[ck] = c + 1
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)                    # these ops run on torch's current stream
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
            self._wrap(ops, "gram", lambda x, att, y, geom, hb, *a, **kw: gkey(geom)),
            self._wrap(ops, "gram_i8", lambda xi, cls, y, geom, *a, **kw: gkey(geom)),
        ]
        ops.lib.effq_prof_enable(SAMPLE)
        self.lib = ops.lib
        return lambda: [u() for u in undo]

    def collect_library(self):
        """Fold the library's sampled records in (after a device synchronise)."""
        from efficientq_amd import _lib
        lib = self.lib
        r = _lib.ProfRecord()
        self.lib_ms = {}
        for i in range(lib.effq_prof_count()):
            _lib.check(lib.effq_prof_read(i, C.byref(r)), "effq_prof_read")
            g = r.geom
            kind = self.KIND[r.kind]
            if kind == "loss" and r.loss_kind == 4:
                key = ("gram_loss@loop", r.c2, r.n)
            elif kind == "loss":
                name = {0: "conv_step", 1: "conv_step_i8", 2: "conv_step_i8s", 3: "conv_step_i8_pair"}[r.loss_kind] + "@loop"
                key = (name, g.N, g.C1, g.C2, g.D, g.H, g.W, g.KD, g.SD)
            else:
                key = (kind, r.c2, r.n)
            self.lib_ms.setdefault(key, []).append(r.ms)
        lib.effq_prof_enable(0)

    @staticmethod
    def _work(key):
        op = key[0].replace("@loop", "")
        if op in ("conv_step", "conv_step_i8", "conv_step_i8s", "conv_step_i8_pair", "gram", "gram_i8"):
            N, c1, c2, D, H, W, k, s = key[1:]
            od, oh, ow = (D + 2 * (k // 2) - k) // s + 1, (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
            V, Vin = N * od * oh * ow, N * D * H * W
            if op == "conv_step":
                return ("mfma", 2.0 * c2 * c1 * k ** 3 * V, PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                        f"k_conv3d* ({c1}->{c2}, {k}^3/s{s}, {N}x{od}x{oh}x{ow} voxels, f32 MFMA)")
            if op == "conv_step_i8s":
                return ("hbm", 4.0 * c2 * V + 1.0 * c1 * Vin, PEAK_HBM_GBS, "GB/s",
                        f"k_conv3d_i8s ({c1}->{c2}, {k}^3/s{s}, {N}x{od}x{oh}x{ow} voxels: 4*c2 B of target + c1 B of "
                        f"level ids per input voxel, i8 MFMA exact)")
            if op == "conv_step_i8_pair":
                # 4*(c1 V_in + c2 V) bytes per evaluation in fp32; here level ids are 1 byte, and ONE launch
                # evaluates TWO iterates (units per launch = 2): the second shares the pass over x and y
                return ("hbm", 2.0 * (4.0 * c2 * V + 1.0 * c1 * Vin), PEAK_HBM_GBS, "GB/s",
                        f"k_conv3d_i8p<2> ({c1}->{c2}, 3^3, {N}x{od}x{oh}x{ow} voxels, TWO iterates per launch: 2 x (4*c2 B "
                        f"of target + c1 B of level ids per voxel) algorithmic, one pass over x and y, i8 MFMA exact)")
            if op == "conv_step_i8":
                return ("hbm", 4.0 * c2 * V + 1.0 * c1 * Vin, PEAK_HBM_GBS, "GB/s",
                        f"k_conv3d_i8 ({c1}->{c2}, 3^3, {N}x{od}x{oh}x{ow} voxels: 4*c2 B of target + c1 B of level "
                        f"ids per voxel, i8 MFMA exact)")
            n = c1 * k ** 3 + 1
            if op == "gram_i8":
                return ("mfma", 1.0 * n * n * V + 2.0 * 4 * c2 * n * V, PEAK_I8_MFMA_TOPS, "TOP/s",
                        f"k_gram_i8 (n={n}, {V} voxels; n^2 V (upper triangle) + 8 c2 n V int8 op, exact)")
            return ("mfma", 2.0 * n * n * V + 2.0 * c2 * n * V, PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                    f"k_gram (n={n}, {V} voxels; 2n^2V+2c2nV flop, upper triangle computed)")
        c2, n = key[1:]
        if op == "gram_loss":
            return ("mfma", 2.0 * c2 * n * n, PEAK_F64_MFMA_TFLOPS, "TFLOP/s",
                    f"k_gram_loss (c2={c2}, n={n}: loss of an iterate from the unweighted Gram system, 2 c2 n^2 fp64 flop "
                    f"on an n x n matrix instead of a pass over the voxels)")
        if op == "inverse":
            return ("mfma", 2.0 * n ** 3, PEAK_F64_MFMA_TFLOPS, "TFLOP/s", f"k_gj_* (n={n}, 2n^3 fp64 flop)")
        if op == "prox":
            if c2 >= 256 and n >= 2048:
                # the 256-row GEMM kernel is > 97 % of this bracket (the split-K reduce takes 5 us, the right-hand side is
                # written by the projection kernel of the previous iteration): reported as a single-kernel op
                return ("mfma", 2.0 * c2 * n * n, PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                        f"k_prox_gemm<2,4,1,2> (+ 5 us k_prox_reduce4) (c2={c2}, n={n}; 2 c2 n^2 flop, f32 MFMA)")
            return ("mfma", 2.0 * c2 * n * n, PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                    f"k_prox_gemm + k_prox_reduce (c2={c2}, n={n}; 2 c2 n^2 flop)")
        nw = c2 * (n - 1)
        if op == "fixed_point":
            # one pass over the weights per call is the least an implementation can do
            return ("hbm", 8.0 * nw, PEAK_HBM_GBS, "GB/s", f"weight-scale fixed point ({nw} weights; latency-bound)")
        return ("hbm", 21.0 * nw, PEAK_HBM_GBS, "GB/s", f"k_project_dual ({nw} weights)")

    def summary(self, steps):
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))
        except (OSError, ValueError):
            traffic = {}
        rows = []
        items = []
        for key, pairs in self.rec.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            items.append((key, ms, self.count[key], "main"))
        for key, ms in self.lib_ms.items():
            if key[0] == "inverse":
                launches, stream = len(ms), "side"
            else:
                launches, stream = len(ms) * SAMPLE, ("loss" if key[0].endswith("@loop") else "main")
                if key[0].startswith("conv_step_i8_pair"):
                    launches = len(ms) * SAMPLE // 2          # one launch per two iterations
            items.append((key, ms, launches, stream))
        busy = {"main": 0.0, "loss": 0.0, "side": 0.0}
        for key, ms, launches, stream in items:
            bound, work, peak, unit, label = self._work(key)
            avg = sum(ms) / len(ms)
            ach = work / (avg * 1e-3) / (1e12 if unit in ("TFLOP/s", "TOP/s") else 1e9)
            tkey = key[0].replace("@loop", "") + "|" + ",".join(str(v) for v in key[1:])
            tr = traffic.get(tkey)
            busy[stream] += avg * launches
            rows.append(dict(kernel=label, stream=stream, bound=bound, achieved=round(ach, 2), peak=peak, unit=unit,
                             frac=round(ach / peak, 4), traffic=(tr["bytes"] if tr else None),
                             traffic_algorithmic=(tr["algorithmic_bytes"] if tr else None),
                             traffic_source=(tr["source"] if tr else None), launches=launches,
                             timed_launches=len(ms), avg_ms=round(avg, 4), total_ms=round(avg * launches, 1),
                             ms_per_step=round(avg * launches / steps, 1), work_per_launch=work,
                             composite=(key[0] in ("inverse", "fixed_point") or
                                        (key[0] == "prox" and not (key[1] >= 256 and key[2] >= 2048)))))
        rows.sort(key=lambda r: -r["total_ms"])
        for r in rows:
            log(f"[ops] {r['total_ms']:9.1f} ms {r['launches']:6d} x {r['avg_ms']:9.4f} ms  {r['frac']:.3f} of {r['bound']} "
                f"peak [{r['stream']}]  {r['kernel'][:88]}")
        if not rows:
            return None, [], busy
        # `roofline` = the largest SINGLE-KERNEL op of the MAIN stream - the critical path (its duration can be checked
        # against the rocprofv3 kernel summary).  Ops of the loss and side streams run beside it and their in-situ
        # durations are mostly time spent waiting for CUs (the 256-channel loss conv takes 0.05 ms alone and 0.3 ms
        # between the workgroups of the prox GEMM); multi-kernel ops (inverse = a few hundred kernels, the multi-launch
        # fixed points) are listed with the others as composite.
        first = next((i for i, r in enumerate(rows) if not r["composite"] and r["stream"] == "main"),
                     next((i for i, r in enumerate(rows) if not r["composite"]), 0))
        rows.insert(0, rows.pop(first))
        return rows[0], rows[1:9], busy


# ---------------------------------------------------------------------------------------------------------------------
def brats_layers(vols, size):
    """(c1, c2, k, stride, V_out, V_in, quantised_input) of the 22 quantised convs of the BraTS net for `vols` volumes."""
    s1 = size // 2
    L = [(4, 32, 3, 2, s1 ** 3, size ** 3, False)]
    res = {32: s1, 64: s1 // 2, 128: s1 // 4, 256: s1 // 8}
    for c in (32, 64, 128):
        L += [(c, c, 3, 1, res[c] ** 3, res[c] ** 3, True)] * 2
        L += [(c, 2 * c, 1, 1, res[2 * c] ** 3, res[2 * c] ** 3, True)]
    L += [(256, 256, 3, 1, res[256] ** 3, res[256] ** 3, True)] * 2
    for c in (256, 128, 64):
        L += [(c, c // 2, 1, 1, res[c] ** 3, res[c] ** 3, True)]
        L += [(c // 2, c // 2, 3, 1, res[c // 2] ** 3, res[c // 2] ** 3, True)] * 2
    L += [(32, 3, 1, 1, s1 ** 3, s1 ** 3, False)]
    return [(c1, c2, k, s, v * vols, vi * vols, q) for (c1, c2, k, s, v, vi, q) in L]


def cpu_baseline(levels, vols, size):
    """The CPU restatement of the reference (oracle/, validated bit for bit against it) timed per COMPONENT on this
    box's host cores, each on a bounded sample, and extrapolated to the benchmark workload with the component's own
    scaling law: conv + MSE and Gram GEMM ~ flops, im2col ~ patch-matrix bytes, LU solve ~ n^3 (+ n^2 c2), fixed
    points ~ elements x iterations.  ~40 s of CPU work."""
    import numpy as np
    import torch.nn.functional as F
    from oracle import effq_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = min(torch.get_num_threads(), avail)
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(0)
    meas = {}

    def timeit(fn, budget=4.0, min_rep=2):
        fn()
        t0, rep = time.time(), 0
        while rep < min_rep or time.time() - t0 < budget:
            fn()
            rep += 1
            if rep >= 50:
                break
        return (time.time() - t0) / rep

    # conv + mse (EfficientQConv.py:118-122): two shapes, seconds per GFLOP
    rates = []
    for c, S in ((32, 32), (128, 8)):
        x = torch.relu(torch.randn(1, c, S, S, S, generator=gen))
        w = torch.randn(c, c, 3, 3, 3, generator=gen) * 0.05
        y = torch.randn(1, c, S, S, S, generator=gen)
        t = timeit(lambda: F.mse_loss(F.conv3d(x, w, None, 1, 1), y).item(), 3.0)
        rates.append(t / (2 * c * c * 27 * S ** 3 / 1e9))
        meas[f"conv_mse_{c}ch_{S}^3_s"] = round(t, 5)
    conv_s_per_gflop = sum(rates) / len(rates)
    # im2col (python triple loop, solver.py:86-111) + Gram GEMMs on 32 ch @ 24^3
    c, S = 32, 24
    xq = torch.relu(torch.randn(1, c, S, S, S, generator=gen))
    yy = torch.randn(1, c, S, S, S, generator=gen)
    t0 = time.time()
    ps = O.ProxSystem(xq, yy, (3, 3, 3), 1, 1, torch.zeros(c, c, 3, 3, 3), torch.zeros(c), None)
    t_gram = time.time() - t0
    n0 = c * 27 + 1
    meas["im2col_gram_32ch_24^3_s"] = round(t_gram, 3)
    gram_s_per_gflop = t_gram / ((2 * n0 * n0 + 2 * c * n0) * S ** 3 / 1e9)
    # LU solve per iteration (solver.py:331): n = 865 and 1729
    sol = {}
    for n, c2 in ((865, 32), (1729, 64)):
        A = torch.randn(n, n, generator=gen)
        A = A @ A.T + n * torch.eye(n)
        B = torch.randn(c2, n, generator=gen)
        sol[n] = timeit(lambda: torch.linalg.solve(A, B.T), 3.0)
        meas[f"lu_solve_n{n}_s"] = round(sol[n], 5)
    solve_coef = sol[1729] / (1729.0 ** 3)            # the largest measured size sets the n^3 law
    # project_by_iter on weights (layer_helper.py:40-70), fp64
    wv = torch.randn(110592, generator=gen) * 0.05
    t_fit = timeit(lambda: O.fit_scale(wv, levels, -1.0, 1.0), 3.0)
    it_w = O.fit_scale(wv, levels, -1.0, 1.0).iters
    fit_s_per_elem_iter = t_fit / (110592 * it_w)
    meas["project_by_iter_110592_weights_s"] = round(t_fit, 4)
    av = torch.relu(torch.randn(32 * 32 ** 3, generator=gen))
    t_act = timeit(lambda: O.fit_scale(av, levels, 0.0, 1.0), 4.0, 1)
    it_a = O.fit_scale(av, levels, 0.0, 1.0).iters
    meas["project_by_iter_1M_activations_s"] = round(t_act, 3)
    # extrapolate to the workload
    total = dict(conv=0.0, gram=0.0, solve=0.0, proj_w=0.0, proj_a=0.0)
    for (c1, c2, k, s, V, Vin, q) in brats_layers(vols, size):
        n = c1 * k ** 3 + 1
        total["conv"] += 201 * conv_s_per_gflop * (2.0 * c1 * c2 * k ** 3 * V / 1e9)
        total["gram"] += gram_s_per_gflop * ((2.0 * n * n + 2.0 * c2 * n) * V / 1e9)
        total["solve"] += 200 * solve_coef * n ** 3
        itw = it_w if (c1 != 4 and c2 != 3) else 300           # first / last layer keep 256 weight levels
        total["proj_w"] += 200 * fit_s_per_elem_iter * c2 * c1 * k ** 3 * itw
        if q:
            total["proj_a"] += (t_act / (32 * 32 ** 3)) * c1 * Vin
    est = sum(total.values())
    return dict(value=round(vols / est, 6), unit="calib-vols/s", cores=cores, kind="port",
                est_seconds_per_calibration=round(est, 1),
                components_s={k: round(v, 1) for k, v in total.items()}, measured=meas,
                sample="oracle components timed on this box's host cores and extrapolated per component to the "
                       f"{vols} x 4x{size}^3 BraTS workload: conv+MSE (2 shapes) ~ flops, im2col+Gram (32 ch, 24^3) ~ "
                       "flops, LU solve (n = 865, 1729) ~ n^3 x 200 per layer, project_by_iter ~ elements x iterations "
                       f"({it_w} at {levels} weight levels, {it_a} on activations); hooks / copies / std() left out "
                       "(favours the CPU)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--vols", type=int, default=16, help="calibration volumes per GPU")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--net", choices=["brats", "lits"], default="brats",
                    help="brats = BASELINE configs[1] (the headline); lits = configs[3] geometry (1x160^3 volumes, widths to 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-subrun", action="store_true")
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
    args, model = build_model(a.levels, device, a.net)
    if a.net == "lits" and a.size == 128:
        a.size = 160
    pristine = {k: v.clone() for k, v in model.state_dict().items()}
    t = time.time()
    ids = range(rank * a.vols, (rank + 1) * a.vols)           # rank r holds its own shard of the volumes
    vols = synth.calib_batch(a.net, ids, a.size).to(device)
    nmod = 4 if a.net == "brats" else 1
    log(f"[rank {rank}] {a.vols} synthetic volumes {nmod}x{a.size}^3 in HBM ({time.time() - t:.1f}s)")

    ops = get_ops(device)
    timer = OpTimer()

    def one_step():
        model.load_state_dict(pristine, strict=True)
        return K.calibrate_model(model, vols, a.net, args.init_stride)

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
    timer.collect_library()
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    total_vols = a.vols * world * a.steps
    roof, others, busy = timer.summary(a.steps)
    from efficientq_amd import qconv as _Q
    exact = bool(_Q.EXACT_INT_DEFAULT)
    if roof is not None:
        roof = dict(roof)
    # 24.55 TFLOP per 4x128^3 BraTS volume, 99.69 per 1x160^3 LiTS volume (conv x 201 + Gram), scaled by voxels
    tfv = (NET_TFLOP_PER_VOLUME * (a.size / 128.0) ** 3) if a.net == "brats" else (99.69 * (a.size / 160.0) ** 3)
    out = {
        "metric": "ptq_calibration_throughput", "value": round(total_vols / dt, 5), "unit": "calib-vols/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
        "wall_clock_s_per_calibration": round(dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i8xi8->i32 (exact) for the Gram systems and per-iteration loss convs of the layers with quantised input, f32/f64 elsewhere"
                 if exact else "f32", "data": "synthetic",
        "config": {"workload": (f"BraTS 3D-UNet fp32->{a.levels}-level PTQ (qlvl_w={a.levels} qlvl_a={a.levels}, "
                                f"q_first=q_last=256,-1), 22 quantised convs, 200 ADMM its/layer, "
                                f"{a.vols} synthetic 4x{a.size}^3 volumes per GPU (BASELINE.json configs[1])")
                   if a.net == "brats" else
                   (f"LiTS 3D-UNet fp32->{a.levels}-level PTQ, 28 quantised convs (widths 32..512), 200 ADMM its/layer, "
                    f"{a.vols} synthetic 1x{a.size}^3 volumes per GPU (BASELINE.json configs[3] geometry)"),
                   "vols_per_gpu": a.vols, "volume": f"{nmod}x{a.size}^3", "parallelism": f"dp{world}",
                   "fp_pass_s": round(res["t1"] - res["t0"], 3), "ptq_pass_s": round(res["t2"] - res["t1"], 3)},
        "roofline": roof,
        "other_kernels": others,
        # (F_conv + F_gram) * N / wall over the f32 matrix peak, with the reference's fp32 ALGORITHMIC flops
        # (the exact-integer kernels do that work on the i8 matrix cores, so the fraction may exceed 1)
        "achieved_mfma": {"algorithmic_tflop_per_volume": tfv, "tflops": round(tfv * total_vols / dt, 1),
                          "peak": PEAK_F32_MFMA_TFLOPS,
                          "frac_of_f32_peak": round(tfv * total_vols / dt / PEAK_F32_MFMA_TFLOPS, 3)},
        # share of the wall clock the ops bracketed on each stream account for (loss and side overlap main)
        "stream_busy_frac": {k: round(v * 1e-3 / dt, 3) for k, v in busy.items()},
    }
    if rank == 0 and world == 1 and exact and not a.no_f32_subrun:
        # like-for-like dtype: the same calibration with every loss and Gram system on the f32 matrix cores
        for m in model.modules():
            if hasattr(m, "lwq_exact_int"):
                m.lwq_exact_int = False
        one_step()
        torch.cuda.synchronize(device)
        t1 = time.time()
        one_step()
        torch.cuda.synchronize(device)
        d1 = time.time() - t1
        out["f32_only"] = {"ms_per_step": round(d1 * 1e3, 1), "value": round(a.vols / d1, 4), "unit": "calib-vols/s",
                           "steps": 1, "dtype": "f32"}
        for m in model.modules():
            if hasattr(m, "lwq_exact_int"):
                m.lwq_exact_int = True
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.net == "brats":
        log("[rank 0] timing the CPU baseline components ...")
        out["cpu_baseline"] = cpu_baseline(a.levels, a.vols, a.size)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
