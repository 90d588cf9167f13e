#!/usr/bin/env python3
"""Headline benchmark: full layer-wise PTQ calibration of the BraTS 3D-UNet (BASELINE.json configs[1]:
fp32 -> 2-bit, qlvl_w=4 qlvl_a=4, 16 synthetic 4x128^3 volumes per GPU).

A "step" is ONE complete calibration (the reference's t2-t0 window, ptqer.py:333-363: FP pass with
target capture + mask pyramid + quantising pass over all 22 quantised convs) of this rank's
volumes, inputs resident in HBM.  With N GPUs the calibration set is vols*N volumes sharded N ways
(data-parallel, Gram / statistics all-reduced over RCCL) => weak scaling; value = total volumes / s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4}] [--vols V] [--size S] [--levels L]

``--gpus N`` with N > 1 starts its own N ranks (``python -m torch.distributed.run --nproc-per-node N``,
one rank per GPU over RCCL) when it was not itself started by a launcher (WORLD_SIZE unset); under a
launcher (WORLD_SIZE set) it is one of the ranks.  The parent process never touches the GPU.
``--config``: 2 = BraTS net, 4/4 levels, 16 volumes per GPU (default, the headline); 3 = BraTS net,
16/16 levels, 8 volumes per GPU (8-way: 64 volumes); 4 = LiTS net, 4/4 levels, 8 volumes 1x160^3 per GPU
(4-way: 32 volumes).  Prints ONE JSON line on rank 0 (progress goes to stderr).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {2: dict(net="brats", levels=4, vols=16, size=128),       # BASELINE.json configs[1]
           3: dict(net="brats", levels=16, vols=8, size=128),       # configs[2]: 64 volumes 8-way
           4: dict(net="lits", levels=4, vols=8, size=160)}         # configs[3]: 32 volumes 4-way


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=2,
                    help="BASELINE.json preset: 2 (default) / 3 / 4, see the module docstring")
    ap.add_argument("--vols", type=int, default=None, help="calibration volumes per GPU (overrides the preset)")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--levels", type=int, default=None)
    ap.add_argument("--net", choices=["brats", "lits"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-subrun", action="store_true")
    ap.add_argument("--no-conv-subrun", action="store_true")
    ap.add_argument("--f32-only", action="store_true",
                    help="evaluate every per-iteration loss on the f32 matrix cores (no exact-integer path)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="only start the ranks, run one all-reduce and report the rank count (no calibration)")
    a = ap.parse_args(argv)
    preset = CONFIGS[a.config]
    for k, v in preset.items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    return a


def launch_ranks(a, argv):
    """--gpus N > 1 outside a launcher: start N ranks as CHILD processes and relay rank 0's JSON line.  This process
    has not touched the GPU (torch is not even imported yet) and never does."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes needs it on this driver)
    print(f"[bench] starting {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        try:
            if out.startswith("{") and "metric" in json.loads(out):
                line = out
                continue
        except ValueError:
            pass
        print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc != 0:
        print(f"[bench] rank launcher exited with {rc}", file=sys.stderr, flush=True)
        return rc
    if line is None:
        print("[bench] the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr, flush=True)
        return 1
    print(line, flush=True)
    return 0


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:
        sys.exit(launch_ranks(_a, sys.argv[1:]))

import ctypes as C  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_I8_MFMA_TOPS = 5033.2        # MI355X dense int8 matrix peak (MI355X_MICROARCH.md: = fp8 dense)
PEAK_F64_MFMA_TFLOPS = 78.6       # MI355X fp64 matrix peak (vendor)
PEAK_BF16_MFMA_TFLOPS = 2516.6    # MI355X dense bf16 matrix peak (MI355X_MICROARCH.md: ~2.5 PF dense = 16 x the f32 rate)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
NET_TFLOP_PER_VOLUME = 24.55      # conv 21.59 + Gram 2.96 TFLOP per 4x128^3 volume (fp32 algorithmic work)
# one iteration in SAMPLE is bracketed by HIP events (an event record is a barrier packet in the queue: ~7 us each)
SAMPLE = int(os.environ.get("EFFQ_BENCH_SAMPLE", "16"))


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def _pmc_mfma_busy():
    """SQ_VALU_MFMA_BUSY_CYCLES share of the prox GEMM measured alone (scripts/pmc_prox.sh -> profiles/*_pmc_prox/summary.txt,
    line `mfma_busy_frac <value>`): the newest committed summary, or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_prox", "summary.txt")), reverse=True):
        try:
            for line in open(path):
                if line.startswith("mfma_busy_frac"):
                    return float(line.split()[1])
        except (OSError, ValueError):
            continue
    return None


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

    KIND = {1: "prox", 2: "fixed_point", 3: "project", 4: "loss", 5: "inverse", 6: "wait"}
    LEVELS = 4          # activation levels of the run (set by main)

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
            if kind == "wait":
                self.wait_ms = getattr(self, "wait_ms", 0.0) + r.ms
                continue
            if kind == "loss" and r.loss_kind == 4:
                key = ("gram_loss@loop", r.c2, r.n)
            elif kind == "loss" and r.loss_kind == 5:
                # one record = one GROUP of iterates (k_gl8 + k_gl8_finish); every second group is bracketed
                V = g.N * g.D * g.H * g.W
                key = ("gram_loss_i8@loop", r.c2, r.n, V)
            elif kind == "loss":
                name = {0: "conv_step", 1: "conv_step_i8", 2: "conv_step_i8s"}[r.loss_kind] + "@loop"
                key = (name, g.N, g.C1, g.C2, g.D, g.H, g.W, g.KD, g.SD)
            else:
                key = (kind, r.c2, r.n)
            self.lib_ms.setdefault(key, []).append(r.ms)
        lib.effq_prof_enable(0)

    @staticmethod
    def _work(key):
        op = key[0].replace("@loop", "")
        if op in ("conv_step", "conv_step_i8", "conv_step_i8s", "gram", "gram_i8"):
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
            if op == "conv_step_i8":
                ops_i8, byts = 2.0 * c2 * c1 * k ** 3 * V, 4.0 * c2 * V + 1.0 * c1 * Vin
                if ops_i8 / byts > PEAK_I8_MFMA_TOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
                    # beyond the ridge (629 op/B): the wide layers on small volumes are bound by the i8 matrix cores and the
                    # L2 stream of their weights, not by HBM (128 -> 128 at 16^3 has ~1400 op/B)
                    return ("mfma", ops_i8, PEAK_I8_MFMA_TOPS, "TOP/s",
                            f"k_conv3d_i8 ({c1}->{c2}, 3^3, {N}x{od}x{oh}x{ow} voxels: 2 c2 c1 27 V int8 op, exact; "
                            f"{ops_i8 / byts:.0f} op per HBM byte)")
                return ("hbm", byts, PEAK_HBM_GBS, "GB/s",
                        f"k_conv3d_i8 ({c1}->{c2}, 3^3, {N}x{od}x{oh}x{ow} voxels: 4*c2 B of target + c1 B of level "
                        f"ids per voxel, i8 MFMA exact)")
            n = c1 * k ** 3 + 1
            if op == "gram_i8":
                return ("mfma", 1.0 * n * n * V + 2.0 * 4 * c2 * n * V, PEAK_I8_MFMA_TOPS, "TOP/s",
                        f"k_gram_i8 (n={n}, {V} voxels; n^2 V (upper triangle) + 8 c2 n V int8 op, exact)")
            return ("mfma", 2.0 * n * n * V + 2.0 * c2 * n * V, PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                    f"k_gram (n={n}, {V} voxels; 2n^2V+2c2nV flop, upper triangle computed)")
        if op == "gram_loss_i8":
            c2, n, V = key[1:]
            nw, grp = n - 1, int(os.environ.get("EFFQ_LOSS_GROUP", "8"))
            P = 1
            while (1 << (8 * P - 1)) - 1 < (OpTimer.LEVELS - 1) ** 2 * V:        # effq_gram_loss_i8_num_planes
                P += 1
            return ("mfma", 2.0 * P * grp * c2 * nw * (nw + 256) / 2.0, PEAK_I8_MFMA_TOPS, "TOP/s",
                    f"k_gl8 + k_gl8_finish (c2={c2}, n={n}: losses of {grp} iterates per launch from the integer Gram system, "
                    f"{P} digit planes x {grp} x c2 nw (nw + 256) int8 op on the upper triangle, exact)")
        c2, n = key[1:]
        if op == "gram_loss":
            return ("mfma", 2.0 * c2 * n * n, PEAK_F64_MFMA_TFLOPS, "TFLOP/s",
                    f"k_gram_loss (c2={c2}, n={n}: loss of an iterate from the unweighted Gram system, 2 c2 n^2 fp64 flop "
                    f"on an n x n matrix instead of a pass over the voxels)")
        if op == "inverse":
            # symmetric Gauss-Jordan: only the tiles on and above the diagonal are updated => n^3 flop, not 2n^3
            return ("mfma", 1.0 * n ** 3, PEAK_F64_MFMA_TFLOPS, "TFLOP/s",
                    f"k_gj_* (n={n}, n^3 fp64 flop: symmetric sweep, upper triangle only)")
        if op == "prox":
            if c2 > 64 and n >= 1024:
                # the GEMM kernel is > 95 % of this bracket (the split-K reduce takes ~5 us, the right-hand side is
                # written by the projection kernel of the previous iteration): reported as a single-kernel op.
                # It runs on the bf16 matrix cores with both fp32 operands split in three (six exact bf16 products per
                # fp32 product, fp32 accumulate): the flops it EXECUTES are 6 x 2 c2 n^2, and that is
                # what is set against the dense bf16 peak
                tile = "256,256,4,2" if c2 > 128 else "128,128,2,4"
                return ("mfma", 6.0 * 2.0 * c2 * n * n, PEAK_BF16_MFMA_TFLOPS, "TFLOP/s",
                        f"k_prox_gemm_b3<{tile}> (+ ~5 us k_prox_reduce4) (c2={c2}, n={n}: the fp32 product 2 c2 n^2 = "
                        f"{2.0 * c2 * n * n / 1e9:.2f} GFLOP evaluated as 6 bf16 MFMA products of operands split in "
                        f"three, {12.0 * c2 * n * n / 1e9:.1f} GFLOP executed)")
            return ("mfma", 2.0 * c2 * n * n, PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                    f"k_prox_gemm + k_prox_reduce (c2={c2}, n={n}; 2 c2 n^2 flop)")
        nw = c2 * (n - 1)
        if op == "fixed_point":
            # one pass over the weights per call is the least an implementation can do
            return ("hbm", 8.0 * nw, PEAK_HBM_GBS, "GB/s", f"weight-scale fixed point ({nw} weights; latency-bound)")
        return ("hbm", 21.0 * nw, PEAK_HBM_GBS, "GB/s", f"k_project_dual ({nw} weights)")

    def summary(self, steps):
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json")))
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
            elif key[0] == "gram_loss_i8@loop":
                launches, stream = len(ms) * 2, "loss"
            else:
                launches, stream = len(ms) * SAMPLE, ("loss" if key[0].endswith("@loop") else "main")
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
                             composite=(key[0] in ("inverse", "fixed_point", "gram_loss_i8@loop") or
                                        (key[0] == "prox" and not (key[1] > 64 and key[2] >= 1024)))))
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
    """The CPU restatement of the reference (oracle/, validated bit for bit against it) timed on this box's host cores.
    Measured for real, on the shapes of the benchmark workload (N = 1 at full resolution, solve-only once):
      * the dominant layer (32 -> 32, 3^3, ONE 64^3 volume): activation scale fit, im2col + Gram, conv + MSE;
      * one torch.linalg.solve per DISTINCT system size of the net (n = 33 ... 6913; the reference solves 200x per layer);
      * project_by_iter on 110 592 weights.
    Volume-dependent parts are exactly linear in the voxel count (im2col, Gram, conv, activation fit) and scaled by
    it; the solves are volume-independent and summed as measured.  ~30-60 s of CPU work."""
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
    layers = brats_layers(vols, size)

    def timeit(fn, budget=3.0, min_rep=1, warm=True):
        if warm:
            fn()
        t0, rep = time.time(), 0
        while rep < min_rep or time.time() - t0 < budget:
            fn()
            rep += 1
            if rep >= 50:
                break
        return (time.time() - t0) / rep

    # ---- the dominant layer at full resolution, ONE volume (32 -> 32, 3^3, 64^3 voxels) ----
    c, S = 32, size // 2
    x1 = torch.relu(torch.randn(1, c, S, S, S, generator=gen))
    y1 = torch.randn(1, c, S, S, S, generator=gen)
    w1 = torch.randn(c, c, 3, 3, 3, generator=gen) * 0.05
    t0 = time.time()
    fit = O.fit_scale(x1, levels, 0.0, 1.0)                   # EfficientQConv.py:68 (fp64, all voxels)
    t_act = time.time() - t0
    it_a = fit.iters
    xq1 = (fit.alpha * fit.b)
    del fit
    meas[f"project_by_iter_32ch_{S}^3_activations_s"] = round(t_act, 3)
    t0 = time.time()
    ps = O.ProxSystem(xq1, y1, (3, 3, 3), 1, 1, torch.zeros(c, c, 3, 3, 3), torch.zeros(c), None)   # solver.py:253-314
    t_gram = time.time() - t0
    del ps
    n0 = c * 27 + 1
    meas[f"im2col_gram_32ch_{S}^3_s"] = round(t_gram, 3)
    gram_s_per_gflop = t_gram / ((2.0 * n0 * n0 + 2.0 * c * n0) * S ** 3 / 1e9)
    t_conv = timeit(lambda: F.mse_loss(F.conv3d(xq1, w1, None, 1, 1), y1).item(), 2.0)            # :118-122
    meas[f"conv_mse_32ch_{S}^3_s"] = round(t_conv, 5)
    conv_s_per_gflop = t_conv / (2.0 * c * c * 27 * S ** 3 / 1e9)
    act_s_per_elem = t_act / x1.numel()
    del x1, y1, xq1
    # ---- one LU solve per distinct system size (solver.py:331) ----
    sol = {}
    for (c1, c2, k, s, V, Vin, q) in layers:
        n = c1 * k ** 3 + 1
        if n in sol:
            continue
        A = torch.randn(n, n, generator=gen)
        A = A @ A.T + n * torch.eye(n)
        B = torch.randn(c2, n, generator=gen)
        sol[n] = timeit(lambda: torch.linalg.solve(A, B.T), 1.0, 1, warm=(n < 3000))
        meas[f"lu_solve_n{n}_s"] = round(sol[n], 5)
        del A, B
    # ---- project_by_iter on weights (layer_helper.py:40-70), fp64 ----
    wv = torch.randn(110592, generator=gen) * 0.05
    t_fit = timeit(lambda: O.fit_scale(wv, levels, -1.0, 1.0), 2.0)
    it_w = O.fit_scale(wv, levels, -1.0, 1.0).iters
    fit_s_per_elem_iter = t_fit / (110592 * it_w)
    meas["project_by_iter_110592_weights_s"] = round(t_fit, 4)
    total = dict(conv=0.0, gram=0.0, solve=0.0, proj_w=0.0, proj_a=0.0)
    for (c1, c2, k, s, V, Vin, q) in layers:
        n = c1 * k ** 3 + 1
        total["conv"] += 201 * conv_s_per_gflop * (2.0 * c1 * c2 * k ** 3 * V / 1e9)
        total["gram"] += gram_s_per_gflop * ((2.0 * n * n + 2.0 * c2 * n) * V / 1e9)
        total["solve"] += 200 * sol[n]
        itw = it_w if (c1 != 4 and c2 != 3) else 300           # first / last layer keep 256 weight levels
        total["proj_w"] += 200 * fit_s_per_elem_iter * c2 * c1 * k ** 3 * itw
        if q:
            total["proj_a"] += act_s_per_elem * c1 * Vin
    est = sum(total.values())
    return dict(value=round(vols / est, 6), unit="calib-vols/s", cores=cores, kind="port",
                est_seconds_per_calibration=round(est, 1),
                components_s={k: round(v, 1) for k, v in total.items()}, measured=meas,
                sample=f"oracle timed on this box's {cores} host threads at the workload's own shapes: the dominant layer "
                       f"(32->32 3^3) on ONE {S}^3 volume (activation fit {it_a} its, im2col + Gram, conv + MSE), one "
                       f"torch.linalg.solve per distinct system size n = {sorted(sol)} (200 per layer in the reference), "
                       f"project_by_iter on 110592 weights ({it_w} its); volume-dependent parts scaled linearly in the voxel "
                       f"count to {vols} x 4x{size}^3, solves summed as measured; hooks / copies / std() left out "
                       "(favours the CPU)")


def spawn_check(a, world, rank, local):
    """--spawn-check: the launch path only (ranks start, rendezvous, one all-reduce), no calibration."""
    use_gpu = torch.cuda.is_available()
    backend = os.environ.get("EFFQ_BENCH_BACKEND", "nccl" if use_gpu else "gloo")
    ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            t = torch.ones(1, device=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
            t = torch.ones(1)
        dist.all_reduce(t)
        ranks = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "spawn_check", "n_gpus": world, "rccl_ranks": ranks, "backend": backend}), flush=True)


def main():
    a = parse_args()
    if a.f32_only:
        os.environ["EFFQ_EXACT_INT"] = "0"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if a.spawn_check:
        return spawn_check(a, world, rank, local)
    # EFFQ_BENCH_BACKEND=gloo: REHEARSAL of the N > 1 flow on a box with fewer GPUs than ranks (ranks share devices,
    # collectives staged through the host by qconv.SumReducer); the product backend is "nccl" = RCCL over xGMI
    backend = os.environ.get("EFFQ_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs visible (one rank per GPU over RCCL)")
    dev_index = local if backend == "nccl" else local % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    rccl_ranks = 1
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # the process group is the rendezvous; the collectives of the calibration (and the ones below) are RCCL calls
            # on the kernels' own stream (efficientq_amd/rccl.py).  No device_id: an eagerly created framework communicator
            # brings HIP streams of its own, which alias the calibration's streams onto shared hardware queues (+20 % per
            # calibration, measured on one GPU with a 1-rank group that was never used)
            dist.init_process_group("nccl")
            from efficientq_amd import rccl as _rccl
            comm = _rccl.get_comm(None)
        if comm is not None:
            rccl_ranks = comm.barrier(device)        # an actual all-reduce over RCCL: the rank count it reports
        else:
            if not dist.is_initialized():
                dist.init_process_group(backend)
            probe = torch.ones(1, device=device) if backend == "nccl" else torch.ones(1)
            dist.all_reduce(probe)
            rccl_ranks = int(probe.item())
        if rccl_ranks != dist.get_world_size() or rccl_ranks != a.gpus:
            raise SystemExit(f"all-reduce over {backend} saw {rccl_ranks} ranks, expected {a.gpus}")

    # EFFQ_DP_FORCE=1 on ONE rank: every data-parallel collective of the path is issued on RCCL (a 1-rank group: each
    # all-reduce is the identity, but it costs what issuing it costs - the stream hop, the event records, the launch);
    # the step-time difference to a plain run is the issue cost of the data-parallel path, measurable on one GPU
    dp_forced = world == 1 and os.environ.get("EFFQ_DP_FORCE", "0") == "1"
    if dp_forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        dist.init_process_group("nccl", rank=0, world_size=1)

    from efficientq_amd import calibrate as K, synth
    from efficientq_amd.hip_ops import get_ops
    args, model = build_model(a.levels, device, a.net)
    pristine = {k: v.clone() for k, v in model.state_dict().items()}
    t = time.time()
    ids = range(rank * a.vols, (rank + 1) * a.vols)           # rank r holds its own shard of the volumes
    vols = synth.calib_batch(a.net, ids, a.size).to(device)
    nmod = 4 if a.net == "brats" else 1
    log(f"[rank {rank}] {a.vols} synthetic volumes {nmod}x{a.size}^3 in HBM ({time.time() - t:.1f}s)")

    ops = get_ops(device)
    OpTimer.LEVELS = a.levels
    timer = OpTimer()

    def one_step():
        model.load_state_dict(pristine, strict=True)
        return K.calibrate_model(model, vols, a.net, args.init_stride)

    def fence():
        if world > 1 and comm is not None:
            comm.barrier(device)
        elif world > 1:
            dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(a.warmup):
        t = time.time()
        one_step()
        torch.cuda.synchronize(device)
        log(f"[rank {rank}] warmup {i}: {time.time() - t:.2f}s")
    unwrap = timer.wrap(ops)
    fence()
    from efficientq_amd.qconv import SumReducer as _SR
    _SR.calls = 0
    t0 = time.time()
    res = None
    for i in range(a.steps):
        res = one_step()
    fence()
    dt = time.time() - t0
    coll_per_step = _SR.calls / max(1, a.steps)
    unwrap()
    timer.collect_library()
    tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1 and comm is not None:
        comm.all_reduce_max_(tmax)
    elif world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    total_vols = a.vols * world * a.steps
    roof, others, busy = timer.summary(a.steps)
    from efficientq_amd import qconv as _Q
    exact = bool(_Q.EXACT_INT_DEFAULT)
    if roof is not None:
        roof = dict(roof)
        if "k_prox_gemm_b3" in roof["kernel"]:
            # the same launch on the fp32 scale (the product it evaluates, against the f32 matrix peak it replaced) and the
            # share of the SIMD cycles its matrix cores were busy (rocprofv3 PMC of the kernel alone, profiles/r03_pmc_prox)
            alg = roof["work_per_launch"] / 6.0
            roof["fp32_equivalent"] = {"achieved": round(alg / (roof["avg_ms"] * 1e-3) / 1e12, 2),
                                       "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(alg / (roof["avg_ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                       "note": "2 c2 n^2 flop of the fp32 product per launch; round 2 ran it on the f32 "
                                               "matrix cores at 0.55 of their peak"}
            busy_alone = _pmc_mfma_busy()
            if busy_alone is not None:
                roof["mfma_busy_frac_alone"] = busy_alone          # read from the committed PMC summary, not a literal
            roof["note"] = ("frac = executed bf16 MFMA flop / dense bf16 peak quoted at 2.4 GHz; the chip holds 1.5 - 1.7 GHz "
                            "in dense bf16 loops on random data (MI355X_MICROARCH.md, DVFS give-back)")
    # 24.55 TFLOP per 4x128^3 BraTS volume, 99.69 per 1x160^3 LiTS volume (conv x 201 + Gram), scaled by voxels
    tfv = (NET_TFLOP_PER_VOLUME * (a.size / 128.0) ** 3) if a.net == "brats" else (99.69 * (a.size / 160.0) ** 3)
    baseline_cfg = {2: "configs[1]", 3: "configs[2]", 4: "configs[3]"}[a.config]
    preset = CONFIGS[a.config]
    if any(getattr(a, k) != v for k, v in preset.items()):
        baseline_cfg += " modified by flags"
    out = {
        "metric": "ptq_calibration_throughput", "value": round(total_vols / dt, 5), "unit": "calib-vols/s",
        "n_gpus": world, "rccl_ranks": rccl_ranks, "collective_backend": (backend if world > 1 else None),
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
        "wall_clock_s_per_calibration": round(dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i8xi8->i32 (exact) for the Gram systems and per-iteration loss convs of the layers with quantised input, f32/f64 elsewhere"
                 if exact else "f32", "data": "synthetic",
        "config": {"workload": (f"BraTS 3D-UNet fp32->{a.levels}-level PTQ (qlvl_w={a.levels} qlvl_a={a.levels}, "
                                f"q_first=q_last=256,-1), 22 quantised convs, 200 ADMM its/layer, "
                                f"{a.vols} synthetic 4x{a.size}^3 volumes per GPU (BASELINE.json {baseline_cfg})")
                   if a.net == "brats" else
                   (f"LiTS 3D-UNet fp32->{a.levels}-level PTQ, 28 quantised convs (widths 32..512), 200 ADMM its/layer, "
                    f"{a.vols} synthetic 1x{a.size}^3 volumes per GPU (BASELINE.json {baseline_cfg})"),
                   "preset": a.config, "vols_per_gpu": a.vols, "total_vols": a.vols * world,
                   "volume": f"{nmod}x{a.size}^3", "levels": a.levels, "parallelism": f"dp{world}",
                   "fp_pass_s": round(res["t1"] - res["t0"], 3), "ptq_pass_s": round(res["t2"] - res["t1"], 3)},
        "roofline": roof,
        "other_kernels": others,
        # NOT a utilisation figure: the fp32 flops the REFERENCE spends on this workload (201 convs per layer
        # + Gram) per second of OUR wall clock.  14 of 22 layers take their 200 losses from the Gram system and the rest
        # run on the i8 matrix cores, so most of that work is not executed here at all.
        "reference_equivalent": {"algorithmic_tflop_per_volume": tfv, "tflops": round(tfv * total_vols / dt, 1),
                                 "note": "reference fp32 flops (201 conv+MSE per layer + Gram) / our wall clock; "
                                         "not a roofline fraction - most of that work is not executed on this path"},
        # share of the wall clock the ops bracketed on each stream account for (loss and side overlap main)
        "stream_busy_frac": {k: round(v * 1e-3 / dt, 3) for k, v in busy.items()},
        # measured WITHOUT a profiler: HIP-event pairs on the main stream around every point where the chain waits for another
        # stream (the inverse of the next rho, the joins that end a layer), summed over the timed steps
        "dp_forced": ({"collectives_per_step": coll_per_step, "rccl_direct": os.environ.get("EFFQ_RCCL_DIRECT", "1") != "0", "layers": 22 if a.net == "brats" else 28,
                       "note": "EFFQ_DP_FORCE=1: one rank, every collective of the data-parallel path issued on RCCL"}
                      if dp_forced else None),
        "main_queue_wait": {"ms_per_step": round(getattr(timer, "wait_ms", 0.0) / a.steps, 2),
                            "frac_of_step": round(getattr(timer, "wait_ms", 0.0) * 1e-3 / dt, 4)},
    }
    solo = rank == 0 and world == 1
    if solo and exact and not a.no_conv_subrun and _Q.GRAM_LOSS_DEFAULT:
        # north_star's named kernel (conv3d_quant_calib_step, here its exact-integer form k_conv3d_i8l2e) stays in view:
        # the same calibration with the per-iteration losses by conv passes over the voxels (EFFQ_GRAM_LOSS=0)
        _Q.GRAM_LOSS_DEFAULT = False
        try:
            one_step()
            torch.cuda.synchronize(device)
            t2 = OpTimer()
            un2 = t2.wrap(ops)
            t1 = time.time()
            one_step()
            torch.cuda.synchronize(device)
            d1 = time.time() - t1
            un2()
            t2.collect_library()
            r2, o2, _ = t2.summary(1)
            conv_rows = [r for r in ([r2] + o2 if r2 else []) if r["kernel"].startswith("k_conv3d_i8 (32->32")]
            out["conv_path"] = {"ms_per_step": round(d1 * 1e3, 1), "value": round(a.vols / d1, 4), "unit": "calib-vols/s",
                                "steps": 1, "note": "EFFQ_GRAM_LOSS=0: every per-iteration loss by a conv pass",
                                "roofline": (conv_rows[0] if conv_rows else None)}
        finally:
            _Q.GRAM_LOSS_DEFAULT = True
    if solo and exact and not a.no_f32_subrun:
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
    if solo and not a.no_cpu_baseline and a.net == "brats":
        log("[rank 0] timing the CPU baseline components ...")
        out["cpu_baseline"] = cpu_baseline(a.levels, a.vols, a.size)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or dp_forced:
        if comm is not None:
            comm.barrier(device)
        elif world > 1:
            dist.barrier()
        from efficientq_amd import rccl as _rccl
        _rccl.close_all()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
