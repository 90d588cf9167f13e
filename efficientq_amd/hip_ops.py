"""Tensor-level front end of the C-ABI library: torch tensors supply device memory and
the HIP stream; every computation is a call into ``libeffq_hip.so``.

No CPU fallback exists here on purpose: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import Geom, check

ADMM_TOL = 1e-5   # layer_helper.py:55
# tensors from this size on are fitted by the bracketed fixed point (effq_fp_bracket_*)
import os as _os
FP_BRACKET_MIN = int(_os.environ.get("EFFQ_FP_BRACKET_MIN", 1 << 18))
COOP_FIXED_POINT = _os.environ.get("EFFQ_COOP_FP", "1") != "0"
# data-parallel activation fit: after DP_GATHER_AFTER all-reduced iterations the ranks exchange their tallies and undecided
# lists once and finish on their own (effq_fp_bracket_export / _import); lists longer than DP_GATHER_MAX_BYTES in all: more
# all-reduced iterations first.  EFFQ_DP_GATHER_FIT=0: one all-reduce per iteration (A/B)
DP_GATHER_FIT = _os.environ.get("EFFQ_DP_GATHER_FIT", "1") != "0"
DP_GATHER_AFTER = int(_os.environ.get("EFFQ_DP_GATHER_AFTER", "4"))
DP_GATHER_MAX_BYTES = int(_os.environ.get("EFFQ_DP_GATHER_MAX_BYTES", str(128 << 20)))
BUCKET_FIXED_POINT = _os.environ.get("EFFQ_BUCKET_FP", "1") != "0"
TRAJ_FIXED_POINT = _os.environ.get("EFFQ_FP_TRAJ", "1") != "0"
SIDE2_STREAM = _os.environ.get("EFFQ_SIDE2", "1") != "0"
SIDE_STREAM = _os.environ.get("EFFQ_SIDE", "1") != "0"      # 0: every inverse on the main stream, ahead of the loop (diagnostic)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def to_ndhwc(x: torch.Tensor) -> torch.Tensor:
    """(N,C,D,H,W) logical -> contiguous (N,D,H,W,C) storage (no copy if already channels_last_3d)."""
    return x.permute(0, 2, 3, 4, 1).contiguous()


def from_ndhwc(t: torch.Tensor) -> torch.Tensor:
    """contiguous (N,D,H,W,C) -> logical (N,C,D,H,W) view (channels_last_3d strides)."""
    return t.permute(0, 4, 1, 2, 3)


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(i) for i in v)


def make_geom(x_shape_ncdhw, c2: int, ksize, stride, padding) -> Geom:
    n, c1, d, h, w = (int(i) for i in x_shape_ncdhw)
    k, s, p = _triple(ksize), _triple(stride), _triple(padding)
    return Geom(n, c1, int(c2), d, h, w, k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2])


def _check_shapes(geom: Geom, x, w=None, bias=None, y=None, att=None):
    """Host-side guard: buffer sizes must match what the kernels index (an out-of-bounds access
    on the device can take the whole node down)."""
    od, oh, ow = geom.out_dims()
    if min(od, oh, ow) <= 0:
        raise _lib.EffqError(f"empty conv output for geometry {[getattr(geom, f[0]) for f in geom._fields_]}")
    want = {"x": (x, geom.N * geom.D * geom.H * geom.W * geom.C1),
            "weight": (w, geom.C2 * geom.C1 * geom.KD * geom.KH * geom.KW),
            "bias": (bias, geom.C2),
            "y": (y, geom.N * od * oh * ow * geom.C2),
            "att": (att, geom.N * od * oh * ow)}
    for name, (t, n) in want.items():
        if t is not None and t.numel() != n:
            raise _lib.EffqError(f"{name} has {t.numel()} elements, geometry needs {n}")


class HipOps:
    """Per-device handle: stream + library-owned workspaces (grown on demand, never shrunk)."""

    def __init__(self, device: torch.device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.EffqError(f"efficientq_amd runs on a HIP device only, got {device} (no CPU fallback)")
        self.lib = _lib.load()
        self.device = device
        self._red_ws = torch.zeros(self.lib.effq_reduce_ws_bytes(), dtype=torch.uint8, device=device)
        self._ws = {}
        self._att_cache = {}
        self._pinned_stream = None
        self._pinned_torch = None
        self._ws_retired = []

    # -- plumbing ---------------------------------------------------------------------------
    @property
    def stream(self):
        # looking the current stream up through torch costs ~4 us per op; a caller that issues hundreds of ops on
        # a known stream pins it with on_stream()
        if self._pinned_stream is not None:
            return self._pinned_stream
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def on_stream(self, stream: Optional["torch.cuda.Stream"]):
        """Pin the HIP stream the following ops launch on (None: follow torch's current stream again).
        Returns the previous pin so that callers can restore it."""
        prev = (self._pinned_stream, self._pinned_torch)
        self._pinned_stream = None if stream is None else C.c_void_p(stream.cuda_stream)
        self._pinned_torch = stream
        return prev

    def restore_stream(self, pin):
        self._pinned_stream, self._pinned_torch = pin if pin is not None else (None, None)

    def loss_stream(self):
        """The stream the per-iteration loss evaluation runs on, one iteration behind the ADMM chain."""
        if getattr(self, "_loss", None) is None:
            self._loss = torch.cuda.Stream(self.device)
        return self._loss

    def side_stream(self):
        """A second HIP stream of this device for work that is independent of the calibration stream."""
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    def side_stream2(self):
        """A third stream: the later inverses of a layer alternate between the two side streams."""
        if getattr(self, "_side2", None) is None:
            self._side2 = torch.cuda.Stream(self.device)
        return self._side2

    def warm_streams(self):
        """Create every stream the calibration uses NOW, in a fixed order: HIP hands out hardware queues in creation order
        (4 by default), and streams that share one run their kernels one after the other.  Created lazily - after a
        communicator had brought its own streams - the two side streams of the inverses landed on one queue (the later
        inverses of the 256-channel layers 44 / 63 ms instead of 28 / 32, +5 % per calibration with a 1-rank RCCL group);
        rccl.DirectComm calls this before ncclCommInitRank."""
        if getattr(self, "_warm", False):
            return
        main = torch.cuda.current_stream(self.device)
        # which streams get their helper, in which order, is empirical (per calibration with a 1-rank RCCL group, one box: none
        # 693 ms, main / loss / side / side2 675, main / side / side2 - the order of a run without a communicator - 691; plain
        # run 657; GPU_MAX_HW_QUEUES = 6 / 8: 766 / 753).  EFFQ_WARM_ORDER = 0 / 1 / 2 selects them (tuning aid)
        order = _os.environ.get("EFFQ_WARM_ORDER", "1")
        loss, side, side2 = self.loss_stream(), self.side_stream(), self.side_stream2()
        sts = () if order == "0" else (main, loss, side, side2) if order == "1" else (main, side, side2)
        for st in sts:
            check(self.lib.effq_spd_inverse_prepare(st.cuda_stream), "effq_spd_inverse_prepare")
        self._warm = True

    def _workspace(self, key: str, nbytes: int) -> torch.Tensor:
        """Library workspace `key`, zero-filled when (re)allocated.  The fill is issued on the stream the NEXT op
        launches on (the pinned stream when one is set): a fill on torch's current stream would race with kernels
        of a pinned loss stream that only waited for an event recorded before the fill.  A replaced buffer stays
        referenced until release_retired() (kernels of another stream may still be reading it, and the caching
        allocator only orders a block against the stream it was allocated on)."""
        cur = self._ws.get(key)
        if cur is None or cur.numel() < nbytes:
            if cur is not None:
                self._ws_retired.append(cur)
            if self._pinned_torch is not None:
                with torch.cuda.stream(self._pinned_torch):
                    cur = torch.zeros(int(nbytes), dtype=torch.uint8, device=self.device)
            else:
                cur = torch.zeros(int(nbytes), dtype=torch.uint8, device=self.device)
            self._ws[key] = cur
        return cur

    def reserve(self, key: str, nbytes: int):
        """Size workspace `key` ahead of a multi-stream section (on the current stream, before its events)."""
        self._workspace(key, nbytes)

    def release_retired(self):
        """Drop replaced workspaces; call only where every stream of this handle has been joined."""
        self._ws_retired.clear()

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        if t.device != self.device and not (t.device.type == "cuda" and self.device.index in (None, t.device.index)):
            raise _lib.EffqError(f"tensor on {t.device}, ops on {self.device}")
        if t.dtype != torch.float32:
            raise _lib.EffqError(f"expected float32, got {t.dtype}")
        return t if t.is_contiguous() else t.contiguous()

    # -- a1/a3 --------------------------------------------------------------------------------
    def quant_dequant_f32(self, x: torch.Tensor, alpha: torch.Tensor, levels: int, lo: float, hi: float,
                          want_idx: bool = False):
        """PTQConv._quantize_act / discretize in fp32 (PTQConv.py:114-116).  alpha: 0-dim device tensor."""
        x = self._f32(x)
        a = self._f32(alpha.reshape(1))
        y = torch.empty_like(x)
        idx = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_idx else None
        check(self.lib.effq_quant_dequant_f32(_ptr(x), _ptr(a), lo, hi, levels, _ptr(y), _ptr(idx), x.numel(),
                                              self.stream), "effq_quant_dequant_f32")
        return (y, idx) if want_idx else y

    def quant_dequant_f64path(self, x: torch.Tensor, state: torch.Tensor, levels: int, lo: float, hi: float,
                              want_b: bool = False, want_idx: bool = False):
        """a*b of project_by_iter (layer_helper.py:66, EfficientQConv.py:70); alpha = state[0] (device double)."""
        x = self._f32(x)
        y = torch.empty_like(x)
        b = torch.empty_like(x) if want_b else None
        idx = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_idx else None
        check(self.lib.effq_quant_dequant_f64path(_ptr(x), _ptr(state), lo, hi, levels, _ptr(y), _ptr(b), _ptr(idx),
                                                  x.numel(), self.stream), "effq_quant_dequant_f64path")
        return y, b, idx

    # -- a2 -----------------------------------------------------------------------------------
    def abs_sum(self, x: torch.Tensor) -> torch.Tensor:
        x = self._f32(x)
        out = torch.empty(2, dtype=torch.float64, device=x.device)
        check(self.lib.effq_abs_sum_f64(_ptr(x), x.numel(), _ptr(out), _ptr(self._red_ws), self.stream),
              "effq_abs_sum_f64")
        return out

    def moments(self, x: torch.Tensor) -> torch.Tensor:
        """[sum, sum of squares, count] in fp64 on the device."""
        x = self._f32(x)
        out = torch.empty(3, dtype=torch.float64, device=x.device)
        check(self.lib.effq_moments_f64(_ptr(x), x.numel(), _ptr(out), _ptr(self._red_ws), self.stream),
              "effq_moments_f64")
        return out

    def new_fp_state(self) -> torch.Tensor:
        # effq_fp_state viewed as 5 doubles: alpha, alpha_prev, sums[2], {iters,done}
        return torch.zeros(5, dtype=torch.float64, device=self.device)

    @staticmethod
    def read_fp_state(state: torch.Tensor):
        host = state.cpu()
        iters, done = host[4:5].view(torch.int32).tolist()
        return float(host[0]), int(iters), int(done)

    def fit_scale(self, x: torch.Tensor, levels: int, lo: float, hi: float, reducer=None,
                  guess_iters: int = 16, state: Optional[torch.Tensor] = None, abs_sums: Optional[torch.Tensor] = None):
        """project_by_iter on the device (layer_helper.py:40-70).

        ``reducer`` (callable on a device fp64 tensor, in place) sums statistics over data-parallel
        ranks; with it the statistics of every iteration are all-reduced before the update.
        Returns (alpha: float, iters: int, state tensor).  Raises RuntimeWarning like the
        reference when the cap 100*L is hit.
        """
        x = self._f32(x)
        n = x.numel()
        st = state if state is not None else self.new_fp_state()
        cap = 100 * levels
        if abs_sums is not None:
            s0 = abs_sums                      # [sum|x|, n], already summed over the data-parallel ranks by the caller
        else:
            s0 = self.abs_sum(x)
            if reducer is not None:
                reducer(s0)
        batch = max(4, int(guess_iters))
        if n >= FP_BRACKET_MIN and levels <= 256:
            return self._fit_scale_bracket(x, n, levels, lo, hi, reducer, batch, st, s0, cap)
        check(self.lib.effq_fp_init(_ptr(st), _ptr(s0), self.stream), "effq_fp_init")
        while True:
            if reducer is None:
                check(self.lib.effq_alpha_fixed_point(_ptr(x), n, levels, lo, hi, ADMM_TOL, cap, batch, _ptr(st),
                                                      _ptr(self._red_ws), self.stream), "effq_alpha_fixed_point")
            else:
                sums = st[2:4]
                done = st[4:5]
                for _ in range(batch):
                    check(self.lib.effq_alpha_stats_f64(_ptr(x), _ptr(st), lo, hi, levels, n, _ptr(sums),
                                                        C.c_void_p(done.data_ptr() + 4), _ptr(self._red_ws),
                                                        self.stream), "effq_alpha_stats_f64")
                    reducer(sums)
                    check(self.lib.effq_fp_update(_ptr(st), ADMM_TOL, cap, self.stream), "effq_fp_update")
            alpha, iters, done = self.read_fp_state(st)   # one host sync per batch
            if done == 1:
                return alpha, iters, st
            if done == 2:
                raise RuntimeWarning(f"Exceed maximum iteration ({cap}) for alpha optimization")
            batch = min(max(8, iters // 2), 256)

    def _fit_scale_bracket(self, x, n, levels, lo, hi, reducer, batch, st, s0, cap):
        """fit_scale on a large tensor by the bracketed fixed point (effq_fp_bracket_*): after the first iterations only
        the values whose level can still change are read.  Same iterates as the per-iteration passes."""
        ws = self._workspace("fp_bracket", self.lib.effq_fp_bracket_ws_bytes(n))
        # unsigned quantiser = post-ReLU input: the first pass already drops the exact zeros
        check(self.lib.effq_fp_bracket_init(_ptr(st), _ptr(s0), n, levels, int(lo == 0.0), _ptr(ws), ws.numel(),
                                            self.stream), "effq_fp_bracket_init")
        gather = (reducer is not None and DP_GATHER_FIT and lo == 0.0 and hasattr(reducer, "all_gather"))
        dp_iters = DP_GATHER_AFTER
        while True:
            if reducer is None:
                check(self.lib.effq_fp_bracket_run(_ptr(x), n, levels, lo, hi, ADMM_TOL, cap, batch, _ptr(st), _ptr(ws),
                                                   self.stream), "effq_fp_bracket_run")
            else:
                sums = st[2:4]
                for _ in range(dp_iters if gather else batch):
                    check(self.lib.effq_fp_bracket_stats(_ptr(x), n, levels, lo, hi, _ptr(st), _ptr(ws), self.stream),
                          "effq_fp_bracket_stats")
                    reducer(sums)
                    check(self.lib.effq_fp_bracket_update(n, levels, lo, hi, ADMM_TOL, cap, _ptr(st), _ptr(ws),
                                                          self.stream), "effq_fp_bracket_update")
                if gather:
                    r = self._fit_scale_gathered(x, n, levels, lo, hi, reducer, st, ws, cap)
                    if not isinstance(r, int):
                        return r
                    dp_iters = r                   # not usable yet (or the gathered fit left its bracket): more all-reduced
                    continue                       # iterations first
            alpha, iters, done = self.read_fp_state(st)   # one host sync per batch
            if done == 1:
                return alpha, iters, st
            if done == 2:
                raise RuntimeWarning(f"Exceed maximum iteration ({cap}) for alpha optimization")
            batch = min(max(8, iters // 2), 256)

    def _fit_scale_gathered(self, x, n, levels, lo, hi, reducer, st, ws, cap):
        """Data-parallel ranks, "gather once" (effq_fp_bracket_export / _import): what is left of every rank's shard under
        the current bracket - four integer tallies and the list of the undecided values - is exchanged with TWO collectives
        (tallies + list lengths + brackets in one all-reduce, the zero-padded lists in one all-gather of a fixed size), and
        every rank finishes the fit on its own, bit-identically.  No host round trip on the way: whether the exchange was
        usable is decided on the device.  Returns (alpha, iters, state), or - when the fit must go on with all-reduced
        iterations: no list yet / lists too long / horizon bracket / an iterate left the bracket of the import - the number of
        such iterations to run before the next attempt."""
        world, rank = reducer.world, reducer.rank
        dev = self.device
        slot = max(4096, min((n + 3) // 4 * 4, DP_GATHER_MAX_BYTES // 4 // max(1, world)))      # floats per rank
        lst = self._workspace("fp_gather_list", 4 * slot).view(torch.float32)
        exp = torch.empty(self.lib.effq_fp_bracket_export_words(), dtype=torch.int64, device=dev)
        check(self.lib.effq_fp_bracket_export(_ptr(ws), n, _ptr(exp), _ptr(lst), slot, self.stream),
              "effq_fp_bracket_export")
        # one int64 message: [0..3] tallies (summed), then one slot triple per rank - list length, bit patterns of the
        # bracket its tallies are valid under (the other ranks add zeros)
        pack = torch.zeros(4 + 3 * world, dtype=torch.int64, device=dev)
        pack[:4] = exp[:4]
        pack[4 + 3 * rank: 7 + 3 * rank] = exp[4:7]
        reducer(pack)
        gathered = reducer.all_gather(lst[:slot])               # second (and last) message of the fit
        m = world * slot
        ws2 = self._workspace("fp_bracket_gathered", self.lib.effq_fp_bracket_ws_bytes(m))
        check(self.lib.effq_fp_bracket_import(_ptr(ws), n, _ptr(pack), world, slot, _ptr(st), _ptr(ws2), ws2.numel(),
                                              self.stream), "effq_fp_bracket_import")
        batch = 12 * levels
        while True:
            check(self.lib.effq_fp_bracket_run(_ptr(gathered), m, levels, lo, hi, ADMM_TOL, cap, batch, _ptr(st), _ptr(ws2),
                                               self.stream), "effq_fp_bracket_run")
            alpha, iters, done = self.read_fp_state(st)         # the fit's one host read (in the common case)
            if done == 1:
                return alpha, iters, st
            if done == 2:
                raise RuntimeWarning(f"Exceed maximum iteration ({cap}) for alpha optimization")
            if done == 4:                                       # exchange not usable / left the bracket of the import:
                check(self.lib.effq_fp_bracket_rebase(_ptr(st), _ptr(ws), n, self.stream), "effq_fp_bracket_rebase")
                # on with all-reduced iterations, from the base - as many as the lists need to shrink into their slots (the
                # undecided share falls by ~0.57 every two iterations at few levels: profiles/r03_fp_bracket_trace_*)
                lens = pack[4::3].cpu().tolist()
                worst = max(lens) if min(lens) >= 0 else 0
                return DP_GATHER_AFTER if worst <= slot else max(2, min(32, int(math.ceil(3.6 * math.log(worst / slot))) + 1))
            batch = min(2 * batch, 256)

    def fp_bracket_diagnostics(self):
        """The header of the bracketed fixed point's workspace after a fit (tests, tuning)."""
        ws = self._ws["fp_bracket"]
        f = ws[:256].view(torch.float64).cpu()
        i = ws[:256].view(torch.int64).cpu()
        return {"blo": float(f[0]), "bhi": float(f[1]), "src": int(i[7]), "G": int(i[10]), "per": int(i[11]),
                "escapes": int(i[12]), "narrowings": int(i[13]), "visited": int(i[14]), "list_total": int(i[15]),
                "density": float(f[17]), "widen": int(i[20]), "base": int(i[21]), "z_total": int(i[22])}

    def weight_fixed_point(self, wstar, dual, v, levels: int, state, guess: int = 16):
        """Projection input v = wstar + dual and its scale fixed point (EfficientQConv.py:108).
        Small tensors: ONE launch, no host round trip (returns None).  Large ones: fused iterations in
        batches with one 40-byte read per batch (returns the iteration count)."""
        n = wstar.numel()
        if n <= self.lib.effq_fp_small_max():
            check(self.lib.effq_fixed_point_small(_ptr(wstar), _ptr(dual), _ptr(v), n, levels, -1.0, 1.0, ADMM_TOL,
                                                  100 * levels, _ptr(state), self.stream), "effq_fixed_point_small")
            return None
        if n <= self.lib.effq_fp_coop_max() and COOP_FIXED_POINT:
            check(self.lib.effq_fixed_point_coop(_ptr(wstar), _ptr(dual), _ptr(v), n, levels, -1.0, 1.0, ADMM_TOL,
                                                 100 * levels, _ptr(state), _ptr(self._red_ws), self.stream),
                  "effq_fixed_point_coop")
            return None
        self.admm_presum(wstar, dual, v)
        _, it, _ = self.fit_scale(v, levels, -1.0, 1.0, guess_iters=guess, state=state)
        return it

    def fixed_point_bucket(self, a, b, v, levels: int, state, lo: float = -1.0, hi: float = 1.0):
        """project_by_iter of v = a + b (b may be None) on the bucketed copy: one workgroup, one launch
        (effq_fixed_point_bucket)."""
        n = a.numel()
        ws = self._workspace("fp_bucket", self.lib.effq_fp_bucket_ws_bytes(n))
        check(self.lib.effq_fixed_point_bucket(_ptr(a), _ptr(b), _ptr(v), n, levels, lo, hi, ADMM_TOL, 100 * levels,
                                               _ptr(state), _ptr(ws), ws.numel(), self.stream),
              "effq_fixed_point_bucket")

    def new_fp_pred(self) -> torch.Tensor:
        """Zero-filled prediction state of effq_fixed_point_traj (nothing known)."""
        return torch.zeros(self.lib.effq_fp_traj_pred_bytes(), dtype=torch.uint8, device=self.device)

    def fixed_point_traj(self, a, b, v, levels: int, state, pred, lo: float = -1.0, hi: float = 1.0):
        """project_by_iter of v = a + b from the previous call's iterates (`pred`), one launch (effq_fixed_point_traj)."""
        n = a.numel()
        ws = self._workspace("fp_traj", self.lib.effq_fp_traj_ws_bytes(n))
        check(self.lib.effq_fixed_point_traj(_ptr(a), _ptr(b), _ptr(v), n, levels, lo, hi, ADMM_TOL, 100 * levels,
                                             _ptr(state), _ptr(pred), _ptr(ws), ws.numel(), self.stream),
              "effq_fixed_point_traj")

    def fixed_point_bucket_rec(self, a, b, v, levels: int, state, pred, lo: float = -1.0, hi: float = 1.0):
        n = a.numel()
        ws = self._workspace("fp_bucket", self.lib.effq_fp_bucket_ws_bytes(n))
        check(self.lib.effq_fixed_point_bucket_rec(_ptr(a), _ptr(b), _ptr(v), n, levels, lo, hi, ADMM_TOL, 100 * levels,
                                                   _ptr(state), _ptr(ws), ws.numel(), _ptr(pred), self.stream),
              "effq_fixed_point_bucket_rec")

    def fixed_point_coop_rec(self, a, b, v, levels: int, state, pred, lo: float = -1.0, hi: float = 1.0):
        check(self.lib.effq_fixed_point_coop_rec(_ptr(a), _ptr(b), _ptr(v), a.numel(), levels, lo, hi, ADMM_TOL,
                                                 100 * levels, _ptr(state), _ptr(self._red_ws), _ptr(pred), self.stream),
              "effq_fixed_point_coop_rec")

    @staticmethod
    def read_fp_pred(pred: torch.Tensor):
        """FptPred of csrc/fp_level.h as a dict (tests, diagnostics)."""
        i32 = pred[:16].view(torch.int32).cpu()
        f = pred[16:16 + 8 * 40].view(torch.float64).cpu()
        i64 = pred[16 + 8 * 40:16 + 8 * 45].view(torch.int64).cpu()
        fn = pred[16 + 8 * 45:16 + 8 * 53].view(torch.float64).cpu()
        j64 = pred[16 + 8 * 53:16 + 8 * 63].view(torch.int64).cpu()
        K = int(i32[0])
        return {"K": K, "e": int(i32[1]), "lo": f[0:8].tolist(), "hi": f[8:16].tolist(), "eps": f[16:24].tolist(),
                "eps_n": fn.tolist(), "calls": int(i64[0]), "warm_iters": int(i64[1]), "full_iters": int(i64[2]),
                "listed": int(i64[3]), "list_max": int(i64[4]), "ring_iters": int(j64[0]), "ring_listed": int(j64[1]),
                "trace_us": [(int(j64[2 + k]) - int(j64[2])) / 100.0 for k in range(5)],
                "mean_us_phase1_setup_loop": [int(j64[7 + k]) / 100.0 / max(int(i64[0]), 1) for k in range(3)]}

    def fp_check(self, state, err_flag):
        check(self.lib.effq_fp_check(_ptr(state), _ptr(err_flag), self.stream), "effq_fp_check")

    # -- a5/a6 --------------------------------------------------------------------------------
    def gram(self, x_ndhwc: torch.Tensor, att: Optional[torch.Tensor], y_ndhwc: torch.Tensor, geom: Geom,
             has_bias: bool, A0: Optional[torch.Tensor] = None, B0: Optional[torch.Tensor] = None):
        """A0 (n x n), B0 (c2 x n) of solver.py:282-314, reference row order.  Accumulates when A0/B0 given."""
        x = self._f32(x_ndhwc)
        y = self._f32(y_ndhwc)
        a = self._f32(att) if att is not None else None
        _check_shapes(geom, x, y=y, att=a)
        n = geom.C1 * geom.KD * geom.KH * geom.KW + int(has_bias)
        acc = int(A0 is not None)
        if A0 is not None and (tuple(A0.shape) != (n, n) or tuple(B0.shape) != (geom.C2, n)):
            raise _lib.EffqError("gram: A0/B0 shapes do not match the geometry")
        if A0 is None:
            A0 = torch.empty(n, n, dtype=torch.float32, device=self.device)
            B0 = torch.empty(geom.C2, n, dtype=torch.float32, device=self.device)
        need = self.lib.effq_gram_ws_bytes(C.byref(geom), int(has_bias))
        ws = self._workspace("gram", need)
        check(self.lib.effq_gram_accum(_ptr(x), _ptr(a), _ptr(y), C.byref(geom), int(has_bias), _ptr(A0), _ptr(B0),
                                       acc, _ptr(ws), ws.numel(), self.stream), "effq_gram_accum")
        return A0, B0

    def gram_i8_supported(self, geom: Geom, act_levels: int) -> bool:
        return bool(self.lib.effq_gram_i8_supported(C.byref(geom), int(act_levels)))

    GRAM_I8_MAX_CLASSES = 16

    def att_classes(self, att: Optional[torch.Tensor]):
        """Voxel list sorted by attention weight for effq_gram_accum_i8: (vox_list int32 padded per class to
        multiples of 128 with -1, chunk_cls int32, cls_w float32, ncls).  None when the mask has more distinct
        values than the kernel takes (the caller then uses the fp32 Gram).  Built by the library (effq_att_classes);
        cached per mask tensor, the pyramid level is shared by several layers."""
        if att is None:
            return (None, None, None, 1)
        key = (att.data_ptr(), att.numel(), att._version)
        hit = self._att_cache.get(key)
        if hit is not None:
            return hit[1]
        flat = self._f32(att).reshape(-1)
        V = flat.numel()
        lst = torch.empty(V + 128 * self.GRAM_I8_MAX_CLASSES, dtype=torch.int32, device=flat.device)
        chunk_cls = torch.empty(V // 128 + self.GRAM_I8_MAX_CLASSES, dtype=torch.int32, device=flat.device)
        cls_w = torch.empty(self.GRAM_I8_MAX_CLASSES, dtype=torch.float32, device=flat.device)
        info = (C.c_int32 * 3)()
        ws = self._workspace("att_cls", self.lib.effq_att_classes_ws_bytes())
        check(self.lib.effq_att_classes(_ptr(flat), V, _ptr(lst), _ptr(chunk_cls), _ptr(cls_w), info, _ptr(ws),
                                        self.stream), "effq_att_classes")
        k, n_list, overflow = int(info[0]), int(info[1]), int(info[2])
        res = None if overflow else (lst[:n_list], chunk_cls[: n_list // 128], cls_w[:k], k)
        if len(self._att_cache) > 16:
            self._att_cache.clear()
        self._att_cache[key] = (att, res)     # holding the mask keeps its address from being reused
        return res

    def gram_i8(self, xidx_ndhwc: torch.Tensor, att_cls, y_ndhwc: torch.Tensor, geom: Geom, has_bias: bool,
                act_alpha: torch.Tensor, act_levels: int, A0: Optional[torch.Tensor] = None,
                B0: Optional[torch.Tensor] = None, unweighted: bool = False):
        """A0/B0 of gram() for an already quantised input, exact on the i8 matrix cores (effq_gram_accum_i8).
        att_cls = att_classes(att).  unweighted=True also returns (Au, Bu): the same sums without the attention weights,
        fp64, no factor 2 - the operands of gram_loss()."""
        if xidx_ndhwc.dtype != torch.uint8 or not xidx_ndhwc.is_contiguous():
            raise _lib.EffqError("gram_i8 wants contiguous uint8 level ids")
        y = self._f32(y_ndhwc)
        _check_shapes(geom, xidx_ndhwc, y=y)
        lst, chunk_cls, cls_w, ncls = att_cls
        if lst is not None:
            od, oh, ow = geom.out_dims()
            if (int(lst.numel()) < geom.N * od * oh * ow or lst.numel() % 128 or
                    chunk_cls.numel() * 128 != lst.numel() or cls_w.numel() != ncls):
                raise _lib.EffqError("gram_i8: voxel list does not match the output volume")
        n = geom.C1 * geom.KD * geom.KH * geom.KW + int(has_bias)
        acc = int(A0 is not None)
        if A0 is not None and (tuple(A0.shape) != (n, n) or tuple(B0.shape) != (geom.C2, n)):
            raise _lib.EffqError("gram_i8: A0/B0 shapes do not match the geometry")
        if A0 is None:
            A0 = torch.empty(n, n, dtype=torch.float32, device=self.device)
            B0 = torch.empty(geom.C2, n, dtype=torch.float32, device=self.device)
        al = self._f32(act_alpha.reshape(1))
        ws = self._workspace("gram_i8", self.lib.effq_gram_i8_ws_bytes(C.byref(geom), int(ncls)))
        Au = Bu = None
        if unweighted:
            if acc:
                raise _lib.EffqError("gram_i8: the unweighted system is not accumulated across calls here")
            Au = torch.empty(n, n, dtype=torch.float64, device=self.device)
            Bu = torch.empty(geom.C2, n, dtype=torch.float64, device=self.device)
        check(self.lib.effq_gram_accum_i8_unw(_ptr(xidx_ndhwc), _ptr(y), C.byref(geom), int(has_bias), _ptr(al),
                                              int(act_levels), _ptr(lst), _ptr(chunk_cls), _ptr(cls_w), int(ncls),
                                              0 if lst is None else int(lst.numel()), _ptr(A0), _ptr(B0), acc,
                                              _ptr(Au), _ptr(Bu), _ptr(ws), ws.numel(), self.stream),
              "effq_gram_accum_i8_unw")
        return (A0, B0, Au, Bu) if unweighted else (A0, B0)

    def gram_f64_supported(self, geom: Geom, has_bias: bool) -> bool:
        return bool(self.lib.effq_gram_f64_supported(C.byref(geom), int(has_bias)))

    def gram_f64(self, x_ndhwc: torch.Tensor, y_ndhwc: torch.Tensor, geom: Geom, has_bias: bool):
        """(Au, Bu): the unweighted fp64 Gram system of a layer with full-precision input (effq_gram_f64) - the
        operands of gram_loss() for the first conv and the classifier."""
        x, y = self._f32(x_ndhwc), self._f32(y_ndhwc)
        _check_shapes(geom, x, y=y)
        n = geom.C1 * geom.KD * geom.KH * geom.KW + int(has_bias)
        Au = torch.empty(n, n, dtype=torch.float64, device=self.device)
        Bu = torch.empty(geom.C2, n, dtype=torch.float64, device=self.device)
        ws = self._workspace("gram_f64", self.lib.effq_gram_f64_ws_bytes(C.byref(geom), int(has_bias)))
        check(self.lib.effq_gram_f64(_ptr(x), _ptr(y), C.byref(geom), int(has_bias), _ptr(Au), _ptr(Bu), _ptr(ws),
                                     ws.numel(), self.stream), "effq_gram_f64")
        return Au, Bu

    def upsample_trilinear(self, x_ndhwc: torch.Tensor, scale) -> torch.Tensor:
        """nn.Upsample(scale_factor=scale, mode='trilinear') on an NDHWC tensor (effq_upsample_trilinear)."""
        x = self._f32(x_ndhwc)
        N, D, H, W, Cc = (int(i) for i in x.shape)
        sd, sh, sw = (int(i) for i in scale)
        out = torch.empty(N, D * sd, H * sh, W * sw, Cc, dtype=torch.float32, device=self.device)
        check(self.lib.effq_upsample_trilinear(_ptr(x), N, D, H, W, Cc, sd, sh, sw, _ptr(out), self.stream),
              "effq_upsample_trilinear")
        return out

    def gram_loss(self, Au: torch.Tensor, Bu: torch.Tensor, syy: torch.Tensor, G: torch.Tensor, b, sqerr=None):
        """Squared error of conv(Qx, G, b) against the FP target from the unweighted Gram system (effq_gram_loss)."""
        c2, n = (int(i) for i in Bu.shape)
        has_b = b is not None
        if (Au.dtype != torch.float64 or Bu.dtype != torch.float64 or syy.dtype != torch.float64 or
                tuple(Au.shape) != (n, n) or G.numel() != c2 * (n - int(has_b))):
            raise _lib.EffqError("gram_loss: operand shapes / dtypes do not match")
        if sqerr is None:
            sqerr = torch.zeros(2, dtype=torch.float64, device=self.device)
        ws = self._workspace("gram_loss", self.lib.effq_gram_loss_ws_bytes(n))
        check(self.lib.effq_gram_loss(_ptr(Au), _ptr(Bu), _ptr(syy), _ptr(self._f32(G)), _ptr(b), c2, n, int(has_b),
                                      _ptr(sqerr), _ptr(ws), ws.numel(), self.stream), "effq_gram_loss")
        return sqerr

    def gram_loss_i8_supported(self, c2: int, n: int, has_b: bool, w_levels: int) -> bool:
        return bool(self.lib.effq_gram_loss_i8_supported(int(c2), int(n), int(has_b), int(w_levels)))

    def gram_loss_i8_planes(self, Au: torch.Tensor, has_b: bool, act_alpha: torch.Tensor, act_levels: int, voxels: int):
        """Balanced base-256 digit planes of K = Aww / s_a^2 (effq_gram_loss_i8_prepare): [P][round_up(nw, 256)][nw] int8.
        `voxels` bounds the entries: K <= (act_levels - 1)^2 * voxels.  Returns None if they need more than 6 planes."""
        n = int(Au.shape[0])
        P = self.lib.effq_gram_loss_i8_num_planes(int((act_levels - 1) ** 2 * voxels))
        if P < 1:
            return None
        nw = n - int(has_b)
        planes = torch.empty(P, (nw + 255) // 256 * 256, nw, dtype=torch.int8, device=self.device)
        assert planes.numel() == self.lib.effq_gram_loss_i8_planes_bytes(n, int(has_b), P)
        err = torch.zeros(1, dtype=torch.int32, device=self.device)
        al = self._f32(act_alpha.reshape(1))
        check(self.lib.effq_gram_loss_i8_prepare(_ptr(Au), n, int(has_b), _ptr(al), int(act_levels), P, _ptr(planes),
                                                 _ptr(err), self.stream), "effq_gram_loss_i8_prepare")
        planes._effq_err = err           # read with the layer's one host read (admm_read) or by the caller
        return planes

    def gram_loss_i8(self, planes, Au, Bu, syy, Gq, b, states, act_alpha, act_levels: int, w_levels: int, hist=None):
        """Losses of `count` stacked iterates (Gq [count, c2, nw] int8, b [count, c2], states [count, 5] fp64)."""
        count, c2 = int(Gq.shape[0]), int(Bu.shape[0])
        n = int(Bu.shape[1])
        has_b = b is not None
        if hist is None:
            hist = torch.zeros(count, 2, dtype=torch.float64, device=self.device)
        ws = self._workspace("gram_loss_i8", self.lib.effq_gram_loss_i8_ws_bytes())
        al = self._f32(act_alpha.reshape(1))
        check(self.lib.effq_gram_loss_i8(_ptr(planes), int(planes.shape[0]), _ptr(Au), _ptr(Bu), _ptr(syy), _ptr(Gq), _ptr(b),
                                         _ptr(states), _ptr(al), int(act_levels), int(w_levels), c2, n, int(has_b), count,
                                         _ptr(hist), _ptr(ws), ws.numel(), self.stream), "effq_gram_loss_i8")
        return hist

    def gram_reduce(self, A0: torch.Tensor, B0: torch.Tensor, reducer):
        """Data-parallel SUM of a layer's Gram system as ONE message: upper triangle of A0 + B0 (effq_gram_pack)."""
        n, c2 = int(A0.shape[0]), int(B0.shape[0])
        buf = torch.empty(self.lib.effq_gram_packed_elems(n, c2), dtype=torch.float32, device=self.device)
        check(self.lib.effq_gram_pack(_ptr(A0), _ptr(B0), n, c2, _ptr(buf), self.stream), "effq_gram_pack")
        reducer(buf)
        check(self.lib.effq_gram_unpack(_ptr(buf), n, c2, _ptr(A0), _ptr(B0), self.stream), "effq_gram_unpack")
        return A0, B0

    # -- f2: bit-packed level ids -------------------------------------------------------------
    @staticmethod
    def storage_bits(levels: int) -> int:
        """Smallest supported field width (1/2/4/8 bits) that holds `levels` level ids."""
        for b in (1, 2, 4, 8):
            if levels <= (1 << b):
                return b
        raise _lib.EffqError(f"{levels} levels do not fit 8 bits")

    def pack_levels(self, idx: torch.Tensor, bits: int) -> torch.Tensor:
        if idx.dtype != torch.uint8 or not idx.is_contiguous():
            raise _lib.EffqError("pack_levels wants contiguous uint8 level ids")
        nb = self.lib.effq_packed_bytes(idx.numel(), int(bits))
        if idx.numel() and nb == 0:
            raise _lib.EffqError(f"unsupported field width {bits}")
        out = torch.empty(int(nb), dtype=torch.uint8, device=idx.device)
        check(self.lib.effq_pack_levels(_ptr(idx), idx.numel(), int(bits), _ptr(out), self.stream), "effq_pack_levels")
        return out

    def unpack_levels(self, packed: torch.Tensor, n: int, bits: int) -> torch.Tensor:
        if packed.dtype != torch.uint8 or packed.numel() != self.lib.effq_packed_bytes(int(n), int(bits)):
            raise _lib.EffqError("unpack_levels: buffer does not match n and the field width")
        out = torch.empty(int(n), dtype=torch.uint8, device=packed.device)
        check(self.lib.effq_unpack_levels(_ptr(packed.contiguous()), int(n), int(bits), _ptr(out), self.stream),
              "effq_unpack_levels")
        return out

    # -- a7 -----------------------------------------------------------------------------------
    def spd_inverse(self, A0: torch.Tensor, has_bias: bool, rho: float, eta: float,
                    out: Optional[torch.Tensor] = None, ws_key: str = "inv") -> torch.Tensor:
        n = A0.shape[0]
        A0 = self._f32(A0)
        if out is None:     # n rows of effq_ainv_ld(n) floats (zero padded, 16-byte aligned rows)
            out = torch.empty(n, self.lib.effq_ainv_ld(n), dtype=torch.float32, device=self.device)
        ws = self._workspace(ws_key, self.lib.effq_spd_inverse_ws_bytes(n))
        check(self.lib.effq_spd_inverse(_ptr(A0), n, int(has_bias), rho, eta, _ptr(out), _ptr(ws), ws.numel(),
                                        self.stream), "effq_spd_inverse")
        return out

    def prox_solve(self, B0, Ainv, W0, b0, G, dual, rho: float, eta: float, wstar, bstar):
        c2, n = B0.shape
        ws = self._workspace("prox", self.lib.effq_prox_ws_bytes(c2, n))
        check(self.lib.effq_prox_solve(_ptr(B0), _ptr(Ainv), _ptr(W0), _ptr(b0), _ptr(G), _ptr(dual), c2, n,
                                       int(b0 is not None), rho, eta, _ptr(wstar), _ptr(bstar), _ptr(ws),
                                       ws.numel(), self.stream), "effq_prox_solve")

    PROX_SHIFT_TERMS = 26      # contraction factor < 1/2 per sweep: 2^-26 is below fp32 resolution

    def prox_solve_shifted(self, B0, Ainv, W0, b0, G, dual, rho: float, eta: float, rho_inv: float, wstar, bstar):
        """prox_solve for A(rho) through the inverse of A(rho_inv), rho_inv >= rho (effq_prox_solve_shifted)."""
        c2, n = B0.shape
        ws = self._workspace("prox", self.lib.effq_prox_ws_bytes(c2, n))
        # sweeps for 2^-26: factor <= d/(rho_inv+eta)
        terms = self.shift_terms(rho, eta, rho_inv)
        check(self.lib.effq_prox_solve_shifted(_ptr(B0), _ptr(Ainv), _ptr(W0), _ptr(b0), _ptr(G), _ptr(dual), c2, n,
                                               int(b0 is not None), rho, eta, rho_inv, terms, _ptr(wstar),
                                               _ptr(bstar), _ptr(ws), ws.numel(), self.stream),
              "effq_prox_solve_shifted")

    # -- the whole ADMM loop of a layer in one binding call ---------------------------------------------
    def admm_run(self, A0, B0, W0, b0, geom: Geom, y_ndhwc, *, xq=None, xidx=None, act_alpha=None, act_levels: int = 0,
                 loss_kind: int = 0, rho: float, rho_max: float, eta: float, iters: int, period: int, levels: int,
                 overlap: bool = True, loss_gram=None, residuals: bool = False):
        """effq_admm_run: enqueue `iters` ADMM iterations (chain on the current stream, per-iteration loss on the
        loss stream, later inverses on the side stream).  Returns a handle with the rings and `hist` (iters x 2
        device doubles, sums of squared errors); no host synchronisation.
        Memory: every iterate is kept until the best one is picked after the loop (G_ring: iters x nw floats, plus an
        int8 copy where the loss is an integer conv): 1.75 GB for a 256 -> 256 3^3 layer, 7 GB for LiTS' 512 -> 512,
        linear in `iters` (200 in the reference, a constructor constant there) - of 288 GB; the reference keeps one
        best iterate but synchronises with the host every iteration to do so."""
        from types import SimpleNamespace
        c2, n = (int(i) for i in B0.shape)
        has_b = b0 is not None
        # convert ONCE and keep the converted tensors alive with the run (the call only enqueues work on three streams:
        # a temporary .contiguous() copy could be handed back to the allocator while kernels still read it)
        W0, A0, B0 = self._f32(W0), self._f32(A0), self._f32(B0)
        b0 = self._f32(b0) if has_b else None
        xq = self._f32(xq) if xq is not None else None
        nw = W0.numel()
        y = self._f32(y_ndhwc)
        planes = None
        if loss_gram is not None and len(loss_gram) == 4:
            loss_kind = 5                         # (Au, Bu, syy, planes): the same with the quadratic form on the i8 cores
            planes = loss_gram[3]
            loss_gram = loss_gram[:3]
        elif loss_gram is not None:
            loss_kind = 4                         # (Au, Bu, syy): losses from the unweighted Gram system
        _check_shapes(geom, xq if loss_kind in (0, 4, 5) else xidx, W0, b0, y)
        if tuple(A0.shape) != (n, n) or nw != c2 * (n - int(has_b)):
            raise _lib.EffqError("admm_run: A0/B0/W0 shapes do not match")
        n_inv = self.lib.effq_admm_num_inverses(float(rho), float(rho_max), int(iters), int(period))
        if n_inv < 1:
            raise _lib.EffqError("admm_run: bad rho schedule")
        dev, f32 = self.device, torch.float32
        ld = self.lib.effq_ainv_ld(n)
        r = SimpleNamespace(iters=int(iters), nw=nw, c2=c2, has_b=has_b)
        r.ainv = torch.empty(n_inv, n, ld, dtype=f32, device=dev)
        r.dual = torch.empty(nw, dtype=f32, device=dev)
        r.wstar = torch.empty(nw, dtype=f32, device=dev)
        r.v = torch.empty(nw, dtype=f32, device=dev)
        r.G_ring = torch.empty(iters, nw, dtype=f32, device=dev)
        r.Gq_ring = torch.empty(iters, nw, dtype=torch.int8, device=dev) if loss_kind in (1, 2, 5) else None
        r.b_ring = torch.empty(iters, c2, dtype=f32, device=dev) if has_b else None
        r.state_ring = torch.zeros(iters, 5, dtype=torch.float64, device=dev)
        r.hist = torch.zeros(iters, 2, dtype=torch.float64, device=dev)
        r.err = torch.zeros(1, dtype=torch.int32, device=dev)
        r.res = torch.zeros(iters, 2, dtype=torch.float64, device=dev) if residuals else None     # lwq_verbose
        main = torch.cuda.current_stream(dev)
        loss_s = self.loss_stream() if overlap else None
        side_s = self.side_stream()
        # workspaces are (zero-)filled on the current stream, ahead of the events the library orders the other
        # streams by
        prox = self._workspace("prox", self.lib.effq_prox_ws_bytes(c2, n))
        inv = self._workspace("inv", self.lib.effq_spd_inverse_ws_bytes(n))
        inv_side = (self._workspace("inv_side", self.lib.effq_spd_inverse_ws_bytes(n))
                    if n_inv > 1 and SIDE_STREAM else None)
        inv_side2 = (self._workspace("inv_side2", self.lib.effq_spd_inverse_ws_bytes(n))
                     if n_inv > 2 and SIDE2_STREAM and SIDE_STREAM else None)
        fpw = (self._workspace("fp_bucket", self.lib.effq_fp_bucket_ws_bytes(nw))
               if nw <= self.lib.effq_fp_bucket_max() and BUCKET_FIXED_POINT else None)
        # weight projection from the previous iteration's iterates (effq_fixed_point_traj)
        traj = TRAJ_FIXED_POINT and bool(self.lib.effq_admm_uses_traj(nw, int(levels)))   # the library's own decision
        tws = self._workspace("fp_traj", self.lib.effq_fp_traj_ws_bytes(nw)) if traj else None
        r.fp_pred = torch.zeros(self.lib.effq_fp_traj_pred_bytes(), dtype=torch.uint8, device=dev) if traj else None
        if loss_kind == 5:
            cws = self._workspace("gram_loss_i8", self.lib.effq_gram_loss_i8_ws_bytes())
        elif loss_kind == 4:
            cws = self._workspace("gram_loss", self.lib.effq_gram_loss_ws_bytes(n))
        elif loss_kind == 1:
            cws = self._workspace("conv_i8", self.lib.effq_conv_i8_ws_bytes(C.byref(geom)))
        elif loss_kind == 2:
            cws = self._workspace("conv_i8s", self.lib.effq_conv_i8s_ws_bytes(C.byref(geom), int(act_levels),
                                                                             int(levels)))
        else:
            cws = self._workspace("conv", self.lib.effq_conv_ws_bytes(C.byref(geom)))
        al = self._f32(act_alpha.reshape(1)) if (act_alpha is not None and loss_kind in (1, 2, 5)) else None
        a = _lib.AdmmRunArgs()
        p = lambda t: None if t is None else t.data_ptr()
        a.A0, a.B0, a.W0, a.b0 = p(A0), p(B0), p(W0), p(b0)
        a.c2, a.n, a.has_bias, a.w_levels = c2, n, int(has_b), int(levels)
        a.iters, a.rho_period = int(iters), int(period)
        a.rho, a.rho_max, a.eta, a.tol = float(rho), float(rho_max), float(eta), ADMM_TOL
        a.geom = geom
        a.loss_kind, a.act_levels = int(loss_kind), int(act_levels)
        a.xq = p(xq) if loss_kind == 0 else None
        a.xidx = p(xidx) if loss_kind in (1, 2) else None
        if loss_kind in (4, 5):
            Au, Bu, syy = loss_gram
            if (Au.dtype != torch.float64 or Bu.dtype != torch.float64 or syy.dtype != torch.float64 or
                    tuple(Au.shape) != (n, n) or tuple(Bu.shape) != (c2, n) or syy.numel() != 1):
                raise _lib.EffqError("admm_run: loss_gram wants fp64 (Au [n,n], Bu [c2,n], syy [1])")
            a.loss_Au, a.loss_Bu, a.loss_syy = p(Au), p(Bu), p(syy)
            if loss_kind == 5:
                a.loss_planes, a.loss_nplanes = p(planes), int(planes.shape[0])
                r.loss_planes = planes
        a.y_fp, a.act_alpha_dev = p(y), p(al)
        a.dual, a.wstar, a.v = p(r.dual), p(r.wstar), p(r.v)
        a.G_ring, a.Gq_ring, a.b_ring = p(r.G_ring), p(r.Gq_ring), p(r.b_ring)
        a.state_ring, a.hist, a.err_flag = p(r.state_ring), p(r.hist), p(r.err)
        a.res_ring = p(r.res)
        a.ainv_pool, a.n_ainv = p(r.ainv), n_inv
        a.prox_ws, a.prox_ws_bytes = p(prox), prox.numel()
        a.red_ws = p(self._red_ws)
        a.fp_ws, a.fp_ws_bytes = p(fpw), (fpw.numel() if fpw is not None else 0)
        a.fp_pred, a.fp_traj_ws, a.fp_traj_ws_bytes = p(r.fp_pred), p(tws), (tws.numel() if tws is not None else 0)
        a.inv_ws, a.inv_ws_bytes = p(inv), inv.numel()
        a.inv_ws_side, a.inv_ws_side_bytes = p(inv_side), (inv_side.numel() if inv_side is not None else 0)
        a.conv_ws, a.conv_ws_bytes = p(cws), cws.numel()
        a.stream_main = main.cuda_stream
        a.stream_loss = loss_s.cuda_stream if loss_s is not None else None
        a.stream_side = side_s.cuda_stream if inv_side is not None else None
        if inv_side2 is not None:
            a.stream_side2 = self.side_stream2().cuda_stream
            a.inv_ws_side2, a.inv_ws_side2_bytes = p(inv_side2), inv_side2.numel()
        r.keep = (A0, B0, W0, b0, y, xq, xidx, al, loss_gram)       # the call only enqueues: keep every operand alive
        check(self.lib.effq_admm_run(C.byref(a)), "effq_admm_run")
        return r

    def admm_select_best(self, run):
        """Earliest iterate with the smallest (already all-reduced) loss: (best_G, best_b, best[loss sum, index])."""
        best_G = torch.empty(run.nw, dtype=torch.float32, device=self.device)
        best_b = torch.empty(run.c2, dtype=torch.float32, device=self.device) if run.has_b else None
        best = torch.empty(2, dtype=torch.float64, device=self.device)
        check(self.lib.effq_admm_select_best(_ptr(run.hist), run.iters, _ptr(run.G_ring), _ptr(run.b_ring), run.nw,
                                             run.c2, _ptr(best_G), _ptr(best_b), _ptr(best), self.stream),
              "effq_admm_select_best")
        return best_G, best_b, best

    @staticmethod
    def admm_read(run, best, extra=None):
        """ONE device->host copy per layer: loss history, best, the last scale, fixed-point iteration counts, error
        (and `extra`, a small fp64 device tensor the caller wants read in the same copy)."""
        err = run.err.to(torch.float64)
        if getattr(run, "loss_planes", None) is not None:       # effq_gram_loss_i8_prepare's flag: 1000 x (1 | 2)
            err = err + 1000.0 * run.loss_planes._effq_err.to(torch.float64)
        parts = [run.hist[:, 0], best, run.state_ring[-1, :1], run.state_ring[:, 4], err]
        if extra is not None:
            parts.append(extra.to(torch.float64).reshape(-1))
        pack = torch.cat(parts).cpu()
        it = run.iters
        w_iters = pack[it + 3: 2 * it + 3].contiguous().view(torch.int32)[0::2].tolist()
        return dict(hist=pack[:it].tolist(), best=pack[it:it + 2].tolist(), alpha_w=float(pack[it + 2]),
                    w_iters=w_iters, err=int(pack[2 * it + 3]),
                    extra=pack[2 * it + 4:].tolist() if extra is not None else None)

    def shift_terms(self, rho: float, eta: float, rho_inv: float) -> int:
        d = rho_inv - rho
        return 1 if d <= 0 else min(64, max(2, int(math.ceil(-26.0 * math.log(2.0) / math.log(d / (rho_inv + eta))))))

    # -- a4 elementwise -------------------------------------------------------------------------
    def admm_presum(self, wstar, dual, v):
        check(self.lib.effq_admm_presum(_ptr(wstar), _ptr(dual), _ptr(v), wstar.numel(), self.stream),
              "effq_admm_presum")

    def admm_project_dual(self, v, wstar, state, levels: int, G, dual, dual_div: float, Gq=None):
        check(self.lib.effq_admm_project_dual(_ptr(v), _ptr(wstar), _ptr(state), levels, _ptr(G), _ptr(dual),
                                              float(dual_div), _ptr(Gq), v.numel(), self.stream),
              "effq_admm_project_dual")

    def conv_i8_supported(self, geom: Geom, act_levels: int, w_levels: int) -> bool:
        return bool(self.lib.effq_conv_i8_supported(C.byref(geom), int(act_levels), int(w_levels)))

    def conv_i8s_supported(self, geom: Geom, act_levels: int, w_levels: int) -> bool:
        return bool(self.lib.effq_conv_i8s_supported(C.byref(geom), int(act_levels), int(w_levels)))

    def conv_step_i8s(self, xidx: torch.Tensor, Gq: torch.Tensor, bias, geom: Geom, y_ndhwc: torch.Tensor,
                      act_alpha: torch.Tensor, act_levels: int, w_state: torch.Tensor, w_levels: int, sqerr,
                      prepare: bool):
        """Exact-integer loss evaluation for short-K layers and up to 256 levels (conv3d_calib_step_i8s).
        prepare=True on the first call of a layer (per-voxel level sums are cached in the workspace)."""
        if xidx.dtype != torch.uint8 or Gq.dtype != torch.int8:
            raise _lib.EffqError("conv_step_i8s wants uint8 level ids and int8 weight operands")
        _check_shapes(geom, xidx, Gq, bias, y_ndhwc)
        al = self._f32(act_alpha.reshape(1))
        ws = self._workspace("conv_i8s", self.lib.effq_conv_i8s_ws_bytes(C.byref(geom), int(act_levels), int(w_levels)))
        check(self.lib.conv3d_calib_step_i8s(_ptr(xidx), _ptr(Gq), _ptr(bias), _ptr(self._f32(y_ndhwc)),
                                             C.byref(geom), _ptr(al), int(act_levels), _ptr(w_state), int(w_levels),
                                             int(bool(prepare)), _ptr(sqerr), _ptr(ws), ws.numel(), self.stream),
              "conv3d_calib_step_i8s")
        return sqerr

    def conv_step_i8(self, xidx: torch.Tensor, Gq: torch.Tensor, bias, geom: Geom, y_ndhwc: torch.Tensor,
                     act_alpha: torch.Tensor, act_levels: int, w_state: torch.Tensor, w_levels: int, sqerr):
        """Exact-integer loss evaluation (conv3d_calib_step_i8)."""
        if xidx.dtype != torch.uint8 or Gq.dtype != torch.int8:
            raise _lib.EffqError("conv_step_i8 wants uint8 level ids and int8 weight numerators")
        _check_shapes(geom, xidx, Gq, bias, y_ndhwc)
        al = self._f32(act_alpha.reshape(1))
        ws = self._workspace("conv_i8", self.lib.effq_conv_i8_ws_bytes(C.byref(geom)))
        check(self.lib.conv3d_calib_step_i8(_ptr(xidx), _ptr(Gq), _ptr(bias), _ptr(self._f32(y_ndhwc)), C.byref(geom),
                                            _ptr(al), int(act_levels), _ptr(w_state), int(w_levels), _ptr(sqerr),
                                            _ptr(ws), ws.numel(), self.stream), "conv3d_calib_step_i8")
        return sqerr

    def conv_i8_out_supported(self, geom: Geom, act_levels: int, w_levels: int) -> bool:
        return bool(self.lib.effq_conv_i8_out_supported(C.byref(geom), int(act_levels), int(w_levels)))

    def conv_forward_i8(self, xidx: torch.Tensor, G: torch.Tensor, bias, geom: Geom, y_ndhwc: torch.Tensor, att,
                        act_alpha: torch.Tensor, act_levels: int, w_state: torch.Tensor, w_levels: int):
        """The quantised forward of a calibrated layer + its final loss on the i8 matrix cores (conv3d_quant_forward_i8):
        `G` = alpha_w * b, the projected weights of the iterate whose fixed-point state is `w_state` (5 device doubles,
        alpha first); returns (out NDHWC fp32, [sum err^2, sum att err^2] device doubles)."""
        if xidx.dtype != torch.uint8:
            raise _lib.EffqError("conv_forward_i8 wants uint8 level ids")
        lm1 = int(w_levels) - 1
        a32 = w_state.reshape(-1)[0].to(torch.float32)
        # the integer numerators 2 * level - (Lw - 1) of the weights: G / f32(alpha) is b = level * d - 1 to an ulp
        Gq = (2.0 * torch.round((self._f32(G) / a32 + 1.0) * (0.5 * lm1)) - lm1).to(torch.int8).contiguous()
        _check_shapes(geom, xidx, Gq, bias, y_ndhwc)
        al = self._f32(act_alpha.reshape(1))
        y = self._f32(y_ndhwc)
        out = torch.empty_like(y)
        sq = torch.zeros(2, dtype=torch.float64, device=self.device)
        att_f = self._f32(att) if att is not None else None
        st = w_state.to(torch.float64).contiguous()
        ws = self._workspace("conv_i8", self.lib.effq_conv_i8_ws_bytes(C.byref(geom)))
        check(self.lib.conv3d_quant_forward_i8(_ptr(xidx), _ptr(Gq), _ptr(self._f32(bias) if bias is not None else None),
                                               _ptr(y), _ptr(att_f), C.byref(geom), _ptr(al), int(act_levels), _ptr(st),
                                               int(w_levels), _ptr(sq), _ptr(out), _ptr(ws), ws.numel(), self.stream),
              "conv3d_quant_forward_i8")
        self._keep_i8 = (Gq, st, att_f, al)           # operands of an enqueued kernel: alive until the next call
        return out, sq

    # -- f3: gradients of the activation quantiser, Adam ----------------------------------------------------
    def act_quant_backward(self, x: torch.Tensor, alpha: torch.Tensor, levels: int, gq: torch.Tensor, want_gx=True):
        """(gx, galpha[device double]) of q = discretize(x/alpha, L, 0, 1)*alpha given gq (effq_act_quant_backward)."""
        x, gq = self._f32(x), self._f32(gq)
        a = self._f32(alpha.reshape(1))
        gx = torch.empty_like(x) if want_gx else None
        ga = torch.empty(1, dtype=torch.float64, device=self.device)
        check(self.lib.effq_act_quant_backward(_ptr(x), _ptr(a), int(levels), _ptr(gq), _ptr(gx), _ptr(ga), x.numel(),
                                               _ptr(self._red_ws), self.stream), "effq_act_quant_backward")
        return gx, ga

    def adam_step(self, p, g, m, v, lr: float, t: int, b1=0.9, b2=0.999, eps=1e-8):
        """torch.optim.Adam step in place on flat fp32 tensors (effq_adam_step)."""
        check(self.lib.effq_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), lr, b1, b2, eps, int(t), p.numel(),
                                      self.stream), "effq_adam_step")

    # -- the conv ---------------------------------------------------------------------------------
    def conv_step(self, x_ndhwc: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], geom: Geom,
                  y_ndhwc: Optional[torch.Tensor] = None, att: Optional[torch.Tensor] = None,
                  act_alpha: Optional[torch.Tensor] = None, act_levels: int = 0, want_out: bool = False,
                  sqerr: Optional[torch.Tensor] = None):
        """conv3d_quant_calib_step.  Returns (out NDHWC or None, sqerr device fp64[2] or None)."""
        x = self._f32(x_ndhwc)
        w = self._f32(weight)
        b = self._f32(bias) if bias is not None else None
        y = self._f32(y_ndhwc) if y_ndhwc is not None else None
        a = self._f32(att) if att is not None else None
        _check_shapes(geom, x, w, b, y, a)
        od, oh, ow = geom.out_dims()
        out = torch.empty(geom.N, od, oh, ow, geom.C2, dtype=torch.float32, device=self.device) if want_out else None
        if y is not None and sqerr is None:
            sqerr = torch.empty(2, dtype=torch.float64, device=self.device)
        al = self._f32(act_alpha.reshape(1)) if act_alpha is not None else None
        ws = self._workspace("conv", self.lib.effq_conv_ws_bytes(C.byref(geom)))
        check(self.lib.conv3d_quant_calib_step(_ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(a), C.byref(geom), _ptr(al),
                                               int(act_levels), _ptr(sqerr), _ptr(out), _ptr(ws), ws.numel(),
                                               self.stream), "conv3d_quant_calib_step")
        return out, sqerr


_OPS = {}


def get_ops(device) -> HipOps:
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = str(device)
    if key not in _OPS:
        _OPS[key] = HipOps(device)
    return _OPS[key]
