"""Quantised Conv3d modules with the reference's class contract, computing on MI355X.

``PTQConv`` mirrors the reference base class (src/models/PTQConv.py:11-174): same
constructor, parameters (``weight``, ``bias``, 0-dim ``alpha_act``/``alpha_w``), mode
setters, ``forward`` dispatch, ``store_int_weight``/``restore_fp_weight``.
``EfficientQConvHIP`` mirrors ``EfficientQConv.ptq`` (src/models/EfficientQConv.py:33-166):
the layer-wise ADMM calibration, with every tensor computation issued through the C-ABI
library (``hip_ops``).  torch supplies memory, streams and the collective only.

Both class names contain ``QConv`` because the reference finds quantised layers by
class-name substring (src/models/model_blk.py:26-34).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import hip_ops
from .hip_ops import from_ndhwc, make_geom, to_ndhwc

# ADMM constants are constructor constants in the reference (EfficientQConv.py:23-26)
LWQ_ITER, LWQ_RHO, LWQ_RHO_MAX, LWQ_ETA, RHO_PERIOD = 200, 10.0, 1000.0, 1.0, 50
import os as _os
EXACT_INT_DEFAULT = _os.environ.get("EFFQ_EXACT_INT", "1") != "0"
# evaluate the loss of iteration i on a second stream while the chain computes iteration i+1
OVERLAP_LOSS_DEFAULT = _os.environ.get("EFFQ_OVERLAP_LOSS", "1") != "0"
# per-iteration losses from the unweighted Gram system (effq_gram_loss) for layers with n = k^3 c1 + 1 up to this size
GRAM_LOSS_DEFAULT = _os.environ.get("EFFQ_GRAM_LOSS", "1") != "0"
GRAM_LOSS_MAX_N = int(_os.environ.get("EFFQ_GRAM_LOSS_MAX_N", "1729"))
FORWARD_I8 = _os.environ.get("EFFQ_FORWARD_I8", "1") != "0"       # the calibrated layer's forward + final loss on the i8 cores
GRAM_LOSS_I8 = _os.environ.get("EFFQ_GRAM_LOSS_I8", "1") != "0"     # wide layers: the quadratic form on the i8 matrix cores
# ... up to this system size: measured per calibration, BraTS (n = 3457, 6913) 674 -> 657 ms with it; LiTS with its
# 512-channel layers (n = 13825) included 2110 -> 2169 ms (5.4 ms per group of 8 iterates beside a chain that is itself
# slowed by three concurrent 57 ms inverses), so those keep the integer conv pass
GRAM_LOSS_I8_MAX_N = int(_os.environ.get("EFFQ_GRAM_LOSS_I8_MAX_N", "8000"))


def get_ops(device):
    """Indirection point: tests substitute a CPU stand-in built on the oracle here."""
    return hip_ops.get_ops(device)


class SumReducer:
    """Data-parallel SUM over calibration-volume shards (RCCL all-reduce; identity on one rank)."""

    calls = 0          # collectives issued by this process (tests, bench)

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        # EFFQ_DP_FORCE=1: run the collectives even on one rank (exercises the RCCL path on a single GPU)
        force = _os.environ.get("EFFQ_DP_FORCE", "0") == "1"
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.group = group
        self.direct = None
        if self.on and dist.get_backend(group) == "nccl":
            # RCCL called through its C ABI on the stream the kernels run on (rccl.py): no hop to the process group's
            # stream and back around every one of the ~50 tiny all-reduces of a layer.  EFFQ_RCCL_DIRECT=0: torch's path
            from . import rccl
            self.direct = rccl.get_comm(group)

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        if self.on:
            SumReducer.calls += 1
            if t.is_cuda and self.direct is not None and t.is_contiguous():
                self.direct.all_reduce_sum_(t)
            elif t.is_cuda and self.dist.get_backend(self.group) == "gloo":
                # rehearsal mode (several ranks on ONE GPU, where RCCL refuses duplicate devices):
                # stage through the host; the product backend is "nccl" (= RCCL over xGMI)
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """Rank-major concatenation of every rank's `t` (same length on every rank)."""
        if not self.on:
            return t
        SumReducer.calls += 1
        if t.is_cuda and self.direct is not None and t.is_contiguous():
            return self.direct.all_gather(t)
        world = self.dist.get_world_size(self.group)
        if t.is_cuda and self.dist.get_backend(self.group) == "gloo":
            h = t.cpu()
            parts = [torch.empty_like(h) for _ in range(world)]
            self.dist.all_gather(parts, h, group=self.group)
            return torch.cat(parts).to(t.device)
        parts = [torch.empty_like(t) for _ in range(world)]
        self.dist.all_gather(parts, t.contiguous(), group=self.group)
        return torch.cat(parts)

    @property
    def world(self) -> int:
        return self.dist.get_world_size(self.group) if self.on else 1

    @property
    def rank(self) -> int:
        return self.dist.get_rank(self.group) if self.on else 0

    def __bool__(self):
        return self.on


class _QuantConvFn(torch.autograd.Function):
    """Quantised forward of a PTQConv with gradients for tune_activation_range (ptqer.py:238-272): the forward is the
    fused act-quant conv kernel; the backward is the same conv kernel on the output gradient (flipped, transposed
    weights) followed by the straight-through backward of the activation quantiser.  Weights take no gradient."""

    @staticmethod
    def forward(ctx, x, alpha, mod):
        ctx.mod = mod
        ctx.save_for_backward(x, alpha)
        # (a fresh tensor, not the permuted view _conv returns: the next block's in-place ReLU writes into it)
        return mod._conv(x, mod.q_act).clone(memory_format=torch.preserve_format)

    @staticmethod
    def backward(ctx, g):
        mod = ctx.mod
        x, alpha = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0]
        if not need_x and not mod.q_act:
            return None, None, None
        gq = mod._dgrad(g)
        if not mod.q_act:
            return gq, None, None
        ops = get_ops(g.device)
        gx, ga = ops.act_quant_backward(to_ndhwc(x.detach()), alpha.detach(), mod.qlvl_act, to_ndhwc(gq), want_gx=need_x)
        return (from_ndhwc(gx) if need_x else None), ga.to(alpha.dtype).reshape(alpha.shape), None


class PTQConv(nn.Conv3d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, q_weight=True, qlvl=8, q_act=True, qlvl_act=8, **kwQ):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        if self.dilation != (1, 1, 1) or self.groups != 1:
            # the reference's solver ignores both (solver.py:86-111); both shipped configs use 1
            raise NotImplementedError("dilation/groups other than 1 are outside the calibrated path")
        self.conv_param = dict(stride=stride, padding=padding, dilation=dilation, groups=groups)
        self.q_act, self.q_weight = q_act, q_weight
        self.qlvl_w, self.qlvl_act = qlvl, qlvl_act
        self.kwQ = kwQ
        self.alpha_act = nn.Parameter(torch.tensor(1.))
        self.alpha_w = nn.Parameter(torch.tensor(1.))
        self.output_fp = None            # FP target, set by the forward hook
        self.name = None
        self.snap_dir = kwQ.get('snap_dir', None)
        self.w_backup = self.b_backup = None
        self._act_inited = False
        self.set_fp()

    # ---- mode flags (PTQConv.py:50-72) --------------------------------------------------------
    def _mode(self, fp=False, quantizing=False, quantized=False, init_act=False):
        self._fp, self._quantizing, self._quantized, self._init_act = fp, quantizing, quantized, init_act

    def set_fp(self):
        self._mode(fp=True)

    def set_quantizing(self):
        self._mode(quantizing=True)

    def set_quantized(self):
        self._mode(quantized=True)

    def set_init_act(self):
        self._mode(init_act=True)

    def qweight_init_iter(self):
        pass

    def qparam_init(self):
        pass

    def perform_quantization(self):
        pass

    def backup_weight(self):
        self.w_backup = self.weight.data.cpu().clone()
        if self.bias is not None:
            self.b_backup = self.bias.data.cpu().clone()

    # ---- device helpers ------------------------------------------------------------------------
    def _geom(self, x):
        return make_geom(x.shape, self.out_channels, self.kernel_size, self.stride, self.padding)

    def _conv(self, x, quantize_act: bool):
        """conv3d (optionally with the fp32 activation quant-dequant fused into the tile load)."""
        ops = get_ops(x.device)
        out, _ = ops.conv_step(to_ndhwc(x.detach()), self.weight.data, None if self.bias is None else self.bias.data,
                               self._geom(x), act_alpha=self.alpha_act.data if quantize_act else None,
                               act_levels=self.qlvl_act if quantize_act else 0, want_out=True)
        return from_ndhwc(out)

    def _dgrad(self, g):
        """Gradient of the conv w.r.t. its input: the conv kernel on g with flipped, transposed weights (stride 1)."""
        if self.stride != (1, 1, 1):
            raise NotImplementedError("input gradient of a strided conv is not needed on the calibrated path")
        ops = get_ops(g.device)
        key = (self.weight.data_ptr(), self.weight._version)
        if getattr(self, "_wt_key", None) != key:
            self._wt = self.weight.data.flip(2, 3, 4).transpose(0, 1).contiguous()
            self._wt_key = key
        geom = make_geom(g.shape, self.in_channels, self.kernel_size, self.stride, self.padding)
        out, _ = ops.conv_step(to_ndhwc(g.detach()), self._wt, None, geom, want_out=True)
        return from_ndhwc(out)

    def _quantize_act(self, x):
        """discretize(x/alpha_act, L, 0, 1) * alpha_act in fp32 (PTQConv.py:114-116)."""
        ops = get_ops(x.device)
        return from_ndhwc(ops.quant_dequant_f32(to_ndhwc(x.detach()), self.alpha_act.data, self.qlvl_act, 0.0, 1.0))

    def _quantize_w(self):
        ops = get_ops(self.weight.device)
        return ops.quant_dequant_f32(self.weight.data, self.alpha_w.data, self.qlvl_w, -1.0, 1.0)

    def init_alpha_act(self, x):
        """PTQConv.py:74-78."""
        ops = get_ops(x.device)
        xn = to_ndhwc(x.detach())
        red = SumReducer()
        if self.qlvl_act < 2:
            # "W,-1" layers (full-precision activations, quirk Q14) still pass through here in the init pass: with
            # num_lvl = -1 project_by_iter's loop never runs (max_iter = -100), so a = mean|x|, and discretize works with
            # delta = 1/(-2): values land on {0, 0.5, 1} - the arithmetic of 3 levels (layer_helper.py:25-37,50-66)
            s0 = ops.abs_sum(xn)
            if red:
                red(s0)
            st = ops.new_fp_state()
            st[0] = s0[0] / s0[1]
            a, levels = st[0].item(), 3
        else:
            a, _, st = ops.fit_scale(xn, self.qlvl_act, 0.0, 1.0, reducer=red or None)
            levels = self.qlvl_act
        self.alpha_act.data = torch.tensor(a, device=x.device)
        self._act_inited = True
        y, _, _ = ops.quant_dequant_f64path(xn, st, levels, 0.0, 1.0)
        return from_ndhwc(y)

    def ptq(self, x):
        raise NotImplementedError

    # ---- storage formats (PTQConv.py:125-152) -----------------------------------------------
    def store_int_weight(self):
        """Level ids as uint8 (<=256 levels) / int32, for storage only.  Uses the saved alpha_w,
        which is the LAST iterate's scale while weight is the BEST iterate's (quirk Q6)."""
        a = self.alpha_w.data
        b = self.weight.data / a
        delta = 2 / (self.qlvl_w - 1)
        w_int = torch.round((b + 1) / delta)
        w_int = w_int.to(torch.uint8) if self.qlvl_w <= 256 else w_int.to(torch.int32)
        self.weight.requires_grad = False
        self.weight.data = w_int.data.cpu()

    def restore_fp_weight(self):
        delta = 2 / (self.qlvl_w - 1)
        b = self.weight.data.float() * delta - 1
        self.weight.data = self.alpha_w.data * b

    # ---- bit-packed storage (row f2; the reference keeps one uint8 per weight) ------------------
    def export_packed_weight(self):
        """Level ids of the quantised weight packed at 1/2/4/8 bits each: dict(data, n, bits, shape, alpha_w, levels).
        Same level ids as store_int_weight(); the module is left untouched."""
        ops = get_ops(self.weight.device)
        delta = 2 / (self.qlvl_w - 1)
        ids = torch.round((self.weight.data / self.alpha_w.data + 1) / delta).to(torch.uint8).contiguous().reshape(-1)
        bits = ops.storage_bits(self.qlvl_w)
        return dict(data=ops.pack_levels(ids, bits).cpu(), n=int(ids.numel()), bits=bits,
                    shape=tuple(self.weight.shape), alpha_w=float(self.alpha_w.data), levels=int(self.qlvl_w))

    def import_packed_weight(self, blob):
        """Inverse of export_packed_weight(): weight = alpha_w * (2*id/(L-1) - 1), like restore_fp_weight()."""
        ops = get_ops(self.weight.device)
        ids = ops.unpack_levels(blob["data"].to(self.weight.device), blob["n"], blob["bits"])
        delta = 2 / (blob["levels"] - 1)
        self.alpha_w.data = torch.tensor(blob["alpha_w"], dtype=self.weight.dtype, device=self.weight.device)
        self.weight.data = (self.alpha_w.data * (ids.float() * delta - 1)).reshape(blob["shape"])

    # ---- forward dispatch (PTQConv.py:154-174) ----------------------------------------------
    def forward(self, x):
        if self._fp:
            return self._conv(x, False)
        if self._quantizing:
            self._ptq_out = None
            self.ptq(x)
            out, self._ptq_out = getattr(self, "_ptq_out", None), None
            return from_ndhwc(out) if out is not None else self._conv(x, self.q_act)
        if self._quantized:
            if torch.is_grad_enabled() and (self.alpha_act.requires_grad or x.requires_grad):
                return _QuantConvFn.apply(x, self.alpha_act, self)      # tune_activation_range (row f3)
            return self._conv(x, self.q_act)
        if self._init_act:
            qact = self.init_alpha_act(x)
            return self._conv(qact, False)
        raise RuntimeError(f"Unknown FP/Quant setting: FP={self._fp}, "
                           f"Quantizing={self._quantizing}, Quantized={self._quantized}")


class EfficientQConvHIP(PTQConv):
    """Layer-wise ADMM calibrator (EfficientQConv.py:13-166) on the HIP library."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, q_weight=True, qlvl=8, q_act=True, qlvl_act=8, **kwQ):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         q_weight, qlvl, q_act, qlvl_act, **kwQ)
        self.lwq_iter, self.lwq_rho, self.lwq_rho_max, self.lwq_eta = LWQ_ITER, LWQ_RHO, LWQ_RHO_MAX, LWQ_ETA
        self.lwq_fold_bn = True
        self.lwq_verbose = kwQ.get('lwq_verbose', False)
        self.mask_pyramid = None
        self.layer_loss = None
        self.last_trace = None          # diagnostics of the last calibration (not in the reference)
        self.lwq_trace = kwQ.get('lwq_trace', False)   # record the per-iteration loss (one host sync each)
        # evaluate the per-iteration losses on the i8 matrix cores (exact int32 accumulation) where supported
        self.lwq_exact_int = kwQ.get('lwq_exact_int', EXACT_INT_DEFAULT)
        self.lwq_overlap_loss = kwQ.get('lwq_overlap_loss', OVERLAP_LOSS_DEFAULT)

    @staticmethod
    def _std(m: torch.Tensor) -> float:
        """Unbiased std from [sum, sumsq, n] (Tensor.std(), EfficientQConv.py:46,48)."""
        s, ss, n = m.tolist() if hasattr(m, "tolist") else m
        return math.sqrt(max(ss - s * s / n, 0.0) / (n - 1))

    def ptq(self, x):
        ops = get_ops(x.device)
        red = SumReducer()
        dev = x.device
        xn = to_ndhwc(x.detach())
        yn = to_ndhwc(self.output_fp.detach().to(dev))
        geom = self._geom(x)
        W0 = self.weight.data.contiguous()
        has_b = self.bias is not None
        b0 = self.bias.data.contiguous() if has_b else None
        c2 = self.out_channels
        nw = W0.numel() // c2

        # rho_scale = max(numel(y)*std(y) / (numel(W)*std(W)), 1) [* mean(att)]     (EfficientQConv.py:43-49, :51-62)
        # the three moment triples are enqueued first and read by ONE device->host copy (each read empties the queue:
        # the device idles for the round trip)
        my = ops.moments(yn)
        syy_local = my[1:2].clone()           # sum y^2 over THIS rank's voxels (the loss from the Gram system adds it)
        mw = ops.moments(W0)
        att = None
        if self.lwq_verbose:
            print(f'Calibrating {self.name}')
        if self.mask_pyramid:
            for mask in self.mask_pyramid:
                if tuple(mask.shape[1:]) == tuple(self.output_fp.shape[2:]):
                    att = mask.to(dev).contiguous()
                    break
        ma = ops.moments(att) if att is not None else None
        # data-parallel: the layer's three statistics triples - targets, attention weights, sum|x| of the input for the
        # start of its scale fit - travel as ONE message (they were three)
        fit_here = bool(self.q_act and not self._act_inited)
        sx = ops.abs_sum(xn) if (fit_here and red and hasattr(ops, "abs_sum")) else None
        if red:
            parts = [t for t in (my, ma, sx) if t is not None]
            pack = red(torch.cat(parts))
            off = 0
            for t in parts:
                t.copy_(pack[off:off + t.numel()])
                off += t.numel()
        mh = torch.cat([my, mw] + ([ma] if ma is not None else [])).tolist()
        y_dim = mh[2]
        rho_scale = max(y_dim * self._std(mh[0:3]) / (W0.numel() * self._std(mh[3:6])), 1.0)
        if ma is not None:
            rho_scale *= mh[6] / mh[8]

        act_iters = 0
        xidx = None
        # exact-integer evaluation of the 200 per-iteration losses (north_star's "int-simulated" forward)
        use_i8 = bool(self.q_act and not self._act_inited and self.lwq_exact_int and
                      getattr(ops, "conv_i8_supported", lambda *a: False)(geom, self.qlvl_act, self.qlvl_w))
        # ... through the direct-gather kernel for short-K layers and the 256-level first/last layers
        use_i8s = bool((not use_i8) and self.q_act and not self._act_inited and self.lwq_exact_int and
                       getattr(ops, "conv_i8s_supported", lambda *a: False)(geom, self.qlvl_act, self.qlvl_w))
        int_conv = use_i8 or use_i8s
        # ... and of the Gram system (exact integer sums, weights applied per attention class)
        use_gi8 = bool(self.q_act and not self._act_inited and self.lwq_exact_int and
                       getattr(ops, "gram_i8_supported", lambda *a: False)(geom, self.qlvl_act))
        att_cls = ops.att_classes(att) if use_gi8 else None
        use_gi8 = att_cls is not None
        if self.q_act:                                                     # (:64-72)
            if self._act_inited:
                xq = to_ndhwc(self._quantize_act(x))
            else:
                kw_fit = dict(abs_sums=sx) if sx is not None else {}
                a_act, act_iters, st = ops.fit_scale(xn, self.qlvl_act, 0.0, 1.0, reducer=red or None,
                                                     guess_iters=12 * self.qlvl_act, **kw_fit)
                self.alpha_act.data = torch.tensor(a_act, dtype=x.dtype, device=dev)
                xq, _, xidx = ops.quant_dequant_f64path(xn, st, self.qlvl_act, 0.0, 1.0, want_idx=int_conv or use_gi8)
        else:
            xq = xn

        rho = self.lwq_rho * rho_scale                                    # (:74-76)
        rho_m = self.lwq_rho_max * rho_scale
        eta = self.lwq_eta * rho_scale

        # Losses of the 200 iterates from the unweighted Gram system of the same integer pass (effq_gram_loss) instead of
        # 200 passes over the voxels, where the voxel count is far above n = k^3 c1 + 1 (the conv is cheaper on the small
        # volumes of the wide layers)
        n_sys = nw + int(has_b)
        loss_gram = None
        use_gl = bool(use_gi8 and GRAM_LOSS_DEFAULT and hasattr(ops, "gram_loss") and n_sys <= GRAM_LOSS_MAX_N and
                      yn.numel() // c2 >= 8 * n_sys)
        # ... and on the wide layers (n above GRAM_LOSS_MAX_N: the fp64 evaluation would cost more than the conv pass) with
        # the quadratic form on the i8 matrix cores: both of its factors are small integers there (effq_gram_loss_i8)
        use_gl8 = bool(use_gi8 and GRAM_LOSS_DEFAULT and GRAM_LOSS_I8 and not use_gl and
                       GRAM_LOSS_MAX_N < n_sys <= GRAM_LOSS_I8_MAX_N and
                       getattr(ops, "gram_loss_i8_supported", lambda *a: False)(c2, n_sys, has_b, self.qlvl_w))
        if use_gi8 and (use_gl or use_gl8):
            A0, B0, Au, Bu = ops.gram_i8(xidx, att_cls, yn, geom, has_b, self.alpha_act.data, self.qlvl_act,
                                         unweighted=True)
            loss_gram = (Au, Bu, syy_local)
            if use_gl8:
                planes = ops.gram_loss_i8_planes(Au, has_b, self.alpha_act.data, self.qlvl_act, yn.numel() // c2)
                loss_gram = (Au, Bu, syy_local, planes) if planes is not None else None
        elif use_gi8:                                                      # (:87-91, solver.py:282-314)
            A0, B0 = ops.gram_i8(xidx, att_cls, yn, geom, has_b, self.alpha_act.data, self.qlvl_act)
        else:
            A0, B0 = ops.gram(xq, att, yn, geom, has_b)
            # full-precision input (first conv, classifier: quirk Q14): the same 200 losses from an fp64 Gram system
            if (not self.q_act and GRAM_LOSS_DEFAULT and self.lwq_exact_int and n_sys <= GRAM_LOSS_MAX_N and
                    yn.numel() // c2 >= 8 * n_sys and
                    getattr(ops, "gram_f64_supported", lambda *a: False)(geom, has_b)):
                Au, Bu = ops.gram_f64(xq, yn, geom, has_b)
                loss_gram = (Au, Bu, syy_local)
        if red:                      # one message per layer: upper triangle of A0 + B0
            if hasattr(ops, "gram_reduce"):
                ops.gram_reduce(A0, B0, red)
            else:
                red(A0)
                red(B0)

        # The 200 iterations (:99-144) are enqueued by ONE library call (effq_admm_run): the serial chain prox ->
        # scale fixed point -> projection/dual on this stream, the loss of iteration i (conv + MSE, used only to pick
        # the best iterate) on a second stream while the chain computes i+1, the inverses of A(rho) for the later rho
        # values (A changes only with rho: 4 inverses per layer, not 200 LU solves) on a third.  Per-iteration results
        # stay in rings, the squared errors in `hist`; the best iterate is selected afterwards - so the data-parallel
        # reduction of the 200 losses is ONE collective per layer.
        import time as _time
        t_loop0 = _time.perf_counter()
        loss_kind = 1 if use_i8 else (2 if use_i8s else 0)
        kw = dict(loss_gram=loss_gram) if loss_gram is not None else {}
        if self.lwq_verbose:
            kw['residuals'] = True        # the per-iteration residual norms of the reference's progress line (:114-127)
        run = ops.admm_run(A0, B0, W0, b0, geom, yn, xq=xq, xidx=xidx, act_alpha=self.alpha_act.data,
                           act_levels=self.qlvl_act, loss_kind=loss_kind, rho=rho, rho_max=rho_m, eta=eta,
                           iters=self.lwq_iter, period=RHO_PERIOD, levels=self.qlvl_w,
                           overlap=self.lwq_overlap_loss, **kw)
        t_enq = _time.perf_counter() - t_loop0     # host time to enqueue the 200 iterations (diagnostic)
        red(run.hist)
        best_G, best_b, best = ops.admm_select_best(run)
        # (:161-166) the final loss - and, from the same pass, the layer's quantised output, which forward() hands to the
        # next layer right after this call (PTQConv.py:160-163 runs the same conv a second time): the fp32 activation
        # quant-dequant is fused into the tile load, exactly as in the quantised forward.  Enqueued BEFORE the layer's
        # one device->host read, which then carries its two sums as well.
        fuse = self.q_act and not self._act_inited
        # ... on the i8 matrix cores where the level ids of the input are at hand and the shape has a kernel (32 -> 32 and
        # 64 -> 64 channels at 3^3: the layers with the most voxels): an exact integer contraction + one multiply-add per
        # output, HBM-bound, instead of the f32 conv (2.6 -> 0.3 ms at 16 x 64^3 voxels)
        fwd_i8 = bool(FORWARD_I8 and fuse and xidx is not None and self.lwq_exact_int and
                      getattr(ops, "conv_i8_out_supported", lambda *a: False)(geom, self.qlvl_act, self.qlvl_w))
        if fwd_i8:
            st_best = run.state_ring.index_select(0, best[1:2].to(torch.long)).reshape(-1)    # the best iterate's scale
            out, fin = ops.conv_forward_i8(xidx, best_G, best_b, geom, yn, att, self.alpha_act.data, self.qlvl_act,
                                           st_best, self.qlvl_w)
        else:
            out, fin = ops.conv_step(xn if fuse else xq, best_G, best_b, geom, yn, att,
                                     act_alpha=self.alpha_act.data if fuse else None,
                                     act_levels=self.qlvl_act if fuse else 0, want_out=True)
        self._ptq_out = out if (fuse or not self.q_act) else None
        red(fin)
        info = ops.admm_read(run, best, extra=fin)                         # ONE host sync for the loop and the final loss
        t_loop = _time.perf_counter() - t_loop0
        if _os.environ.get("EFFQ_FP_TRAJ_STATS") and getattr(run, "fp_pred", None) is not None:     # diagnostic
            print(f"[fp_traj] {getattr(self, 'name', '?')}: {ops.read_fp_pred(run.fp_pred)}", flush=True)
        a_w, w_iters, hist, best_h = info["alpha_w"], info["w_iters"], info["hist"], info["best"]
        if self.lwq_verbose and getattr(run, "res", None) is not None:
            # "print every 10 admm iters" (EfficientQConv.py:124-127), after the loop: the iterations are enqueued as a whole
            res, r_i = run.res.cpu(), rho
            for i in range(self.lwq_iter):
                if i % 10 == 0:
                    print(f'ADMM iter {i + 1}: primal residual = {float(res[i, 0]) ** 0.5:.4f}, '
                          f'dual residual = {r_i * float(res[i, 1]) ** 0.5:.4f}, rho = {r_i:.4f}, eta = {eta:.4f}, '
                          f'loss = {hist[i] / (y_dim * 1.0):.7f}.')
                if i % RHO_PERIOD == 0:
                    r_i = r_i * 2 if r_i * 2 <= rho_m else rho_m
        if info["err"] >= 1000:
            raise RuntimeError(f'{self.name}: the unweighted Gram system is not the integer system effq_gram_loss_i8 '
                               f'expects (flag {info["err"] // 1000})')
        if info["err"] != 0:                                               # layer_helper.py:62-64
            if info["err"] == 2:
                raise RuntimeWarning(f'Exceed maximum iteration ({100 * self.qlvl_w}) for alpha optimization')
            # state 3 = the grid barrier of the cooperative fixed point timed out (its workgroups were not co-resident):
            # its barrier words in the reduction workspace are left non-zero - clear them before reporting
            if hasattr(ops, "_red_ws"):
                ops._red_ws.zero_()
            raise RuntimeError(f'weight-scale fixed point did not finish (device state {info["err"]}: grid barrier '
                               f'time-out of the cooperative kernel)')
        if hasattr(ops, "release_retired"):
            ops.release_retired()        # every stream was joined and the host has synchronised
        self.weight.data = best_G.reshape(self.weight.shape)               # (:147-158)
        if has_b:
            self.bias.data = best_b
        self.alpha_w.data = torch.tensor(a_w, dtype=x.dtype, device=dev)   # LAST iterate's scale (quirk Q6)
        fin_h = info["extra"]
        numel = y_dim * 1.0
        lossf = (fin_h[1] if att is not None else fin_h[0]) / numel
        if self.layer_loss is not None:
            self.layer_loss.append(f'{self.name:45s}:{lossf}')
        self.last_trace = dict(rho_scale=rho_scale, best_iter=int(best_h[1]), best_mse=best_h[0] / numel,
                               final_mse=fin_h[0] / numel, layer_loss=lossf, act_iters=act_iters,
                               w_iters=w_iters, alpha_w=a_w, loss_history=[h / numel for h in hist],
                               exact_int=int_conv or loss_gram is not None, exact_gram=use_gi8,
                               gram_loss=loss_gram is not None, host_enqueue_s=t_enq, admm_loop_s=t_loop)

    def compute_quant_error(self, output_fp, Qw, Qact):
        """EfficientQConv.py:168-172."""
        ops = get_ops(Qact.device)
        _, sq = ops.conv_step(to_ndhwc(Qact), Qw, None if self.bias is None else self.bias.data,
                              self._geom(Qact), to_ndhwc(output_fp))
        return sq[0].item() / output_fp.numel()
