"""TEST-ONLY stand-in for efficientq_amd.hip_ops.HipOps built on the CPU oracle.

It lets the not-gpu tests drive the PRODUCT's host logic (qconv.EfficientQConvHIP.ptq,
calibrate.calibrate_model, the data-parallel reductions) on a CPU, including world_size-2 gloo
runs.  It is never importable from the package: tests install it by monkeypatching
``efficientq_amd.qconv.get_ops``.  Each op restates the arithmetic the oracle / reference uses
(torch CPU), e.g. the prox step is a fresh ``torch.linalg.solve`` like solver.py:331.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import effq_oracle as O


def _ncdhw(t):
    return t.permute(0, 4, 1, 2, 3)


class OracleOps:
    def __init__(self):
        self.device = torch.device("cpu")

    # a1/a3
    def quant_dequant_f32(self, x, alpha, levels, lo, hi, want_idx=False):
        y = O.discretize(x / alpha, levels, lo, hi) * alpha
        if want_idx:
            return y, O.quant_index(x / alpha, levels, lo, hi).to(torch.uint8)
        return y

    def quant_dequant_f64path(self, x, state, levels, lo, hi, want_b=False, want_idx=False):
        a = state[0].item()
        b = O.discretize(x.double() / a, levels, lo, hi).float()
        return a * b, (b if want_b else None), None

    # a2
    def abs_sum(self, x):
        return torch.tensor([x.double().abs().sum().item(), float(x.numel())], dtype=torch.float64)

    def moments(self, x):
        xd = x.double()
        return torch.tensor([xd.sum().item(), (xd * xd).sum().item(), float(x.numel())], dtype=torch.float64)

    def new_fp_state(self):
        return torch.zeros(5, dtype=torch.float64)

    def fit_scale(self, x, levels, lo, hi, reducer=None, guess_iters=16, state=None, abs_sums=None):
        st = state if state is not None else self.new_fp_state()
        xd = x.double()
        if abs_sums is not None:
            s0 = abs_sums
        else:
            s0 = self.abs_sum(x)
            if reducer is not None:
                reducer(s0)
        a, a_old, n, cap = (s0[0] / s0[1]).item(), -999.0, 0, 100 * levels
        while abs(a - a_old) > 1e-5 and n < cap:
            b = O.discretize(xd / a, levels, lo, hi)
            sums = torch.stack([(b * xd).sum(), (b * b).sum()])
            if reducer is not None:
                reducer(sums)
            a_old, a = a, (sums[0] / sums[1]).item()
            n += 1
        if n == cap:
            raise RuntimeWarning(f"Exceed maximum iteration ({cap}) for alpha optimization")
        st[0] = a
        return a, n, st

    def weight_fixed_point(self, wstar, dual, v, levels, state, guess=16):
        self.admm_presum(wstar, dual, v)
        _, it, _ = self.fit_scale(v, levels, -1.0, 1.0, state=state)
        return it

    def fp_check(self, state, err_flag):
        pass      # fit_scale above raises on the spot, like the reference

    @staticmethod
    def read_fp_state(state):
        return state[0].item(), 0, 1

    # a5/a6
    def gram(self, x_ndhwc, att, y_ndhwc, geom, has_bias, A0=None, B0=None):
        k = (geom.KD, geom.KH, geom.KW)
        x, y = _ncdhw(x_ndhwc), _ncdhw(y_ndhwc)
        c2, c1 = geom.C2, geom.C1
        w0 = torch.zeros(c2, c1, *k)
        ps = O.ProxSystem(x, y, k, (geom.SD, geom.SH, geom.SW), (geom.PD, geom.PH, geom.PW), w0,
                          torch.zeros(c2) if has_bias else None, att)
        if A0 is not None:
            A0 += ps.A0
            B0 += ps.B0
            return A0, B0
        return ps.A0.clone(), ps.B0.clone()

    # a7: the "inverse" is just the assembled system matrix; prox_solve does the reference's LU solve
    def spd_inverse(self, A0, has_bias, rho, eta, out=None):
        n = A0.shape[0]
        eye = torch.eye(n)
        if has_bias:
            q = torch.eye(n)
            q[-1, -1] = 0
            return A0 + rho * q + eta * eye
        return A0 + (rho + eta) * eye

    def prox_solve(self, B0, A, W0, b0, G, dual, rho, eta, wstar, bstar):
        c2, n = B0.shape
        if b0 is not None:
            B = B0 + eta * torch.cat([W0.reshape(c2, -1), b0.unsqueeze(1)], dim=1)
            B[:, : n - 1] += rho * (G - dual).reshape(c2, -1)
        else:
            B = B0 + rho * (G - dual).reshape(c2, -1) + eta * W0.reshape(c2, -1)
        what = torch.linalg.solve(A, B.T).T
        if b0 is not None:
            wstar.copy_(what[:, :-1].reshape(wstar.shape))
            bstar.copy_(what[:, -1])
        else:
            wstar.copy_(what.reshape(wstar.shape))

    # a4
    def admm_presum(self, wstar, dual, v):
        v.copy_(wstar + dual)

    def admm_project_dual(self, v, wstar, state, levels, G, dual, dual_div, Gq=None):
        a = state[0].item()
        b = O.discretize(v.double() / a, levels, -1.0, 1.0).float()
        g = a * b
        d = wstar - g + dual
        if dual_div != 1.0:
            d = d / dual_div
        G.copy_(g)
        dual.copy_(d)

    # the whole loop (the product issues it through effq_admm_run; same order of operations, EfficientQConv.py:99-144)
    def admm_run(self, A0, B0, W0, b0, geom, y_ndhwc, *, xq=None, xidx=None, act_alpha=None, act_levels=0,
                 loss_kind=0, rho, rho_max, eta, iters, period, levels, overlap=True, residuals=False):
        from types import SimpleNamespace
        has_b = b0 is not None
        c2 = B0.shape[0]
        G = W0.clone()
        dual = torch.zeros_like(W0)
        wstar, v = torch.empty_like(W0), torch.empty_like(W0)
        r = SimpleNamespace(iters=iters, nw=W0.numel(), c2=c2, has_b=has_b, G_ring=[], b_ring=[] if has_b else None,
                            hist=torch.zeros(iters, 2, dtype=torch.float64), w_iters=[], alpha_w=None)
        st = self.new_fp_state()
        r.res = torch.zeros(iters, 2, dtype=torch.float64) if residuals else None
        for i in range(iters):
            A = self.spd_inverse(A0, has_b, rho, eta)
            bstar = torch.empty(c2) if has_b else None
            self.prox_solve(B0, A, W0, b0, G, dual, rho, eta, wstar, bstar)
            r.w_iters.append(self.weight_fixed_point(wstar, dual, v, levels, st))
            dual_div = 1.0
            if i % period == 0:
                dual_div = 2.0 if rho * 2 <= rho_max else rho_max / rho
            Gn = torch.empty_like(W0)
            self.admm_project_dual(v, wstar, st, levels, Gn, dual, dual_div)
            if residuals:
                r.res[i, 0] = ((wstar.double() - Gn.double()) ** 2).sum()
                r.res[i, 1] = ((Gn.double() - G.double()) ** 2).sum()
            _, sq = self.conv_step(xq, Gn, bstar, geom, y_ndhwc)
            r.hist[i] = sq
            r.G_ring.append(Gn)
            if has_b:
                r.b_ring.append(bstar)
            G = Gn
            if i % period == 0:
                rho = rho * 2 if rho * 2 <= rho_max else rho_max
        r.alpha_w = st[0].item()
        return r

    def admm_select_best(self, run):
        h = run.hist[:, 0].tolist()
        bi = 0
        for i in range(1, run.iters):
            if h[i] < h[bi]:
                bi = i
        best = torch.tensor([h[bi], float(bi)], dtype=torch.float64)
        return run.G_ring[bi].clone(), (run.b_ring[bi].clone() if run.has_b else None), best

    @staticmethod
    def admm_read(run, best, extra=None):
        return dict(hist=run.hist[:, 0].tolist(), best=best.tolist(), alpha_w=run.alpha_w, w_iters=run.w_iters, err=0,
                    extra=extra.double().reshape(-1).tolist() if extra is not None else None)

    def conv_step(self, x_ndhwc, weight, bias, geom, y_ndhwc=None, att=None, act_alpha=None, act_levels=0,
                  want_out=False, sqerr=None):
        x = _ncdhw(x_ndhwc)
        if act_alpha is not None:
            x = O.quantize_act_f32(x, act_alpha, act_levels)
        w = weight.reshape(geom.C2, geom.C1, geom.KD, geom.KH, geom.KW)
        out = F.conv3d(x, w, bias, (geom.SD, geom.SH, geom.SW), (geom.PD, geom.PH, geom.PW))
        if y_ndhwc is not None:
            y = _ncdhw(y_ndhwc)
            d2 = (out - y) ** 2
            if sqerr is None:
                sqerr = torch.zeros(2, dtype=torch.float64)
            # F.mse_loss / weighted mean of the reference, kept as SUMS (mean * numel)
            sqerr[0] = d2.mean().item() * d2.numel()
            sqerr[1] = (att.unsqueeze(1) * d2).mean().item() * d2.numel() if att is not None else sqerr[0]
        return (out.permute(0, 2, 3, 4, 1).contiguous() if want_out else None), sqerr


def install(monkeypatch):
    import efficientq_amd.qconv as Q
    ops = OracleOps()
    monkeypatch.setattr(Q, "get_ops", lambda device: ops)
    return ops
