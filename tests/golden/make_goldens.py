"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run only in the build container (``/root/reference`` present):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

The reference is imported read-only (``sys.dont_write_bytecode``); its one
missing dependency on the orchestrator path, ``nibabel``, is replaced by a
no-op stand-in module.  Only inputs and outputs (data)
are written; no reference source travels.  Fixtures: G1..G10 of SURVEY.md 8(c), G11 for row f1, G12 for row f3,
g5b / g5d (wide layers: reference self-spread and the fp64 anchor), g5e (a 32 -> 32 layer on V >> n voxels), g6c / g6d (GPU hook
behaviour; wide whole nets), g6e (LiTS init_stride 2,2,1), g6f (whole nets at 16 / 16 levels).
With no arguments EVERY fixture is regenerated.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
torch.set_num_threads(8)

_nib = types.ModuleType("nibabel")


class _Nii:
    def __init__(self, *a, **k):
        pass

    def to_filename(self, f):
        pass


_nib.Nifti1Image, _nib.load = _Nii, (lambda f: None)
sys.modules["nibabel"] = _nib

import models  # noqa: E402
from models import layer_helper, solver, fold_bn, factoryQ, factory_blk  # noqa: E402
from models.PTQConv import PTQConv  # noqa: E402
import definer  # noqa: E402
import ptqer  # noqa: E402

EQ_MOD = sys.modules["models.EfficientQConv"]


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KB")


def relu_gauss(gen, *shape):
    return torch.relu(torch.randn(*shape, generator=gen))


# ---------------------------------------------------------------- G1 discretize
def g1():
    gen = torch.Generator().manual_seed(101)
    out = {}
    for L in (4, 16, 256):
        for (lo, hi) in ((-1, 1), (0, 1)):
            d = (hi - lo) / (L - 1)
            v = torch.randn(3072, generator=gen) * 0.8
            # exact half-way points between levels, level points, and out of range
            k = torch.arange(0, min(L, 256) - 1, dtype=torch.float32)
            half = (k + 0.5) * d + lo
            lev = k * d + lo
            extra = torch.tensor([lo - 1.5, hi + 2.0, lo, hi, 0.0, -0.0, 1e-8, -1e-8])
            v = torch.cat([v, half, lev, extra])[:4096].contiguous()
            q32 = layer_helper.discretize(v, L, lo, hi)
            q64 = layer_helper.discretize(v.double(), L, lo, hi)
            tag = f"L{L}_{'w' if lo < 0 else 'a'}"
            out[f"{tag}_in"] = v
            out[f"{tag}_q32"] = q32
            out[f"{tag}_q64"] = q64
            # PTQConv._quantize_act style: x/alpha then *alpha (fp32), PTQConv.py:114-116
            alpha = torch.tensor(0.7341)
            out[f"{tag}_qdq32"] = layer_helper.discretize(v / alpha, L, lo, hi) * alpha
    save("g1_discretize.npz", **out)


# ---------------------------------------------------------------- G2 project_by_iter
def g2():
    gen = torch.Generator().manual_seed(202)
    out = {}
    act = relu_gauss(gen, 2, 8, 16, 16, 16)
    wgt = torch.randn(16, 16, 3, 3, 3, generator=gen) * 0.05
    out["act"] = act
    out["wgt"] = wgt
    for L in (4, 16, 256):
        cnt = {"n": 0}
        orig = layer_helper.discretize

        def counting(*a, **k):
            cnt["n"] += 1
            return orig(*a, **k)

        layer_helper.discretize = counting
        try:
            a, b = layer_helper.project_by_iter(act, L, 0, 1)
            out[f"act_L{L}_alpha"] = np.float64(a)
            out[f"act_L{L}_iters"] = np.int64(cnt["n"] - 1)
            out[f"act_L{L}_idx"] = torch.round(b * (L - 1)).to(torch.uint8)
            out[f"act_L{L}_b"] = b if L <= 16 else b[:1, :1]
            cnt["n"] = 0
            a, b = layer_helper.project_by_iter(wgt, L, -1, 1)
            out[f"wgt_L{L}_alpha"] = np.float64(a)
            out[f"wgt_L{L}_iters"] = np.int64(cnt["n"] - 1)
            out[f"wgt_L{L}_idx"] = torch.round((b + 1) * (L - 1) / 2).to(torch.uint8)
        finally:
            layer_helper.discretize = orig
    save("g2_project.npz", **out)


# ---------------------------------------------------------------- G3 Gram, G4 solve
def g3_g4():
    gen = torch.Generator().manual_seed(303)
    out = {}
    cases = [
        ("k3s1p1", 3, 1, 1, True, True),
        ("k3s221p1", 3, (2, 2, 1), 1, True, True),
        ("k1s1p0", 1, 1, 0, True, False),
        ("k3s1p1_nobias_noatt", 3, 1, 1, False, False),
    ]
    for tag, k, s, p, has_b, has_att in cases:
        N, c1, c2 = 2, 3, 5
        x = relu_gauss(gen, N, c1, 5, 6, 7)
        w = torch.randn(c2, c1, k, k, k, generator=gen) * 0.2
        b = torch.randn(c2, generator=gen) * 0.1 if has_b else None
        y = F.conv3d(x, w, b, s, p) + 0.05 * torch.randn(
            F.conv3d(x, w, b, s, p).shape, generator=gen)
        att = torch.randint(1, 4, y[:, 0].shape, generator=gen).float() if has_att else None
        qs = solver.QuadraSolver(x, y, k, k, k, s, p, device="cpu", mu=0, eta=1.3, W0=w.clone(),
                                 att=att, b0=b.clone() if has_b else None)
        out[f"{tag}_x"], out[f"{tag}_y"], out[f"{tag}_w"] = x, y, w
        if has_b:
            out[f"{tag}_b"] = b
        if has_att:
            out[f"{tag}_att"] = att
        out[f"{tag}_A0"], out[f"{tag}_B0"] = qs.A0, qs.B0
        G = w + 0.01 * torch.randn(w.shape, generator=gen)
        out[f"{tag}_G"] = G
        res = qs.solve(7.5, 1.3, G)
        if has_b:
            out[f"{tag}_wstar"], out[f"{tag}_bstar"] = res
        else:
            out[f"{tag}_wstar"] = res
    save("g3g4_gram_solve.npz", **out)


# ---------------------------------------------------------------- G5 one layer ptq
def run_layer(c1, c2, k, stride, pad, N, S, L_w, L_a, q_act, seed, with_mask, bias=True, iters=None, inputs=None):
    """iters: run only the first `iters` ADMM iterations (lwq_iter is an instance attribute, EfficientQConv.py:23);
    the loss history then has `iters` entries and rec["wstar0"/"bstar0"] hold the FIRST proximal solve.
    inputs: dict(w, b, x_fp, x, y, mask) built elsewhere (tests/golden_inputs.py) instead of drawn here."""
    gen = torch.Generator().manual_seed(seed)
    conv = models.EfficientQConv(c1, c2, k, stride, pad, 1, 1, bias, q_weight=True, qlvl=L_w,
                                 q_act=q_act, qlvl_act=L_a)
    if iters is not None:
        conv.lwq_iter = int(iters)
    if inputs is not None:
        with torch.no_grad():
            conv.weight.copy_(inputs["w"])
            conv.bias.copy_(inputs["b"])
        conv.set_fp()
        y = conv(inputs["x_fp"]).detach()
        assert torch.equal(y, inputs["y"]), "the FP target is exact by construction: every conv returns the same bits"
        x = inputs["x"].clone()
    else:
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (1.0 / (c1 * k ** 3) ** 0.5))
            if bias:
                conv.bias.copy_(torch.randn(c2, generator=gen) * 0.1)
        x_fp = relu_gauss(gen, N, c1, S, S, S)
        conv.set_fp()
        y = conv(x_fp).detach()
        # quantised-upstream stand-in: perturbed input (quirk Q9)
        x = torch.relu(x_fp + 0.05 * torch.randn(x_fp.shape, generator=gen))
    conv.output_fp = y
    conv.name = f"layer_c{c1}_{c2}_k{k}"
    conv.layer_loss = []
    if with_mask:
        m_full = inputs["mask"].clone() if inputs is not None else torch.randint(1, 4, y[:, 0].shape, generator=gen).float()
        m_half = torch.ones(N, *[d // 2 for d in y.shape[2:]])
        conv.mask_pyramid = [m_half, m_full]
    w_in, b_in = conv.weight.data.clone(), (conv.bias.data.clone() if bias else None)

    losses, alphas = [], []
    orig_mse, orig_proj = F.mse_loss, EQ_MOD.project_by_iter

    def mse_spy(*a, **kw):
        r = orig_mse(*a, **kw)
        losses.append(r.item())
        return r

    first_proj = []

    def proj_spy(v, L, lo=-1., hi=1.):
        a, b = orig_proj(v, L, lo, hi)
        alphas.append((lo, a))
        if lo < 0 and not first_proj:
            first_proj.append(torch.round((b.detach() - lo) / ((hi - lo) / (L - 1))).to(torch.uint8))
        return a, b

    first_solve = []
    orig_solve = solver.QuadraSolver.solve

    def solve_spy(self_, *a, **kw):
        r = orig_solve(self_, *a, **kw)
        if not first_solve:
            first_solve.append(tuple(t.detach().clone() for t in r) if isinstance(r, tuple) else (r.detach().clone(),))
        return r

    F.mse_loss, EQ_MOD.project_by_iter = mse_spy, proj_spy
    solver.QuadraSolver.solve = solve_spy
    try:
        with torch.no_grad():
            conv.ptq(x)
    finally:
        F.mse_loss, EQ_MOD.project_by_iter = orig_mse, orig_proj
        solver.QuadraSolver.solve = orig_solve
    n_it = conv.lwq_iter
    aw_hist = np.array([a for lo, a in alphas if lo < 0], dtype=np.float64)
    rec = dict(x=x, y=y, w_in=w_in, loss_hist=np.array(losses[:n_it], dtype=np.float64),
               final_mse=np.float64(losses[n_it]), aw_hist=aw_hist, wstar0=first_solve[0][0], G0idx=first_proj[0],
               weight=conv.weight.data, alpha_w=conv.alpha_w.data, alpha_act=conv.alpha_act.data,
               layer_loss=np.float64(float(conv.layer_loss[0].split(":")[1])),
               meta=np.array([c1, c2, k, pad, N, S, L_w, L_a, int(q_act), int(with_mask)]),
               stride=np.array(solver.triplet(stride)))
    if bias:
        rec["b_in"] = b_in
        rec["bias"] = conv.bias.data
        rec["bstar0"] = first_solve[0][1]
    if with_mask:
        rec["mask_full"] = m_full
    # the forward that feeds the next layer (PTQConv.py:157-162)
    conv.set_quantized()
    rec["fwd_q"] = conv(x).detach()
    return rec


def g5():
    out = {}
    for tag, kw in {
        "L4": dict(c1=8, c2=8, k=3, stride=1, pad=1, N=2, S=12, L_w=4, L_a=4, q_act=True, seed=505,
                   with_mask=True),
        "L16": dict(c1=8, c2=8, k=3, stride=1, pad=1, N=2, S=12, L_w=16, L_a=16, q_act=True, seed=506,
                    with_mask=False),
        "first": dict(c1=2, c2=8, k=3, stride=2, pad=1, N=2, S=12, L_w=256, L_a=256, q_act=False,
                      seed=507, with_mask=True),
        "k1": dict(c1=8, c2=16, k=1, stride=1, pad=0, N=2, S=8, L_w=4, L_a=4, q_act=True, seed=508,
                   with_mask=False),
    }.items():
        rec = run_layer(**kw)
        rec.pop("wstar0", None), rec.pop("bstar0", None), rec.pop("G0idx", None)   # (kept out of this fixture: see g5d)
        for k, v in rec.items():
            out[f"{tag}_{k}"] = v
        print(tag, "layer_loss", rec["layer_loss"], "best", int(np.argmin(rec["loss_hist"])))
    save("g5_layer_ptq.npz", **out)


def g5b():
    """Reference self-spread at the dominant widths: the SAME 32->32 (and 64->64) 3^3 layer calibrated by the
    reference with 1 and with 8 BLAS threads.  The two runs differ only in the summation order
    inside the library GEMM / conv kernels; whatever distance separates them is the reference's own reproducibility
    floor for this layer shape, and the bar of the product tests is anchored on it."""
    out = {}
    # (c64: V = 2 * 12^3 = 3456 output voxels for n = 1729 unknowns: an over-determined system like the real layers)
    for tag, kw in G5B_CASES.items():
        sub = 2 if tag == "c64" else 1          # fwd_q is kept on every sub-th voxel per axis (fixture size)
        recs = {}
        for nt in (1, 8):
            torch.set_num_threads(nt)
            recs[nt] = run_layer(**kw)
            recs[nt].pop("wstar0", None), recs[nt].pop("bstar0", None), recs[nt].pop("G0idx", None)
        torch.set_num_threads(8)
        base = recs[8]
        for k in ("x", "y", "w_in", "b_in", "mask_full", "meta", "stride"):
            out[f"{tag}_{k}"] = base[k]
        out[f"{tag}_mask_full"] = base["mask_full"].to(torch.uint8)
        for nt, rec in recs.items():
            assert torch.equal(rec["x"], base["x"]) and torch.equal(rec["y"], base["y"])
            rec["fwd_q"] = rec["fwd_q"][:, :, ::sub, ::sub, ::sub].contiguous()
            for k in ("loss_hist", "final_mse", "aw_hist", "weight", "bias", "alpha_w", "alpha_act", "layer_loss",
                      "fwd_q"):
                out[f"{tag}_t{nt}_{k}"] = rec[k]
        out[f"{tag}_fwd_sub"] = np.int64(sub)
        a, b = recs[1], recs[8]
        L = kw["L_w"]
        lv = lambda t: torch.round((t / t.abs().max() + 1) * (L - 1) / 2)
        spread = dict(
            layer_loss=abs(a["layer_loss"] - b["layer_loss"]) / b["layer_loss"],
            best_mse=abs(a["loss_hist"].min() - b["loss_hist"].min()) / b["loss_hist"].min(),
            idx_mismatch=(lv(a["weight"]) != lv(b["weight"])).float().mean().item(),
            out_rel_mse=(((a["fwd_q"] - b["fwd_q"]) ** 2).mean() / (b["fwd_q"] ** 2).mean()).item(),
            hist_first5=float(np.max(np.abs(a["loss_hist"][:5] - b["loss_hist"][:5]) / b["loss_hist"][:5])),
            hist_max=float(np.max(np.abs(a["loss_hist"] - b["loss_hist"]) / b["loss_hist"])))
        for k, v in spread.items():
            out[f"{tag}_spread_{k}"] = np.float64(v)
        print(tag, {k: f"{v:.3e}" for k, v in spread.items()}, "layer_loss t1/t8", a["layer_loss"], b["layer_loss"],
              "best it", int(np.argmin(a["loss_hist"])), int(np.argmin(b["loss_hist"])))
    save("g5b_wide_layers.npz", **out)


G5B_CASES = {
    "c32": dict(c1=32, c2=32, k=3, stride=1, pad=1, N=2, S=12, L_w=4, L_a=4, q_act=True, seed=2024, with_mask=True),
    "c64": dict(c1=64, c2=64, k=3, stride=1, pad=1, N=2, S=12, L_w=4, L_a=4, q_act=True, seed=2025, with_mask=True),
}


def g5d():
    """fp64 ANCHOR for the wide layers of g5b.  For the same 32->32 / 64->64 layers:
      * the reference's FIRST proximal solve w*_0, b*_0 and its first 8 losses, with 1 and with 8 BLAS threads
        (lwq_iter = 8: the prefix of the 200-iteration runs of g5b, checked against them);
      * the oracle's fp64 evaluation of the same arithmetic (oracle.calibrate_layer(dtype=float64), whose fp32 mode
        reproduces the reference bit for bit): w*_0, b*_0, the whole loss history, final weight levels, layer_loss.
    The product test asserts that the HIP path is no farther from the fp64 values than the reference's own runs."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import effq_oracle as O
    g5b_file = np.load(os.path.join(HERE, "g5b_wide_layers.npz"))
    out = {}
    for tag, kw in G5B_CASES.items():
        recs = {}
        for nt in (1, 8):
            torch.set_num_threads(nt)
            recs[nt] = run_layer(iters=8, **kw)
            assert np.array_equal(recs[nt]["loss_hist"], g5b_file[f"{tag}_t{nt}_loss_hist"][:8]), "prefix of the g5b run"
        torch.set_num_threads(8)
        base = recs[8]
        assert np.array_equal(base["x"].numpy(), g5b_file[f"{tag}_x"])
        N = base["x"].shape[0]
        pyr = [torch.ones(N, *[d // 2 for d in base["y"].shape[2:]]), base["mask_full"]]
        # the switch off reproduces the reference (fp32) ...
        chk = O.calibrate_layer(base["x"], base["y"], base["w_in"], base["b_in"], 1, 1, qlvl_w=kw["L_w"],
                                qlvl_act=kw["L_a"], mask_pyramid=pyr, iters=8)
        assert np.array_equal(np.array(chk.loss_history), base["loss_hist"]), "oracle fp32 == reference"
        assert torch.equal(chk.wstar0, base["wstar0"])
        # ... and switched on it is the anchor
        r = O.calibrate_layer(base["x"], base["y"], base["w_in"], base["b_in"], 1, 1, qlvl_w=kw["L_w"],
                              qlvl_act=kw["L_a"], mask_pyramid=pyr, dtype=torch.float64)
        L = kw["L_w"]
        idx = torch.round((r.weight / r.weight.abs().max() + 1) * (L - 1) / 2).to(torch.uint8)
        out[f"{tag}_f64_loss_hist"] = np.array(r.loss_history, dtype=np.float64)
        out[f"{tag}_f64_wstar0"] = r.wstar0.float()          # (fp32 storage: 6e-8 relative, the distances are >= 1e-6)
        out[f"{tag}_f64_bstar0"] = r.bstar0
        out[f"{tag}_f64_weight_idx"] = idx
        # iteration 0 in fp64: v_0 = w*_0 (dual = 0), its scale and level ids, and each weight's distance (in level
        # units) from the nearest rounding boundary - a run may differ from fp64 at iteration 0 only where that is tiny
        fit0 = O.fit_scale(r.wstar0, L, -1.0, 1.0)
        dl_ = 2.0 / (L - 1)
        u0 = (torch.clamp(r.wstar0 / fit0.alpha, -1.0, 1.0) + 1.0) / dl_
        out[f"{tag}_f64_aw0"] = np.float64(fit0.alpha)
        out[f"{tag}_f64_G0idx"] = torch.round(u0).to(torch.uint8)
        out[f"{tag}_f64_margin0"] = (u0 - torch.floor(u0) - 0.5).abs().float()
        out[f"{tag}_f64_layer_loss"] = np.float64(r.layer_loss)
        out[f"{tag}_f64_alpha_act"] = np.float64(r.alpha_act)
        for nt, rec in recs.items():
            out[f"{tag}_t{nt}_wstar0"] = rec["wstar0"]
            out[f"{tag}_t{nt}_bstar0"] = rec["bstar0"]
            out[f"{tag}_t{nt}_loss8"] = rec["loss_hist"]
            out[f"{tag}_t{nt}_G0idx"] = rec["G0idx"]
            out[f"{tag}_t{nt}_aw0"] = np.float64(rec["aw_hist"][0])
            mm = rec["G0idx"] != out[f"{tag}_f64_G0idx"]
            print(tag, f"t{nt}: iteration-0 level ids differ from fp64 at {int(mm.sum())} of {mm.numel()} weights, "
                  f"largest boundary margin among them {float(out[f'{tag}_f64_margin0'][mm].max()) if mm.any() else 0:.2e}")
        w64 = r.wstar0
        d = {nt: ((recs[nt]["wstar0"].double() - w64).norm() / w64.norm()).item() for nt in (1, 8)}
        dl = {nt: np.abs(recs[nt]["loss_hist"][:5] - np.array(r.loss_history[:5])) / np.array(r.loss_history[:5])
              for nt in (1, 8)}
        print(tag, "w*0 rel distance to fp64: t1 %.3e t8 %.3e" % (d[1], d[8]), "loss[0:5] rel distance t1", dl[1],
              "t8", dl[8], "fp64 layer_loss", r.layer_loss, "best it", r.best_iter)
    save("g5d_wide_fp64_anchor.npz", **out)


def g5e():
    """The regime the bench runs in: ONE 32 -> 32 3^3 layer calibrated by the reference on
    V >> n voxels - one 1 x 32 x 32^3 and one 1 x 32 x 48^3 volume (V / n = 38 and 128; g5b: 4), with an attention mask, with 1
    and with 8 BLAS threads, plus the oracle's fp64 evaluation of the same arithmetic.  The tensors are rebuilt from a
    seed by tests/golden_inputs.py (the FP target is exact in fp32 by construction); the fixture keeps their checksums,
    what the reference produced, and the reference's own t1-vs-t8 distances."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import effq_oracle as O
    from tests import golden_inputs as GI
    out = {}
    for tag, case in GI.G5E_CASES.items():
        S = case["S"]
        inp = GI.wide_layer_inputs(S, case["seed"])
        y64 = F.conv3d(inp["x_fp"].double(), inp["w"].double(), inp["b"].double(), 1, 1)
        assert torch.equal(inp["y"].double(), y64), "target exact in fp32"
        kw = dict(c1=32, c2=32, k=3, stride=1, pad=1, N=1, S=S, L_w=4, L_a=4, q_act=True, seed=case["seed"],
                  with_mask=True, inputs=inp)
        recs = {}
        for nt in (1, 8):
            torch.set_num_threads(nt)
            recs[nt] = run_layer(**kw)
        torch.set_num_threads(8)
        out[f"{tag}_meta"] = recs[8]["meta"]
        out[f"{tag}_seed"] = np.int64(case["seed"])
        for k in ("w", "b", "x", "y", "mask"):
            out[f"{tag}_chk_{k}"] = GI.checksums(inp[k])
        out[f"{tag}_x_sub"] = inp["x"][:, ::8, ::8, ::8, ::8]
        out[f"{tag}_y_sub"] = inp["y"][:, ::8, ::8, ::8, ::8]
        sub = 4
        for nt, rec in recs.items():
            assert torch.equal(rec["x"], inp["x"]) and torch.equal(rec["y"], inp["y"])
            fq = rec["fwd_q"]
            out[f"{tag}_t{nt}_fwd_q_sub"] = fq[:, :, ::sub, ::sub, ::sub].contiguous()
            for k in ("loss_hist", "final_mse", "aw_hist", "weight", "bias", "alpha_w", "alpha_act", "layer_loss",
                      "wstar0", "bstar0", "G0idx"):
                out[f"{tag}_t{nt}_{k}"] = rec[k]
        out[f"{tag}_fwd_sub"] = np.int64(sub)
        a, b = recs[1], recs[8]
        L = 4
        lv = lambda t: torch.round((t / t.abs().max() + 1) * (L - 1) / 2)
        spread = dict(
            layer_loss=abs(a["layer_loss"] - b["layer_loss"]) / b["layer_loss"],
            best_mse=abs(a["loss_hist"].min() - b["loss_hist"].min()) / b["loss_hist"].min(),
            idx_mismatch=(lv(a["weight"]) != lv(b["weight"])).float().mean().item(),
            out_rel_mse=(((a["fwd_q"] - b["fwd_q"]) ** 2).mean() / (b["fwd_q"] ** 2).mean()).item(),
            hist_max=float(np.max(np.abs(a["loss_hist"] - b["loss_hist"]) / b["loss_hist"])))
        for k, v in spread.items():
            out[f"{tag}_spread_{k}"] = np.float64(v)
        print(tag, "reference t1 vs t8:", {k: f"{v:.3e}" for k, v in spread.items()}, "layer_loss", a["layer_loss"],
              b["layer_loss"], "best it", int(np.argmin(a["loss_hist"])), int(np.argmin(b["loss_hist"])))
        # the oracle: fp32 mode == reference (8 threads), fp64 mode = anchor
        pyr = [torch.ones(1, S // 2, S // 2, S // 2), inp["mask"]]
        chk = O.calibrate_layer(inp["x"], inp["y"], inp["w"], inp["b"], 1, 1, qlvl_w=4, qlvl_act=4, mask_pyramid=pyr)
        d_or = abs(chk.layer_loss - b["layer_loss"]) / b["layer_loss"]
        idx_or = (lv(chk.weight) != lv(b["weight"])).float().mean().item()
        print(tag, "oracle fp32 vs reference t8: layer_loss rel", d_or, "idx mismatch", idx_or,
              "loss hist equal:", np.array_equal(np.array(chk.loss_history), b["loss_hist"]))
        out[f"{tag}_oracle32_layer_loss"] = np.float64(chk.layer_loss)
        r = O.calibrate_layer(inp["x"], inp["y"], inp["w"], inp["b"], 1, 1, qlvl_w=4, qlvl_act=4, mask_pyramid=pyr,
                              dtype=torch.float64)
        out[f"{tag}_f64_loss_hist"] = np.array(r.loss_history, dtype=np.float64)
        out[f"{tag}_f64_wstar0"] = r.wstar0.float()
        out[f"{tag}_f64_bstar0"] = r.bstar0
        out[f"{tag}_f64_weight_idx"] = lv(r.weight.float()).to(torch.uint8)
        out[f"{tag}_f64_layer_loss"] = np.float64(r.layer_loss)
        out[f"{tag}_f64_alpha_act"] = np.float64(r.alpha_act)
        for nt in (1, 8):
            print(tag, f"t{nt} vs fp64: layer_loss rel", abs(recs[nt]["layer_loss"] - r.layer_loss) / r.layer_loss,
                  "idx mismatch", (lv(recs[nt]["weight"]) != lv(r.weight.float())).float().mean().item(),
                  "w*0 rel", ((recs[nt]["wstar0"].double() - r.wstar0.double()).norm() / r.wstar0.double().norm()).item())
    save("g5e_wide_layer_many_voxels.npz", **out)


# ---------------------------------------------------------------- G6 whole do_ptq
def _shape3(S):
    return tuple(S) if isinstance(S, (tuple, list)) else (S, S, S)


def tiny_args(task, L, S, nmod, ncls, multi_label=None, init_stride="1", width="8,16,8",
              depth="1,1,1", root="/tmp/effq_gold"):
    return argparse.Namespace(
        pretrain=f"{root}/round1_fp.pkl", resume=None, device="cpu", task=task, round="1", suffix="",
        config=None, test_fp=False, no_test=True, save_nii=False, bin_label=None,
        multi_label=multi_label, model="UResQ", nMod=nmod, nClass=ncls, init_stride=init_stride,
        width=width, depth=depth, dilation=None, nla="relu", norm="bn", drop_rate=0.5, ds="simple",
        init_kernel=3, hetero_dim=True, blk="mid", qconv="effq", qlvl_w=L, qlvl_a=L,
        q_first="256,-1", q_last="256,-1", lwq_dataid=0, lwq_batchsz=2,
        lwq_patchsz=",".join(str(v) for v in _shape3(S)), lwq_verbose=False)


def randomise(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.Conv3d):
                fan = m.weight[0].numel()
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan) ** 0.5)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
            if isinstance(m, nn.BatchNorm3d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)


def _copying_hook(m, i, o):
    """hooks.py:5-6 as it behaves with the model on a GPU (the reference's intended device, entrance.py --device): there
    `o.detach().cpu()` is a COPY taken before the next block's in-place ReLU (factoryQ.py:76-77) rewrites the conv's
    output.  On the CPU - the only device this container has - `.cpu()` returns the tensor itself, and the targets of the
    convs that feed an in-place ReLU silently become relu(y).  The g6c fixtures are generated with the copy restored."""
    m.output_fp = o.detach().clone()


def run_do_ptq(task, L, S, width="8,16,8", copy_targets=False, threads=8, seed=606, lits_stride="1"):
    """The reference's real do_ptq (ptqer.py:282-387) on a seeded random network and seeded volumes; returns what it
    produced.  copy_targets: hooks.py:5-6 as it behaves on a GPU (see _copying_hook).  lits_stride: init_stride of the
    LiTS-style net ("2,2,1" = config/lits_ptq.yaml: anisotropic first conv, mask pyramid and final up-sampling)."""
    root = "/tmp/effq_gold"
    torch.set_num_threads(threads)
    orig_hook = ptqer.forward_hook
    if copy_targets:
        ptqer.forward_hook = _copying_hook
    os.makedirs(root + "/snap", exist_ok=True)
    if task == "lits":
        args = tiny_args("lits", L, S, 1, 3, width=width, init_stride=lits_stride)
    else:
        args = tiny_args("brats", L, S, 2, 4, multi_label="brats", init_stride="2,2,2", width=width)
    QConv, Qinfo, kwQ = definer.get_conv_class(args)
    mc, _ = definer.get_model_cube(args, QConv, kwQ)
    model = mc["model"]
    randomise(model, seed)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    torch.save({"state_dict": model.state_dict()}, args.pretrain)
    nmod = args.nMod
    g = torch.Generator().manual_seed(1)
    vols = torch.randn(2, nmod, *_shape3(S), generator=g)
    if task == "brats":
        zz = torch.arange(S).float() - (S - 1) / 2
        r = (zz[:, None, None] ** 2 + zz[None, :, None] ** 2 + zz[None, None, :] ** 2).sqrt()
        vols = vols * (r < 0.45 * S).float()

    class DS(torch.utils.data.Dataset):
        def __len__(s):
            return 2

        def __getitem__(s, i):
            return vols[i], torch.zeros(*_shape3(S)).long()

        def use_fix_transform(s):
            pass

    cube = types.SimpleNamespace(trainseqloader=torch.utils.data.DataLoader(DS(), 1, shuffle=False))
    captured = {}
    orig_set_mask = ptqer.set_mask

    def spy_set_mask(m, pyr):
        captured["pyr"] = [p.clone() for p in pyr]
        return orig_set_mask(m, pyr)

    ptqer.set_mask = spy_set_mask

    def snap(name, compress=False):
        captured[name] = {k: v.clone() for k, v in model.state_dict().items()}

    tester = types.SimpleNamespace(test_as_is=lambda *a, **k: None, snapshot=snap)
    # capture output_q / output_fp through extract_nii
    outs = []
    orig_extract = ptqer.extract_nii

    def spy_extract(o, task="lits"):
        outs.append(o.detach().clone())
        return orig_extract(o, task=task)

    ptqer.extract_nii = spy_extract
    try:
        ptqer.do_ptq(args, mc, cube, tester, root + "/snap")
    finally:
        ptqer.set_mask, ptqer.extract_nii = orig_set_mask, orig_extract
        ptqer.forward_hook = orig_hook
        torch.set_num_threads(8)
    with open(root + "/snap/layer_loss.txt") as f:
        ll = f.read().strip().split("\n")
    with open(root + "/snap/class_voxel_nums.txt") as f:
        nums = [int(float(t)) for t in f.read().split()]
    with open(root + "/snap/time_cost.txt") as f:
        print("reference time_cost:", f.read())
    return dict(ll=ll, nums=nums, outs=outs, captured=captured, sd0=sd0, vols=vols, L=L, S=S)


def g6(task="lits", L=4, S=16, tag="g6_tiny_lits_L4", brats_zero=False, copy_targets=False, lits_stride="1",
       second_run_threads=None):
    """second_run_threads: also run the reference with that many BLAS threads and store its layer losses and FP-vs-Q
    agreement (`t1_*`): the reference's own reproducibility floor for this case."""
    r = run_do_ptq(task, L, S, copy_targets=copy_targets, lits_stride=lits_stride)
    ll, nums, outs, captured, sd0, vols = r["ll"], r["nums"], r["outs"], r["captured"], r["sd0"], r["vols"]
    sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4), slice(None, None, 4))
    out = {"vols_seed": np.int64(1), "vols_check": vols[:, :, ::8, ::8, ::8],
           "output_q_sub": outs[0][-1][sub], "output_fp_sub": outs[1][-1][sub],
           "output_q_stats": np.array([outs[0][-1].double().mean().item(), outs[0][-1].double().std().item()]),
           "output_fp_stats": np.array([outs[1][-1].double().mean().item(), outs[1][-1].double().std().item()]),
           "agree": np.float64(((outs[0][-1] > 0) == (outs[1][-1] > 0)).float().mean().item()),
           "layer_names": np.array([l.split(":")[0].strip() for l in ll]),
           "layer_loss": np.array([float(l.split(":")[1]) for l in ll], dtype=np.float64),
           "class_nums": np.array(nums, dtype=np.int64),
           "meta": np.array([L, _shape3(S)[0]]), "shape": np.array(_shape3(S)), "init_stride": np.array(lits_stride if task == "lits" else "2,2,2")}
    if second_run_threads:
        r2 = run_do_ptq(task, L, S, copy_targets=copy_targets, lits_stride=lits_stride, threads=second_run_threads)
        o2 = r2["outs"]
        out["t1_layer_loss"] = np.array([float(l.split(":")[1]) for l in r2["ll"]], dtype=np.float64)
        out["t1_agree"] = np.float64(((o2[0][-1] > 0) == (o2[1][-1] > 0)).float().mean().item())
        out["t1_threads"] = np.int64(second_run_threads)
        print(tag, "reference self-spread: layer_loss", np.abs(out["t1_layer_loss"] - out["layer_loss"]) / out["layer_loss"],
              "agreement", float(out["agree"]), float(out["t1_agree"]))
    for i, p in enumerate(captured["pyr"]):
        out[f"pyr{i}"] = p.to(torch.uint8)
        assert (p == p.to(torch.uint8).float()).all()
    for k, v in sd0.items():
        out["sd0/" + k] = v
    for k, v in captured["state_in_fp.pkl"].items():
        out["sdq/" + k] = v
    for k, v in captured["state_in_int8.pkl"].items():
        if k.endswith(".weight") and v.dtype == torch.uint8:
            out["sdi/" + k] = v
    save(tag + ".npz", **out)
    print("\n".join(ll))


def g6d(tasks=("lits", "brats")):
    """Whole-network goldens at the widths that matter: the reference's real do_ptq on a
    width 32,64,32 net (n = 865 / 1729 systems), LiTS 2 x 1x32^3 and BraTS 2 x 2x64^3 with zero background, 4/4 levels.
    Per task: the GPU hook behaviour (targets copied, _copying_hook) and the CPU hook behaviour (targets aliased), each
    with 8 and with 1 BLAS thread - the second run is the reference's own reproducibility floor per layer and for the
    FP-vs-Q agreement.  The start weights are NOT stored: they are `randomise(model, 606)`, which the
    product's synth.randomise_network reproduces (a checksum per tensor is stored instead)."""
    for task, S in (("lits", 32), ("brats", 64)):
        if task not in tasks:
            continue
        runs = {"copy_t8": run_do_ptq(task, 4, S, "32,64,32", True, 8),
                "copy_t1": run_do_ptq(task, 4, S, "32,64,32", True, 1),
                "alias_t8": run_do_ptq(task, 4, S, "32,64,32", False, 8),
                "alias_t1": run_do_ptq(task, 4, S, "32,64,32", False, 1)}
        base = runs["copy_t8"]
        out = {"meta": np.array([4, S]), "vols_seed": np.int64(1), "net_seed": np.int64(606),
               "vols_check": base["vols"][:, :, ::8, ::8, ::8],
               "layer_names": np.array([l.split(":")[0].strip() for l in base["ll"]]),
               "class_nums": np.array(base["nums"], dtype=np.int64)}
        for k, v in base["sd0"].items():
            if v.dtype.is_floating_point:
                out["sd0sum/" + k] = np.float64(v.double().sum().item())
        for i, p in enumerate(base["captured"]["pyr"]):
            out[f"pyr{i}"] = p.to(torch.uint8)
            assert (p == p.to(torch.uint8).float()).all()
        sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4), slice(None, None, 4))
        for tag, r in runs.items():
            oq, of = r["outs"][0][-1], r["outs"][1][-1]
            out[f"{tag}/layer_loss"] = np.array([float(l.split(":")[1]) for l in r["ll"]], dtype=np.float64)
            out[f"{tag}/agree"] = np.float64(((oq > 0) == (of > 0)).float().mean().item())
            out[f"{tag}/output_q_sub"] = oq[sub]
            out[f"{tag}/out_rel_mse_vs_fp"] = np.float64((((oq - of) ** 2).mean() / (of ** 2).mean()).item())
            assert r["nums"] == base["nums"]
            sdq = r["captured"]["state_in_fp.pkl"]
            sdi = r["captured"]["state_in_int8.pkl"]
            for k, v in sdq.items():
                if k.endswith("alpha_w") or k.endswith("alpha_act"):
                    out[f"{tag}/sdq/{k}"] = v
                elif k.endswith(".bias") and k[:-5] + ".alpha_w" in sdq:
                    out[f"{tag}/sdq/{k}"] = v
            if tag == "copy_t8":
                out["output_fp_sub"] = of[sub]
            for k, v in sdi.items():
                if k.endswith(".weight") and v.dtype == torch.uint8 and tag == "copy_t8":
                    out[f"{tag}/sdi/{k}"] = v
        a, b = runs["copy_t1"], runs["copy_t8"]
        la = np.array([float(l.split(":")[1]) for l in a["ll"]])
        lb = np.array([float(l.split(":")[1]) for l in b["ll"]])
        print(task, "self-spread of layer_loss (t1 vs t8):", np.abs(la - lb) / lb)
        print(task, "agree copy t8/t1/alias:", out["copy_t8/agree"], out["copy_t1/agree"], out["alias_t8/agree"])
        save(f"g6d_wide_{task}_L4.npz", **out)


# ---------------------------------------------------------------- G7 BN fold
def g7():
    gen = torch.Generator().manual_seed(707)
    blk = factoryQ.NLAConvBN_3d(4, 6, 3, 1, 1, 1, bias=False)
    randomise(blk, 708)
    blk.eval()
    x = torch.randn(2, 4, 6, 6, 6, generator=gen)
    w0 = blk.conv.weight.data.clone()
    bn = blk.bn
    rec = dict(x=x, w=w0, gamma=bn.weight.data.clone(), beta=bn.bias.data.clone(),
               mean=bn.running_mean.clone(), var=bn.running_var.clone(), eps=np.float64(bn.eps),
               y_before=blk(x.clone()).detach())
    fold_bn.search_fold_and_remove_bn(blk)
    rec.update(w_fold=blk.conv.weight.data.clone(), b_fold=blk.conv.bias.data.clone(),
               y_after=blk(x.clone()).detach())
    save("g7_bnfold.npz", **rec)


# ---------------------------------------------------------------- G8 att map / pyramid
def g8():
    gen = torch.Generator().manual_seed(808)
    out = {}
    for task, C in (("lits", 3), ("brats", 3)):
        logits = torch.randn(3, 2, C, 8, 8, 8, generator=gen) * 2
        if task == "lits":
            logits[:, :, 0] += 1.5
        data = torch.randn(2, 2, 8, 8, 8, generator=gen)
        data[:, :, :2] = 0
        data[:, :, :, :, 6:] = 0
        body = (data[:, 0] != 0).bool() if task == "brats" else torch.ones_like(data[:, 0]).bool()
        wmap, nums = ptqer.get_att_weight_map(logits, torch.ones_like(data[:, 0]).bool(), "p:0.5", task=task)
        for st in ("1", "2,2,2", "2,2,1"):
            if st != "1":
                # ("2,2,1" = config/lits_ptq.yaml: avg_pool3d(out, (2,2,1)) as the pyramid's first level, ptqer.py:148-150)
                sf = tuple(float(v) for v in st.split(","))
                lg = F.interpolate(logits[-1], scale_factor=sf, mode="trilinear")[None]
                dt = F.interpolate(data, scale_factor=sf, mode="nearest")
                bd = (dt[:, 0] != 0).bool() if task == "brats" else torch.ones_like(dt[:, 0]).bool()
                wm, nm = ptqer.get_att_weight_map(lg, torch.ones_like(dt[:, 0]).bool(), "p:0.5", task=task)
                pyr = ptqer.get_mask_pyramid(lg, bd, wm, st, num_lvls=3, task=task)
                key = f"{task}_s2" if st == "2,2,2" else f"{task}_s221"
                out[f"{key}_logits"], out[f"{key}_data"] = lg, dt
            else:
                wm, nm = wmap, nums
                pyr = ptqer.get_mask_pyramid(logits, body, wmap, st, num_lvls=3, task=task)
                key = f"{task}_s1"
                out[f"{key}_logits"], out[f"{key}_data"] = logits, data
            out[f"{key}_nums"] = np.array(nm, dtype=np.int64)
            out[f"{key}_wvals"] = np.array([wm[i] for i in range(len(wm))], dtype=np.float64)
            for i, p in enumerate(pyr):
                out[f"{key}_pyr{i}"] = p.to(torch.uint8)
                assert (p == p.to(torch.uint8).float()).all()
    save("g8_attmask.npz", **out)


# ---------------------------------------------------------------- G9 int storage
def g9():
    gen = torch.Generator().manual_seed(909)
    out = {}
    for L in (4, 16, 256):
        conv = PTQConv(4, 6, 3, 1, 1, qlvl=L)
        a_best = 0.0831
        idx = torch.randint(0, L, conv.weight.shape, generator=gen)
        q = a_best * (idx.float() * (2 / (L - 1)) - 1)
        conv.weight.data = q.clone()
        conv.alpha_w.data = torch.tensor(a_best * 1.0007)      # mismatched scale (quirk Q6)
        conv.store_int_weight()
        stored = conv.weight.data.clone()
        conv.restore_fp_weight()
        out[f"L{L}_q"], out[f"L{L}_alpha"] = q, conv.alpha_w.data.clone()
        out[f"L{L}_int"], out[f"L{L}_restored"] = stored, conv.weight.data.clone()
    save("g9_intweight.npz", **out)


# ---------------------------------------------------------------- G10 ResBlock mid
def g10():
    gen = torch.Generator().manual_seed(1010)
    rb = factory_blk.ResBlockWithType(4, 4, 0.5, 1, factoryQ.ReLU(True), nn.Conv3d, nn.BatchNorm3d, "mid")
    randomise(rb, 1011)
    rb.eval()
    x = torch.randn(2, 4, 6, 6, 6, generator=gen)
    y = rb(x.clone())
    rec = {"x": x, "y": y.detach()}
    for k, v in rb.state_dict().items():
        rec["sd/" + k] = v
    save("g10_resblock_mid.npz", **rec)


def g11():
    """Row f1: sliding-window split / stitch (utils/transforms.py:784-852) and the FP-vs-Q Dice helper
    (utils/metrics.py:21-25, 119-148)."""
    from utils import transforms as tfm
    from utils import metrics as M
    gen = torch.Generator().manual_seed(11)
    out = {}
    for tag, shape, psz, ov in (("a", (2, 2, 12, 13, 14), 6, 2), ("b", (1, 1, 12, 12, 12), (6, 12, 6), (2, 0, 3)),
                                ("c", (1, 1, 7, 8, 9), 7, 3)):
        img = torch.randn(*shape, generator=gen)
        patches = tfm.image_to_patch3d(img, psz, ov)
        out[f"{tag}_img"] = img
        out[f"{tag}_npatch"] = np.int64(len(patches))
        out[f"{tag}_patches"] = torch.stack(patches)
        # "network output": 2 heads x (a function of the patch), stitched like validate_seg does (validate.py:240-245)
        preds = [torch.stack([p * 2.0 + 1.0, p.flip(1) - 0.5]) for p in patches]
        out[f"{tag}_stitched"] = tfm.patch_to_image3d(img, preds, psz, ov)
    logits = torch.randn(2, 3, 6, 7, 8, generator=gen)
    tgt_l = torch.randint(0, 3, (2, 6, 7, 8), generator=gen)
    tgt_b = torch.randint(0, 2, (2, 3, 6, 7, 8), generator=gen)
    out["m_logits"], out["m_tgt_lits"], out["m_tgt_brats"] = logits, tgt_l, tgt_b
    out["m_dice_lits"] = torch.stack([d.float() for d in M.validate_vs_label(logits, tgt_l, "lits")])
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):        # the brats branch prints the shape
        out["m_dice_brats"] = torch.stack([d.float() for d in M.validate_vs_label(logits, tgt_b, "brats")])
    out["m_dice_empty"] = M.dice(torch.zeros(4, dtype=torch.bool), torch.zeros(4, dtype=torch.bool)).float()
    save("g11_sliding_window.npz", **out)


# ---------------------------------------------------------------- G12 tune_activation_range (row f3)
def g12():
    """The reference's tune_activation_range (ptqer.py:238-272; dead code there, run in isolation here) on the tiny
    lits net: alpha_act initialised by the reference's own init pass (need_init=True), then 50 Adam steps."""
    root = "/tmp/effq_gold"
    os.makedirs(root, exist_ok=True)
    S, L = 16, 4
    args = tiny_args("lits", L, S, 1, 3)
    QConv, Qinfo, kwQ = definer.get_conv_class(args)
    mc, _ = definer.get_model_cube(args, QConv, kwQ)
    model = mc["model"]
    randomise(model, 707)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    model.eval()
    fold_bn.search_fold_and_remove_bn(model)
    vols = torch.randn(2, 1, S, S, S, generator=torch.Generator().manual_seed(12))
    ptqer.set_fp(model)
    with torch.no_grad():
        output_fp = model(vols).detach()
    losses = ptqer.tune_activation_range(model, output_fp, vols, max_iter=50, need_init=True)
    names, alphas = [], []
    for n, m in model.named_modules():
        if isinstance(m, PTQConv):
            names.append(n)
            alphas.append(float(m.alpha_act.data))
    # the alphas right after the init pass: run it again on a fresh copy
    mc2, _ = definer.get_model_cube(args, QConv, kwQ)
    m2 = mc2["model"]
    m2.load_state_dict(sd0)
    m2.eval()
    fold_bn.search_fold_and_remove_bn(m2)
    ptqer.set_init_alpha(m2)
    with torch.no_grad():
        m2(vols)
    init = [float(m.alpha_act.data) for n, m in m2.named_modules() if isinstance(m, PTQConv)]
    out = {f"sd0/{k}": v for k, v in sd0.items()}
    out.update(vols_seed=np.int64(12), meta=np.array([S, L]), loss_all=np.array(losses, dtype=np.float64),
               alpha_final=np.array(alphas, dtype=np.float64), alpha_init=np.array(init, dtype=np.float64),
               layer_names=np.array(names))
    print("g12 loss first/last", losses[0], losses[-1], "alphas", [f"{a:.4f}" for a in alphas])
    save("g12_tune_act.npz", **out)


if __name__ == "__main__":
    ALL = ["g1", "g2", "g3g4", "g5", "g5b", "g5d", "g5e", "g6", "g6b", "g6c", "g6d", "g6e", "g6f", "g7", "g8", "g9", "g10",
           "g11", "g12"]
    which = sys.argv[1:] or ALL
    with torch.no_grad():
        for name, fn in (("g1", g1), ("g2", g2), ("g3g4", g3_g4), ("g7", g7), ("g8", g8), ("g9", g9), ("g10", g10),
                         ("g11", g11)):
            if name in which:
                fn()
    if "g5" in which:
        g5()
    if "g5b" in which:
        g5b()
    if "g5d" in which:
        g5d()                      # (reads g5b_wide_layers.npz: after g5b)
    if "g6" in which:
        g6("lits", 4, 32, "g6_tiny_lits_L4")
    if "g6b" in which:
        g6("brats", 4, 64, "g6_tiny_brats_L4")
    if "g6c" in which:
        g6("lits", 4, 32, "g6c_tiny_lits_L4", copy_targets=True)
        g6("brats", 4, 64, "g6c_tiny_brats_L4", copy_targets=True)
    if "g5e" in which:
        g5e()
    if "g6e" in which:
        # LiTS geometry: init_stride "2,2,1" through the real do_ptq, GPU hook behaviour
        # (the pyramid pools five more times after the initial stride, ptqer.py:154-167: the strided axes need 64 voxels)
        g6("lits", 4, (64, 64, 32), "g6e_tiny_lits_s221_L4", copy_targets=True, lits_stride="2,2,1", second_run_threads=1)
    if "g6f" in which:
        # configs[2] arithmetic on a whole net: 16 / 16 levels, GPU hook behaviour
        g6("lits", 16, 32, "g6f_tiny_lits_L16", copy_targets=True, second_run_threads=1)
        g6("brats", 16, 64, "g6f_tiny_brats_L16", copy_targets=True, second_run_threads=1)
    if "g6d" in which:
        g6d()
    if "g6d_lits" in which:
        g6d(("lits",))
    if "g6d_brats" in which:
        g6d(("brats",))
    if "g12" in which:
        g12()
