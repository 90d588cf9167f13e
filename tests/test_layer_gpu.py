"""Layer- and model-level parity of the HIP path on a real MI355X (-m gpu): the product's
EfficientQConvHIP.ptq / calibrate_model against reference goldens (g5, g6) and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import effq_oracle as O
from tests.test_host_cpu import _layer_from_gold, _tiny, T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _to_dev(conv):
    conv.to(DEV)
    conv.output_fp = conv.output_fp.to(DEV)
    if conv.mask_pyramid:
        conv.mask_pyramid = [m.to(DEV) for m in conv.mask_pyramid]
    return conv


@pytest.mark.parametrize("tag", ["L4", "L16", "first", "k1"])
def test_layer_calibration_matches_reference(gold, tag):
    """north_star bar: per-layer quantised output within 1e-3 relative MSE of the reference path;
    quantiser indices of the calibrated weights bit-exact at 4 levels."""
    g = gold("g5_layer_ptq.npz")
    conv, x, (L_w, L_a, q_act) = _layer_from_gold(g, tag)
    _to_dev(conv)
    conv.set_quantizing()
    with torch.no_grad():
        out = conv(x.to(DEV))
    want_loss = float(g[f"{tag}_layer_loss"])
    got_loss = float(conv.layer_loss[0].split(":")[1])
    assert abs(got_loss - want_loss) <= 1e-3 * want_loss, (got_loss, want_loss)
    wg = T(g[f"{tag}_weight"])
    w = conv.weight.data.cpu()
    lv = lambda t: torch.round((t / t.abs().max() + 1) * (L_w - 1) / 2)
    mism = (lv(w) != lv(wg)).float().mean().item()
    if L_w == 4:
        assert mism == 0.0
    else:
        assert mism <= 0.05          # 16+/256 levels: flat plateau, a few % of indices move
    if q_act:
        a_ref = float(g[f"{tag}_alpha_act"])
        assert abs(conv.alpha_act.item() - a_ref) <= 1e-6 * a_ref
    ref_out = T(g[f"{tag}_fwd_q"])
    rel_mse = ((out.cpu() - ref_out) ** 2).mean() / (ref_out ** 2).mean()
    assert rel_mse <= 1e-3
    tr = conv.last_trace
    # the best-iterate loss is what the reference selected on, to 1e-3
    assert abs(tr["best_mse"] - float(g[f"{tag}_loss_hist"].min())) <= 1e-3 * float(g[f"{tag}_loss_hist"].min())


def test_lwq_verbose_progress_line_on_the_device(gold, capsys):
    """The reference's progress line (EfficientQConv.py:114-127) from the HIP path: residual norms of every iteration from
    one small launch per iteration (effq_admm_run_args.res_ring), printed after the loop; against the oracle."""
    import oracle.effq_oracle as O
    from tests.test_host_cpu import _progress_lines
    g = gold("g5_layer_ptq.npz")
    conv, x, (L_w, L_a, q_act) = _layer_from_gold(g, "L4")
    conv.lwq_verbose = True
    _to_dev(conv)
    conv.set_quantizing()
    with torch.no_grad():
        conv(x.to(DEV))
    lines = _progress_lines(capsys.readouterr().out)
    assert [l[0] for l in lines] == list(range(1, 200, 10))
    c1, c2, k, pad, N, S, _, _, _, with_mask = [int(v) for v in g["L4_meta"]]
    mp = [m.cpu() for m in conv.mask_pyramid] if with_mask else None
    ref = O.calibrate_layer(x, T(g["L4_y"]), T(g["L4_w_in"]), T(g["L4_b_in"]), tuple(int(v) for v in g["L4_stride"]), pad,
                            qlvl_w=L_w, qlvl_act=L_a, q_act=q_act, mask_pyramid=mp)
    for it, pres, dres, rho, eta, loss in lines:
        i = it - 1
        assert abs(rho - ref.rho_history[i]) <= 1e-4 * ref.rho_history[i] + 1e-4
        assert abs(pres - ref.primal_res[i]) <= 5e-3 * ref.primal_res[i] + 5e-3, (i, pres, ref.primal_res[i])
        assert abs(dres - ref.dual_res[i]) <= 5e-3 * ref.dual_res[i] + 5e-3, (i, dres, ref.dual_res[i])   # (late: a flip or none)
        assert abs(loss - ref.loss_history[i]) <= 1e-3 * ref.loss_history[i] + 1e-7


def _rel_mse(a, b):
    return (((a - b) ** 2).mean() / (b ** 2).mean()).item()


@pytest.mark.parametrize("tag", ["c32", "c64"])
def test_wide_layer_calibration_within_reference_self_spread(gold, tag):
    """The dominant BraTS shape (32->32 3^3) and a 64->64 layer against the REFERENCE run twice on the same
    inputs - with 1 and with 8 BLAS threads (tests/golden/make_goldens.py:g5b).  The two reference runs sit
    S_out ~ 3e-2 output rel-MSE, S_idx ~ 5-22 % weight indices and S_ll ~ 9e-4 layer_loss apart: on these layers
    the discrete ADMM trajectory amplifies last-ulp differences of the Gram sums, so north_star's 1e-3 output
    bar is met by the reference against itself only on small layers (G5).  Bars here: north_star's 1e-3, or a
    small multiple of the reference's own spread where that is larger; what precedes the divergence (scales,
    rho_scale, the first iterations) is held to the tight bar."""
    g = gold("g5b_wide_layers.npz")
    conv, x, (L_w, L_a, q_act) = _layer_from_gold(g, tag)
    conv.mask_pyramid[1] = conv.mask_pyramid[1].float()
    conv.lwq_trace = True
    _to_dev(conv)
    conv.set_quantizing()
    with torch.no_grad():
        out = conv(x.to(DEV)).cpu()
    sp = {k: float(g[f"{tag}_spread_{k}"]) for k in ("layer_loss", "best_mse", "idx_mismatch", "out_rel_mse",
                                                      "hist_first5")}
    got_loss = float(conv.layer_loss[0].split(":")[1])
    hist = np.array(conv.last_trace["loss_history"])
    lv = lambda t: torch.round((t / t.abs().max() + 1) * (L_w - 1) / 2)
    w = conv.weight.data.cpu()
    rep = {}
    for nt in (1, 8):
        ref_hist = g[f"{tag}_t{nt}_loss_hist"]
        a_ref = float(g[f"{tag}_t{nt}_alpha_act"])
        assert abs(conv.alpha_act.item() - a_ref) <= 1e-6 * a_ref
        # The first iterations: north_star's bar / 10, or 5x the largest distance the two reference runs have shown
        # between themselves up to that iteration.  One weight index that flips at a rounding tie moves the loss of an
        # iteration by 2e-4..1e-3 at 64 channels (measured by perturbing w* by 1e-5 relative, the accuracy of the
        # reference's own fp32 LU solve at cond(A) = 1.6e4: tests/diagnostics/debug_wide.py); the reference's runs with 1 and 8
        # threads are 7e-4 apart at iteration 0 already and 1.7 % by iteration 4, and both are 2e-4 away from exact
        # fp64 arithmetic at iteration 1, where they happen to agree with each other to 2e-7.
        self5 = np.abs(g[f"{tag}_t1_loss_hist"][:5] - g[f"{tag}_t8_loss_hist"][:5]) / g[f"{tag}_t8_loss_hist"][:5]
        bar5 = np.maximum(1e-4, 5 * np.maximum.accumulate(self5))
        assert np.all(np.abs(hist[:5] - ref_hist[:5]) <= bar5 * ref_hist[:5]), (hist[:5], ref_hist[:5], bar5)
        d_ll = abs(got_loss - float(g[f"{tag}_t{nt}_layer_loss"])) / float(g[f"{tag}_t{nt}_layer_loss"])
        d_best = abs(hist.min() - ref_hist.min()) / ref_hist.min()
        d_idx = (lv(w) != lv(T(g[f"{tag}_t{nt}_weight"]))).float().mean().item()
        sub = int(g[f"{tag}_fwd_sub"])
        d_out = _rel_mse(out[:, :, ::sub, ::sub, ::sub], T(g[f"{tag}_t{nt}_fwd_q"]))
        rep[nt] = (d_ll, d_best, d_idx, d_out)
        # plateau values: the (att-weighted) layer_loss and the (unweighted) best iterate loss measure the same plateau;
        # bar = north_star's 1e-3 or 3x the larger of the two distances the reference keeps from itself
        plateau = max(sp["layer_loss"], sp["best_mse"])
        assert d_ll <= max(1e-3, 3 * plateau), (d_ll, sp)
        assert d_best <= max(1e-3, 3 * plateau), (d_best, sp)
        assert d_idx <= 2 * sp["idx_mismatch"], (d_idx, sp)
        assert d_out <= max(1e-3, 2 * sp["out_rel_mse"]), (d_out, sp)      # the assertion r1 had dropped
    print(f"{tag}: reference self-spread {sp}; hip vs t1/t8 (layer_loss, best, idx, out rel-MSE) = {rep}")


@pytest.mark.parametrize("tag", ["s32", "s48"])
def test_wide_layer_on_many_voxels_against_reference_and_fp64(gold, tag):
    """The regime the bench runs in: ONE 32 -> 32 3^3 layer on V = 32^3 / 48^3 voxels of one volume
    (V / n = 38 / 128), through the DEFAULT path (exact-integer Gram, per-iteration losses from the Gram system), against
    the reference run with 1 and with 8 BLAS threads and the oracle's fp64 evaluation of the same arithmetic (g5e; inputs
    rebuilt from the seed, FP target exact by construction).  What the fixture shows: more voxels do NOT make the
    reference agree with itself - its two runs separate within the first iterations and end 3.4 - 3.8 % of the ids,
    2.0 - 2.3e-2 output rel-MSE and 8e-4 (32^3) / 3.5e-3 (48^3) layer_loss apart, up to 1.9e-3 from the fp64 plateau.  So,
    with FIXED bars:
      * everything that precedes the trajectory is held to the tight bar: alpha_act 1e-6; the first proximal solve within
        1e-5 of fp64 (reference: 5 - 7e-6); level ids of iteration 0 equal to fp64's except where the fp64 pre-image lies
        within 1e-4 level units of a rounding boundary, at most 8 of them; loss of iteration 0 within 1e-5 + 5e-4 per such
        id of fp64's;
      * the plateau: layer_loss and best loss within 6e-3 of the fp64 anchor and of either reference run (the reference
        itself: 3.5e-3 between its runs, 1.9e-3 from fp64); weight ids and quantised output within 5 % / 5e-2 rel-MSE of
        the 8-thread run (the reference's own two runs: 3.8 % / 2.3e-2) - north_star's 1e-3 is not met by the reference
        against itself at 32 channels at ANY voxel count, on the output or on the layer loss."""
    from efficientq_amd.qconv import EfficientQConvHIP
    from tests import golden_inputs as GI
    g = gold("g5e_wide_layer_many_voxels.npz")
    S = GI.G5E_CASES[tag]["S"]
    inp = GI.wide_layer_inputs(S, int(g[f"{tag}_seed"]))
    for k in ("w", "b", "x", "y", "mask"):
        assert torch.equal(GI.checksums(inp[k]), T(g[f"{tag}_chk_{k}"])), k
    conv = EfficientQConvHIP(32, 32, 3, 1, 1, 1, 1, True, q_weight=True, qlvl=4, q_act=True, qlvl_act=4, lwq_trace=True)
    conv.weight.data, conv.bias.data = inp["w"].clone(), inp["b"].clone()
    conv.output_fp, conv.name, conv.layer_loss = inp["y"], "layer", []
    conv.mask_pyramid = [torch.ones(1, S // 2, S // 2, S // 2), inp["mask"]]
    _to_dev(conv)
    import efficientq_amd.qconv as Q
    ops = Q.get_ops(torch.device(DEV))
    runs, calls, orig = [], [], ops.admm_run

    def spy(*a, **kw):
        calls.append((a, kw))
        runs.append(orig(*a, **kw))
        return runs[-1]
    ops.admm_run = spy
    try:
        conv.set_quantizing()
        with torch.no_grad():
            out = conv(inp["x"].to(DEV)).cpu()
    finally:
        del ops.admm_run                   # the instance attribute shadowing the method
    torch.cuda.synchronize()
    tr = dict(conv.last_trace)
    assert tr["gram_loss"], "the default path of the bench's dominant layers"
    # w*_0 (which the run overwrites) formed again from the run's operands by the entry points iteration 0 goes through
    (A0, B0, W0, b0, geom, yn), kw = calls[0]
    Ainv = ops.spd_inverse(A0, True, 2 * kw["rho"], kw["eta"])
    wst, bst = torch.empty_like(W0), torch.empty_like(b0)
    ops.prox_solve_shifted(B0, Ainv, W0, b0, W0, torch.zeros_like(W0), kw["rho"], kw["eta"], 2 * kw["rho"], wst, bst)
    torch.cuda.synchronize()
    tr["wstar0"] = wst
    hist = np.array(tr["loss_history"])
    got_ll = float(conv.layer_loss[0].split(":")[1])
    lv = lambda t: torch.round((t / t.abs().max() + 1) * 1.5)
    w = conv.weight.data.cpu()
    f_hist, f_ll = g[f"{tag}_f64_loss_hist"], float(g[f"{tag}_f64_layer_loss"])
    a64 = float(g[f"{tag}_f64_alpha_act"])
    assert abs(conv.alpha_act.item() - a64) <= 1e-6 * a64
    # ---- iteration 0 against fp64
    w64 = T(g[f"{tag}_f64_wstar0"]).double()
    ws0 = tr["wstar0"].double().cpu().reshape(w64.shape) if "wstar0" in tr else None
    rep = {}
    if ws0 is not None:
        rep["w*0 vs fp64"] = ((ws0 - w64).norm() / w64.norm()).item()
        assert rep["w*0 vs fp64"] <= 1e-5, rep
        # level ids of iteration 0: fp64's, except on a rounding boundary
        from oracle import effq_oracle as O
        fit = O.fit_scale(ws0, 4, -1.0, 1.0)
        u = (torch.clamp(ws0 / fit.alpha, -1.0, 1.0) + 1.0) / (2.0 / 3.0)
        fit64 = O.fit_scale(w64, 4, -1.0, 1.0)
        u64 = (torch.clamp(w64 / fit64.alpha, -1.0, 1.0) + 1.0) / (2.0 / 3.0)
        flips = torch.round(u) != torch.round(u64)
        margin = (u64 - torch.floor(u64) - 0.5).abs()
        rep["iteration-0 ids off fp64"] = int(flips.sum())
        assert int(flips.sum()) <= 8 and (not flips.any() or float(margin[flips].max()) <= 1e-4), (rep, margin[flips])
        assert abs(hist[0] - f_hist[0]) <= (1e-5 + 5e-4 * int(flips.sum())) * f_hist[0], (hist[0], f_hist[0], rep)
    else:
        assert abs(hist[0] - f_hist[0]) <= 4e-3 * f_hist[0]
    # ---- the plateau, fixed bars
    refs = {nt: (float(g[f"{tag}_t{nt}_layer_loss"]), g[f"{tag}_t{nt}_loss_hist"].min()) for nt in (1, 8)}
    refs["f64"] = (f_ll, f_hist.min())
    for k, (ll, best) in refs.items():
        rep[f"layer_loss vs {k}"] = abs(got_ll - ll) / ll
        rep[f"best vs {k}"] = abs(hist.min() - best) / best
        assert rep[f"layer_loss vs {k}"] <= 6e-3 and rep[f"best vs {k}"] <= 6e-3, rep
    sub = int(g[f"{tag}_fwd_sub"])
    rep["ids vs t8"] = (lv(w) != lv(T(g[f"{tag}_t8_weight"]))).float().mean().item()
    rep["ids vs fp64"] = (lv(w) != T(g[f"{tag}_f64_weight_idx"]).float()).float().mean().item()
    rep["out rel-MSE vs t8"] = _rel_mse(out[:, :, ::sub, ::sub, ::sub], T(g[f"{tag}_t8_fwd_q_sub"]))
    print(f"{tag}: {rep}; reference t1 vs t8: ids {float(g[f'{tag}_spread_idx_mismatch']):.4f}, out rel-MSE "
          f"{float(g[f'{tag}_spread_out_rel_mse']):.4f}, layer_loss {float(g[f'{tag}_spread_layer_loss']):.2e}")
    assert rep["ids vs t8"] <= 5e-2 and rep["out rel-MSE vs t8"] <= 5e-2, rep
    assert len(torch.unique(w)) <= 4


@pytest.mark.parametrize("task,fname", [("brats", "g6_tiny_brats_L4.npz"), ("lits", "g6_tiny_lits_L4.npz"),
                                        ("brats", "g6c_tiny_brats_L4.npz"), ("lits", "g6c_tiny_lits_L4.npz"),
                                        ("lits", "g6e_tiny_lits_s221_L4.npz"),
                                        ("lits", "g6f_tiny_lits_L16.npz"), ("brats", "g6f_tiny_brats_L16.npz")])
def test_whole_calibration_matches_reference(gold, monkeypatch, task, fname):
    """Whole do_ptq window against the reference, in both of its behaviours: g6_* = as it runs on the CPU (the hook's
    `.cpu()` aliases the conv output there, so the targets of the convs feeding an in-place ReLU are overwritten);
    g6c_* = with the copy a GPU run makes (the product's default; tests/golden/make_goldens.py:_copying_hook).
    g6e = the LiTS geometry of config/lits_ptq.yaml, init_stride "2,2,1" on 64 x 64 x 32 volumes: anisotropic first conv,
    avg_pool3d(out, (2,2,1)) at the head of the mask pyramid (ptqer.py:148-150), final up-sampling x (2,2,1).
    g6f = 16 / 16 levels (BASELINE configs[2] arithmetic) through the whole net.  Both with the GPU hook behaviour."""
    from efficientq_amd import calibrate as K
    monkeypatch.setattr(K, "ALIAS_FP_TARGETS", fname.startswith("g6_"))
    g = gold(fname)
    L = int(g["meta"][0])
    stride = str(g["init_stride"]) if "init_stride" in g.files else None
    args, model, _ = _tiny(task, L=L, init_stride=stride if task == "lits" else None)
    if stride is not None:
        assert args.init_stride == stride
    model.load_state_dict({k[4:]: T(g[k]) for k in g.files if k.startswith("sd0/")}, strict=False)
    model.eval()
    K.search_fold_and_remove_bn(model)
    model.to(DEV)
    S = int(g["meta"][1])
    shape = tuple(int(v) for v in g["shape"]) if "shape" in g.files else (S, S, S)
    nmod = 1 if task == "lits" else 2
    vols = torch.randn(2, nmod, *shape, generator=torch.Generator().manual_seed(int(g["vols_seed"])))
    if task == "brats":
        zz = torch.arange(S).float() - (S - 1) / 2
        r = (zz[:, None, None] ** 2 + zz[None, :, None] ** 2 + zz[None, None, :] ** 2).sqrt()
        vols = vols * (r < 0.45 * S).float()
    assert torch.equal(vols[:, :, ::8, ::8, ::8], T(g["vols_check"]))
    K.set_name(model)
    res = K.calibrate_model(model, vols.to(DEV), task, args.init_stride)
    names = [l.split(":")[0].strip() for l in res["layer_loss"]]
    assert names == g["layer_names"].tolist()
    got = np.array([float(l.split(":")[1]) for l in res["layer_loss"]])
    want = g["layer_loss"]
    # The class census and the masks come from an argmax over the FP logits: a voxel whose two largest logits agree to
    # the last bits may fall to the other class when the convs sum in another order (g6e: ONE voxel of 262 144 moves from
    # class 0 to class 2 against the reference's CPU convs).  Exact wherever no such voxel exists; otherwise the census
    # keeps its total and moves by at most 2 voxels per class, the masks differ in at most 1e-5 of their voxels.
    nums_ref = g["class_nums"].tolist()
    assert sum(res["nums"]) == sum(nums_ref) and max(abs(a - b) for a, b in zip(res["nums"], nums_ref)) <= 2, (res["nums"], nums_ref)
    for i, m in enumerate(res["pyramid"]):
        ref_m = T(g[f"pyr{i}"]).float()
        assert m.shape == ref_m.shape and (m.cpu() != ref_m).float().mean().item() <= 1e-5, i
    if res["nums"] == nums_ref and "s221" not in fname:
        for i, m in enumerate(res["pyramid"]):
            assert torch.equal(m.cpu(), T(g[f"pyr{i}"]).float())
    sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4), slice(None, None, 4))
    assert torch.allclose(res["output_fp"][-1][sub].cpu(), T(g["output_fp_sub"]), atol=2e-5)
    print(f"{fname}: layer_loss distance to the reference {np.abs(got - want) / want}")
    # first layers see identical inputs: 1e-3 relative; later layers drift with the kept plateau iterate
    assert np.all(np.abs(got[:2] - want[:2]) <= 1e-3 * want[:2]), (got, want)
    # (observed: <= 1 % on most layers, up to 8.5 % - towards the LOWER loss - on one late layer of the tiny nets)
    assert np.all(np.abs(got - want) <= 1.2e-1 * want), (got, want)
    assert abs(got.sum() - want.sum()) <= 4e-2 * want.sum()
    agree = ((res["output_q"][-1] > 0) == (res["output_fp"][-1] > 0)).float().mean().item()
    # Dice proxy within 1 pt on the tiny nets - of the NEARER of the reference's runs where the fixture holds two (g6e /
    # g6f: 8 and 1 BLAS threads).  On the 16-level BraTS-style net the reference's own two runs are 4.1 pt apart (0.9230 /
    # 0.9641: random-init logits crowd the sigmoid threshold, and one layer's kept plateau iterate moves them across it)
    refs = [float(g["agree"])] + ([float(g["t1_agree"])] if "t1_agree" in g.files else [])
    print(f"{fname}: FP-vs-Q agreement hip {agree:.4f}, reference {refs}")
    assert min(abs(agree - r) for r in refs) <= 1e-2, (agree, refs)
    assert res["t2"] > res["t1"] > res["t0"]


@pytest.mark.parametrize("task", ["lits", "brats"])
@pytest.mark.parametrize("hook", ["copy", "alias"])
def test_whole_calibration_at_32_and_64_channels_matches_reference(gold, monkeypatch, task, hook):
    """Whole do_ptq window on a width 32,64,32 net (systems of n = 865 / 1729 unknowns - the widths that carry the BraTS
    and LiTS nets) against the reference's real do_ptq (tests/golden/make_goldens.py:g6d), in both hook behaviours.
    g6d also holds the SAME reference run with 1 BLAS thread instead of 8: the distance between the reference's two runs
    is its own reproducibility floor, per layer and for the FP-vs-Q agreement (the Dice proxy).  Bars:
      * masks, class census: exact; FP output: 2e-5 of its largest value;
      * the first layer (identical inputs on both sides, n = 28 / 55: no plateau lottery): layer_loss within
        north_star's 1e-3 (observed 1e-7);
      * every later layer: within max(1e-2, 3 x the largest distance the reference's own two runs show on any layer UP TO
        that one) - a layer inherits the drift of everything upstream, and HIP-vs-t8 and t1-vs-t8 are each ONE draw of
        the same plateau-iterate lottery (the 32-channel layer right behind the first conv already sits 8.5e-4 from
        itself between the reference's runs: north_star's 1e-3 is the noise floor there, not a bar);
      * FP-vs-Q agreement: within north_star's 0.1 pt of the nearer of the reference's two runs (which are 0.08 pt
        apart)."""
    from efficientq_amd import calibrate as K, synth
    monkeypatch.setattr(K, "ALIAS_FP_TARGETS", hook == "alias")
    g = gold(f"g6d_wide_{task}_L4.npz")
    args, model, _ = _tiny(task, width="32,64,32")
    synth.randomise_network(model, int(g["net_seed"]))
    for k, v in model.state_dict().items():
        if f"sd0sum/{k}" in g.files:          # the generator's start weights, by checksum
            assert abs(v.double().sum().item() - float(g[f"sd0sum/{k}"])) <= 1e-9 * max(1.0, abs(float(g[f"sd0sum/{k}"]))), k
    model.eval()
    K.search_fold_and_remove_bn(model)
    model.to(DEV)
    S = int(g["meta"][1])
    nmod = 1 if task == "lits" else 2
    vols = torch.randn(2, nmod, S, S, S, generator=torch.Generator().manual_seed(int(g["vols_seed"])))
    if task == "brats":
        zz = torch.arange(S).float() - (S - 1) / 2
        r = (zz[:, None, None] ** 2 + zz[None, :, None] ** 2 + zz[None, None, :] ** 2).sqrt()
        vols = vols * (r < 0.45 * S).float()
    assert torch.equal(vols[:, :, ::8, ::8, ::8], T(g["vols_check"]))
    K.set_name(model)
    res = K.calibrate_model(model, vols.to(DEV), task, args.init_stride)
    names = [l.split(":")[0].strip() for l in res["layer_loss"]]
    assert names == g["layer_names"].tolist()
    assert res["nums"] == g["class_nums"].tolist()
    for i, m in enumerate(res["pyramid"]):
        assert torch.equal(m.cpu(), T(g[f"pyr{i}"]).float())
    sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4), slice(None, None, 4))
    fp_ref = T(g["output_fp_sub"])
    assert torch.allclose(res["output_fp"][-1][sub].cpu(), fp_ref, rtol=0, atol=2e-5 * fp_ref.abs().max().item())
    got = np.array([float(l.split(":")[1]) for l in res["layer_loss"]])
    want = g[f"{hook}_t8/layer_loss"]
    self_spread = np.abs(g[f"{hook}_t1/layer_loss"] - want) / want
    d = np.abs(got - want) / want
    agree = ((res["output_q"][-1] > 0) == (res["output_fp"][-1] > 0)).float().mean().item()
    refs = [float(g[f"{hook}_t8/agree"]), float(g[f"{hook}_t1/agree"])]
    print(f"{task}/{hook}: layer_loss distance to the reference {d}; reference t1-vs-t8 {self_spread}; agreement hip "
          f"{agree:.5f} reference {refs}")
    assert d[0] <= 1e-3, (d, got, want)
    assert np.all(d <= np.maximum(1e-2, 3 * np.maximum.accumulate(self_spread))), (d, self_spread)
    assert min(abs(agree - r) for r in refs) <= 1e-3, (agree, refs)


def test_workspace_growth_inside_the_overlapped_loop_is_ordered_on_the_loss_stream():
    """a loss-stream workspace that grows mid-layer used to be zero-filled on torch's current
    stream, unordered against the pinned loss stream.  Drop the integer-conv workspaces so that they grow inside
    the loop, once with the two-stream overlap and once serial: the per-iteration losses (exact integer sums,
    deterministic) must be identical."""
    from efficientq_amd.qconv import EfficientQConvHIP, get_ops
    cases = {"i8 32->32": dict(c1=32, c2=32, k=3, s=1, p=1, S=40, N=2, L=4),
             "i8s 4->32 s2, 256 levels": dict(c1=4, c2=32, k=3, s=2, p=1, S=64, N=4, L=256)}
    for name, c in cases.items():
        hist = {}
        for overlap in (True, False):
            gen = torch.Generator().manual_seed(5)
            conv = EfficientQConvHIP(c["c1"], c["c2"], c["k"], c["s"], c["p"], 1, 1, True, q_weight=True,
                                     qlvl=c["L"], q_act=True, qlvl_act=c["L"], lwq_trace=True,
                                     lwq_overlap_loss=overlap)
            with torch.no_grad():
                conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (c["c1"] * 27)) ** 0.5)
                conv.bias.copy_(torch.randn(c["c2"], generator=gen) * 0.1)
            x = torch.relu(torch.randn(c["N"], c["c1"], c["S"], c["S"], c["S"], generator=gen))
            y = torch.nn.functional.conv3d(x, conv.weight.data, conv.bias.data, c["s"], c["p"])
            conv.output_fp, conv.name, conv.layer_loss = y, "ws", []
            _to_dev(conv)
            ops = get_ops(torch.device(DEV))
            torch.cuda.synchronize()
            for key in ("conv_i8", "conv_i8s"):
                ops._ws.pop(key, None)
            conv.set_quantizing()
            with torch.no_grad():
                conv(x.to(DEV))
            assert conv.last_trace["exact_int"], name
            hist[overlap] = np.array(conv.last_trace["loss_history"])
        assert np.array_equal(hist[True], hist[False]), (name, np.abs(hist[True] - hist[False]).max())


def test_cooperative_fixed_point_time_out_is_contained():
    """a barrier time-out of the cooperative weight fixed point (k_fp_coop) used to leave its barrier
    words non-zero, and the launches of the following ADMM iterations - already enqueued - ran on them unsynchronised.
    Now the first time-out POISONS the workspace: later launches return at once with done = 3, ptq raises RuntimeError
    after the layer and zero-fills the workspace, and the next calibration is bit-identical to an undisturbed one.
    Part A forces the path deterministically (poison word set by hand); part B lets real barriers time out (spin limit
    of one poll): whichever way the race goes the run either succeeds with the undisturbed result or raises cleanly."""
    from efficientq_amd.qconv import EfficientQConvHIP, get_ops

    def run():
        gen = torch.Generator().manual_seed(21)
        c1, c2, S, N = 64, 32, 8, 2
        conv = EfficientQConvHIP(c1, c2, 3, 1, 1, 1, 1, True, q_weight=True, qlvl=32, q_act=True, qlvl_act=4, lwq_trace=True)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (c1 * 27)) ** 0.5)
            conv.bias.copy_(torch.randn(c2, generator=gen) * 0.1)
        x = torch.relu(torch.randn(N, c1, S, S, S, generator=gen))
        conv.output_fp = torch.nn.functional.conv3d(x, conv.weight.data, conv.bias.data, 1, 1)
        conv.name, conv.layer_loss = "l", []
        _to_dev(conv)
        conv.set_quantizing()
        with torch.no_grad():
            conv(x.to(DEV))
        return np.array(conv.last_trace["loss_history"]), conv.weight.data.cpu().clone()

    ops = get_ops(torch.device(DEV))
    assert 32768 < 64 * 32 * 27 <= ops.lib.effq_fp_coop_max()        # 32 levels, 55 296 weights: the cooperative kernel
    base_hist, base_w = run()
    words = ops._red_ws.view(torch.int32)
    poison = (8 * 2048 * 4 + 64 + 8) // 4                            # counter[2] in the tail of the reduction workspace
    # ---- A: poisoned workspace ----
    words[poison] = 1
    with pytest.raises(RuntimeError, match="fixed point did not finish"):
        run()
    torch.cuda.synchronize()
    assert int(ops._red_ws.count_nonzero()) == 0                     # the handler left the workspace clean
    hist, w = run()
    assert np.array_equal(hist, base_hist) and torch.equal(w, base_w)
    # ---- B: real time-outs ----
    ops.lib.effq_fp_coop_set_spin_limit(1)
    try:
        try:
            hist, w = run()
            assert np.array_equal(hist, base_hist) and torch.equal(w, base_w)
        except RuntimeError as e:
            assert "fixed point did not finish" in str(e)
    finally:
        ops.lib.effq_fp_coop_set_spin_limit(0)
    torch.cuda.synchronize()
    assert int(ops._red_ws.count_nonzero()) == 0
    hist, w = run()
    assert np.array_equal(hist, base_hist) and torch.equal(w, base_w)


def test_exact_int_and_fp32_loss_paths_agree_on_a_layer():
    """Same layer calibrated with the Gram system and the per-iteration losses on the i8 matrix cores (exact
    integer sums) and on the f32 matrix cores: the first iteration losses agree to fp32 rounding (until the
    first weight index flips on a rounding tie, iteration 5 here), the result to the plateau tolerance."""
    from efficientq_amd.qconv import EfficientQConvHIP
    res = {}
    for exact in (True, False):
        gen = torch.Generator().manual_seed(99)
        c, S, N = 32, 12, 2
        conv = EfficientQConvHIP(c, c, 3, 1, 1, 1, 1, True, q_weight=True, qlvl=4, q_act=True, qlvl_act=4,
                                 lwq_exact_int=exact, lwq_trace=True)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (c * 27)) ** 0.5)
            conv.bias.copy_(torch.randn(c, generator=gen) * 0.1)
        x_fp = torch.relu(torch.randn(N, c, S, S, S, generator=gen))
        y = torch.nn.functional.conv3d(x_fp, conv.weight.data, conv.bias.data, 1, 1)
        x = torch.relu(x_fp + 0.05 * torch.randn(x_fp.shape, generator=gen))
        conv.output_fp, conv.name, conv.layer_loss = y, "l", []
        _to_dev(conv)
        conv.set_quantizing()
        with torch.no_grad():
            conv(x.to(DEV))
        assert conv.last_trace["exact_int"] == exact and conv.last_trace["exact_gram"] == exact
        res[exact] = (np.array(conv.last_trace["loss_history"]), float(conv.layer_loss[0].split(":")[1]))
    hi, hf = res[True][0], res[False][0]
    assert np.all(np.abs(hi[:4] - hf[:4]) <= 5e-6 * hf[:4]), (hi[:5], hf[:5])
    assert abs(res[True][1] - res[False][1]) <= 5e-3 * res[False][1]


def test_losses_from_the_gram_system_equal_the_conv_losses_on_a_layer(monkeypatch):
    """The loss only picks the best iterate: with the 200 losses taken from the unweighted Gram system (effq_gram_loss)
    or from 200 exact-integer conv passes the ADMM chain is the same, so ALL 200 losses must agree to the fp32 rounding
    of the conv epilogue (2e-6) and the winners must be equivalent (layer_loss and output within north_star's 1e-3).  With an
    attention mask (the loop's loss is unweighted, the Gram system is not: quirk Q5)."""
    import efficientq_amd.qconv as Q
    res = {}
    for gram_loss in (True, False):
        monkeypatch.setattr(Q, "GRAM_LOSS_DEFAULT", gram_loss)
        gen = torch.Generator().manual_seed(7)
        c, S, N = 32, 20, 2
        conv = Q.EfficientQConvHIP(c, c, 3, 1, 1, 1, 1, True, q_weight=True, qlvl=4, q_act=True, qlvl_act=4, lwq_trace=True)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (c * 27)) ** 0.5)
            conv.bias.copy_(torch.randn(c, generator=gen) * 0.1)
        x_fp = torch.relu(torch.randn(N, c, S, S, S, generator=gen))
        y = torch.nn.functional.conv3d(x_fp, conv.weight.data, conv.bias.data, 1, 1)
        x = torch.relu(x_fp + 0.05 * torch.randn(x_fp.shape, generator=gen))
        conv.output_fp, conv.name, conv.layer_loss = y, "l", []
        conv.mask_pyramid = [torch.randint(0, 4, (N, S, S, S), generator=gen).float()]
        _to_dev(conv)
        conv.set_quantizing()
        with torch.no_grad():
            out = conv(x.to(DEV))
        tr = conv.last_trace
        assert tr["gram_loss"] == gram_loss and tr["exact_int"] and tr["exact_gram"]
        res[gram_loss] = (np.array(tr["loss_history"]), tr["best_iter"], conv.weight.data.cpu().clone(), out.cpu().clone(),
                          float(conv.layer_loss[0].split(":")[1]))
    hg, hc = res[True][0], res[False][0]
    assert hg.shape == (200,) and np.all(np.abs(hg - hc) <= 2e-6 * hc), np.abs(hg / hc - 1).max()
    # the winner: on the plateau several iterates lie within the conv path's own rounding of each other (here 115 and 117),
    # so the two paths may pick different ones among them - never one that the other path sees as worse by more than that
    bg, bc = res[True][1], res[False][1]
    assert abs(hc[bg] - hc[bc]) <= 4e-6 * hc[bc] and abs(hg[bg] - hg[bc]) <= 4e-6 * hg[bc], (bg, bc)
    assert abs(res[True][4] - res[False][4]) <= 1e-3 * res[False][4]
    assert _rel_mse(res[True][3], res[False][3]) <= 1e-3


@pytest.mark.parametrize("tag", ["c32", "c64"])
def test_wide_layer_first_iterations_against_the_fp64_anchor(gold, tag):
    """Which side is closer to exact arithmetic where the HIP path and the reference disagree?
    g5d holds, for the 32->32 / 64->64 layers of g5b, the oracle's fp64 evaluation of the same arithmetic (anchor) and the
    reference's first proximal solve w*_0 with 1 and 8 BLAS threads.  Fixed bars, nothing fitted:
      * w*_0: the HIP solve (exact-integer Gram, fp64 inverse of the NEXT rho, 2^-26 contraction sweeps, f32 GEMM) is no
        farther from fp64 than the reference's fp32 LU solves are (relative l2 distance, and max distance in level units);
      * level ids after iteration 0 differ from fp64's only where fp64's own pre-image lies closer to a rounding boundary
        than the HIP solve's distance from fp64 at that weight (i.e. every flip is a tie broken by the last bits), and
        there are at most 4 such weights (the reference: 0 - 1 of 110 592);
      * loss of iteration 0: 2e-5 + 1e-3 per flipped weight (one flip moves it by 2e-4 .. 1e-3: the reference's 1-thread
        run sits 7.5e-4 from fp64 at iteration 0 of the 64-channel layer with ONE flipped weight);
      * losses of iterations 1 - 4: no farther from fp64 than the farther of the two reference runs up to that iteration
        (or 1e-4)."""
    ga, g = gold("g5d_wide_fp64_anchor.npz"), gold("g5b_wide_layers.npz")
    import efficientq_amd.qconv as Q
    L_w = 4
    dl = 2.0 / (L_w - 1)
    w64 = T(ga[f"{tag}_f64_wstar0"]).double()
    aw64 = float(ga[f"{tag}_f64_aw0"])
    idx64 = T(ga[f"{tag}_f64_G0idx"])
    margin = T(ga[f"{tag}_f64_margin0"]).double()
    u64 = (torch.clamp(w64 / aw64, -1, 1) + 1) / dl

    def dist(w, aw):
        w = w.double()
        u = (torch.clamp(w / aw, -1, 1) + 1) / dl
        return ((w - w64).norm() / w64.norm()).item(), (u - u64).abs()

    ref = {}
    for nt in (1, 8):
        d2, du = dist(T(ga[f"{tag}_t{nt}_wstar0"]), float(ga[f"{tag}_t{nt}_aw0"]))
        ref[nt] = dict(d2=d2, du_max=du.max().item(), flips=int((T(ga[f"{tag}_t{nt}_G0idx"]) != idx64).sum()),
                       dloss=np.abs(ga[f"{tag}_t{nt}_loss8"][:5] - ga[f"{tag}_f64_loss_hist"][:5]) /
                       ga[f"{tag}_f64_loss_hist"][:5])
    # ---- the HIP path: the full run with the run handle and the operands of effq_admm_run captured; w*_0 (which the
    # run overwrites) is formed again from those operands by the entry points iteration 0 goes through ----
    conv, x, _ = _layer_from_gold(g, tag)
    conv.mask_pyramid[1] = conv.mask_pyramid[1].float()
    conv.lwq_trace = True
    _to_dev(conv)
    ops = Q.get_ops(torch.device(DEV))
    runs, calls, orig = [], [], ops.admm_run

    def spy(*a, **kw):
        calls.append((a, kw))
        runs.append(orig(*a, **kw))
        return runs[-1]
    ops.admm_run = spy
    try:
        conv.set_quantizing()
        with torch.no_grad():
            conv(x.to(DEV))
    finally:
        del ops.admm_run                   # the instance attribute shadowing the method
    torch.cuda.synchronize()
    run, ((A0, B0, W0, b0, geom, yn), kw) = runs[0], calls[0]
    rho, eta = kw["rho"], kw["eta"]
    Ainv = ops.spd_inverse(A0, True, 2 * rho, eta)                 # iteration 0 is solved through the NEXT rho's inverse
    wst, bst = torch.empty_like(W0), torch.empty_like(b0)
    ops.prox_solve_shifted(B0, Ainv, W0, b0, W0, torch.zeros_like(W0), rho, eta, 2 * rho, wst, bst)
    torch.cuda.synchronize()
    w_hip = wst.cpu().reshape(w64.shape)
    aw_hip = float(run.state_ring[0, 0].item())
    G0 = run.G_ring[0].cpu().reshape(w64.shape).double()
    idx_hip = torch.round((G0 / aw_hip + 1) / dl).to(torch.uint8)
    assert torch.equal(idx_hip, torch.round((torch.clamp(w_hip.double() / aw_hip, -1, 1) + 1) / dl).to(torch.uint8))
    d2, du = dist(w_hip, aw_hip)
    flips = idx_hip != idx64
    nflip = int(flips.sum())
    hist = np.array(conv.last_trace["loss_history"])
    f64h = ga[f"{tag}_f64_loss_hist"]
    dloss = np.abs(hist[:5] - f64h[:5]) / f64h[:5]
    print(f"{tag}: w*_0 rel distance to fp64: hip {d2:.3e}, reference t1 {ref[1]['d2']:.3e} t8 {ref[8]['d2']:.3e}; max "
          f"distance in level units hip {du.max().item():.3e}, t1 {ref[1]['du_max']:.3e} t8 {ref[8]['du_max']:.3e}; flipped "
          f"level ids at iteration 0: hip {nflip}, t1 {ref[1]['flips']} t8 {ref[8]['flips']}; loss[0:5] rel distance to "
          f"fp64: hip {dloss}, t1 {ref[1]['dloss']}, t8 {ref[8]['dloss']}")
    assert d2 <= max(ref[1]["d2"], ref[8]["d2"]), (d2, ref)
    assert du.max().item() <= max(ref[1]["du_max"], ref[8]["du_max"]), (du.max().item(), ref)
    assert nflip <= 4
    if nflip:
        assert bool((margin[flips] <= 1.01 * du[flips] + 1e-12).all()), (margin[flips], du[flips])
    assert dloss[0] <= 2e-5 + 1e-3 * nflip, (dloss, nflip)
    bar = np.maximum(1e-4, np.maximum.accumulate(np.maximum(ref[1]["dloss"], ref[8]["dloss"])))
    assert np.all(dloss[1:] <= bar[1:]), (dloss, bar)


@pytest.mark.parametrize("c1,c2,k,stride,pad,S", [(4, 32, 3, 2, 1, 24), (32, 3, 1, 1, 0, 12)])
def test_losses_from_the_fp64_gram_system_equal_the_conv_losses_on_full_precision_input_layers(monkeypatch, c1, c2, k,
                                                                                                stride, pad, S):
    """First conv / classifier (q_first = q_last = "256,-1": 256-level weights, full-precision input, quirk Q14): the 200
    losses from the fp64 Gram system (effq_gram_f64 + effq_gram_loss) against 200 f32 conv passes - same ADMM chain, so
    every loss agrees to the conv epilogue's fp32 rounding (2e-6) and the winners are equivalent."""
    import efficientq_amd.qconv as Q
    res = {}
    for gram_loss in (True, False):
        monkeypatch.setattr(Q, "GRAM_LOSS_DEFAULT", gram_loss)
        gen = torch.Generator().manual_seed(11)
        N = 2
        conv = Q.EfficientQConvHIP(c1, c2, k, stride, pad, 1, 1, True, q_weight=True, qlvl=256, q_act=False,
                                   qlvl_act=256, lwq_trace=True)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (c1 * k ** 3)) ** 0.5)
            conv.bias.copy_(torch.randn(c2, generator=gen) * 0.1)
        x_fp = torch.randn(N, c1, S, S, S, generator=gen)
        y = torch.nn.functional.conv3d(x_fp, conv.weight.data, conv.bias.data, stride, pad)
        x = x_fp + 0.05 * torch.randn(x_fp.shape, generator=gen)
        conv.output_fp, conv.name, conv.layer_loss = y, "l", []
        conv.mask_pyramid = [torch.randint(1, 4, tuple(y[:, 0].shape), generator=gen).float()]
        _to_dev(conv)
        conv.set_quantizing()
        with torch.no_grad():
            out = conv(x.to(DEV))
        tr = conv.last_trace
        assert tr["gram_loss"] == gram_loss
        res[gram_loss] = (np.array(tr["loss_history"]), tr["best_iter"], out.cpu().clone(),
                          float(conv.layer_loss[0].split(":")[1]))
    hg, hc = res[True][0], res[False][0]
    assert hg.shape == (200,) and np.all(np.abs(hg - hc) <= 2e-6 * hc), np.abs(hg / hc - 1).max()
    bg, bc = res[True][1], res[False][1]
    assert abs(hc[bg] - hc[bc]) <= 4e-6 * hc[bc] and abs(hg[bg] - hg[bc]) <= 4e-6 * hg[bc], (bg, bc)
    assert abs(res[True][3] - res[False][3]) <= 1e-3 * res[False][3]
    assert _rel_mse(res[True][2], res[False][2]) <= 1e-3


def test_data_parallel_two_ranks_on_one_gpu_matches_single_rank(tmp_path):
    """The sharded HIP path (volumes split over 2 ranks, Gram / statistics / losses all-reduced) against the
    unsharded one.  Both ranks share cuda:0; the collective is gloo with host staging (RCCL refuses two ranks
    on one device) -- the reduction points exercised are exactly the ones RCCL serves on a multi-GPU node."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "dp")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29622",
                        os.path.join(root, "tests", "dp_worker_gpu.py"), out],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    single = torch.load(out + "_single.pt")
    dp0, dp1 = torch.load(out + "_rank0.pt"), torch.load(out + "_rank1.pt")
    for k in dp0["sd"]:
        assert torch.equal(dp0["sd"][k], dp1["sd"][k]), k          # replicas in lock step, bit for bit
    assert dp0["nums"] == single["nums"]
    a, b = np.array(dp0["loss"]), np.array(single["loss"])
    assert abs(a[0] - b[0]) <= 1e-6 * b[0], (a, b)      # first layer: identical inputs, sums re-associated only
    assert np.all(np.abs(a - b) <= 1e-1 * b), (a, b)    # later layers: plateau drift
    assert abs(a.sum() - b.sum()) <= 3e-2 * b.sum()


def test_collectives_on_rccl_with_one_rank_are_the_identity(tmp_path):
    """The data-parallel code path on the REAL backend ("nccl" = RCCL), one rank: every collective the sharded
    calibration issues runs on RCCL on the stream the product uses it on; the result is bit-identical to the run
    without collectives, and a layer costs a bounded number of them (statistics 3, activation fixed point ~45 at 4
    levels, Gram 1, loss history 1, final loss 1)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "nccl1.pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dp_worker_nccl.py"), out], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = torch.load(out)
    for run in ("forced", "via_torch"):                  # RCCL through its C ABI on the kernels' stream / through torch
        for k in d["plain"]["sd"]:
            assert torch.equal(d[run]["sd"][k], d["plain"]["sd"][k]), (run, k)
        assert d[run]["loss"] == d["plain"]["loss"] and d[run]["nums"] == d["plain"]["nums"], run
    layers = len(d["plain"]["loss"])
    # <= ~60 collectives per layer at 4 levels (the 200 per-iteration losses travel as ONE message, the three statistics
    # triples of a layer as one), + the class census
    assert 0 < d["forced"]["collectives"] <= 66 * layers + 8, (d["forced"]["collectives"], layers)
    assert d["forced"]["collectives"] == d["via_torch"]["collectives"]
    assert d["forced"]["direct_calls"] >= d["forced"]["collectives"] - 8         # all but the host-side census went direct


def test_tune_activation_range_matches_reference(gold):
    """Row f3: the reference's tune_activation_range (ptqer.py:238-272) run in isolation on the tiny lits net (G12: init
    pass + 50 Adam steps at lr 5e-4) against the product (conv / input-gradient / quantiser-backward / Adam on the HIP
    library).  alpha_act after the init pass: 1e-6; loss history and final alpha_act: 2e-3 (a step moves an alpha by
    <= 5e-4 whatever the size of its gradient, so fp32 summation-order differences in a gradient near zero can cost a
    few steps' worth of drift)."""
    from efficientq_amd import calibrate as K
    from efficientq_amd.qconv import PTQConv
    from efficientq_amd.tune import tune_activation_range
    g = gold("g12_tune_act.npz")
    args, model, _ = _tiny("lits")
    model.load_state_dict({k[4:]: T(g[k]) for k in g.files if k.startswith("sd0/")}, strict=False)
    model.eval()
    K.search_fold_and_remove_bn(model)
    model.to(DEV)
    S = int(g["meta"][0])
    vols = torch.randn(2, 1, S, S, S, generator=torch.Generator().manual_seed(int(g["vols_seed"]))).to(DEV)
    K.set_fp(model)
    with torch.no_grad():
        output_fp = model(vols).detach()
    mods = [(n, m) for n, m in model.named_modules() if isinstance(m, PTQConv)]
    assert [n for n, _ in mods] == g["layer_names"].tolist()
    K.set_init_alpha(model)
    with torch.no_grad():
        model(vols)
    init = np.array([m.alpha_act.item() for _, m in mods])
    assert np.all(np.abs(init - g["alpha_init"]) <= 1e-6 * g["alpha_init"]), (init, g["alpha_init"])
    losses = tune_activation_range(model, output_fp, vols, max_iter=50, need_init=True)
    want = g["loss_all"]
    assert len(losses) == 50
    assert np.all(np.abs(np.array(losses) - want) <= 2e-3 * want), (losses[:5], want[:5], losses[-3:], want[-3:])
    final = np.array([m.alpha_act.item() for _, m in mods])
    assert np.all(np.abs(final - g["alpha_final"]) <= 2e-3 * g["alpha_final"]), (final, g["alpha_final"])
    assert losses[-1] < losses[0]


def test_cli_ptq_mission_writes_reference_artifacts(tmp_path):
    """`entrance ptq` with the reference's flag names on a tiny synthetic problem: layer_loss.txt,
    time_cost.txt, class_voxel_nums.txt and the three snapshots (uint8 weights in the int8 ones)."""
    from efficientq_amd import entrance
    snap = str(tmp_path / "snap")
    entrance.main(["ptq", "--task", "lits", "--qconv", "effq", "--qlvl_w", "4", "--qlvl_a", "4", "--q_first", "256,-1",
                   "--q_last", "256,-1", "--width", "8,16,8", "--depth", "1,1,1", "--init_stride", "1", "--nMod", "1",
                   "--nClass", "3", "--blk", "mid", "--ds", "simple", "--hetero_dim", "--drop_rate", "0.5",
                   "--lwq_batchsz", "2", "--lwq_patchsz", "32,32,32", "--synthetic", "--no_test", "--snap_dir", snap])
    import os
    for f in ("layer_loss.txt", "time_cost.txt", "class_voxel_nums.txt", "state_in_fp.pkl", "state_in_int8.pkl",
              "state_in_int8_compress.npz"):
        assert os.path.exists(os.path.join(snap, f)), f
    lines = open(os.path.join(snap, "layer_loss.txt")).read().strip().split("\n")
    assert len(lines) == 10 and lines[0].startswith(f"{'conv0.conv':45s}:")
    sd = torch.load(os.path.join(snap, "state_in_int8.pkl"))["state_dict"]
    w = sd["u_blocks.UResBlock1.Layer1.block1.conv.weight"]
    assert w.dtype == torch.uint8 and int(w.max()) <= 3
    fp = torch.load(os.path.join(snap, "state_in_fp.pkl"))["state_dict"]["u_blocks.UResBlock1.Layer1.block1.conv.weight"]
    assert len(torch.unique(fp)) <= 4


def test_mixed_precision_search_respects_budget_and_improves_with_bits():
    """Row f4 harness on the tiny net (cheap layer_loss sensitivities): the greedy assignment stays within the bit
    budget, uses only the candidate levels, and more bits make the calibrated network better end to end (relative MSE of
    the quantised output against the FP output)."""
    from efficientq_amd import calibrate as K, config as Cf, mixed, synth
    args = Cf.make_args(dict(Cf.TINY_NET, width="8,16,8"), 4, 4)
    QConv, _, kwQ = Cf.get_conv_class(args)

    def build():
        m = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
        synth.randomise_network(m, 3)
        m.eval(); K.search_fold_and_remove_bn(m); m.to(DEV); K.set_name(m)
        return m
    vols = torch.randn(2, 1, 16, 16, 16, generator=torch.Generator().manual_seed(5)).to(DEV)
    res = mixed.search(build, vols, "lits", args.init_stride, [2.0, 3.0, 4.0], levels=(4, 8, 16),
                       sensitivity="layer_loss")
    assert [r["budget_bits"] for r in res] == [2.0, 3.0, 4.0]
    for r in res:
        assert r["avg_bits"] <= r["budget_bits"] + 1e-9
        assert set(r["levels"].values()) <= {4, 8, 16} and len(r["levels"]) == 8
    assert set(res[0]["levels"].values()) == {4} and set(res[2]["levels"].values()) == {16}
    assert res[2]["output_error"] < res[1]["output_error"] < res[0]["output_error"]


def test_mixed_precision_search_on_the_brats_net():
    """BASELINE configs[4] on its real net (BraTS 3D-UNet, 20 searched layers, 2 volumes of 4 x 64^3; one GPU of the 8 the
    config names: budgets are independent replicas).  End-to-end sensitivities (41 calibrations), then for each budget:
    the average stays within it, only candidate levels are used, 2 / 4 bits reproduce the uniform maps bit for bit (same
    map => same calibration: the path is deterministic), and the CHOSEN map at 2.5 and 3 bits is no worse end to end than
    the uniform map of the next-lower uniform budget (2 bits).  On this small sample the 4-level ACTIVATIONS set the floor:
    uniform 2-bit and uniform 4-bit weights are only 5 - 13 % apart, and the end-to-end error of one and the same map
    moves by up to 8 % when the summation order of any kernel changes (ADMM trajectories are chaotic; observed between two
    builds of round 3: uniform 2-bit 0.00595 / 0.00547, 3-bit map 0.00506 / 0.00558) - so the ordering assertions leave
    15 %; the sweep on 4 volumes of 128^3 with activation levels following (profiles/r03_mixed_precision_brats.jsonl)
    shows the effect of the search itself."""
    from efficientq_amd import calibrate as K, config as Cf, mixed, synth
    args = Cf.make_args(Cf.BRATS_NET, 4, 4)
    QConv, _, kwQ = Cf.get_conv_class(args)

    def build():
        m = Cf.get_model_cube(args, QConv, kwQ)[0]["model"]
        synth.randomise_network(m, 0)
        m.eval(); K.search_fold_and_remove_bn(m); m.to(DEV); K.set_name(m)
        return m
    vols = synth.calib_batch("brats", range(2), 64).to(DEV)
    budgets = [2.0, 2.5, 3.0, 4.0]
    res = mixed.search(build, vols, "brats", args.init_stride, budgets, levels=(4, 8, 16))
    u4 = mixed.uniform(build, vols, "brats", args.init_stride, 4)
    u16 = mixed.uniform(build, vols, "brats", args.init_stride, 16)
    print([(r["budget_bits"], round(r["avg_bits"], 3), r["output_error"], r["agreement"]) for r in res], u4, u16)
    for r, b in zip(res, budgets):
        assert r["budget_bits"] == b and r["avg_bits"] <= b + 1e-9
        assert set(r["levels"].values()) <= {4, 8, 16} and len(r["levels"]) == 20
    assert set(res[0]["levels"].values()) == {4} and set(res[3]["levels"].values()) == {16}
    # same maps => same calibration (deterministic): the uniform runs are reproduced
    assert abs(res[0]["output_error"] - u4["output_error"]) <= 1e-6 * u4["output_error"]
    assert abs(res[3]["output_error"] - u16["output_error"]) <= 1e-6 * u16["output_error"]
    assert res[1]["avg_bits"] > 2.2 and res[2]["avg_bits"] > 2.7          # the budget is spent
    assert res[1]["output_error"] <= 1.15 * u4["output_error"] and res[2]["output_error"] <= 1.15 * u4["output_error"]
    assert res[2]["output_error"] <= 1.15 * res[1]["output_error"]
    assert res[3]["output_error"] <= 1.15 * res[2]["output_error"]
    assert res[3]["agreement"] >= res[0]["agreement"] - 1e-3


def test_packed_weight_export_round_trip():
    """Row f2: a calibrated 4-level layer exported at 2 bits per weight and re-imported gives the same weights
    as the reference's uint8 round trip (store_int_weight / restore_fp_weight, PTQConv.py:125-152)."""
    from efficientq_amd.qconv import EfficientQConvHIP
    gen = torch.Generator().manual_seed(5)
    c, S, N = 16, 8, 2
    conv = EfficientQConvHIP(c, c, 3, 1, 1, 1, 1, True, q_weight=True, qlvl=4, q_act=True, qlvl_act=4)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * 0.1)
        conv.bias.copy_(torch.randn(c, generator=gen) * 0.1)
    x = torch.relu(torch.randn(N, c, S, S, S, generator=gen))
    conv.output_fp = torch.nn.functional.conv3d(x, conv.weight.data, conv.bias.data, 1, 1)
    conv.name, conv.layer_loss = "l", []
    _to_dev(conv)
    conv.set_quantizing()
    with torch.no_grad():
        conv(x.to(DEV))
    blob = conv.export_packed_weight()
    assert blob["bits"] == 2 and blob["data"].numel() == (conv.weight.numel() * 2 + 7) // 8
    w_before = conv.weight.data.clone()
    conv.store_int_weight()
    ids = conv.weight.data.clone()
    conv.weight.data = conv.weight.data.to(DEV)
    conv.restore_fp_weight()
    w_ref = conv.weight.data.clone()
    conv.import_packed_weight(blob)
    assert torch.equal(conv.weight.data, w_ref)
    assert int(ids.max()) <= 3
    # (weights are the best iterate's, alpha_w the last iterate's - quirk Q6 - so the round trip is not w_before)
    assert (w_ref - w_before).abs().max() <= 0.5 * w_before.abs().max()


def test_quantised_sliding_window_inference_and_dice_proxy(gold, monkeypatch):
    """Row f1 on the width 32,64,32 LiTS net of g6d: calibrate on the GPU, then
      * the one-patch sliding window is the plain quantised forward, bit for bit;
      * OVERLAPPED half-size patches run through the HIP quantised forward (every conv = conv3d_quant_calib_step with the
        activation quantiser fused) and stitched by the product (utils/transforms.py:784-852) against the ORACLE: the same
        calibrated weights on the CPU, every patch from oracle.split_patches through oracle conv arithmetic
        (tests/cpu_backend.py: F.conv3d on oracle.discretize'd activations), stitched by oracle.stitch_patches.  The two
        convs sum in different orders, so an activation within an ulp of a rounding boundary may take the other level on
        one side (a local difference of one level step): bars = relative MSE 1e-6 and 99.9 % of the voxels within 1e-5 of
        the largest logit;
      * the FP-vs-quantised agreement (the Dice proxy) of the stitched output within north_star's 0.1 pt of the nearer of
        the reference's two runs of the same calibration (g6d)."""
    import copy
    from efficientq_amd import calibrate as K, evaluate as E, synth
    import efficientq_amd.qconv as Q
    from tests import cpu_backend
    g = gold("g6d_wide_lits_L4.npz")
    args, model, _ = _tiny("lits", width="32,64,32")
    synth.randomise_network(model, int(g["net_seed"]))
    model.eval()
    K.search_fold_and_remove_bn(model)
    model.to(DEV)
    S = int(g["meta"][1])
    vols_cpu = torch.randn(2, 1, S, S, S, generator=torch.Generator().manual_seed(int(g["vols_seed"])))
    assert torch.equal(vols_cpu[:, :, ::8, ::8, ::8], T(g["vols_check"]))
    vols = vols_cpu.to(DEV)
    K.set_name(model)
    fp_model = copy.deepcopy(model)                        # calibration overwrites the weights in place
    res = K.calibrate_model(model, vols, "lits", args.init_stride)
    K.set_quantized(model)
    psz, ov = S // 2, S // 8
    with torch.no_grad():
        whole = torch.stack(list(model(vols)))
        one = E.sliding_window_forward(model, vols, S, 0)
        assert torch.equal(one, whole)
        # the calibration hands each layer the output of its i8 forward (an exact integer contraction), the calibrated
        # model runs the f32 conv: the same numbers to fp32 rounding, and a level of a later activation may flip on it
        dq = (whole[-1] - res["output_q"][-1]).abs()
        topq = res["output_q"][-1].abs().max().item()
        print(f"quantised forward vs the calibration's own output: max {dq.max().item() / topq:.2e}, mean {dq.mean().item() / topq:.2e} of the largest logit")
        assert dq.mean().item() <= 1e-6 * topq and (dq > 1e-4 * topq).float().mean().item() <= 1e-4
        sw = E.sliding_window_forward(model, vols, psz, ov)
    assert sw.shape == whole.shape and torch.isfinite(sw).all()
    dice_q, out_q, out_fp = E.fp_vs_quantised_dice(model, vols, "lits", fp_model=fp_model, patch_size=S, overlap=0)
    assert torch.allclose(out_fp, res["output_fp"][-1], atol=2e-5 * out_fp.abs().max().item())
    dice_q2, _, _ = E.fp_vs_quantised_dice(model, vols, "lits", fp_logits=res["output_fp"][-1], patch_size=S, overlap=0)
    assert all(abs(float(a) - float(b)) <= 1e-3 for a, b in zip(dice_q, dice_q2))
    assert len(dice_q) == out_q.shape[1] and all(0.0 <= float(d) <= 1.0 for d in dice_q)
    agree = ((out_q > 0) == (out_fp > 0)).float().mean().item()
    refs = [float(g["copy_t8/agree"]), float(g["copy_t1/agree"])]
    assert min(abs(agree - r) for r in refs) <= 1e-3, (agree, refs)          # 0.1 pt
    # ---- the oracle side: same weights, CPU, oracle arithmetic per patch, oracle stitching
    sw_hip = sw.cpu()
    cpu_model = copy.deepcopy(model).cpu()
    cpu_backend.install(monkeypatch)
    K.set_quantized(cpu_model)
    with torch.no_grad():
        preds = [torch.stack(list(cpu_model(pt.contiguous()))) for pt in O.split_patches(vols_cpu, psz, ov)]
    want = O.stitch_patches(vols_cpu, preds, psz, ov)
    assert want.shape == sw_hip.shape
    d = (sw_hip - want).abs()
    top = want.abs().max().item()
    rel = _rel_mse(sw_hip, want)
    q999 = torch.quantile(d.flatten()[:: max(1, d.numel() // 4_000_000)], 0.999).item()
    print(f"stitched HIP vs oracle: rel-MSE {rel:.2e}, 99.9 % of |diff| <= {q999 / top:.2e} of the largest logit, max "
          f"{d.max().item() / top:.2e}; agreement {agree:.5f} (reference {refs})")
    assert rel <= 1e-6 and q999 <= 1e-5 * top, (rel, q999 / top)
