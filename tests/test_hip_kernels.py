"""Kernel-level parity: every C-ABI entry point against the CPU oracle / reference goldens.
Runs on a real MI355X only (-m gpu).  Bit-exact for quantiser indices; stated tolerances elsewhere."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import effq_oracle as O

pytestmark = pytest.mark.gpu

T = torch.from_numpy


@pytest.fixture(scope="module")
def ops():
    from efficientq_amd.hip_ops import get_ops
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return get_ops("cuda:0")


def dev(t):
    return t.to("cuda:0")


# ------------------------------------------------------------------ a1 / a3
def test_quant_dequant_f32_matches_reference_goldens(ops, gold):
    g = gold("g1_discretize.npz")
    alpha = torch.tensor(0.7341)
    for L in (4, 16, 256):
        for tag, lo, hi in (("w", -1.0, 1.0), ("a", 0.0, 1.0)):
            v = T(g[f"L{L}_{tag}_in"])
            want = T(g[f"L{L}_{tag}_qdq32"])
            got, idx = ops.quant_dequant_f32(dev(v), dev(alpha), L, lo, hi, want_idx=True)
            assert torch.equal(got.cpu(), want), (L, tag)
            want_idx = O.quant_index(v / alpha, L, lo, hi)
            assert torch.equal(idx.cpu().long(), want_idx)
            # alpha = 1 reproduces plain discretize (layer_helper.py:25-37)
            got1 = ops.quant_dequant_f32(dev(v), dev(torch.tensor(1.0)), L, lo, hi)
            assert torch.equal(got1.cpu(), T(g[f"L{L}_{tag}_q32"]))


def test_quant_dequant_ragged_and_empty(ops):
    gen = torch.Generator().manual_seed(3)
    for n in (1, 3, 5, 1023, 4097):
        v = torch.randn(n, generator=gen)
        a = torch.tensor(0.37)
        got = ops.quant_dequant_f32(dev(v), dev(a), 16, -1.0, 1.0)
        assert torch.equal(got.cpu(), O.discretize(v / a, 16, -1, 1) * a)
    e = ops.quant_dequant_f32(dev(torch.empty(0)), dev(torch.tensor(1.0)), 4, 0.0, 1.0)
    assert e.numel() == 0


# ------------------------------------------------------------------ a2
@pytest.mark.parametrize("L", [4, 16, 256])
def test_fit_scale_matches_reference_goldens(ops, gold, L):
    g = gold("g2_project.npz")
    for name, lo, hi, to_idx in (("act", 0.0, 1.0, lambda b: torch.round(b * (L - 1))),
                                 ("wgt", -1.0, 1.0, lambda b: torch.round((b + 1) * (L - 1) / 2))):
        v = T(g[name])
        alpha, iters, st = ops.fit_scale(dev(v), L, lo, hi)
        want_a = float(g[f"{name}_L{L}_alpha"])
        assert abs(alpha - want_a) <= 1e-11 * abs(want_a), (alpha, want_a)
        assert iters == int(g[f"{name}_L{L}_iters"])
        y, b, idx = ops.quant_dequant_f64path(dev(v), st, L, lo, hi, want_b=True, want_idx=True)
        assert torch.equal(idx.cpu(), T(g[f"{name}_L{L}_idx"]))          # quantiser indices bit-exact
        assert torch.equal(to_idx(b.cpu()).to(torch.uint8), T(g[f"{name}_L{L}_idx"]))
        # a*b (EfficientQConv.py:70): python double * fp32 tensor
        fit = O.fit_scale(v, L, lo, hi)
        assert torch.equal(b.cpu(), fit.b)
        assert torch.allclose(y.cpu(), fit.alpha * fit.b, rtol=1e-6, atol=0)


def test_fit_scale_with_reducer_equals_unsharded(ops):
    gen = torch.Generator().manual_seed(11)
    v = torch.relu(torch.randn(2, 8, 12, 12, 12, generator=gen))
    a0, it0, _ = ops.fit_scale(dev(v), 4, 0.0, 1.0)
    a1, it1, _ = ops.fit_scale(dev(v), 4, 0.0, 1.0, reducer=lambda t: t)   # identity all-reduce
    assert a0 == a1 and it0 == it1


def test_fit_scale_raises_at_cap(ops):
    v = torch.randn(4096, generator=torch.Generator().manual_seed(1))
    import efficientq_amd.hip_ops as H
    old = H.ADMM_TOL
    H.ADMM_TOL = -1.0
    try:
        with pytest.raises(RuntimeWarning):
            ops.fit_scale(dev(v), 4, -1.0, 1.0)
    finally:
        H.ADMM_TOL = old


def test_moments(ops):
    gen = torch.Generator().manual_seed(5)
    for n in (7, 4096, 1 << 20):
        v = torch.randn(n, generator=gen) * 3 + 0.5
        m = ops.moments(dev(v)).cpu()
        vd = v.double()
        assert m[2].item() == n
        assert abs(m[0].item() - vd.sum().item()) <= 1e-9 * vd.abs().sum().item()
        assert abs(m[1].item() - (vd * vd).sum().item()) <= 1e-12 * (vd * vd).sum().item()


# ------------------------------------------------------------------ a5 / a6 / a7
GRAM_CASES = [("k3s1p1", 3, 1, 1), ("k3s221p1", 3, (2, 2, 1), 1), ("k1s1p0", 1, 1, 0),
              ("k3s1p1_nobias_noatt", 3, 1, 1)]


def _ndhwc(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


@pytest.mark.parametrize("tag,k,s,p", GRAM_CASES)
def test_gram_and_solve_match_reference_goldens(ops, gold, tag, k, s, p):
    from efficientq_amd.hip_ops import make_geom
    g = gold("g3g4_gram_solve.npz")
    x, y, w = T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), T(g[f"{tag}_w"])
    b = T(g[f"{tag}_b"]) if f"{tag}_b" in g else None
    att = T(g[f"{tag}_att"]) if f"{tag}_att" in g else None
    geom = make_geom(x.shape, y.shape[1], k, s, p)
    A0, B0 = ops.gram(dev(_ndhwc(x)), dev(att) if att is not None else None, dev(_ndhwc(y)), geom, b is not None)
    wantA, wantB = T(g[f"{tag}_A0"]), T(g[f"{tag}_B0"])
    # fp32 Gram: summation order differs from the reference's BLAS; tolerance 2e-6 of the matrix scale
    assert (A0.cpu() - wantA).abs().max() <= 2e-6 * wantA.abs().max()
    assert (B0.cpu() - wantB).abs().max() <= 2e-6 * wantB.abs().max()
    assert torch.equal(A0, A0.T)                                   # exactly symmetric
    # prox solve from the reference's own A0/B0 (isolates the solver): |dW| <= 1e-5 * scale
    rho, eta = 7.5, 1.3
    Ainv = ops.spd_inverse(dev(wantA), b is not None, rho, eta)
    G = T(g[f"{tag}_G"])
    wstar = torch.empty_like(dev(w))
    bstar = torch.empty(w.shape[0], device="cuda:0") if b is not None else None
    ops.prox_solve(dev(wantB), Ainv, dev(w), dev(b) if b is not None else None, dev(G), dev(torch.zeros_like(G)),
                   rho, eta, wstar, bstar)
    want_w = T(g[f"{tag}_wstar"])
    assert (wstar.cpu() - want_w).abs().max() <= 1e-5 * want_w.abs().max()
    if b is not None:
        want_b = T(g[f"{tag}_bstar"])
        assert (bstar.cpu() - want_b).abs().max() <= 1e-5 * max(want_b.abs().max(), want_w.abs().max())


@pytest.mark.parametrize("c1,c2,k,s,p,S", [(8, 8, 3, 1, 1, 10), (32, 32, 3, 1, 1, 8), (4, 32, 3, 2, 1, 12),
                                           (16, 32, 1, 1, 0, 8), (40, 24, 3, 1, 1, 6)])
def test_gram_vs_oracle(ops, c1, c2, k, s, p, S):
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(c1 * 100 + c2)
    N = 2
    x = torch.relu(torch.randn(N, c1, S, S + 1, S + 2, generator=gen))
    w = torch.randn(c2, c1, k, k, k, generator=gen) * 0.1
    b = torch.randn(c2, generator=gen)
    y = F.conv3d(x, w, b, s, p)
    att = torch.randint(1, 4, y[:, 0].shape, generator=gen).float()
    ps = O.ProxSystem(x, y, (k, k, k), s, p, w, b, att)
    geom = make_geom(x.shape, c2, k, s, p)
    A0, B0 = ops.gram(dev(_ndhwc(x)), dev(att), dev(_ndhwc(y)), geom, True)
    assert (A0.cpu() - ps.A0).abs().max() <= 3e-6 * ps.A0.abs().max()
    assert (B0.cpu() - ps.B0).abs().max() <= 3e-6 * ps.B0.abs().max()
    # sharded accumulation over volumes == unsharded (multi-GPU partial sums)
    g1 = make_geom(x[:1].shape, c2, k, s, p)
    A1, B1 = ops.gram(dev(_ndhwc(x[:1])), dev(att[:1]), dev(_ndhwc(y[:1])), g1, True)
    A1, B1 = ops.gram(dev(_ndhwc(x[1:])), dev(att[1:]), dev(_ndhwc(y[1:])), g1, True, A1, B1)
    assert (A1 - A0).abs().max() <= 3e-6 * ps.A0.abs().max()
    assert (B1 - B0).abs().max() <= 3e-6 * ps.B0.abs().max()


@pytest.mark.parametrize("c1,c2,k,s,p,S,La,att_kind,bias", [
    (32, 32, 3, 1, 1, 8, 4, "int", True),        # the BraTS 32-channel layer shape in small
    (16, 24, 3, 1, 1, 7, 16, "float", True),     # c2 padded to 32, non-integer class weights, odd sizes
    (32, 16, 3, 2, 1, 9, 4, "none", True),       # stride 2, no attention
    (48, 3, 1, 1, 0, 6, 128, "int", False),      # 1x1, no bias, the widest level range
    (64, 32, 3, 1, 1, 6, 4, "many", True),       # 16 distinct weights
])
def test_gram_i8_exact_vs_f32_and_oracle(ops, c1, c2, k, s, p, S, La, att_kind, bias):
    """effq_gram_accum_i8 (integer Gram on the i8 matrix cores, weights applied per sorted class) against the
    fp64 value of the same sums, the fp32 kernel and the oracle's getA0B0 on xhat = alpha*k/(La-1)."""
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(c1 * 131 + c2 + La)
    N = 2
    shape = (N, c1, S, S + 1, S + 2)
    idx = torch.randint(0, La, shape, generator=gen).to(torch.uint8)
    alpha = torch.tensor(0.7312, dtype=torch.float32)
    xhat = (alpha.double() * idx.double() / (La - 1)).float()
    w = torch.randn(c2, c1, k, k, k, generator=gen) * 0.1
    b = torch.randn(c2, generator=gen) if bias else None
    y = F.conv3d(torch.relu(torch.randn(shape, generator=gen)), w, b, s, p) * 3.7
    vshape = y[:, 0].shape
    if att_kind == "int":
        att = torch.randint(1, 5, vshape, generator=gen).float()
    elif att_kind == "float":
        att = torch.tensor([1.0, 2.5, 17.25])[torch.randint(0, 3, vshape, generator=gen)]
    elif att_kind == "many":
        att = torch.randint(1, 17, vshape, generator=gen).float()
    else:
        att = None
    geom = make_geom(shape, c2, k, s, p)
    assert ops.gram_i8_supported(geom, La)
    cls = ops.att_classes(dev(att) if att is not None else None)
    assert cls is not None
    A0, B0 = ops.gram_i8(dev(_ndhwc(idx)), cls, dev(_ndhwc(y)), geom, bias, dev(alpha), La)
    # fp64 reference of the same sums
    X = torch.from_numpy(O.patch_matrix(idx.float().numpy(), (k, k, k), s, p).copy()).double()
    X = X * (alpha.double() / (La - 1))
    if bias:
        X = torch.cat([X, torch.ones(1, X.shape[1], dtype=torch.float64)])
    a = att.reshape(1, -1).double() if att is not None else torch.ones(1, X.shape[1], dtype=torch.float64)
    ymat = y.permute(1, 0, 2, 3, 4).reshape(c2, -1).double()
    wantA, wantB = 2 * (X * a) @ X.T, 2 * (ymat * a) @ X.T
    assert (A0.cpu().double() - wantA).abs().max() <= 2e-7 * wantA.abs().max()      # fp32 output rounding only
    assert (B0.cpu().double() - wantB).abs().max() <= 2e-7 * wantB.abs().max()
    assert torch.equal(A0, A0.T)
    # the fp32 kernel on xhat agrees to its own rounding
    Af, Bf = ops.gram(dev(_ndhwc(xhat)), dev(att) if att is not None else None, dev(_ndhwc(y)), geom, bias)
    assert (A0 - Af).abs().max() <= 3e-6 * wantA.abs().max()
    assert (B0 - Bf).abs().max() <= 3e-6 * wantB.abs().max()
    # sharded accumulation == unsharded
    g1 = make_geom((1,) + shape[1:], c2, k, s, p)
    c0 = ops.att_classes(dev(att[:1].contiguous())) if att is not None else cls
    c1_ = ops.att_classes(dev(att[1:].contiguous())) if att is not None else cls
    A1, B1 = ops.gram_i8(dev(_ndhwc(idx[:1])), c0, dev(_ndhwc(y[:1])), g1, bias, dev(alpha), La)
    A1, B1 = ops.gram_i8(dev(_ndhwc(idx[1:])), c1_, dev(_ndhwc(y[1:])), g1, bias, dev(alpha), La, A1, B1)
    assert (A1 - A0).abs().max() <= 3e-7 * wantA.abs().max()
    assert (B1 - B0).abs().max() <= 3e-7 * wantB.abs().max()


def test_gram_i8_rejects_what_it_cannot_do(ops):
    from efficientq_amd.hip_ops import make_geom
    assert not ops.gram_i8_supported(make_geom((1, 4, 8, 8, 8), 32, 3, 2, 1), 4)       # C1 % 16
    assert not ops.gram_i8_supported(make_geom((1, 32, 8, 8, 8), 32, 3, 1, 1), 256)    # levels beyond int8
    assert not ops.gram_i8_supported(make_geom((1, 16, 8, 8, 8), 8, 5, 1, 2), 4)       # 125 taps
    att = torch.arange(17 * 8, dtype=torch.float32, device="cuda:0").reshape(1, 17, 8, 1)
    assert ops.att_classes(att) is None                                              # too many distinct weights


@pytest.mark.parametrize("n,c2", [(28, 8), (217, 16), (865, 32), (130, 64)])
def test_spd_inverse_and_prox(ops, n, c2):
    gen = torch.Generator().manual_seed(n)
    X = torch.randn(n, 3 * n, generator=gen)
    A0 = (2 * X @ X.T).float()
    rho, eta = 30.0, 3.0
    A = A0.double().clone()
    d = torch.full((n,), rho + eta, dtype=torch.float64)
    d[-1] = eta
    A += torch.diag(d)
    Ainv_pad = ops.spd_inverse(dev(A0), True, rho, eta)
    Ainv = Ainv_pad.cpu()[:, :n]
    assert Ainv_pad.shape[1] % 32 == 0 and Ainv_pad[:, n:].abs().sum().item() == 0
    want = torch.linalg.inv(A)
    assert (Ainv.double() - want).abs().max() <= 2e-7 * want.abs().max() + 1e-12
    B0 = torch.randn(c2, n, generator=gen) * 10
    W0 = torch.randn(c2, n - 1, generator=gen)
    b0 = torch.randn(c2, generator=gen)
    G = torch.randn(c2, n - 1, generator=gen)
    dual = torch.randn(c2, n - 1, generator=gen) * 0.1
    wstar = torch.empty(c2, n - 1, device="cuda:0")
    bstar = torch.empty(c2, device="cuda:0")
    ops.prox_solve(dev(B0), Ainv_pad, dev(W0), dev(b0), dev(G), dev(dual), rho, eta, wstar, bstar)
    Bm = B0.double() + eta * torch.cat([W0, b0[:, None]], 1).double()
    Bm[:, :-1] += rho * (G - dual).double()
    wantW = torch.linalg.solve(A, Bm.T).T
    got = torch.cat([wstar.cpu(), bstar.cpu()[:, None]], 1).double()
    assert (got - wantW).abs().max() <= 2e-5 * wantW.abs().max()
    # the solve for rho/2 through the SAME inverse (effq_prox_solve_shifted: iteration 0 of the ADMM schedule,
    # whose rho doubles right after it) equals the direct solve of that system
    rho0 = rho / 2
    A_half = A - torch.diag(torch.cat([torch.full((n - 1,), rho - rho0, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)]))
    Bm0 = B0.double() + eta * torch.cat([W0, b0[:, None]], 1).double()
    Bm0[:, :-1] += rho0 * (G - dual).double()
    want0 = torch.linalg.solve(A_half, Bm0.T).T
    ops.prox_solve_shifted(dev(B0), Ainv_pad, dev(W0), dev(b0), dev(G), dev(dual), rho0, eta, rho, wstar, bstar)
    got0 = torch.cat([wstar.cpu(), bstar.cpu()[:, None]], 1).double()
    assert (got0 - want0).abs().max() <= 3e-5 * want0.abs().max()


@pytest.mark.parametrize("n,c2", [(1025, 256), (1301, 200), (2081, 512), (3457, 128), (865, 32)])
def test_prox_product_is_fp32_grade_on_every_kernel_variant(ops, n, c2):
    """What = Bm * Ainv of the prox solve against the fp64 product of the SAME fp32 operands.  For c2 > 128 and
    n >= 1024 the product runs on the bf16 matrix cores with both operands split in three (k_prox_gemm_b3: six exact
    bf16 products per fp32 product, terms below 2^-24 dropped); the smaller shapes on the f32 matrix cores.  Both must
    be fp32-grade: error <= 4e-7 of |Bm| |Ainv| accumulated over K (an fp32 dot product of length n in random order sits
    at ~1e-7 sqrt(n) / n of that bound) - and row / column remainders, the bias column and the K split must be right."""
    gen = torch.Generator().manual_seed(n + c2)
    lda = ops.lib.effq_ainv_ld(n)
    S = torch.randn(n, n, generator=gen) * 1e-3
    S = 0.5 * (S + S.T)                                            # the kernels use the symmetry of A^-1
    Ainv = torch.zeros(n, lda)
    Ainv[:, :n] = S
    B0 = torch.randn(c2, n, generator=gen) * 10
    W0 = torch.randn(c2, n - 1, generator=gen)
    b0 = torch.randn(c2, generator=gen)
    G = torch.randn(c2, n - 1, generator=gen)
    dual = torch.randn(c2, n - 1, generator=gen) * 0.1
    rho, eta = 30.0, 3.0
    wstar = torch.empty(c2, n - 1, device="cuda:0")
    bstar = torch.empty(c2, device="cuda:0")
    ops.prox_solve(dev(B0), dev(Ainv), dev(W0), dev(b0), dev(G), dev(dual), rho, eta, wstar, bstar)
    # Bm exactly as the build kernel forms it in fp32 (solver.py:316-322 op order)
    Bm = torch.cat([(B0[:, :-1] + np.float32(eta) * W0) + np.float32(rho) * (G - dual),
                    (B0[:, -1] + np.float32(eta) * b0)[:, None]], 1)
    want = Bm.double() @ S.double()
    bound = Bm.double().abs() @ S.double().abs()
    got = torch.cat([wstar.cpu(), bstar.cpu()[:, None]], 1).double()
    assert ((got - want).abs() <= 4e-7 * bound).all(), ((got - want).abs() / bound).max()
    assert (got - want).abs().max() <= 2e-6 * want.abs().max()


# ------------------------------------------------------------------ the conv entry point
CONV_CASES = [
    # c1, c2, k, stride, pad, spatial
    (32, 32, 3, 1, 1, (8, 8, 16)),
    (8, 8, 3, 1, 1, (9, 7, 11)),        # ragged tiles
    (4, 32, 3, 2, 1, (16, 16, 16)),     # first layer, stride 2
    (1, 8, 3, 1, 1, (8, 8, 8)),         # single modality
    (2, 8, 3, (2, 2, 1), 1, (10, 12, 9)),
    (64, 128, 1, 1, 0, (4, 4, 8)),      # 1x1x1
    (32, 3, 1, 1, 0, (8, 8, 8)),        # classifier
    (64, 64, 3, 1, 1, (4, 8, 8)),       # two channel slabs
    (48, 40, 3, 1, 1, (4, 4, 8)),       # odd widths
    (4, 32, 3, 2, 1, (24, 24, 48)),     # first layer: 3 x 3 x 3 tiles of the LDS-staged kernel, the centre one interior
    (4, 32, 3, 2, 1, (10, 14, 18)),     # first layer, ragged output tiles (5 x 7 x 9)
    (4, 32, 3, 1, 1, (12, 12, 24)),     # the same kernel at stride 1
    (64, 3, 1, 1, 0, (5, 6, 7)),        # classifier kernel: 64 channels, voxel count not a multiple of its 64-voxel body
    (128, 2, 1, 1, 0, (4, 4, 6)),
    (256, 4, 1, 1, 0, (3, 4, 5)),
    (1, 32, 3, 2, 1, (24, 24, 48)),     # single-modality first conv (LiTS): LDS-staged kernel, interior + border tiles
    (1, 32, 3, 2, 1, (10, 14, 18)),     # ragged output tiles
    (1, 32, 3, 1, 1, (12, 12, 24)),     # stride 1
    (1, 32, 3, (2, 2, 1), 1, (24, 24, 24)),   # the LiTS stride
]


@pytest.mark.parametrize("c1,c2,k,s,p,sp", CONV_CASES)
def test_conv_step_vs_torch_fp32(ops, c1, c2, k, s, p, sp):
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(c1 * 1000 + c2 + k)
    N = 2
    x = torch.relu(torch.randn(N, c1, *sp, generator=gen))
    w = torch.randn(c2, c1, k, k, k, generator=gen) * (1.0 / (c1 * k ** 3) ** 0.5)
    b = torch.randn(c2, generator=gen) * 0.1
    ref = F.conv3d(x, w, b, s, p)
    y = ref + 0.1 * torch.randn(ref.shape, generator=gen)
    att = torch.randint(1, 4, y[:, 0].shape, generator=gen).float()
    geom = make_geom(x.shape, c2, k, s, p)
    out, sq = ops.conv_step(dev(_ndhwc(x)), dev(w), dev(b), geom, dev(_ndhwc(y)), dev(att), want_out=True)
    got = out.permute(0, 4, 1, 2, 3).cpu()
    # fp32 fma chains in a different order than oneDNN: 1e-5 of the output scale
    assert (got - ref).abs().max() <= 1e-5 * ref.abs().max()
    d2 = (ref.double() - y.double()) ** 2
    sq = sq.cpu()
    assert abs(sq[0].item() - d2.sum().item()) <= 1e-5 * d2.sum().item()
    assert abs(sq[1].item() - (att[:, None].double() * d2).sum().item()) <= 1e-5 * (att[:, None] * d2).sum().item()
    # loss-only call (no output tensor) and no-mask call agree with the fused one
    _, sq2 = ops.conv_step(dev(_ndhwc(x)), dev(w), dev(b), geom, dev(_ndhwc(y)), None)
    sq2 = sq2.cpu()
    if (c1 in (1, 4) and c2 == 32 and k == 3) or (k == 1 and c2 <= 4 and c1 % 32 == 0):
        # loss-only calls of these two layer shapes run the direct-gather kernels (conv3d_direct.hip):
        # a different fp32 summation order than the tiled kernel
        assert abs(sq2[0].item() - sq[0].item()) <= 2e-6 * sq[0].item() and sq2[1].item() == sq2[0].item()
        assert abs(sq2[0].item() - d2.sum().item()) <= 1e-5 * d2.sum().item()
    else:
        assert sq2[0].item() == sq[0].item() and sq2[1].item() == sq2[0].item()
    # bias-free forward
    out3, _ = ops.conv_step(dev(_ndhwc(x)), dev(w), None, geom, want_out=True)
    ref3 = F.conv3d(x, w, None, s, p)
    assert (out3.permute(0, 4, 1, 2, 3).cpu() - ref3).abs().max() <= 1e-5 * ref3.abs().max()


def test_conv_step_fused_act_quant_matches_quantized_forward(ops):
    """PTQConv.forward in quantized mode (PTQConv.py:163-167): quantiser indices bit-exact, so the
    fused path must equal conv(quant_dequant_f32(x)) exactly (same kernel, same fma order)."""
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(77)
    x = torch.relu(torch.randn(2, 16, 8, 8, 8, generator=gen))
    w = torch.randn(16, 16, 3, 3, 3, generator=gen) * 0.05
    b = torch.randn(16, generator=gen) * 0.1
    alpha = torch.tensor(0.8123)
    geom = make_geom(x.shape, 16, 3, 1, 1)
    xq = ops.quant_dequant_f32(dev(_ndhwc(x)), dev(alpha), 4, 0.0, 1.0)
    a, _ = ops.conv_step(xq, dev(w), dev(b), geom, want_out=True)
    f, _ = ops.conv_step(dev(_ndhwc(x)), dev(w), dev(b), geom, act_alpha=dev(alpha), act_levels=4, want_out=True)
    assert torch.equal(a, f)
    ref = O.quantized_forward(x, w, b, alpha, 4, True, 1, 1)
    assert (f.permute(0, 4, 1, 2, 3).cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()


def test_reductions_are_run_to_run_deterministic(ops):
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(9)
    x = torch.relu(torch.randn(2, 32, 8, 16, 16, generator=gen))
    w = torch.randn(32, 32, 3, 3, 3, generator=gen) * 0.03
    y = torch.randn(2, 32, 8, 16, 16, generator=gen)
    geom = make_geom(x.shape, 32, 3, 1, 1)
    xs, ys, ws = dev(_ndhwc(x)), dev(_ndhwc(y)), dev(w)
    vals = {tuple(ops.conv_step(xs, ws, None, geom, ys)[1].cpu().tolist()) for _ in range(5)}
    assert len(vals) == 1
    ms = {tuple(ops.moments(xs).cpu().tolist()) for _ in range(5)}
    assert len(ms) == 1


@pytest.mark.parametrize("L", [4, 16, 256])
def test_single_launch_weight_fixed_point_matches_goldens(ops, gold, L):
    """effq_fixed_point_small: v = w* + dual formed on the fly, whole project_by_iter in one launch."""
    g = gold("g2_project.npz")
    wgt = T(g["wgt"])
    half = dev(wgt * 0.75)
    rest = dev(wgt - wgt * 0.75)
    v = torch.empty_like(half)
    st = ops.new_fp_state()
    assert ops.weight_fixed_point(half, rest, v, L, st) is None
    alpha, iters, done = ops.read_fp_state(st)
    vsum = (wgt * 0.75) + (wgt - wgt * 0.75)
    fit = O.fit_scale(vsum, L, -1, 1)
    assert torch.equal(v.cpu(), vsum)
    assert done == 1 and iters == fit.iters and abs(alpha - fit.alpha) <= 1e-11 * fit.alpha
    err = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    ops.fp_check(st, err)
    assert err.item() == 0
    # fused multi-workgroup iterations (the large-tensor path) agree with the single-launch path
    a2, it2, _ = ops.fit_scale(v, L, -1.0, 1.0)
    assert it2 == iters and abs(a2 - alpha) <= 1e-12 * alpha


@pytest.mark.parametrize("c1,c2,La,Lw,sp", [(32, 32, 4, 4, (8, 8, 16)), (32, 64, 16, 16, (9, 7, 11)),
                                             (64, 64, 4, 4, (8, 8, 8)), (64, 32, 16, 4, (5, 6, 9)),
                                             (32, 32, 128, 128, (4, 4, 8)), (128, 128, 4, 4, (4, 8, 8)),
                                             (256, 64, 16, 16, (4, 4, 8)), (128, 32, 16, 4, (5, 6, 9)),
                                             # 3 x 3 x 3 tiles: the centre tile takes the interior fast path
                                             (32, 32, 4, 4, (24, 12, 24)), (64, 64, 4, 4, (12, 12, 24)),
                                             (64, 64, 16, 16, (12, 10, 24)),      # 64 ch, not tile-divisible
                                             (512, 64, 4, 4, (4, 4, 8)), (512, 128, 16, 16, (3, 5, 9))])   # 2-plane tile
def test_exact_int_conv_step_equals_fp32_path(ops, c1, c2, La, Lw, sp):
    """conv3d_calib_step_i8 (i8 MFMA, exact int32 accumulation) against conv3d_quant_calib_step on the
    SAME quantised operands: identical loss up to the fp32 path's own rounding (<= 2e-6 relative)."""
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(c1 + 7 * c2 + La)
    N = 2
    x = torch.relu(torch.randn(N, *sp, c1, generator=gen))                    # NDHWC
    geom = make_geom((N, c1, *sp), c2, 3, 1, 1)
    assert ops.conv_i8_supported(geom, La, Lw)
    a_act, _, st_a = ops.fit_scale(dev(x), La, 0.0, 1.0)
    xq, _, xidx = ops.quant_dequant_f64path(dev(x), st_a, La, 0.0, 1.0, want_idx=True)
    alpha_act = torch.tensor(a_act, dtype=torch.float32, device="cuda:0")
    wst = dev(torch.randn(c2, c1, 3, 3, 3, generator=gen) * 0.05)
    dual = torch.zeros_like(wst)
    v = torch.empty_like(wst)
    st_w = ops.new_fp_state()
    ops.weight_fixed_point(wst, dual, v, Lw, st_w)
    if wst.numel() > ops.lib.effq_fp_small_max():
        pass
    G = torch.empty_like(wst)
    Gq = torch.empty(wst.shape, dtype=torch.int8, device="cuda:0")
    ops.admm_project_dual(v, wst, st_w, Lw, G, dual, 1.0, Gq)
    a_w = ops.read_fp_state(st_w)[0]
    # Gq really is the signed numerator of G
    assert torch.equal(G.cpu(), np.float32(a_w) * (Gq.cpu().float() / (Lw - 1)).double().float()) or \
        torch.allclose(G.cpu(), np.float32(a_w) * Gq.cpu().float() / (Lw - 1), rtol=2e-7, atol=0)
    b = dev(torch.randn(c2, generator=gen) * 0.1)
    y = dev(torch.randn(N, *sp, c2, generator=gen))
    _, sq32 = ops.conv_step(xq, G, b, geom, y, None)
    sq8 = torch.zeros(2, dtype=torch.float64, device="cuda:0")
    ops.conv_step_i8(xidx, Gq, b, geom, y, alpha_act, La, st_w, Lw, sq8)
    s32, s8 = sq32.cpu().tolist(), sq8.cpu().tolist()
    assert abs(s8[0] - s32[0]) <= 2e-6 * s32[0], (s8, s32)
    assert s8[1] == s8[0]
    # fp64 ground truth of the same integer model
    out = torch.nn.functional.conv3d(xidx.cpu().permute(0, 4, 1, 2, 3).double(), Gq.cpu().double(), None, 1, 1)
    s = float(np.float32(a_act)) * float(np.float32(a_w)) / ((La - 1) * (Lw - 1))
    ref = ((out * s + b.cpu().double().view(1, -1, 1, 1, 1) - y.cpu().permute(0, 4, 1, 2, 3).double()) ** 2).sum().item()
    assert abs(s8[0] - ref) <= 1e-6 * ref
    # deterministic
    sq8b = torch.zeros(2, dtype=torch.float64, device="cuda:0")
    ops.conv_step_i8(xidx, Gq, b, geom, y, alpha_act, La, st_w, Lw, sq8b)
    assert sq8b.cpu().tolist() == s8


@pytest.mark.parametrize("c,sp,La,Lw,with_att,with_bias", [(32, (8, 4, 8), 4, 4, True, True), (32, (24, 12, 24), 4, 4, False, True),
                                                         (32, (16, 8, 16), 16, 16, True, False), (64, (4, 4, 8), 4, 4, True, True),
                                                         (64, (12, 12, 24), 4, 4, True, True), (64, (8, 8, 8), 16, 16, False, False)])
def test_quantised_forward_on_the_i8_matrix_cores(ops, c, sp, La, Lw, with_att, with_bias):
    """conv3d_quant_forward_i8 (the calibrated layer's forward + final loss from one exact integer pass) against the f32
    conv on the same quantised operands (conv3d_quant_calib_step with want_out) and against fp64: output, plain and
    attention-weighted sums; the weight numerators are recovered from G and the iterate's scale."""
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(c + La + sp[0])
    N = 2
    x = torch.relu(torch.randn(N, *sp, c, generator=gen))
    geom = make_geom((N, c, *sp), c, 3, 1, 1)
    assert ops.conv_i8_out_supported(geom, La, Lw)
    a_act, _, st_a = ops.fit_scale(dev(x), La, 0.0, 1.0)
    xq, _, xidx = ops.quant_dequant_f64path(dev(x), st_a, La, 0.0, 1.0, want_idx=True)
    alpha_act = torch.tensor(a_act, dtype=torch.float32, device="cuda:0")
    wst = dev(torch.randn(c, c, 3, 3, 3, generator=gen) * 0.05)
    dual, v, st_w = torch.zeros_like(wst), torch.empty_like(wst), ops.new_fp_state()
    ops.weight_fixed_point(wst, dual, v, Lw, st_w)
    G = torch.empty_like(wst)
    Gq = torch.empty(wst.shape, dtype=torch.int8, device="cuda:0")
    ops.admm_project_dual(v, wst, st_w, Lw, G, dual, 1.0, Gq)
    b = dev(torch.randn(c, generator=gen) * 0.1) if with_bias else None
    y = dev(torch.randn(N, *sp, c, generator=gen))
    att = dev(torch.tensor([0.25, 1.0, 3.5])[torch.randint(0, 3, (N, *sp), generator=gen)]) if with_att else None
    out32, sq32 = ops.conv_step(xq, G, b, geom, y, att, want_out=True)
    out8, sq8 = ops.conv_forward_i8(xidx, G, b, geom, y, att, alpha_act, La, st_w, Lw)
    torch.cuda.synchronize()
    assert torch.equal(ops._keep_i8[0].reshape(Gq.shape), Gq)          # the numerators, recovered from G / alpha
    scale = out32.abs().max().item()
    assert (out8 - out32).abs().max().item() <= 3e-6 * scale
    s32, s8 = sq32.cpu().tolist(), sq8.cpu().tolist()
    assert abs(s8[0] - s32[0]) <= 2e-6 * s32[0] and abs(s8[1] - s32[1]) <= 2e-6 * s32[1], (s8, s32)
    if not with_att:
        assert abs(s8[1] - s8[0]) <= 1e-7 * s8[0]           # two fp32 chains of the same terms
    # fp64 ground truth of the integer model
    a_w = ops.read_fp_state(st_w)[0]
    ref = torch.nn.functional.conv3d(xidx.cpu().permute(0, 4, 1, 2, 3).double(), Gq.cpu().double(), None, 1, 1)
    ref = ref * (float(np.float32(a_act)) * float(np.float32(a_w)) / ((La - 1) * (Lw - 1)))
    if with_bias:
        ref = ref + b.cpu().double().view(1, -1, 1, 1, 1)
    assert (out8.cpu().permute(0, 4, 1, 2, 3).double() - ref).abs().max().item() <= 3e-7 * scale
    err2 = (ref - y.cpu().permute(0, 4, 1, 2, 3).double()) ** 2
    assert abs(s8[0] - err2.sum().item()) <= 1e-6 * err2.sum().item()
    if with_att:
        want = (att.cpu().double().unsqueeze(1) * err2).sum().item()
        assert abs(s8[1] - want) <= 1e-6 * want


@pytest.mark.parametrize("c1,c2,k,pad,sp,La,Lw,att_kind,with_bias", [
    (32, 32, 3, 1, (8, 8, 16), 4, 4, "none", True), (32, 32, 3, 1, (9, 7, 11), 4, 4, "classes0", True),
    (64, 64, 3, 1, (8, 8, 8), 16, 16, "classes", False), (32, 64, 1, 0, (6, 5, 7), 4, 4, "classes", True),
    (16, 8, 3, 1, (5, 6, 9), 4, 16, "none", True)])
def test_loss_from_the_unweighted_gram_system_equals_the_conv_loss(ops, c1, c2, k, pad, sp, La, Lw, att_kind, with_bias):
    """effq_gram_loss: sum (conv(Qx, G, b) - y)^2 = sum_c g_c^T Au g_c - 2 g_c . Bu_c + sum y^2 with the UNWEIGHTED fp64
    system that effq_gram_accum_i8_unw produces beside the attention-weighted A0 / B0.  Against the fp64 evaluation of the
    same integer model: <= 1e-9 (the attention weights - zeros included - must not enter); against the conv entry point
    (fp32 epilogue): <= 2e-6; A0 / B0 themselves are unchanged by asking for the by-product."""
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(c1 + 3 * c2 + La + k)
    N = 2
    x = torch.relu(torch.randn(N, *sp, c1, generator=gen))
    geom = make_geom((N, c1, *sp), c2, k, 1, pad)
    od, oh, ow = geom.out_dims()
    a_act, _, st_a = ops.fit_scale(dev(x), La, 0.0, 1.0)
    xq, _, xidx = ops.quant_dequant_f64path(dev(x), st_a, La, 0.0, 1.0, want_idx=True)
    alpha_act = torch.tensor(a_act, dtype=torch.float32, device="cuda:0")
    y = dev(torch.randn(N, od, oh, ow, c2, generator=gen))
    att = None
    if att_kind != "none":
        lo = 0 if att_kind == "classes0" else 1
        att = dev(torch.randint(lo, 4, (N, od, oh, ow), generator=gen).float())
    cls = ops.att_classes(att)
    A0, B0, Au, Bu = ops.gram_i8(xidx, cls, y, geom, with_bias, alpha_act, La, unweighted=True)
    A0b, B0b = ops.gram_i8(xidx, cls, y, geom, with_bias, alpha_act, La)
    assert torch.equal(A0, A0b) and torch.equal(B0, B0b)
    # the unweighted system against fp64 on the host: the integer model xhat = s * level id, s = alpha_act / (La - 1)
    sx = float(np.float32(a_act)) / (La - 1)
    X = O.patch_matrix(xidx.cpu().permute(0, 4, 1, 2, 3).float().numpy(), (k, k, k), 1, pad,
                       ones_row=with_bias).astype(np.float64)
    X[:c1 * k ** 3] *= sx
    Y = y.cpu().double().reshape(-1, c2).numpy().T
    assert np.abs(Au.cpu().numpy() - X @ X.T).max() <= 1e-12 * np.abs(X @ X.T).max()
    assert np.abs(Bu.cpu().numpy() - Y @ X.T).max() <= 2e-8 * np.abs(Y @ X.T).max()     # y rides in 32-bit fixed point
    # an iterate: projected weights of a perturbed solution
    wst = dev(torch.randn(c2, c1, k, k, k, generator=gen) * 0.05)
    dual, v, G = torch.zeros_like(wst), torch.empty_like(wst), torch.empty_like(wst)
    st_w = ops.new_fp_state()
    ops.weight_fixed_point(wst, dual, v, Lw, st_w)
    Gq = torch.empty(wst.shape, dtype=torch.int8, device="cuda:0")
    ops.admm_project_dual(v, wst, st_w, Lw, G, dual, 1.0, Gq)
    b = dev(torch.randn(c2, generator=gen) * 0.1) if with_bias else None
    syy = (y.double() ** 2).sum().reshape(1)
    got = ops.gram_loss(Au, Bu, syy, G, b).cpu().tolist()
    out = torch.nn.functional.conv3d(xidx.cpu().permute(0, 4, 1, 2, 3).double() * sx, G.cpu().double(),
                                     None if b is None else b.cpu().double(), 1, pad)
    ref = ((out - y.cpu().permute(0, 4, 1, 2, 3).double()) ** 2).sum().item()
    assert abs(got[0] - ref) <= 2e-8 * ref, (got, ref)
    assert got[1] == got[0]
    _, sq32 = ops.conv_step(xq, G, b, geom, y, None)
    assert abs(sq32.cpu().tolist()[0] - got[0]) <= 2e-6 * ref
    assert ops.gram_loss(Au, Bu, syy, G, b).cpu().tolist() == got          # deterministic


@pytest.mark.parametrize("c1,c2,sp,La,Lw,count,with_bias", [
    (64, 64, (6, 5, 7), 4, 4, 5, True),            # nw = 1728: 6.75 row tiles of 256 (a partial one), 2 digit planes
    (128, 128, (5, 4, 6), 4, 4, 8, True),          # the 128-channel geometry of the BraTS net, a full group
    (64, 32, (9, 8, 10), 16, 16, 3, False),        # 16 / 16 levels, no bias, 3 planes
    (128, 256, (4, 4, 4), 4, 8, 2, True)])
def test_losses_of_a_group_of_iterates_on_the_i8_matrix_cores(ops, c1, c2, sp, La, Lw, count, with_bias):
    """effq_gram_loss_i8 (the per-iteration losses of the 128- / 256-channel layers): the quadratic form g^T Au g of every
    iterate of a group as an exact integer <K, J^T J> on the i8 matrix cores (K = the Gram system of the level ids in
    balanced base-256 digit planes, J = the int8 level numerators of the iterate), bias and cross terms in fp64.  Against
    the fp64 evaluation of the same integer model (the contract of conv3d_calib_step_i8: out = f32(alpha_a) f32(alpha_w)
    / ((La-1)(Lw-1)) * (J . k) + b): <= 2e-8 (the fixed-point y inside Bu; the quadratic form itself is exact); against the exact-integer conv pass it replaces: <= 2e-6 (that kernel's
    fp32 epilogue); bit-identical run to run; a stacked group equals the iterates one at a time."""
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(c1 + 3 * c2 + La + 7 * count)
    N, k, pad = 2, 3, 1
    x = torch.relu(torch.randn(N, *sp, c1, generator=gen))
    geom = make_geom((N, c1, *sp), c2, k, 1, pad)
    od, oh, ow = geom.out_dims()
    a_act, _, st_a = ops.fit_scale(dev(x), La, 0.0, 1.0)
    xq, _, xidx = ops.quant_dequant_f64path(dev(x), st_a, La, 0.0, 1.0, want_idx=True)
    alpha_act = torch.tensor(a_act, dtype=torch.float32, device="cuda:0")
    y = dev(torch.randn(N, od, oh, ow, c2, generator=gen))
    cls = ops.att_classes(dev(torch.randint(1, 4, (N, od, oh, ow), generator=gen).float()))
    n = c1 * 27 + int(with_bias)
    assert ops.gram_loss_i8_supported(c2, n, with_bias, Lw)
    A0, B0, Au, Bu = ops.gram_i8(xidx, cls, y, geom, with_bias, alpha_act, La, unweighted=True)
    V = N * od * oh * ow
    planes = ops.gram_loss_i8_planes(Au, with_bias, alpha_act, La, V)
    torch.cuda.synchronize()
    assert int(planes._effq_err.item()) == 0
    # the planes ARE the integer system: K = sum_p 256^p D_p = sum_v k k^T
    nw = c1 * 27
    X = O.patch_matrix(xidx.cpu().permute(0, 4, 1, 2, 3).float().numpy(), (k, k, k), 1, pad).astype(np.int64)
    K = X @ X.T
    Kp = sum((256 ** q) * planes[q, :nw].cpu().numpy().astype(np.int64) for q in range(planes.shape[0]))
    assert np.array_equal(Kp, K) and not planes[:, nw:].any()
    # `count` iterates: projected weights of perturbed solutions, stacked like the rings of effq_admm_run
    Gq = torch.empty(count, c2, nw, dtype=torch.int8, device="cuda:0")
    G = torch.empty(count, c2, nw, device="cuda:0")
    states = torch.zeros(count, 5, dtype=torch.float64, device="cuda:0")
    b = dev(torch.randn(count, c2, generator=gen) * 0.1) if with_bias else None
    for j in range(count):
        wst = dev(torch.randn(c2, nw, generator=gen) * (0.03 + 0.01 * j))
        dual, v = torch.zeros_like(wst), torch.empty_like(wst)
        ops.weight_fixed_point(wst, dual, v, Lw, states[j])
        ops.admm_project_dual(v, wst, states[j], Lw, G[j], dual, 1.0, Gq[j])
    syy = (y.double() ** 2).sum().reshape(1)
    hist = ops.gram_loss_i8(planes, Au, Bu, syy, Gq, b, states, alpha_act, La, Lw)
    got = hist.cpu().numpy()
    assert np.array_equal(got[:, 0], got[:, 1])
    xk = xidx.cpu().permute(0, 4, 1, 2, 3).double()
    yd = y.cpu().permute(0, 4, 1, 2, 3).double()
    sa = float(np.float32(a_act)) / (La - 1)
    for j in range(count):
        sw = float(np.float32(states[j, 0].item())) / (Lw - 1)
        Jw = Gq[j].cpu().double().reshape(c2, c1, k, k, k)
        out = torch.nn.functional.conv3d(xk, Jw, None, 1, pad) * (sa * sw)
        if with_bias:
            out = out + b[j].cpu().double().reshape(1, c2, 1, 1, 1)
        ref = ((out - yd) ** 2).sum().item()
        assert abs(got[j, 0] - ref) <= 2e-8 * ref, (j, got[j, 0], ref)       # (Bu carries y in 32-bit fixed point)
        # the exact-integer conv pass it replaces (fp32 epilogue)
        sq8 = torch.zeros(2, dtype=torch.float64, device="cuda:0")
        if ops.conv_i8_supported(geom, La, Lw):
            ops.conv_step_i8(xidx, Gq[j].reshape(c2, c1, k, k, k), None if b is None else b[j], geom, y, alpha_act, La,
                             states[j], Lw, sq8)
            assert abs(sq8.cpu().tolist()[0] - got[j, 0]) <= 2e-6 * ref, (j, sq8.cpu().tolist(), got[j, 0])
        one = ops.gram_loss_i8(planes, Au, Bu, syy, Gq[j:j + 1], None if b is None else b[j:j + 1], states[j:j + 1],
                               alpha_act, La, Lw).cpu().numpy()
        assert one[0, 0] == got[j, 0], (j, one, got[j])
    assert np.array_equal(ops.gram_loss_i8(planes, Au, Bu, syy, Gq, b, states, alpha_act, La, Lw).cpu().numpy(), got)


@pytest.mark.parametrize("c1,c2,k,s,pad,sp,with_bias", [
    (4, 32, 3, 2, 1, (16, 14, 18), True),          # BraTS first conv (n = 109), ragged voxel count
    (32, 3, 1, 1, 0, (9, 7, 11), True),            # classifier (n = 33, c2 = 3)
    (1, 32, 3, (2, 2, 1), 1, (12, 10, 9), True),   # LiTS first conv, stride 2,2,1 (n = 28)
    (4, 8, 3, 1, 1, (5, 6, 7), False),             # no bias (n = 108)
    (2, 64, 3, 1, 0, (6, 6, 6), True)])            # no padding, c2 = 64
def test_fp64_gram_system_of_a_full_precision_input(ops, c1, c2, k, s, pad, sp, with_bias):
    """effq_gram_f64: Au = sum xhat xhat^T, Bu = sum y xhat^T in fp64 (products of fp32 values are exact, the sums differ
    from numpy's by rounding order only: <= 1e-13 of the largest entry), reference im2col row order, deterministic; and the
    loss effq_gram_loss forms from it equals the fp64 conv loss <= 1e-10 and the conv entry point's <= 2e-6."""
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(c1 * 11 + c2 + k)
    N = 2
    x = torch.randn(N, *sp, c1, generator=gen)
    geom = make_geom((N, c1, *sp), c2, k, s, pad)
    assert ops.gram_f64_supported(geom, with_bias)
    od, oh, ow = geom.out_dims()
    y = torch.randn(N, od, oh, ow, c2, generator=gen)
    Au, Bu = ops.gram_f64(dev(x), dev(y), geom, with_bias)
    X = O.patch_matrix(x.permute(0, 4, 1, 2, 3).numpy(), (k, k, k), s, pad, ones_row=with_bias,
                       dtype=np.float64)
    Y = y.double().reshape(-1, c2).numpy().T
    assert Au.shape == (X.shape[0], X.shape[0]) and Bu.shape == (c2, X.shape[0])
    assert np.abs(Au.cpu().numpy() - X @ X.T).max() <= 1e-13 * np.abs(X @ X.T).max()
    assert np.abs(Bu.cpu().numpy() - Y @ X.T).max() <= 1e-13 * max(np.abs(Y @ X.T).max(), 1.0)
    assert torch.equal(Au, Au.T)
    Au2, Bu2 = ops.gram_f64(dev(x), dev(y), geom, with_bias)
    assert torch.equal(Au, Au2) and torch.equal(Bu, Bu2)                       # deterministic
    G = dev(torch.randn(c2, c1, k, k, k, generator=gen) * 0.1)
    b = dev(torch.randn(c2, generator=gen) * 0.1) if with_bias else None
    syy = (dev(y).double() ** 2).sum().reshape(1)
    got = ops.gram_loss(Au, Bu, syy, G, b).cpu().tolist()
    out = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), G.cpu().double(),
                                     None if b is None else b.cpu().double(), s, pad)
    ref = ((out - y.permute(0, 4, 1, 2, 3).double()) ** 2).sum().item()
    assert abs(got[0] - ref) <= 1e-10 * ref, (got, ref)
    _, sq32 = ops.conv_step(dev(x), G, b, geom, dev(y), None)
    assert abs(sq32.cpu().tolist()[0] - got[0]) <= 2e-6 * ref


def test_fp64_gram_system_rejects_what_it_cannot_do(ops):
    from efficientq_amd.hip_ops import make_geom
    assert not ops.gram_f64_supported(make_geom((1, 32, 8, 8, 8), 32, 3, 1, 1), True)      # n = 865 > 128
    assert not ops.gram_f64_supported(make_geom((1, 4, 8, 8, 8), 128, 1, 1, 0), True)      # c2 > 64


@pytest.mark.parametrize("c1,c2,k,s,p,La,Lw,sp", [
    (4, 32, 3, 2, 1, 256, 256, (12, 10, 14)),     # the first conv of the BraTS net (q_first = 256)
    (32, 3, 1, 1, 0, 256, 256, (6, 7, 9)),        # the classifier (q_last = 256)
    (32, 64, 1, 1, 0, 4, 4, (6, 6, 6)),           # 1x1x1 convs of the up/down paths
    (64, 32, 1, 1, 0, 16, 4, (5, 6, 7)),
    (128, 64, 1, 1, 0, 4, 16, (4, 4, 6)),
    (4, 32, 3, 1, 1, 4, 256, (7, 8, 9)),          # only the weights beyond int8
    (16, 32, 2, 2, 0, 256, 4, (8, 8, 8)),         # only the activations beyond int8, even kernel
])
def test_short_k_exact_int_conv_step(ops, c1, c2, k, s, p, La, Lw, sp):
    """conv3d_calib_step_i8s (direct-gather i8 MFMA, up to 256 levels through offset operands) against the fp64
    value of the same integer model and against the fp32 conv path on the same quantised operands."""
    from efficientq_amd.hip_ops import make_geom
    gen = torch.Generator().manual_seed(c1 + 7 * c2 + La + 3 * Lw)
    N = 2
    x = torch.relu(torch.randn(N, *sp, c1, generator=gen) + 0.3)               # NDHWC
    geom = make_geom((N, c1, *sp), c2, k, s, p)
    assert ops.conv_i8s_supported(geom, La, Lw)
    a_act, _, st_a = ops.fit_scale(dev(x), La, 0.0, 1.0)
    xq, _, xidx = ops.quant_dequant_f64path(dev(x), st_a, La, 0.0, 1.0, want_idx=True)
    alpha_act = torch.tensor(a_act, dtype=torch.float32, device="cuda:0")
    wst = dev(torch.randn(c2, c1, k, k, k, generator=gen) * 0.05)
    dual = torch.zeros_like(wst)
    v = torch.empty_like(wst)
    st_w = ops.new_fp_state()
    ops.weight_fixed_point(wst, dual, v, Lw, st_w)
    G = torch.empty_like(wst)
    Gq = torch.empty(wst.shape, dtype=torch.int8, device="cuda:0")
    ops.admm_project_dual(v, wst, st_w, Lw, G, dual, 1.0, Gq)
    a_w = ops.read_fp_state(st_w)[0]
    num = (2 * Gq.cpu().double() + 1) if Lw > 128 else Gq.cpu().double()       # signed numerators 2*level-(Lw-1)
    assert torch.allclose(G.cpu().double(), float(np.float32(a_w)) * num / (Lw - 1), rtol=3e-7, atol=0)
    b = dev(torch.randn(c2, generator=gen) * 0.1)
    od, oh, ow = geom.out_dims()
    y = dev(torch.randn(N, od, oh, ow, c2, generator=gen))
    _, sq32 = ops.conv_step(xq, G, b, geom, y, None)
    sq8 = torch.zeros(2, dtype=torch.float64, device="cuda:0")
    ops.conv_step_i8s(xidx, Gq, b, geom, y, alpha_act, La, st_w, Lw, sq8, True)
    s32, s8 = sq32.cpu().tolist(), sq8.cpu().tolist()
    out = torch.nn.functional.conv3d(xidx.cpu().permute(0, 4, 1, 2, 3).double(), num, None, s, p)
    sc = float(np.float32(a_act)) * float(np.float32(a_w)) / ((La - 1) * (Lw - 1))
    ref = ((out * sc + b.cpu().double().view(1, -1, 1, 1, 1) - y.cpu().permute(0, 4, 1, 2, 3).double()) ** 2).sum().item()
    assert abs(s8[0] - ref) <= 1e-6 * ref, (s8, ref)
    assert abs(s8[0] - s32[0]) <= 3e-6 * s32[0], (s8, s32)
    assert s8[1] == s8[0]
    # later calls of the layer reuse the cached level sums; deterministic
    sq8b = torch.zeros(2, dtype=torch.float64, device="cuda:0")
    ops.conv_step_i8s(xidx, Gq, b, geom, y, alpha_act, La, st_w, Lw, sq8b, False)
    assert sq8b.cpu().tolist() == s8


def test_short_k_exact_int_conv_rejects_what_it_cannot_do(ops):
    from efficientq_amd.hip_ops import make_geom
    assert not ops.conv_i8s_supported(make_geom((1, 32, 8, 8, 8), 32, 3, 1, 1), 4, 4)      # K = 864
    assert not ops.conv_i8s_supported(make_geom((1, 8, 8, 8, 8), 32, 1, 1, 0), 4, 4)       # C1 = 8
    assert not ops.conv_i8s_supported(make_geom((1, 256, 8, 8, 8), 128, 1, 1, 0), 4, 4)    # too many B operands
    assert not ops.conv_i8s_supported(make_geom((1, 4, 8, 8, 8), 32, 3, 2, 1), 4, 300)


@pytest.mark.parametrize("n,L", [(40000, 4), (110592, 4), (442368, 16), (1769472, 4), (300001, 256), (7077888, 4)])
def test_cooperative_weight_fixed_point_matches_oracle(ops, n, L):
    """effq_fixed_point_coop (several workgroups, grid barrier per iteration): same alpha / iteration count as the
    fp64 oracle, v = w* + dual formed on the fly, identical on repetition (deterministic combination order)."""
    gen = torch.Generator().manual_seed(n + L)
    w = torch.randn(n, generator=gen) * 0.05
    du = torch.randn(n, generator=gen) * 0.005
    assert ops.lib.effq_fp_small_max() < n <= ops.lib.effq_fp_coop_max()
    v = torch.empty(n, device="cuda:0")
    st = ops.new_fp_state()
    assert ops.weight_fixed_point(dev(w), dev(du), v, L, st) is None          # no host round trip
    alpha, iters, done = ops.read_fp_state(st)
    vsum = w + du
    fit = O.fit_scale(vsum, L, -1, 1)
    assert torch.equal(v.cpu(), vsum)
    assert done == 1 and iters == fit.iters, (done, iters, fit.iters)
    assert abs(alpha - fit.alpha) <= 1e-11 * fit.alpha
    st2 = ops.new_fp_state()
    ops.weight_fixed_point(dev(w), dev(du), v, L, st2)
    assert ops.read_fp_state(st2) == (alpha, iters, done)


def test_attention_class_lists_are_built_on_the_device(ops):
    """effq_att_classes: the voxel list of a mask, grouped by attention weight (segments padded to multiples of 128 with
    -1, classes in ascending order of weight) - checked as sets against torch; 17 distinct weights -> not representable."""
    gen = torch.Generator().manual_seed(3)
    for shape, vals in (((2, 9, 10, 11), [1.0, 2.0, 3.0]), ((1, 4, 4, 4), [2.0]), ((3, 16, 16, 16), [1.0, 5.0, 2.0, 7.0, 3.0])):
        idx = torch.randint(0, len(vals), shape, generator=gen)
        att = dev(torch.tensor(vals)[idx].float().contiguous())
        ops._att_cache.clear()
        lst, chunk_cls, cls_w, k = ops.att_classes(att)
        want_vals = sorted(set(torch.unique(att).cpu().tolist()))
        assert k == len(want_vals) and cls_w.cpu().tolist() == want_vals
        lst_h, cc = lst.cpu(), chunk_cls.cpu()
        assert lst_h.numel() % 128 == 0 and cc.numel() * 128 == lst_h.numel()
        flat = att.reshape(-1).cpu()
        pos = 0
        for c, v in enumerate(want_vals):
            members = torch.nonzero(flat == v).reshape(-1)
            padded = (members.numel() + 127) // 128 * 128
            seg = lst_h[pos: pos + padded]
            assert torch.equal(torch.sort(seg[seg >= 0]).values, members.to(torch.int32))
            assert (seg < 0).sum().item() == padded - members.numel()
            assert torch.all(cc[pos // 128: (pos + padded) // 128] == c)
            pos += padded
        assert pos == lst_h.numel()
    many = dev(torch.arange(17).float().repeat(100).reshape(1, 17, 10, 10).contiguous())
    ops._att_cache.clear()
    assert ops.att_classes(many) is None


@pytest.mark.parametrize("n,c2", [(33, 3), (109, 32), (865, 32)])
def test_gram_system_packs_into_one_message_and_back(ops, n, c2):
    """effq_gram_pack / effq_gram_unpack: upper triangle of A0 + B0, exact round trip; summing two packed partial systems
    equals packing their sum (what the data-parallel all-reduce does)."""
    gen = torch.Generator().manual_seed(n)
    mk = lambda: (lambda m: (m + m.T).contiguous())(torch.randn(n, n, generator=gen))
    A1, A2 = dev(mk()), dev(mk())
    B1, B2 = dev(torch.randn(c2, n, generator=gen)), dev(torch.randn(c2, n, generator=gen))
    bufs = []
    for A, B in ((A1, B1), (A2, B2)):
        buf = torch.empty(ops.lib.effq_gram_packed_elems(n, c2), dtype=torch.float32, device="cuda:0")
        from efficientq_amd.hip_ops import check, _ptr
        check(ops.lib.effq_gram_pack(_ptr(A), _ptr(B), n, c2, _ptr(buf), ops.stream), "pack")
        bufs.append(buf)
    assert bufs[0].numel() == n * (n + 1) // 2 + c2 * n
    iu = torch.triu_indices(n, n)
    assert torch.equal(bufs[0][: n * (n + 1) // 2].cpu(), A1.cpu()[iu[0], iu[1]])
    A, B = A1.clone(), B1.clone()
    got = ops.gram_reduce(A, B, lambda t: t.add_(bufs[1]))          # "all-reduce" over two ranks
    assert torch.equal(got[0], A1 + A2) and torch.equal(got[1], B1 + B2)
    assert torch.equal(got[0], got[0].T)


# ------------------------------------------------------------------ a2: bucketed single-workgroup fixed point
def _run_bucket(ops, a, b, L, lo=-1.0, hi=1.0):
    v = torch.empty(a.numel(), device="cuda:0") if b is not None else None
    st = ops.new_fp_state()
    ops.fixed_point_bucket(dev(a), None if b is None else dev(b), v, L, st, lo, hi)
    return ops.read_fp_state(st), v


@pytest.mark.parametrize("L", [4, 16, 256])
def test_bucketed_fixed_point_matches_reference_goldens(ops, gold, L):
    """effq_fixed_point_bucket against G2 (the reference's project_by_iter on weights AND on ReLU activations):
    alpha <= 1e-11 relative, equal iteration count, and the level ids at that alpha bit-exact."""
    g = gold("g2_project.npz")
    for key, lo in (("wgt", -1.0), ("act", 0.0)):
        x = T(g[key]).reshape(-1).contiguous()
        (alpha, iters, done), _ = _run_bucket(ops, x, None, L, lo, 1.0)
        want_a, want_it = float(g[f"{key}_L{L}_alpha"]), int(g[f"{key}_L{L}_iters"])
        assert done == 1 and iters == want_it, (key, L, iters, want_it)
        assert abs(alpha - want_a) <= 1e-11 * want_a
        st = ops.new_fp_state()
        st[0] = alpha
        _, _, idx = ops.quant_dequant_f64path(dev(x), st, L, lo, 1.0, want_idx=True)
        assert torch.equal(idx.cpu().reshape(-1), T(g[f"{key}_L{L}_idx"]).reshape(-1).to(torch.uint8))


@pytest.mark.parametrize("n,L", [(96, 256), (3456, 256), (2048, 4), (8191, 4), (16385, 16), (27648, 4), (27648, 16),
                                 (32768, 8), (32768, 256), (20000, 2), (5, 4), (1, 4), (32769, 4), (110592, 4),
                                 (110592, 16), (442368, 4), (300001, 256), (1769472, 4), (7077888, 4)])
def test_bucketed_fixed_point_matches_oracle(ops, n, L):
    gen = torch.Generator().manual_seed(7 * n + L)
    w = torch.randn(n, generator=gen) * 0.05
    du = torch.randn(n, generator=gen) * 0.005
    (alpha, iters, done), v = _run_bucket(ops, w, du, L)
    vsum = w + du
    assert torch.equal(v.cpu(), vsum)
    fit = O.fit_scale(vsum, L, -1, 1)
    assert done == 1 and iters == fit.iters, (done, iters, fit.iters)
    assert abs(alpha - fit.alpha) <= 1e-11 * abs(fit.alpha)
    (a2, i2, d2), _ = _run_bucket(ops, w, du, L)                     # deterministic: bit-identical on repetition
    assert (a2, i2, d2) == (alpha, iters, done)
    if n <= ops.lib.effq_fp_small_max():                              # and the all-values kernel agrees
        st = ops.new_fp_state()
        vv = torch.empty(n, device="cuda:0")
        ops.weight_fixed_point(dev(w), dev(du), vv, L, st)
        a3, i3, d3 = ops.read_fp_state(st)
        assert i3 == iters and abs(a3 - alpha) <= 1e-12 * abs(alpha)


def test_bucketed_fixed_point_across_scales_with_one_workspace(ops):
    """The multi-workgroup path reuses its workspace: calls on data of very different scale, back to back."""
    gen = torch.Generator().manual_seed(77)
    base = torch.randn(150000, generator=gen)
    for scale in (0.05, 0.5, 0.005, 50.0, 0.05):
        x = (base * scale).contiguous()
        fit = O.fit_scale(x, 4, -1, 1)
        (alpha, iters, done), _ = _run_bucket(ops, x, None, 4)
        assert done == 1 and iters == fit.iters and abs(alpha - fit.alpha) <= 1e-11 * fit.alpha, (scale, alpha, fit.alpha)


@pytest.mark.parametrize("L", [2, 3, 4, 5, 16, 256])
def test_bucketed_fixed_point_on_adversarial_values(ops, L):
    """Values sitting ON level boundaries of intermediate scales, exact zeros, tiny negatives (where 1 - |v/a| rounds
    to 1), duplicates, a heavy outlier (almost everything in one bucket), all-equal tensors."""
    gen = torch.Generator().manual_seed(1234 + L)
    base = torch.randn(20000, generator=gen) * 0.1
    fit0 = O.fit_scale(base, L, -1, 1)
    d = 2.0 / (L - 1)
    cases = {}
    # boundaries (k - 0.5) * d - 1 at the converged scale and at the start scale, hit exactly and one ulp either side
    bnd = torch.tensor([(k - 0.5) * d - 1.0 for k in range(1, L)], dtype=torch.float64)
    pts = []
    for a in (fit0.alpha, base.abs().double().mean().item()):
        p = (bnd * a).float()
        pts += [p, torch.nextafter(p, torch.tensor(10.0)), torch.nextafter(p, torch.tensor(-10.0))]
    cases["on boundaries"] = torch.cat([base] + pts * 7)
    cases["zeros and tiny"] = torch.cat([base, torch.zeros(500), -torch.zeros(300), torch.full((200,), -1e-30),
                                         torch.full((200,), 1e-30), torch.full((100,), -1e-42)])
    cases["duplicates"] = torch.round(base * 50) / 50
    out = base.clone()
    out[0] = 500.0
    cases["outlier"] = out
    cases["all equal"] = torch.full((5000,), 0.37)
    cases["two values"] = torch.cat([torch.full((3000,), -0.2), torch.full((2000,), 0.9)])
    big = {f"{k} (multi-workgroup path)": torch.cat([v, torch.randn(40000, generator=gen) * 0.1])
           for k, v in cases.items() if k in ("on boundaries", "zeros and tiny", "outlier")}
    big["all equal (multi-workgroup path)"] = torch.full((50000,), 0.37)
    cases.update(big)
    for name, x in cases.items():
        try:
            fit = O.fit_scale(x, L, -1, 1)
        except RuntimeWarning:
            fit = None
        (alpha, iters, done), _ = _run_bucket(ops, x.contiguous(), None, L)
        if fit is None:
            assert done == 2, name
            continue
        assert done == 1 and iters == fit.iters, (name, L, done, iters, fit.iters)
        assert abs(alpha - fit.alpha) <= 1e-11 * abs(fit.alpha), (name, alpha, fit.alpha)


def _run_small(ops, x, du, L):
    """The all-values single-workgroup kernel k_fp_small (effq_fixed_point_small): the many-level path of the product (the
    sorted-value kernel these cases were written for was removed in round 4: 3 x slower)."""
    v = torch.empty(x.numel(), device="cuda:0")
    st = ops.new_fp_state()
    ops.weight_fixed_point(dev(x), dev(du if du is not None else torch.zeros_like(x)), v, L, st)
    return ops.read_fp_state(st), v


@pytest.mark.parametrize("L", [32, 64, 100, 256])
@pytest.mark.parametrize("n", [1, 63, 96, 864, 3456, 4096])
def test_many_level_fixed_point_matches_oracle(ops, n, L):
    """k_fp_small at many levels (<= 4096 values: the 256-level weights of the first / last conv): alpha <= 1e-11 of the
    fp64 restatement of project_by_iter (layer_helper.py:40-70), SAME iteration count, v = w* + dual stored."""
    gen = torch.Generator().manual_seed(n * 7 + L)
    w = torch.randn(n, generator=gen) * 0.07
    du = torch.randn(n, generator=gen) * 0.01
    try:
        fit = O.fit_scale(w + du, L, -1, 1)
    except RuntimeWarning:
        fit = None
    (alpha, iters, done), v = _run_small(ops, w, du, L)
    assert torch.equal(v.cpu(), w + du)
    if fit is None:
        assert done == 2
        return
    assert done == 1 and iters == fit.iters, (done, iters, fit.iters)
    assert abs(alpha - fit.alpha) <= 1e-11 * abs(fit.alpha), (alpha, fit.alpha)


@pytest.mark.parametrize("L", [64, 256])
def test_many_level_fixed_point_on_adversarial_values(ops, L):
    """Values ON rounding boundaries of the converged and of the start scale (and one ulp either side), zeros, signed
    zeros, denormals, duplicates, an outlier, all-equal / two-valued tensors: the level counts
    must be exactly the reference's."""
    gen = torch.Generator().manual_seed(4321 + L)
    base = torch.randn(3000, generator=gen) * 0.1
    fit0 = O.fit_scale(base, L, -1, 1)
    d = 2.0 / (L - 1)
    bnd = torch.tensor([(k - 0.5) * d - 1.0 for k in range(1, L)], dtype=torch.float64)
    pts = []
    for a in (fit0.alpha, base.abs().double().mean().item()):
        p = (bnd * a).float()
        pts += [p[::4], torch.nextafter(p, torch.tensor(10.0))[1::4], torch.nextafter(p, torch.tensor(-10.0))[2::4]]
    cases = {"on boundaries": torch.cat([base] + pts)[:4096],
             "zeros and tiny": torch.cat([base, torch.zeros(300), -torch.zeros(200), torch.full((100,), -1e-30),
                                          torch.full((100,), 1e-30), torch.full((50,), -1e-42)]),
             "duplicates": torch.round(base * 50) / 50,
             "all equal": torch.full((2000,), 0.37),
             "two values": torch.cat([torch.full((1500,), -0.2), torch.full((1000,), 0.9)])}
    out = base.clone()
    out[0] = 500.0
    cases["outlier"] = out
    for name, x in cases.items():
        try:
            fit = O.fit_scale(x, L, -1, 1)
        except RuntimeWarning:
            fit = None
        (alpha, iters, done), _ = _run_small(ops, x.contiguous(), None, L)
        if fit is None:
            assert done == 2, name
            continue
        assert done == 1 and iters == fit.iters, (name, L, done, iters, fit.iters)
        assert abs(alpha - fit.alpha) <= 1e-11 * abs(fit.alpha), (name, alpha, fit.alpha)


@pytest.mark.parametrize("bits,levels,n", [(2, 4, 27648), (4, 16, 1001), (1, 2, 77), (8, 256, 513), (2, 3, 5), (4, 16, 0)])
def test_bit_packed_level_storage_round_trip(ops, bits, levels, n):
    """Row f2: level ids packed at 1/2/4/8 bits (the reference keeps one uint8 per weight, PTQConv.py:125-152)."""
    gen = torch.Generator().manual_seed(bits * 1000 + n)
    idx = torch.randint(0, levels, (n,), generator=gen).to(torch.uint8)
    assert ops.storage_bits(levels) == bits
    packed = ops.pack_levels(dev(idx), bits)
    assert packed.numel() == (n * bits + 7) // 8
    # little-endian bit stream, checked against numpy
    want = np.zeros(packed.numel(), dtype=np.uint8)
    for i, v in enumerate(idx.numpy()):
        want[(i * bits) // 8] |= np.uint8((int(v) << ((i * bits) % 8)) & 0xFF)
    assert np.array_equal(packed.cpu().numpy(), want)
    assert torch.equal(ops.unpack_levels(packed, n, bits).cpu(), idx)


@pytest.mark.parametrize("C,sp,scale", [(32, (4, 6, 5), (2, 2, 2)), (3, (5, 4, 7), (2, 2, 2)), (64, (3, 3, 4), (2, 2, 1)),
                                         (8, (1, 2, 3), (1, 2, 2))])
def test_trilinear_upsampling_matches_the_framework(ops, C, sp, scale):
    """effq_upsample_trilinear (channels-last) against nn.Upsample(scale_factor, mode='trilinear') of the reference's
    decoder: same source-index and interpolation arithmetic in fp32 (<= 1e-6 of the value range; the framework's build
    may contract to FMAs)."""
    gen = torch.Generator().manual_seed(C)
    x = torch.randn(2, C, *sp, generator=gen)
    want = torch.nn.functional.interpolate(x, scale_factor=tuple(float(s) for s in scale), mode="trilinear")
    got = ops.upsample_trilinear(dev(_ndhwc(x)), scale).permute(0, 4, 1, 2, 3).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max() <= 1e-6 * want.abs().max()
