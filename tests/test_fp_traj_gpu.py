"""effq_fixed_point_traj (the weight projection of an ADMM iteration from the previous iteration's iterates,
EfficientQConv.py:108 -> layer_helper.py:40-70) against the oracle and the older fixed points: same iterates, same
iteration count, whatever the predictions are worth."""
import pytest
import torch

from oracle import effq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from efficientq_amd.hip_ops import get_ops
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return get_ops("cuda:0")


def dev(t):
    return t.to("cuda:0")


def _old(ops, w, du, L):
    """The fixed point the loop used before: bucketed up to 2^19 values, cooperative above."""
    st, v = ops.new_fp_state(), torch.empty_like(w)
    if w.numel() <= (1 << 19):
        ops.fixed_point_bucket(w, du, v, L, st)
    else:
        assert ops.weight_fixed_point(w, du, v, L, st) is None
    a, it, done = ops.read_fp_state(st)
    assert done == 1
    return a, it, v


def _clustered(n, gen, alpha=0.07, spread=0.15):
    """Weights as the ADMM loop sees them: clustered around the four levels of a scale."""
    lv = torch.tensor([-1.0, -1 / 3, 1 / 3, 1.0])[torch.randint(0, 4, (n,), generator=gen)]
    return (alpha * (lv + spread * torch.randn(n, generator=gen))).float()


@pytest.mark.parametrize("n", [16384, 27648, 110592, 442368, 1769472, 1769473 + 4096])
@pytest.mark.parametrize("L", [4, 16])
def test_cold_call_equals_the_oracle(ops, n, L):
    if L == 16 and n > 500000:
        pytest.skip("covered at 4 levels")
    gen = torch.Generator().manual_seed(n + L)
    w = torch.randn(n, generator=gen) * 0.05
    du = torch.randn(n, generator=gen) * 0.01
    fit = O.fit_scale(w + du, L, -1, 1)
    st, v, pred = ops.new_fp_state(), torch.empty(n, device="cuda:0"), ops.new_fp_pred()
    ops.fixed_point_traj(dev(w), dev(du), v, L, st, pred)
    a, it, done = ops.read_fp_state(st)
    assert torch.equal(v.cpu(), w + du)
    assert done == 1 and it == fit.iters and abs(a - fit.alpha) <= 1e-11 * fit.alpha
    p = ops.read_fp_pred(pred)
    assert p["K"] == min(it, 8) and p["calls"] == 1 and p["full_iters"] == it and p["warm_iters"] == 0


@pytest.mark.parametrize("n,L", [(27648, 4), (110592, 4), (442368, 4), (1769472, 4), (110592, 16)])
def test_a_drifting_sequence_stays_exact_and_turns_warm(ops, n, L):
    """Forty calls on a slowly changing tensor (what consecutive ADMM iterations look like): every call equals the older
    fixed point on the same values; after the first one nearly every iterate is served from the tallies and the list."""
    gen = torch.Generator().manual_seed(3 * n + L)
    base = dev(_clustered(n, gen))
    du = dev(torch.randn(n, generator=gen) * 0.002)
    noise = dev(torch.randn(n, generator=gen))
    pred, st = ops.new_fp_pred(), ops.new_fp_state()
    v = torch.empty_like(base)
    for k in range(40):
        step = 2e-3 * (0.85 ** k)                      # relative change per call: 2e-3 ... 3e-6
        w = base * (1.0 + step) + 0.07 * step * noise
        if k == 25:
            w = w * 1.08                               # a jump (rho doubles): the predictions are off for one call
        ops.fixed_point_traj(w, du, v, L, st, pred)
        a, it, done = ops.read_fp_state(st)
        a0, it0, v0 = _old(ops, w, du, L)
        assert torch.equal(v, v0)
        assert done == 1 and it == it0, (k, it, it0)
        assert abs(a - a0) <= 1e-13 * a0, (k, a, a0)
        base = w / (1.08 if k == 25 else 1.0)
    p = ops.read_fp_pred(pred)
    assert p["calls"] == 40
    assert p["warm_iters"] >= 0.8 * (p["warm_iters"] + p["full_iters"]), p
    assert p["list_max"] <= (0.5 if L == 4 else 1.0) * n, p       # (the call after the jump lists the most)


def test_recorded_iterates_of_the_older_kernels_warm_the_next_call(ops):
    for n in (110592, 1769472):
        gen = torch.Generator().manual_seed(n)
        w = dev(_clustered(n, gen))
        du = dev(torch.randn(n, generator=gen) * 0.002)
        pred, st, v = ops.new_fp_pred(), ops.new_fp_state(), torch.empty(n, device="cuda:0")
        if n <= (1 << 19):
            ops.fixed_point_bucket_rec(w, du, v, 4, st, pred)
        else:
            ops.fixed_point_coop_rec(w, du, v, 4, st, pred)
        a0, it0, _ = ops.read_fp_state(st)
        p0 = ops.read_fp_pred(pred)
        assert p0["K"] == min(it0, 8) and p0["calls"] == 1
        w2 = w * 1.0005
        ops.fixed_point_traj(w2, du, v, 4, st, pred)
        a, it, done = ops.read_fp_state(st)
        a1, it1, _ = _old(ops, w2, du, 4)
        assert done == 1 and it == it1 and abs(a - a1) <= 1e-13 * a1
        p = ops.read_fp_pred(pred)
        assert p["full_iters"] == 0 and p["warm_iters"] == it, p


def test_run_to_run_identical_bits_and_wrong_predictions_are_harmless(ops):
    n = 442368
    gen = torch.Generator().manual_seed(5)
    w = dev(_clustered(n, gen))
    du = dev(torch.randn(n, generator=gen) * 0.002)
    v = torch.empty_like(w)
    seed, st = ops.new_fp_pred(), ops.new_fp_state()
    ops.fixed_point_traj(w, du, v, 4, st, seed)                     # cold: fills the predictions
    want = ops.read_fp_state(st)
    outs = []
    for _ in range(3):
        pred = seed.clone()
        ops.fixed_point_traj(w, du, v, 4, st, pred)                 # warm, same predictions every time
        outs.append(ops.read_fp_state(st))
    assert outs[0] == outs[1] == outs[2]
    assert outs[0][1] == want[1] and abs(outs[0][0] - want[0]) <= 1e-13 * want[0]
    # predictions of another tensor altogether (scale off by 3): every iterate falls back to a full pass
    pred = seed.clone()
    ops.fixed_point_traj(w * 3.0, du, v, 4, st, pred)
    a, it, done = ops.read_fp_state(st)
    fit = O.fit_scale((w * 3.0 + du).cpu(), 4, -1, 1)
    assert done == 1 and it == fit.iters and abs(a - fit.alpha) <= 1e-11 * fit.alpha


@pytest.mark.parametrize("scale", [8.0, 64.0, 1.0 / 64.0])
def test_predictions_of_a_tensor_of_another_magnitude_fall_back_to_the_full_pass(ops, scale):
    """phase 1 tallies level * u with the unit of the PREVIOUS call's sum|v|.  A tensor that grew 8 x / 64 x since
    (or shrank 64 x: every bracket misses) must cost nothing but speed: the tallies are summed mod 2^64 in unsigned
    arithmetic (no signed overflow), phase 2 discards them where they could have wrapped, every iterate takes the full
    pass, and the result is the older fixed point's."""
    n, L = 442368, 4
    gen = torch.Generator().manual_seed(99)
    base = dev(_clustered(n, gen))
    du = dev(torch.randn(n, generator=gen) * 0.002)
    pred, st = ops.new_fp_pred(), ops.new_fp_state()
    v = torch.empty_like(base)
    for _ in range(3):                                  # warm predictions on the unscaled tensor
        ops.fixed_point_traj(base, du, v, L, st, pred)
    before = ops.read_fp_pred(pred)
    w2, du2 = base * scale, du * scale
    ops.fixed_point_traj(w2, du2, v, L, st, pred)
    a, it, done = ops.read_fp_state(st)
    a_old, it_old, v_old = _old(ops, w2, du2, L)
    assert done == 1 and it == it_old and abs(a - a_old) <= 1e-13 * a_old, (a, a_old, it, it_old)
    assert torch.equal(v, v_old)
    after = ops.read_fp_pred(pred)
    assert after["full_iters"] - before["full_iters"] >= it - 1, (before, after)      # the predictions were worth nothing
    # and the next call on the scaled tensor is warm again
    ops.fixed_point_traj(w2, du2, v, L, st, pred)
    a3, it3, done3 = ops.read_fp_state(st)
    assert done3 == 1 and it3 == it_old and abs(a3 - a_old) <= 1e-13 * a_old


def test_aliased_output_is_rejected(ops):
    """k_fpt's operands are __restrict__ and its last workgroup re-reads v: v_out must not alias a or b."""
    from efficientq_amd._lib import EffqError
    w = dev(torch.randn(32768) * 0.05)
    du = dev(torch.randn(32768) * 0.01)
    with pytest.raises(EffqError):
        ops.fixed_point_traj(w, du, w, 4, ops.new_fp_state(), ops.new_fp_pred())


@pytest.mark.parametrize("c2,nwrow,has_b,nsplit", [(64, 1728, 1, 16), (128, 3456, 1, 9), (32, 2048, 0, 3)])
def test_k_slices_of_the_prox_product_added_up_in_the_prologue(ops, c2, nwrow, has_b, nsplit):
    """effq_fixed_point_traj_parts (internal: what effq_admm_run calls instead of k_prox_reduce4 + effq_fixed_point_traj): the
    K slices of the product summed in slice order inside the kernel - w*, b*, v and the scale bit-identical to the two
    launches, cold and warm."""
    import ctypes as C
    from efficientq_amd.hip_ops import ADMM_TOL
    lib = ops.lib
    fn = lib.effq_fixed_point_traj_parts
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                   C.c_void_p]
    gen = torch.Generator().manual_seed(c2 + nwrow)
    ldp = ((nwrow + has_b + 31) // 32) * 32
    n = c2 * nwrow
    w = _clustered(n, gen).reshape(c2, nwrow)
    du = dev(torch.randn(n, generator=gen) * 0.002)
    pred_a, pred_b = ops.new_fp_pred(), ops.new_fp_pred()
    for rep in range(3):                                   # cold, then warm on a drifting tensor
        w = w * (1.0 + 2e-4 * rep) + 1e-5 * torch.randn(c2, nwrow, generator=gen)
        full = torch.zeros(c2, ldp)
        full[:, :nwrow] = w
        if has_b:
            full[:, nwrow] = torch.randn(c2, generator=gen)
        # slices that add up to `full` in slice order only approximately - the REFERENCE is their sum in that order
        parts = torch.randn(nsplit, c2, ldp, generator=gen) * 0.01
        parts[0] = full - parts[1:].sum(0)
        parts = dev(parts.contiguous())
        acc = parts[0].clone()
        for z in range(1, nsplit):
            acc = acc + parts[z]
        w_sum, b_sum = acc[:, :nwrow].contiguous().reshape(-1), (acc[:, nwrow].contiguous() if has_b else None)
        st_a, st_b = ops.new_fp_state(), ops.new_fp_state()
        v_a, v_b = torch.empty(n, device="cuda:0"), torch.empty(n, device="cuda:0")
        ops.fixed_point_traj(w_sum, du, v_a, 4, st_a, pred_a)
        ws = ops._workspace("fp_traj", lib.effq_fp_traj_ws_bytes(n))
        wst, bst = torch.empty(n, device="cuda:0"), torch.empty(c2, device="cuda:0")
        rc = fn(parts.data_ptr(), nsplit, ldp, c2, nwrow, has_b, du.data_ptr(), wst.data_ptr(), bst.data_ptr(), v_b.data_ptr(),
                4, -1.0, 1.0, ADMM_TOL, 400, st_b.data_ptr(), pred_b.data_ptr(), ws.data_ptr(), ws.numel(), ops.stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(wst, w_sum) and torch.equal(v_a, v_b)
        if has_b:
            assert torch.equal(bst, b_sum)
        assert ops.read_fp_state(st_a) == ops.read_fp_state(st_b)
