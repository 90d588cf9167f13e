"""The bracketed fixed point (effq_fp_bracket_*: project_by_iter on activation-sized tensors, layer_helper.py:40-70)
against the reference goldens, the oracle and the per-iteration-pass kernels: same iterates, same iteration count."""
import ctypes as C

import pytest
import torch

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


class _Forced:
    """fit_scale routed through the bracketed path (min size 1) or kept off it."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        import efficientq_amd.hip_ops as H
        self.H, self.old = H, H.FP_BRACKET_MIN
        H.FP_BRACKET_MIN = 1 if self.on else 1 << 62

    def __exit__(self, *a):
        self.H.FP_BRACKET_MIN = self.old


@pytest.mark.parametrize("L", [4, 16, 256])
def test_bracketed_fit_matches_reference_goldens(ops, gold, L):
    g = gold("g2_project.npz")
    for name, lo, hi in (("act", 0.0, 1.0), ("wgt", -1.0, 1.0)):
        v = T(g[name])
        with _Forced(True):
            alpha, iters, st = ops.fit_scale(dev(v), L, lo, hi)
        want_a = float(g[f"{name}_L{L}_alpha"])
        assert abs(alpha - want_a) <= 1e-11 * abs(want_a), (alpha, want_a)
        assert iters == int(g[f"{name}_L{L}_iters"])
        _, _, idx = ops.quant_dequant_f64path(dev(v), st, L, lo, hi, want_idx=True)
        assert torch.equal(idx.cpu(), T(g[f"{name}_L{L}_idx"]))


def _cases():
    gen = torch.Generator().manual_seed(123)
    relu = torch.relu(torch.randn(4_000_003, generator=gen) * 1.7 + 0.2)           # ragged length, half zeros
    signed = torch.randn(3_000_000, generator=gen) * 0.3 + 0.05
    heavy = torch.relu(torch.randn(2_500_000, generator=gen)) ** 3                # heavy tail: slow convergence
    sparse = torch.zeros(1_200_000)
    sparse[::977] = torch.rand(sparse[::977].numel(), generator=gen) + 0.5        # almost everything is zero
    return [("relu", relu, 0.0, 1.0), ("signed", signed, -1.0, 1.0), ("heavy", heavy, 0.0, 1.0),
            ("sparse", sparse, 0.0, 1.0)]


@pytest.mark.parametrize("L", [4, 16])
def test_bracketed_fit_equals_the_per_iteration_passes(ops, L):
    for name, v, lo, hi in _cases():
        x = dev(v)
        with _Forced(False):
            a0, it0, _ = ops.fit_scale(x, L, lo, hi)
        with _Forced(True):
            a1, it1, _ = ops.fit_scale(x, L, lo, hi)
            dg = ops.fp_bracket_diagnostics()
            a2, it2, _ = ops.fit_scale(x, L, lo, hi, reducer=lambda t: t)          # stats / update halves
            a3, it3, _ = ops.fit_scale(x, L, lo, hi)                               # run to run
        assert it1 == it0, (name, L, it0, it1)
        assert abs(a1 - a0) <= 1e-12 * abs(a0), (name, L, a0, a1)
        assert a2 == a1 and it2 == it1 and a3 == a1 and it3 == it1                # integer sums: deterministic
        # the point of it: far fewer values read than iterations x n
        if it0 >= 12:
            assert dg["visited"] <= 0.5 * it0 * v.numel(), (name, L, dg, it0)
        assert dg["widen"] <= 2, (name, L, dg)          # escapes from a bracket that was meant to hold the limit


def test_bracketed_fit_against_the_oracle_on_an_unaligned_view(ops):
    gen = torch.Generator().manual_seed(5)
    base = dev(torch.relu(torch.randn(300_001, generator=gen)))
    x = base[1:]                                                                   # 4-byte aligned only
    assert x.data_ptr() % 16 != 0
    with _Forced(True):
        a, it, _ = ops.fit_scale(x, 4, 0.0, 1.0)
    fit = O.fit_scale(x.cpu(), 4, 0, 1)
    assert it == fit.iters and abs(a - fit.alpha) <= 1e-11 * fit.alpha


def test_bracketed_fit_adversarial_values(ops):
    """Values on level boundaries of the converged scale, duplicates, tiny and huge magnitudes."""
    gen = torch.Generator().manual_seed(9)
    base = torch.relu(torch.randn(500_000, generator=gen))
    fit = O.fit_scale(base, 4, 0, 1)
    a = fit.alpha
    edge = torch.tensor([a / 6, a / 2, 5 * a / 6, a], dtype=torch.float64).float()
    nudged = torch.cat([edge, torch.nextafter(edge, torch.tensor(0.0)), torch.nextafter(edge, torch.tensor(9.0))])
    for extra in (nudged.repeat(4000), torch.full((50_000,), 1e-30), torch.tensor([1e6, 3e5] * 10)):
        v = torch.cat([base, extra])
        want = O.fit_scale(v, 4, 0, 1)
        with _Forced(True):
            got_a, got_it, _ = ops.fit_scale(dev(v), 4, 0.0, 1.0)
        assert got_it == want.iters and abs(got_a - want.alpha) <= 1e-11 * want.alpha
    const = torch.full((400_000,), 0.37)                                          # one distinct value
    want = O.fit_scale(const, 16, 0, 1)
    with _Forced(True):
        got_a, got_it, _ = ops.fit_scale(dev(const), 16, 0.0, 1.0)
    assert got_it == want.iters and abs(got_a - want.alpha) <= 1e-11 * want.alpha


def test_bracketed_fit_sharded_over_two_ranks_equals_unsharded(ops):
    """Two data-parallel ranks emulated on one device with the C entry points: each holds half of the tensor, its own
    state and workspace; the two sums are added between the stats and update halves (what SumReducer does)."""
    from efficientq_amd.hip_ops import ADMM_TOL, _ptr
    from efficientq_amd._lib import check
    gen = torch.Generator().manual_seed(77)
    v = torch.relu(torch.randn(1_500_000, generator=gen) + 0.3)
    L, lo, hi = 4, 0.0, 1.0
    with _Forced(True):
        a_all, it_all, _ = ops.fit_scale(dev(v), L, lo, hi)
    cut = 611_111
    parts = [dev(v[:cut].clone()), dev(v[cut:].clone())]
    lib, stream = ops.lib, ops.stream
    s0 = [ops.abs_sum(p) for p in parts]
    tot = s0[0] + s0[1]
    sts = [ops.new_fp_state() for _ in parts]
    wss = [torch.zeros(lib.effq_fp_bracket_ws_bytes(p.numel()), dtype=torch.uint8, device="cuda:0") for p in parts]
    for p, st, ws in zip(parts, sts, wss):
        check(lib.effq_fp_bracket_init(_ptr(st), _ptr(tot), p.numel(), L, 1, _ptr(ws), ws.numel(), stream), "init")
    for _ in range(100 * L):
        for p, st, ws in zip(parts, sts, wss):
            check(lib.effq_fp_bracket_stats(_ptr(p), p.numel(), L, lo, hi, _ptr(st), _ptr(ws), stream), "stats")
        both = sts[0][2:4] + sts[1][2:4]
        for p, st, ws in zip(parts, sts, wss):
            st[2:4] = both
            check(lib.effq_fp_bracket_update(p.numel(), L, lo, hi, ADMM_TOL, 100 * L, _ptr(st), _ptr(ws), stream), "update")
        if ops.read_fp_state(sts[0])[2] != 0:
            break
    a0, it0, d0 = ops.read_fp_state(sts[0])
    a1, it1, d1 = ops.read_fp_state(sts[1])
    assert d0 == 1 and d1 == 1 and a0 == a1 and it0 == it1                         # ranks in lock step
    assert it0 == it_all and abs(a0 - a_all) <= 1e-13 * a_all


def test_bracketed_fit_raises_at_the_cap(ops):
    import efficientq_amd.hip_ops as H
    v = torch.randn(300_000, generator=torch.Generator().manual_seed(1))
    old = H.ADMM_TOL
    H.ADMM_TOL = -1.0
    try:
        with _Forced(True), pytest.raises(RuntimeWarning):
            ops.fit_scale(dev(v), 4, -1.0, 1.0)
    finally:
        H.ADMM_TOL = old


@pytest.mark.parametrize("L,after", [(4, 4), (16, 6), (4, 2)])
def test_gather_once_two_emulated_ranks_finish_on_their_own(ops, L, after):
    """Data-parallel "gather once" (effq_fp_bracket_export / _import): two ranks emulated on one device run `after`
    all-reduced iterations, exchange their four integer tallies (summed) and their undecided lists (zero-padded, rank-major),
    and EACH finishes the fit alone on the gathered list: same alpha and iteration count as the unsharded fit (1e-13), the
    two ranks bit-identical, no collective after the exchange.  No exchange while there is no list or while the bracket is
    a horizon (16 levels: export says -2); an imported fit whose iterates leave its bracket ends with state.done = 4 and the
    ranks go on with all-reduced iterations - the loop below is the driver's (hip_ops._fit_scale_gathered)."""
    from efficientq_amd.hip_ops import ADMM_TOL, _ptr
    from efficientq_amd._lib import check
    gen = torch.Generator().manual_seed(78 + L)
    v = torch.relu(torch.randn(1_200_000, generator=gen) + 0.2)
    lo, hi = 0.0, 1.0
    with _Forced(True):
        a_all, it_all, _ = ops.fit_scale(dev(v), L, lo, hi)
    cut = 500_003
    parts = [dev(v[:cut].clone()), dev(v[cut:].clone())]
    lib, stream = ops.lib, ops.stream
    tot = ops.abs_sum(parts[0]) + ops.abs_sum(parts[1])
    sts = [ops.new_fp_state() for _ in parts]
    wss = [torch.zeros(lib.effq_fp_bracket_ws_bytes(p.numel()), dtype=torch.uint8, device="cuda:0") for p in parts]
    for p, st, ws in zip(parts, sts, wss):
        check(lib.effq_fp_bracket_init(_ptr(st), _ptr(tot), p.numel(), L, 1, _ptr(ws), ws.numel(), stream), "init")
    exchanges, reduced_iters, res = 0, 0, None
    words = lib.effq_fp_bracket_export_words()
    slot = 1 << 20                                                 # floats per rank in the fixed-size all-gather
    for _round in range(200):
        for _ in range(after):                                     # all-reduced iterations
            for p, st, ws in zip(parts, sts, wss):
                check(lib.effq_fp_bracket_stats(_ptr(p), p.numel(), L, lo, hi, _ptr(st), _ptr(ws), stream), "stats")
            both = sts[0][2:4] + sts[1][2:4]
            for p, st, ws in zip(parts, sts, wss):
                st[2:4] = both
                check(lib.effq_fp_bracket_update(p.numel(), L, lo, hi, ADMM_TOL, 100 * L, _ptr(st), _ptr(ws), stream), "update")
            reduced_iters += 1
        if ops.read_fp_state(sts[0])[2] != 0:
            res = [ops.read_fp_state(st) for st in sts]
            break
        exps = [torch.zeros(words, dtype=torch.int64, device="cuda:0") for _ in parts]
        lists = [torch.full((slot,), 7.0, device="cuda:0") for _ in parts]       # (garbage the export must overwrite)
        for p, ws, ex, ls in zip(parts, wss, exps, lists):
            check(lib.effq_fp_bracket_export(_ptr(ws), p.numel(), _ptr(ex), _ptr(ls), slot, stream), "export")
        pack = torch.zeros(4 + 3 * 2, dtype=torch.int64, device="cuda:0")        # what the all-reduce leaves on every rank
        pack[:4] = exps[0][:4] + exps[1][:4]
        pack[4:7], pack[7:10] = exps[0][4:7], exps[1][4:7]
        gathered = torch.cat(lists)                                               # what the all-gather leaves on every rank
        lens = [int(ex[4].item()) for ex in exps]
        for ls, k in zip(lists, lens):
            assert not ls[max(k, 0):].any() if 0 <= k <= slot else not ls.any()  # zero-filled behind the list
        res = []
        for p, st, ws in zip(parts, sts, wss):                     # each rank on its own from here
            ws2 = torch.zeros(lib.effq_fp_bracket_ws_bytes(gathered.numel()), dtype=torch.uint8, device="cuda:0")
            check(lib.effq_fp_bracket_import(_ptr(ws), p.numel(), _ptr(pack), 2, slot, _ptr(st), _ptr(ws2), ws2.numel(),
                                             stream), "import")
            for _ in range(40):
                check(lib.effq_fp_bracket_run(_ptr(gathered), gathered.numel(), L, lo, hi, ADMM_TOL, 100 * L, 16, _ptr(st),
                                              _ptr(ws2), stream), "run")
                if ops.read_fp_state(st)[2] != 0:
                    break
            res.append(ops.read_fp_state(st))
        assert res[0] == res[1], res                               # bit-identical ranks, usable exchange or not
        if res[0][2] == 4:                                         # not usable / left its bracket: back to all-reduced iterations
            for p, st, ws in zip(parts, sts, wss):
                check(lib.effq_fp_bracket_rebase(_ptr(st), _ptr(ws), p.numel(), stream), "rebase")
            continue
        exchanges += 1
        break
    a, it, done = res[0]
    assert res[0] == res[1] and done == 1 and it == it_all and abs(a - a_all) <= 1e-13 * a_all, (res, a_all, it_all)
    print(f"L={L}: {it} iterations, {reduced_iters} of them all-reduced, {exchanges} exchange(s)")
    if L == 4:
        assert exchanges == 1 and reduced_iters <= 2 * after, (exchanges, reduced_iters)
    # an exchange whose brackets do not contain the iterate: refused on the device
    st = ops.new_fp_state()
    ws = torch.zeros(lib.effq_fp_bracket_ws_bytes(parts[0].numel()), dtype=torch.uint8, device="cuda:0")
    check(lib.effq_fp_bracket_init(_ptr(st), _ptr(tot), parts[0].numel(), L, 1, _ptr(ws), ws.numel(), stream), "init")
    for _ in range(3):
        check(lib.effq_fp_bracket_run(_ptr(parts[0]), parts[0].numel(), L, lo, hi, ADMM_TOL, 100 * L, 1, _ptr(st), _ptr(ws),
                                      stream), "run")
    pack = torch.zeros(7, dtype=torch.int64, device="cuda:0")
    pack[4] = 10
    pack[5:7] = torch.tensor([1e-9, 2e-9], dtype=torch.float64).view(torch.int64).to("cuda:0")
    some = torch.rand(4096, device="cuda:0")
    ws2 = torch.zeros(lib.effq_fp_bracket_ws_bytes(some.numel()), dtype=torch.uint8, device="cuda:0")
    check(lib.effq_fp_bracket_import(_ptr(ws), parts[0].numel(), _ptr(pack), 1, some.numel(), _ptr(st), _ptr(ws2), ws2.numel(),
                                     stream), "import")
    a_before = ops.read_fp_state(st)
    check(lib.effq_fp_bracket_run(_ptr(some), some.numel(), L, lo, hi, ADMM_TOL, 100 * L, 4, _ptr(st), _ptr(ws2), stream), "run")
    assert ops.read_fp_state(st) == (a_before[0], a_before[1], 4)               # refused: nothing ran


def test_fit_scale_with_a_one_rank_reducer_takes_the_gather_path(ops):
    """ops.fit_scale with a reducer that offers all_gather (what SumReducer is): world size 1, identity collectives - the
    gather-once driver must return the plain fit's result, with a handful of collectives instead of one per iteration."""
    class Red:
        world, rank, calls = 1, 0, 0

        def __call__(self, t):
            Red.calls += 1
            return t

        def all_gather(self, t):
            Red.calls += 1
            return t.clone()
    gen = torch.Generator().manual_seed(5)
    x = dev(torch.relu(torch.randn(2_000_000, generator=gen)))
    for L in (4, 16):
        with _Forced(True):
            a0, it0, _ = ops.fit_scale(x, L, 0.0, 1.0)
            Red.calls = 0
            a1, it1, _ = ops.fit_scale(x, L, 0.0, 1.0, reducer=Red())
        assert it1 == it0 and abs(a1 - a0) <= 1e-13 * a0, (L, a0, a1, it0, it1)
        print(f"L={L}: {it0} iterations, {Red.calls} collectives")
        if L == 4:
            assert Red.calls <= 16 < it0, (L, Red.calls, it0)
        else:               # 16 levels: horizon brackets for most of the fit (nothing to exchange), then the endgame
            assert Red.calls < it0, (L, Red.calls, it0)
