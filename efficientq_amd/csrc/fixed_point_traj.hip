// project_by_iter (layer_helper.py:40-70) for the weight projection INSIDE the ADMM loop (EfficientQConv.py:108), in ONE
// launch, using what the previous ADMM iteration's projection learnt.
//
// v_k = w*_k + dual_{k-1} differs little from v_{k-1}, so the i-th iterate of this call's fixed point lies within 1e-3
// (early) ... 1e-9 (late) of the i-th iterate of the previous call.  FptPred (fp_level.h) carries those iterates - seven
// single ones and the hull of the tail - each with two relative margins: a NARROW one (2.5 x the last drift) and a WIDE
// one (the envelope of the recent drifts: they are noisy).  The level of a value is monotone in the scale
// (fixed_point_bracket.hip), so:
//   phase 1, all workgroups, one pass: v = a + b is formed and stored; every value is classified, per slot j, at the
//     two ends of the wide bracket - equal levels: it adds (level * u, level, level^2) to the wide tally of slot j,
//     u = rint(v * 2^e) - and otherwise at the ends of the narrow one (equal -> ring tally of slot j and bit j of the
//     value's ring mask; different -> bit j of its narrow mask).  Values with a narrow bit go to the narrow list (a
//     fraction of a per cent), values with ring bits only to the ring list (a few per cent).  The unit 2^-e comes from
//     the PREVIOUS call's sum|v|: should |v| have grown so much that the tallies wrap (they are summed mod 2^64, in
//     unsigned arithmetic: no undefined behaviour), phase 2 notices (tallies_ok) and every iterate takes the full pass;
//   phase 2, the last workgroup to finish: sum|v| (partials in workgroup order: deterministic) gives alpha_0; iterate i
//     inside its narrow bracket costs a scan of the narrow list (held in registers) on top of three tallies; inside the
//     wide bracket only, a scan of both lists; outside both, a pass of this one workgroup over all of v (slow, rare:
//     the callers run the first calls of a layer and the calls after a change of rho on the older kernels, which
//     record their iterates).
// Levels are exactly the reference's for every iterate; the sums are integers (exact for |v| >= 4e-5 of the mean at
// 2^21 values, otherwise rounded to 2^-e): run-to-run deterministic, alpha within ~1e-15 of the fp64 kernels'.
// levels <= 16 (packed level counts), n <= 2^23.
#include "common.h"
#include "fp_level.h"
#include "fp_traj.h"

namespace effq {

// the K slices of the prox product, not yet added up: w*[r][k] = sum_z part[z * slab + r * ldp + k] in slice order
// (k_prox_reduce4's order: same bits); b*[r] = the column nwrow of row r.  The trajectory kernel adds them in its
// prologue - one launch per ADMM iteration less, and w* is written once instead of written, read and written.
struct FptParts {
  const float* part;
  size_t slab;            // floats per slice (c2 * ldp)
  int nsplit, ldp, nwrow, c2;
  float* wstar;           // [c2][nwrow] out
  float* bstar;           // [c2] out, or NULL (no bias)
};

template <bool PARTS>
__global__ __launch_bounds__(FPT_T) void k_fpt(const float* __restrict__ a, const float* __restrict__ b2,
                                               float* __restrict__ v_out, size_t n, FptWs w, FptPred* pred,
                                               effq_fp_state* st, double lo, double hi, double d, int levels, double tol,
                                               int max_iter, const FptParts pp) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams
  __shared__ FptSmem sm;
  const int tid = threadIdx.x, wg = blockIdx.x;
  FptCtx cx;
  cx.tr0 = wall_clock64();
  // the operands first: their loads are in flight while the predictions are fetched and the bracket ends set up
  float va[FPT_EPT], vb[FPT_EPT], vv[FPT_EPT];
  bool ok[FPT_EPT];
  const size_t base = (size_t)wg * FPT_WGV + tid;
  if (PARTS) {
    size_t off[FPT_EPT];
#pragma unroll
    for (int e = 0; e < FPT_EPT; ++e) {
      const size_t i = base + (size_t)e * FPT_T;
      const unsigned ic = (unsigned)((i < n) ? i : (n - 1));
      const unsigned r = ic / (unsigned)pp.nwrow;
      ok[e] = i < n;
      off[e] = (size_t)r * pp.ldp + (ic - r * (unsigned)pp.nwrow);
      vb[e] = b2[ic];
      va[e] = pp.part[off[e]];
    }
    for (int z = 1; z < pp.nsplit; ++z) {        // slice order: deterministic, k_prox_reduce4's sums
      float t[FPT_EPT];
#pragma unroll
      for (int e = 0; e < FPT_EPT; ++e) t[e] = pp.part[(size_t)z * pp.slab + off[e]];
#pragma unroll
      for (int e = 0; e < FPT_EPT; ++e) va[e] += t[e];
    }
#pragma unroll
    for (int e = 0; e < FPT_EPT; ++e)
      if (ok[e]) pp.wstar[base + (size_t)e * FPT_T] = va[e];
    if (wg == 0 && pp.bstar != nullptr && tid < pp.c2) {
      float t = pp.part[(size_t)tid * pp.ldp + pp.nwrow];
      for (int z = 1; z < pp.nsplit; ++z) t += pp.part[(size_t)z * pp.slab + (size_t)tid * pp.ldp + pp.nwrow];
      pp.bstar[tid] = t;
    }
  } else {
#pragma unroll
    for (int e = 0; e < FPT_EPT; ++e) {
      const size_t i = base + (size_t)e * FPT_T;
      const size_t ic = (i < n) ? i : (n - 1);
      ok[e] = i < n;
      va[e] = a[ic];
      vb[e] = (b2 != nullptr) ? b2[ic] : 0.0f;
    }
  }
#pragma unroll
  for (int e = 0; e < FPT_EPT; ++e) {
    vv[e] = (b2 != nullptr) ? (va[e] + vb[e]) : va[e];
    if (ok[e] && v_out != nullptr) v_out[base + (size_t)e * FPT_T] = vv[e];
  }
  fpt_phase1<FPT_EPT>(sm, vv, ok, wg, w, pred, lo, hi, d, cx);
  // hand-off to the last workgroup: release / ticket / acquire (cdna_hip_programming.md Guideline 16)
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(&w.hdr->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sm.last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!sm.last) return;
  cx.tr1 = wall_clock64();
  __builtin_amdgcn_s_setprio(3);
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  const bool vec_ok = (v_out != nullptr) && ((reinterpret_cast<uintptr_t>(v_out) & 15u) == 0);
  fpt_phase2<FPT_CR, FPT_FC, FPT_PF>(sm, cx, w, pred, st, (v_out != nullptr) ? v_out : (PARTS ? pp.wstar : a), vec_ok, n, (int)gridDim.x, levels, tol, max_iter, nullptr,
             nullptr);
}

}  // namespace effq
using namespace effq;

extern "C" {

size_t effq_fp_traj_max(void) { return FPT_MAXN; }
size_t effq_fp_traj_ws_bytes(size_t n) { return fpt_ws_bytes(n); }
size_t effq_fp_traj_pred_bytes(void) { return sizeof(FptPred); }

int effq_fixed_point_traj(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                          double tol, int max_iter, effq_fp_state* state_dev, void* pred_dev, void* ws, size_t ws_bytes,
                          void* stream) {
  EFFQ_CHECK_ARG(a && state_dev && pred_dev && ws && n > 0 && levels >= 2 && levels <= 16 && hi > lo && max_iter > 0);
  EFFQ_CHECK_ARG(n <= FPT_MAXN);
  EFFQ_CHECK_ARG(b == nullptr || v_out != nullptr);
  // k_fpt declares its operands __restrict__ and its last workgroup re-reads v_out (or a) after other workgroups wrote it
  EFFQ_CHECK_ARG(v_out == nullptr || (v_out != a && v_out != b));
  if (ws_bytes < fpt_ws_bytes(n)) {
    set_error("fixed_point_traj: workspace %zu < %zu bytes", ws_bytes, fpt_ws_bytes(n));
    return EFFQ_ERR_WORKSPACE;
  }
  const double d = (hi - lo) / (double)(levels - 1);
  FptParts none;
  memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(k_fpt<false>, dim3((unsigned)fpt_groups(n)), dim3(FPT_T), 0, as_stream(stream), a, b, v_out, n,
                     fpt_carve(ws, n), reinterpret_cast<FptPred*>(pred_dev), state_dev, lo, hi, d, levels, tol, max_iter, none);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

// internal (admm_run.hip): the same on the K slices of the prox product (effq_prox_solve_prebuilt_parts): w* = their sum is
// written to wstar_out ([c2][nwrow]), b* to bstar_out, v = w* + dual to v_out
int effq_fixed_point_traj_parts(const float* part, int nsplit, int ldp, int c2, int nwrow, int has_bias, const float* dual,
                                float* wstar_out, float* bstar_out, float* v_out, int levels, double lo, double hi, double tol,
                                int max_iter, effq_fp_state* state_dev, void* pred_dev, void* ws, size_t ws_bytes,
                                void* stream) {
  const size_t n = (size_t)c2 * (size_t)nwrow;
  EFFQ_CHECK_ARG(part && dual && wstar_out && v_out && state_dev && pred_dev && ws && nsplit >= 1 && c2 > 0 && nwrow > 0);
  EFFQ_CHECK_ARG(ldp >= nwrow + (has_bias ? 1 : 0) && levels >= 2 && levels <= 16 && hi > lo && max_iter > 0 && n <= FPT_MAXN);
  EFFQ_CHECK_ARG((!has_bias || bstar_out != nullptr) && c2 <= FPT_T && v_out != wstar_out && v_out != dual);
  if (ws_bytes < fpt_ws_bytes(n)) {
    set_error("fixed_point_traj: workspace %zu < %zu bytes", ws_bytes, fpt_ws_bytes(n));
    return EFFQ_ERR_WORKSPACE;
  }
  FptParts pp;
  pp.part = part; pp.slab = (size_t)c2 * (size_t)ldp; pp.nsplit = nsplit; pp.ldp = ldp; pp.nwrow = nwrow; pp.c2 = c2;
  pp.wstar = wstar_out; pp.bstar = has_bias ? bstar_out : nullptr;
  const double d = (hi - lo) / (double)(levels - 1);
  hipLaunchKernelGGL(k_fpt<true>, dim3((unsigned)fpt_groups(n)), dim3(FPT_T), 0, as_stream(stream),
                     static_cast<const float*>(nullptr), dual, v_out, n, fpt_carve(ws, n),
                     reinterpret_cast<FptPred*>(pred_dev), state_dev, lo, hi, d, levels, tol, max_iter, pp);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
