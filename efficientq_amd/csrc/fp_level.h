// Level index of the reference quantiser (layer_helper.py:25-37 evaluated in fp64) with an fp32 screen, shared by the
// fixed points that classify values at several scales per pass (fixed_point_bracket.hip, fixed_point_traj.hip).
#pragma once
#include <math.h>

namespace effq {

// u = (v / a - lo) / d evaluated in fp32 is off by <= 3e-5 at 256 levels; it is accepted unless it lies within 2e-4 of a
// rounding boundary, where the reference's own arithmetic (IEEE fp64 divisions, round-half-even) decides: the level is
// exactly the reference's (the screen of level_accum in quant_reduce.hip).
struct FpLevel {
  float c1, c0, lmax;
  double a;
};
__device__ __forceinline__ FpLevel fp_level_consts(double a, double lo, double hi, double d) {
  FpLevel c;
  const double rd = 1.0 / d;
  c.c1 = (float)((1.0 / a) * rd);
  c.c0 = (float)(-lo * rd);
  c.lmax = (float)rint((hi - lo) * rd);
  c.a = a;
  return c;
}
// (not inlined: the fallback is taken for ~4 values in 10 000, and its two IEEE divisions are ~100 instructions that
// would otherwise be copied into every unrolled call site)
__device__ __attribute__((noinline)) inline int fp_level_exact(float v, double a, double lo, double hi, double d) {
  double t = (double)v / a;
  t = fmin(fmax(t, lo), hi);
  return (int)rint((t - lo) / d);
}
__device__ __forceinline__ int fp_level(float v, const FpLevel& c, double lo, double hi, double d) {
  float u = __builtin_fmaf(v, c.c1, c.c0);
  u = fminf(fmaxf(u, 0.0f), c.lmax);
  const float rf = rintf(u);
  if (!(fabsf(u - rf) < 0.4998f)) return fp_level_exact(v, c.a, lo, hi, d);
  return (int)rf;
}

// ---- prediction of a fixed point's iterates from the previous call on (nearly) the same tensor -----------------------
// The weight projection of ADMM iteration k runs project_by_iter on v_k = w*_k + dual_{k-1}, which differs little from
// v_{k-1}: the i-th iterate of call k lies within ~1e-3 (early) ... 1e-9 (late) of the i-th iterate of call k - 1.
// Slots 0 .. FPT_SLOTS-2 hold one iterate each, the last slot the hull of all later ones and of the final scale.
constexpr int FPT_SLOTS = 8;
struct FptPred {
  int K;                      // valid slots (0 = nothing known: cold)
  int e;                      // exponent of the integer unit the tallies of the next call use (q = 2^-e)
  int e_valid, pad;
  double lo[FPT_SLOTS], hi[FPT_SLOTS];      // the iterate (lo = hi), or the hull of the tail
  double eps[FPT_SLOTS];                    // relative margin of the WIDE bracket around it (envelope of the recent drifts)
  double nlo[FPT_SLOTS], nhi[FPT_SLOTS];    // the same of the call in progress (becomes lo / hi at its end)
  long long calls, warm_iters, full_iters, listed, list_max;      // diagnostics
  double eps_n[FPT_SLOTS];                  // margin of the NARROW bracket (follows the last drift closely), <= eps
  long long ring_iters, ring_listed;        // diagnostics: iterates served by the wide bracket, entries of its list
  long long trace[8];                       // diagnostics: 100 MHz time stamps of the last workgroup of the last call
};
constexpr double FPT_EPS_MIN = 1e-4, FPT_EPS_MAX = 0.03, FPT_EPS_NEW = 0.01;

// Recorder, used by whichever kernel runs the fixed point: fpt_note(i, alpha_i) by ONE thread for every classification
// scale in order; then, after a barrier (or by the same thread), fpt_finish_slot for j = 0 .. FPT_SLOTS-1 (any threads)
// and fpt_finish_head once.  pred == nullptr: nothing is recorded.
__device__ __forceinline__ void fpt_note(FptPred* p, int i, double alpha) {
  if (p == nullptr) return;
  if (i < FPT_SLOTS) {
    p->nlo[i] = alpha;
    p->nhi[i] = alpha;
  } else {
    p->nlo[FPT_SLOTS - 1] = fmin(p->nlo[FPT_SLOTS - 1], alpha);
    p->nhi[FPT_SLOTS - 1] = fmax(p->nhi[FPT_SLOTS - 1], alpha);
  }
}
__device__ __forceinline__ void fpt_finish_slot(FptPred* p, int j, int iters, double alpha_final) {
  if (p == nullptr) return;
  const int K = (iters < FPT_SLOTS) ? iters : FPT_SLOTS;
  if (j >= K) return;
  double l = p->nlo[j], h = p->nhi[j];
  if (j == K - 1) {
    l = fmin(l, alpha_final);
    h = fmax(h, alpha_final);
  }
  double eps = FPT_EPS_NEW, eps_n = FPT_EPS_NEW;
  if (j < p->K && p->K <= FPT_SLOTS) {
    // the drift from call to call is noisy (ADMM iterates oscillate: factors of 5 - 10 between consecutive calls), so the
    // margin follows its recent MAXIMUM: 4 x the last drift, and never below 0.85 of the previous margin.  Replayed on
    // the oracle's iterates of a 32 -> 32 layer (tests/diagnostics/traj_policy_sim.py): 1.3 % of the iterates fall
    // outside their bracket, all of them in the calls right after the start or a change of rho (3 x drift alone: 9 %)
    const double drift = fmax(fabs(l - p->lo[j]), fabs(h - p->hi[j])) / fabs(h);
    eps = fmin(fmax(fmax(4.0 * drift, 0.85 * p->eps[j]), FPT_EPS_MIN), FPT_EPS_MAX);
    // ... while the narrow bracket bets on the next drift being like the last one: most iterates land in it, and its
    // list is a fraction of the wide one's; an iterate that lands between the two costs a scan of the wide list
    eps_n = fmin(fmax(2.5 * drift, FPT_EPS_MIN), eps);
  }
  p->lo[j] = l;
  p->hi[j] = h;
  p->eps[j] = eps;
  p->eps_n[j] = eps_n;
}
// (after every fpt_finish_slot of the call: they read the old K)
__device__ __forceinline__ void fpt_finish_head(FptPred* p, int iters, double sum_abs, int levels) {
  if (p == nullptr) return;
  p->K = (iters < FPT_SLOTS) ? iters : FPT_SLOTS;
  // unit of the next call's integer tallies: (levels - 1) * sum|v| * 2^e < 2^59 leaves a factor 4 for growth
  int e = 0;
  const double bound = (double)(levels - 1) * sum_abs;
  const bool ok = bound > 0.0 && bound < 1e300;
  if (ok) e = 58 - ilogb(bound);
  p->e = e;
  p->e_valid = ok ? 1 : 0;
  p->calls += 1;
}

}  // namespace effq
