// The projection + dual update of an ADMM iteration (EfficientQConv.py:108-111, 129-137) as device code shared by the
// stand-alone kernels (quant_reduce.hip) and by the single-workgroup fixed points that run it as their epilogue
// (k_fp_small, k_fps: one launch per iteration less on the layers whose weights fit one workgroup).
#pragma once
#include "common.h"

namespace effq {

__device__ __forceinline__ double disc64(double x, double alpha, double lo, double hi, double d, double* idx) {
  double t = x / alpha;
  t = fmin(fmax(t, lo), hi);
  double r = rint((t - lo) / d);
  *idx = r;
  return r * d + lo;
}

struct LevelConsts {
  float c1, c0, lmax;
  double alpha, lo, hi, d;
  bool need_sx;
};
__device__ __forceinline__ LevelConsts level_consts(double alpha, double lo, double hi, double d) {
  LevelConsts c;
  const double rd = 1.0 / d;
  c.c1 = (float)((1.0 / alpha) * rd);
  c.c0 = (float)(-lo * rd);
  c.lmax = (float)rint((hi - lo) * rd);
  c.alpha = alpha; c.lo = lo; c.hi = hi; c.d = d;
  c.need_sx = lo != 0.0;
  return c;
}

// Optional extra output of the projection: Bm = B0 + eta [W0|b0] + rho (G - dual) for the next iteration's prox solve,
// element for element what k_build_b4 (solve.hip) computes - one launch per ADMM iteration less.  The bias column and the
// zero padding of Bm do not depend on the iterate: they stay as the first build of the layer left them.
struct ProjNext {
  float* Bm;
  const float* B0;
  const float* W0;
  int nwrow, n, ldb;       // weights per output channel, row length of B0, row length of Bm
  float rho, eta;
};

// Four consecutive weights: 16-byte accesses, one (row, column) split with 32-bit arithmetic, and the level index from the
// fp32 evaluation of level_accum (the reference's fp64 arithmetic decides within 2e-4 of a rounding boundary: exact).
__device__ __forceinline__ void proj4_apply(unsigned q, const float* v, const float* wstar, double alpha, float alpha32,
                                            const LevelConsts& lc, double d, float* G, float* dual, float dual_div,
                                            int8_t* Gq, int lm1, const ProjNext& nx) {
  const size_t i = (size_t)q * 4;
  const float4 vv = *reinterpret_cast<const float4*>(v + i), ww = *reinterpret_cast<const float4*>(wstar + i),
               dd = *reinterpret_cast<const float4*>(dual + i);
  const float ve[4] = {vv.x, vv.y, vv.z, vv.w}, we[4] = {ww.x, ww.y, ww.z, ww.w}, de[4] = {dd.x, dd.y, dd.z, dd.w};
  float ge[4], du[4];
  int ri[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float u = __builtin_fmaf(ve[e], lc.c1, lc.c0);
    u = fminf(fmaxf(u, 0.0f), lc.lmax);
    float rf = rintf(u);
    if (!(fabsf(u - rf) < 0.4998f)) {
      double r;
      disc64((double)ve[e], alpha, -1.0, 1.0, d, &r);
      rf = (float)r;
    }
    ri[e] = (int)rf;
    const float b = (float)((double)rf * d + -1.0);       // disc64's r * d + lo
    ge[e] = alpha32 * b;
    float t = (we[e] - ge[e]) + de[e];                    // EfficientQConv.py:111
    if (dual_div != 1.0f) t = t / dual_div;               // "dual /= 2" or "dual /= rho_m/rho" (:131-136)
    du[e] = t;
  }
  *reinterpret_cast<float4*>(G + i) = make_float4(ge[0], ge[1], ge[2], ge[3]);
  *reinterpret_cast<float4*>(dual + i) = make_float4(du[0], du[1], du[2], du[3]);
  if (Gq != nullptr) {
    char4 c;
    if (lm1 >= 128) {
      c = make_char4((signed char)(ri[0] - 128), (signed char)(ri[1] - 128), (signed char)(ri[2] - 128), (signed char)(ri[3] - 128));
    } else {
      c = make_char4((signed char)(2 * ri[0] - lm1), (signed char)(2 * ri[1] - lm1), (signed char)(2 * ri[2] - lm1),
                     (signed char)(2 * ri[3] - lm1));
    }
    *reinterpret_cast<char4*>(Gq + i) = c;
  }
  if (nx.Bm != nullptr) {                       // right-hand side of the NEXT prox solve (k_build_b4's arithmetic)
    const unsigned i32 = q * 4u;
    const unsigned r = i32 / (unsigned)nx.nwrow, k = i32 - r * (unsigned)nx.nwrow;   // a group of 4 never straddles rows
    const float* bp = nx.B0 + (size_t)r * (size_t)nx.n + k;                            // (rows of B0 are n long: unaligned)
    const float4 w0 = *reinterpret_cast<const float4*>(nx.W0 + i);
    const float w0e[4] = {w0.x, w0.y, w0.z, w0.w};
    float be[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = bp[e] + nx.eta * w0e[e];
      be[e] = t + nx.rho * (ge[e] - du[e]);
    }
    *reinterpret_cast<float4*>(nx.Bm + (size_t)r * (size_t)nx.ldb + k) = make_float4(be[0], be[1], be[2], be[3]);
  }
}

// The same projection as the EPILOGUE of a single-workgroup fixed point (G == nullptr: none): every thread of the
// workgroup knows the converged scale, the operands are a few tens of KB - a separate launch costs more than the work.
struct ProjFused {
  const float* wstar;
  float* G;
  float* dual;
  int8_t* Gq;
  int32_t* err_flag;
  double d;
  float dual_div;
  int lm1;
  unsigned n4;
  ProjNext nx;
};
__device__ __forceinline__ void proj_fused_epilogue(const ProjFused& pf, const float* v, double alpha, int done, int tid,
                                                    int nthr) {
  if (tid == 0 && pf.err_flag != nullptr && done != 1) *pf.err_flag = (done == 2) ? 2 : 3;
  const float alpha32 = (float)alpha;
  const LevelConsts lc = level_consts(alpha, -1.0, 1.0, pf.d);
  // One workgroup walks the whole tensor: the operand lines of the next groups are touched first (the stores of a group
  // may alias the loads of the next as far as the compiler knows, so it would otherwise wait for one round trip per group)
  constexpr int U = 4;
  for (unsigned q0 = (unsigned)tid; q0 < pf.n4; q0 += (unsigned)nthr * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned q = q0 + (unsigned)u * (unsigned)nthr;
      if (q < pf.n4) {
        const size_t i = (size_t)q * 4;
        __builtin_prefetch(v + i, 0, 3);
        __builtin_prefetch(pf.wstar + i, 0, 3);
        __builtin_prefetch(pf.dual + i, 1, 3);
        if (pf.nx.Bm != nullptr) __builtin_prefetch(pf.nx.W0 + i, 0, 3);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned q = q0 + (unsigned)u * (unsigned)nthr;
      if (q < pf.n4)
        proj4_apply(q, v, pf.wstar, alpha, alpha32, lc, pf.d, pf.G, pf.dual, pf.dual_div, pf.Gq, pf.lm1, pf.nx);
    }
  }
}

}  // namespace effq
