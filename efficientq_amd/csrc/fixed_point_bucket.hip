// project_by_iter (layer_helper.py:40-70) on a value-bucketed copy of the tensor.
//
// The fixed point  a <- sum(b*v)/sum(b*b),  b = discretize(v/a)  only ever asks, per iteration, how many values -
// and which sum of values - lie below each of the L-1 level boundaries.  Instead of re-classifying all n values in
// fp64 on every iteration (k_fp_small / k_fp_coop: 7 us per iteration at 27 k values on one CU, or a grid barrier per
// iteration on many), the values are
//   1. summed (sum|v|, max|v|), counted into B equal-width buckets over [-max|v|, max|v|] and regrouped by bucket,
//      with an exact integer sum per bucket (unit = bucket width * 2^-36) and exclusive prefix tables,
//   2. iterated on: a level boundary falls into one bucket (rarely two); every bucket below it is "below" as a whole
//      (prefix tables), only the values of the boundary bucket are looked at one by one.
// Classification of a single value: the level function of the reference, rint((clamp(v/a) - lo)/d), is monotone in v,
// and its k-th boundary sits at t_k = a*(lo + (k-0.5)*d) up to a few fp64 roundings: |error| <= ~4 ulp64 * |t_k| plus
// a * 2^-52 from the subtraction of lo (the boundary at v = 0: 1 - |v/a| rounds to 1 for tiny negative v).  Values
// outside the guard band t_k -+ max(|t_k| * 2^-20, a * 2^-30) are therefore classified by comparison; values INSIDE it
// (a 1e-6 relative sliver) by the reference's own arithmetic (fpb_level: IEEE fp64 divisions, round-half-even).  The
// level counts are exactly the reference's; the sums are fp64 sums of the same products in another order, built from
// integer partial sums (exact, order-independent: run-to-run and rank-to-rank deterministic), i.e. alpha agrees to
// ~1e-14 relative and the iteration count is the same.  Tested against the reference goldens (G2), the oracle, the
// all-values kernels and adversarial inputs (values on boundaries, zeros, tiny negatives, duplicates, outliers).
//
// Two paths: n <= 32768 - ONE workgroup, values and tables in LDS (k_fps); larger - four launches with global
// atomics (k_fpg_sum / k_fpg_count / k_fpg_scan / k_fpg_scatter_iter), the last workgroup to finish regrouping iterates.
#include "common.h"
#include "fp_level.h"
#include "project_dual.h"

namespace effq {

#ifdef EFFQ_TRACE
__device__ long long g_fpb_trace[1024];
#define FPB_TRACE(i) do { if (threadIdx.x == 0 && (i) < 1024) g_fpb_trace[(i)] = clock64(); } while (0)
#else
#define FPB_TRACE(i) do { } while (0)
#endif

constexpr int FPB_SHIFT = 36;    // integer sums carry (v - origin) in units of bucket_width * 2^-36
constexpr int FPB_TI = 256;      // threads of the iteration phase
constexpr int FPB_CR = 4;        // boundary-bucket values cached in registers per lane

__device__ __forceinline__ int fpb_level(double x, double alpha, double lo, double hi, double d) {
  // layer_helper.py:25-37 in fp64, exactly as disc64 in quant_reduce.hip
  double t = x / alpha;
  t = fmin(fmax(t, lo), hi);
  return (int)rint((t - lo) / d);
}

struct FpbGeo {
  float R, scale;        // bucket(v) = clamp(floor((v + R) * scale), 0, B-1)   (fp32 arithmetic: monotone in v)
  double R64, w, rq, q;  // bucket b starts at b*w - R64; integer unit q = w * 2^-SHIFT, rq = 1/q
  int B;
};

__device__ __forceinline__ FpbGeo fpb_geo(float R, int B) {
  if (!(R >= 1e-30f)) R = 1e-30f;
  if (!(R <= 3.0e38f)) R = 3.0e38f;
  FpbGeo g;
  g.B = B;
  g.R = R;
  g.scale = (float)B / (2.0f * R);
  g.R64 = (double)R;
  g.w = 2.0 * g.R64 / (double)B;
  g.q = ldexp(g.w, -FPB_SHIFT);
  g.rq = 1.0 / g.q;
  return g;
}

__device__ __forceinline__ int fpb_bucket(float v, const FpbGeo& g) {
  float f = floorf((v + g.R) * g.scale);
  f = fminf(fmaxf(f, 0.0f), (float)(g.B - 1));
  return (int)f;
}
// one fp32 step down (dir < 0) or up; saturates at +-FLT_MAX.  Integer arithmetic on the ordered key of the value.
__device__ __forceinline__ float fpb_step(float v, int dir) {
  unsigned u = __float_as_uint(v);
  unsigned key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  if (dir < 0)
    key = (key > 0x00800000u) ? key - 1u : 0x00800000u;     // key(-FLT_MAX)
  else
    key = (key < 0xff7fffffu) ? key + 1u : 0xff7fffffu;     // key(+FLT_MAX)
  u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
  return __uint_as_float(u);
}
// integer contribution of v relative to the lower edge of bucket (b - 1): positive for every v of buckets >= b
// (values beyond the range are clamped INTO the end buckets and may lie far outside them: the conversion saturates
// nowhere near that: |v - origin| / q < 2^63 needs |v| < 2^27 * range; beyond it the kernel reports non-convergence)
__device__ __forceinline__ long long fpb_units(float v, int b, const FpbGeo& g) {
  return __double2ll_rn(((double)v - ((double)(b - 1) * g.w - g.R64)) * g.rq);
}

struct FpbShared {
  double thrS[2][257];       // per iteration parity: S(T_k) = sum of the values below boundary k
  unsigned thrC[2][257];     //                        C(T_k) = their count
  double part[2][2][FPB_TI / 64];
};

// exclusive prefix tables over the buckets: flat (one array each), or segmented (prefix inside segments of 1024
// buckets + the prefix over the segment totals: what the multi-workgroup scan leaves behind)
struct TabFlat {
  const unsigned* off;
  const double* spre;
  __device__ __forceinline__ unsigned cnt_below(int b) const { return off[b]; }
  __device__ __forceinline__ double sum_below(int b) const { return spre[b]; }
};
struct TabSeg {
  const unsigned* off;       // [B]   exclusive prefix inside the bucket's segment
  const double* spre;        // [B]
  const unsigned* segc;      // [S+1] exclusive prefix over the segment totals
  const double* segs;        // [S+1]
  int B;
  __device__ __forceinline__ unsigned cnt_below(int b) const { return (b >= B) ? segc[B >> 10] : off[b] + segc[b >> 10]; }
  __device__ __forceinline__ double sum_below(int b) const { return (b >= B) ? segs[B >> 10] : spre[b] + segs[b >> 10]; }
};

// The iteration phase: FPB_TI threads, prefix tables `tab` and the regrouped values `vals` in whatever memory they
// point to.  Returns through *st (thread 0).
template <typename Tab>
__device__ __forceinline__ void fpb_iterate(const float* __restrict__ vals, const Tab& tab, const FpbGeo& g, size_t n,
                                            double tot_abs, double lo, double hi, double d, int levels, double tol,
                                            int max_iter, FpbShared& sh, effq_fp_state* st, FptPred* pred = nullptr,
                                            double* alpha_out = nullptr, int* done_out = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nthr = levels - 1;                 // level boundaries k = 1 .. L-1
  int p2 = 1;
  while (p2 < nthr) p2 <<= 1;
  int gs = FPB_TI / p2;                        // lanes per boundary (a power of two, one wave at most)
  if (gs > 64) gs = 64;
  if (gs < 1) gs = 1;
  const int k = tid / gs + 1, gl = tid % gs;
  const bool active = (k <= nthr);
  double alpha = tot_abs / (double)n, alpha_prev = -999.0;
  double last0 = 0.0, last1 = 0.0;
  int it = 0, done = 0;
  if (tid < 2) {
    sh.thrC[tid][0] = 0u;
    sh.thrS[tid][0] = 0.0;
    sh.thrC[tid][levels] = tab.cnt_below(g.B);
    sh.thrS[tid][levels] = tab.sum_below(g.B);
  }
  // cached boundary segment of this lane's boundary: values seg0 + gl + r*gs, their integer units
  int c_jlo = -1, c_jhi = -1;
  unsigned c_seg0 = 0, c_seg1 = 0;
  double c_spre = 0.0, c_origin = 0.0;
  float cv[FPB_CR];
  long long cq[FPB_CR];
  const double kpos = lo + ((double)k - 0.5) * d;
  while (!done) {
    FPB_TRACE(16 + 4 * it);
    if (!(alpha > 0.0) || !(alpha < 1e300)) {  // NaN / non-positive scale: the reference would spin to its cap
      done = 2;
      break;
    }
    const int par = it & 1;
    if (tid == 0) fpt_note(pred, it, alpha);     // (the iterates seed the next call's predictions: fixed_point_traj.hip)
    if (active) {
      const double tk = alpha * kpos;
      const double guard = fmax(fabs(tk) * 9.5367431640625e-07, alpha * 9.313225746154785e-10);   // 2^-20, 2^-30
      // band ends in fp32, pushed one fp32 step outwards (the rounding of the conversion may only widen the band)
      const float vlo = fpb_step((float)(tk - guard), -1), vhi = fpb_step((float)(tk + guard), +1);
      const int jlo = fpb_bucket(vlo, g), jhi = fpb_bucket(vhi, g);
      if (jlo != c_jlo || jhi != c_jhi) {
        c_jlo = jlo;
        c_jhi = jhi;
        c_seg0 = tab.cnt_below(jlo);
        c_seg1 = tab.cnt_below(jhi + 1);
        c_spre = tab.sum_below(jlo);
        c_origin = (double)(jlo - 1) * g.w - g.R64;
#pragma unroll
        for (int r = 0; r < FPB_CR; ++r) {
          const unsigned i = c_seg0 + (unsigned)gl + (unsigned)(r * gs);
          cv[r] = (i < c_seg1) ? vals[i] : 3.0e38f;       // padding sorts above every boundary
          cq[r] = (i < c_seg1) ? fpb_units(cv[r], jlo, g) : 0ll;
        }
      }
      unsigned cnt = 0;
      long long qs = 0;
      bool inband = false;
#pragma unroll
      for (int r = 0; r < FPB_CR; ++r) {
        const float v = cv[r];
        const bool below = (v <= vlo);
        inband |= (!below && v < vhi);
        cnt += below ? 1u : 0u;
        qs += below ? cq[r] : 0ll;
      }
      // values inside the guard band (a 1e-6 relative sliver around the boundary): the reference's own arithmetic.
      // A wave-uniform branch, so the two fp64 divisions per value are not paid when no lane needs them.
      if (__builtin_amdgcn_ballot_w64(inband) != 0ull) {
#pragma unroll
        for (int r = 0; r < FPB_CR; ++r) {
          const float v = cv[r];
          if (v > vlo && v < vhi && fpb_level((double)v, alpha, lo, hi, d) < k) {
            ++cnt;
            qs += cq[r];
          }
        }
      }
      if (c_seg0 + (unsigned)(FPB_CR * gs) < c_seg1) {     // segments longer than the register cache: rare
        for (unsigned i = c_seg0 + (unsigned)gl + (unsigned)(FPB_CR * gs); i < c_seg1; i += (unsigned)gs) {
          const float v = vals[i];
          bool below = (v <= vlo);
          if (!below && v < vhi) below = fpb_level((double)v, alpha, lo, hi, d) < k;
          if (below) {
            ++cnt;
            qs += fpb_units(v, jlo, g);
          }
        }
      }
      // integer sums across the group: exact in any order.  Per lane qs < 2^42 (values < 2^39 each): two 21-bit limbs
      // keep every partial sum of a 64-lane group below 2^32; a third limb only where a lane exceeds 2^42
      if (gs > 1) {
        const unsigned long long uq = (unsigned long long)qs;
        unsigned l0 = (unsigned)(uq & 0x1fffffu), l1 = (unsigned)((uq >> 21) & 0x1fffffu), l2 = (unsigned)(uq >> 42);
        cnt = group_sum_u32(cnt, gs);
        l0 = group_sum_u32(l0, gs);
        l1 = group_sum_u32(l1, gs);
        if (__builtin_amdgcn_ballot_w64(l2 != 0u) != 0ull) l2 = group_sum_u32(l2, gs);
        qs = (long long)(((unsigned long long)l2 << 42) + ((unsigned long long)l1 << 21) + (unsigned long long)l0);
      }
      if (gl == 0) {
        sh.thrC[par][k] = c_seg0 + cnt;
        sh.thrS[par][k] = c_spre + ((double)cnt * c_origin + g.q * (double)qs);
      }
    }
    FPB_TRACE(17 + 4 * it);
    __syncthreads();
    FPB_TRACE(18 + 4 * it);
    // level m holds the values between boundaries m and m+1: sum b*v = sum_m b_m S_m, sum b*b = sum_m b_m^2 C_m
    double t0 = 0.0, t1 = 0.0;
    if (levels <= 8) {                          // every thread adds the few terms itself, in level order
      double sprev = 0.0;
      unsigned cprev = 0u;
      for (int m = 0; m < levels; ++m) {
        const double sn = sh.thrS[par][m + 1];
        const unsigned cn = sh.thrC[par][m + 1];
        const double bm = (double)m * d + lo;
        t0 += bm * (sn - sprev);
        t1 += bm * bm * (double)(cn - cprev);
        sprev = sn;
        cprev = cn;
      }
    } else {
      double acc0 = 0.0, acc1 = 0.0;
      if (tid < levels) {
        const double bm = (double)tid * d + lo;
        acc0 = bm * (sh.thrS[par][tid + 1] - sh.thrS[par][tid]);
        acc1 = bm * bm * (double)(sh.thrC[par][tid + 1] - sh.thrC[par][tid]);
      }
      acc0 = wave_sum_f64_dpp(acc0);
      acc1 = wave_sum_f64_dpp(acc1);
      if (lane == 0) {
        sh.part[par][0][wid] = acc0;
        sh.part[par][1][wid] = acc1;
      }
      __syncthreads();
#pragma unroll
      for (int w = 0; w < FPB_TI / 64; ++w) {  // every thread adds the wave partials in wave order: same bits everywhere
        t0 += sh.part[par][0][w];
        t1 += sh.part[par][1][w];
      }
    }
    FPB_TRACE(19 + 4 * it);
    const double a_new = t0 / t1;
    ++it;
    if (it >= max_iter)
      done = 2;
    else if (!(fabs(a_new - alpha) > tol))
      done = 1;
    alpha_prev = alpha;
    alpha = a_new;
    last0 = t0;
    last1 = t1;
  }
  if (tid == 0) {
    st->alpha = alpha;
    st->alpha_prev = alpha_prev;
    st->sums[0] = last0;
    st->sums[1] = last1;
    st->iters = it;
    st->done = done;
    if (pred != nullptr) {
      for (int j = 0; j < FPT_SLOTS; ++j) fpt_finish_slot(pred, j, it, alpha);
      fpt_finish_head(pred, it, tot_abs, levels);
    }
  }
  if (alpha_out != nullptr) *alpha_out = alpha;      // (every thread of the iteration phase holds the same values)
  if (done_out != nullptr) *done_out = done;
}

// ---- path S: one workgroup, everything in LDS ---------------------------------------------------------------------------
constexpr int FPS2_T = 1024;
constexpr int FPS2_B = 2048;
constexpr int FPS2_MAXN = 32768;

template <int PER>   // register slots per thread (values stay in registers through the three build passes)
__global__ __launch_bounds__(FPS2_T) void k_fps(const float* __restrict__ a, const float* b2, float* v_out, size_t n,
                                                effq_fp_state* st, double lo, double hi, double d, int levels, double tol,
                                                int max_iter, FptPred* pred, ProjFused pf) {
  __builtin_amdgcn_s_setprio(3);        // latency-bound, shares its CU with loss-conv waves

  constexpr int B = FPS2_B;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // LDS: spre[B+1] f64 | off[B+2] u32 | vals[n] f32
  double* spre = reinterpret_cast<double*>(smem_raw);
  unsigned* off = reinterpret_cast<unsigned*>(spre + (B + 1));
  float* vals = reinterpret_cast<float*>(off + (B + 2));
  __shared__ FpbShared sh;
  __shared__ double s_red[16];
  __shared__ float s_max[16];
  __shared__ unsigned s_wcnt[16];
  __shared__ double s_wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  constexpr bool KEEP_RANK = PER <= 16;         // 32 slots + 16 rank registers would spill at 1024 threads (128 VGPRs)
  unsigned* cur = reinterpret_cast<unsigned*>(spre);   // !KEEP_RANK: pass 2 draws slots from a second counter array
  for (int i = tid; i < B + 2; i += FPS2_T) {          // (the space of spre, which is written after pass 2)
    off[i] = 0u;
    if (!KEEP_RANK) cur[i] = 0u;
  }
  FPB_TRACE(0);
  // ---- pass 0: v (kept in registers), sum|v|, max|v| -------------------------------------------------------------
  float vr[PER];
  double sabs = 0.0;
  float mx = 0.0f;
  // loads first, on clamped indices (no branch in front of a load: they stay in flight together), 16 slots at a time
  constexpr int LB = (PER < 16) ? PER : 16;
#pragma unroll
  for (int h = 0; h < PER / LB; ++h) {
    float bv[LB];
#pragma unroll
    for (int s = 0; s < LB; ++s) {
      const size_t i = (size_t)tid + (size_t)(h * LB + s) * FPS2_T;
      const size_t ic = (i < n) ? i : (n - 1);
      vr[h * LB + s] = a[ic];
      bv[s] = (b2 != nullptr) ? b2[ic] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < LB; ++s) {
      const size_t i = (size_t)tid + (size_t)(h * LB + s) * FPS2_T;
      float v = (b2 != nullptr) ? (vr[h * LB + s] + bv[s]) : vr[h * LB + s];
      if (i < n) {
        if (v_out != nullptr) v_out[i] = v;
      } else {
        v = 0.0f;
      }
      vr[h * LB + s] = v;
      sabs += fabs((double)v);
      mx = fmaxf(mx, fabsf(v));
    }
  }
  sabs = wave_sum(sabs);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
  if (lane == 0) {
    s_red[wid] = sabs;
    s_max[wid] = mx;
  }
  __syncthreads();
  double tot = 0.0;
  float R = 0.0f;
#pragma unroll
  for (int w = 0; w < FPS2_T / 64; ++w) {       // every thread adds the wave partials in wave order: same bits everywhere
    tot += s_red[w];
    R = fmaxf(R, s_max[w]);
  }
  const FpbGeo g = fpb_geo(R, B);
  FPB_TRACE(1);
  // ---- pass 1: bucket counts (the returned count is the value's rank inside its bucket) ---------------------------
  unsigned rank2[KEEP_RANK ? (PER + 1) / 2 : 1];   // two 16-bit ranks per register (n <= 32768 < 2^16)
#pragma unroll
  for (int s = 0; s < (KEEP_RANK ? (PER + 1) / 2 : 1); ++s) rank2[s] = 0u;
#pragma unroll
  for (int s = 0; s < PER; ++s) {
    const size_t i = (size_t)tid + (size_t)s * FPS2_T;
    if (i < n) {
      const unsigned r = atomicAdd(&off[fpb_bucket(vr[s], g)], 1u);
      if (KEEP_RANK) rank2[s >> 1] |= r << (16 * (s & 1));
    }
  }
  __syncthreads();
  FPB_TRACE(2);
  // ---- exclusive scan of the counts (2 buckets per thread, fixed shape) ---------------------------------------------------
  constexpr int BPT = B / FPS2_T;
  unsigned c_loc[BPT], c_thr = 0;
#pragma unroll
  for (int j = 0; j < BPT; ++j) {
    c_loc[j] = off[tid * BPT + j];
    c_thr += c_loc[j];
  }
  unsigned c_inc = c_thr;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned cu = __shfl_up(c_inc, o, 64);
    if (lane >= o) c_inc += cu;
  }
  if (lane == 63) s_wcnt[wid] = c_inc;
  __syncthreads();
  unsigned c_run = c_inc - c_thr;
  for (int w = 0; w < wid; ++w) c_run += s_wcnt[w];
#pragma unroll
  for (int j = 0; j < BPT; ++j) {
    off[tid * BPT + j] = c_run;
    c_run += c_loc[j];
  }
  if (tid == FPS2_T - 1) off[B] = c_run;        // = n
  __syncthreads();
  FPB_TRACE(3);
  // ---- pass 2: regroup the values by bucket (LDS) --------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < PER; ++s) {
    const size_t i = (size_t)tid + (size_t)s * FPS2_T;
    if (i < n) {
      const int b = fpb_bucket(vr[s], g);
      const unsigned r = KEEP_RANK ? ((rank2[s >> 1] >> (16 * (s & 1))) & 0xffffu) : atomicAdd(&cur[b], 1u);
      vals[off[b] + r] = vr[s];
    }
  }
  __syncthreads();
  FPB_TRACE(4);
  // ---- pass 3: integer sum per bucket, exclusive scan of the bucket sums.  Every thread walks a contiguous chunk of
  // the REGROUPED values (balanced; consecutive values share a bucket) and adds one integer per run of equal buckets
  // to isum[] - integers, so any order gives the same bits.  isum aliases spre (each thread later overwrites exactly
  // the entries it has read).
  unsigned long long* isum = reinterpret_cast<unsigned long long*>(spre);
#pragma unroll
  for (int j = 0; j < BPT; ++j) isum[tid * BPT + j] = 0ull;
  __syncthreads();
  {
    const unsigned chunk = (unsigned)((n + FPS2_T - 1) / FPS2_T) | 1u;      // odd stride: conflict-free LDS reads
    const unsigned c0 = (unsigned)tid * chunk;
    const unsigned c1 = (c0 + chunk < (unsigned)n) ? c0 + chunk : (unsigned)n;
    int cb = -1;
    long long acc = 0;
#pragma unroll 4
    for (unsigned i = c0; i < c1; ++i) {
      const float v = vals[i];
      const int b = fpb_bucket(v, g);
      if (b != cb) {
        if (cb >= 0) atomicAdd(&isum[cb], (unsigned long long)acc);
        cb = b;
        acc = 0;
      }
      acc += fpb_units(v, b, g);
    }
    if (cb >= 0) atomicAdd(&isum[cb], (unsigned long long)acc);
  }
  __syncthreads();
  double s_loc[BPT], s_thr = 0.0;
#pragma unroll
  for (int j = 0; j < BPT; ++j) {
    const int b = tid * BPT + j;
    s_loc[j] = (double)c_loc[j] * ((double)(b - 1) * g.w - g.R64) + g.q * (double)(long long)isum[b];
    s_thr += s_loc[j];
  }
  double s_inc = s_thr;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double su = __shfl_up(s_inc, o, 64);
    if (lane >= o) s_inc += su;
  }
  if (lane == 63) s_wsum[wid] = s_inc;
  __syncthreads();
  double s_run = s_inc - s_thr;
  {
    double base = 0.0;
    for (int w = 0; w < wid; ++w) base += s_wsum[w];
    s_run += base;
  }
#pragma unroll
  for (int j = 0; j < BPT; ++j) {
    spre[tid * BPT + j] = s_run;
    s_run += s_loc[j];
  }
  if (tid == FPS2_T - 1) spre[B] = s_run;       // = sum v
  __syncthreads();
  FPB_TRACE(5);
  if (tid >= FPB_TI) return;                    // the barriers below only count the surviving waves
  TabFlat tab{off, spre};
  double alpha_fin = 0.0;
  int done_fin = 0;
  fpb_iterate(vals, tab, g, n, tot, lo, hi, d, levels, tol, max_iter, sh, st, pred, &alpha_fin, &done_fin);
  // the projection + dual update of this ADMM iteration as the epilogue of the same launch (the FPB_TI surviving threads;
  // v_out was stored before the build passes' barriers)
  if (pf.G != nullptr) proj_fused_epilogue(pf, v_out, alpha_fin, done_fin, tid, FPB_TI);
}

// ---- path G: large tensors, four launches ------------------------------------------------------------------------------------
// k_fpg_sum   : v = a + b2 -> v_out, |v| partials and max|v| per workgroup; the last workgroup to finish (ticket) adds
//               them in workgroup order: sum|v| (deterministic) and the bucket range R = max|v| (nothing is clamped).
// k_fpg_count : counts the values into the buckets (workgroup-private LDS histograms merged with one global atomic per
//               non-empty bucket) and adds their integer units; leaves every value's rank inside its bucket.
// k_fpg_scan  : exclusive prefixes of counts and sums, counters cleared for the next call.
// k_fpg_scatter_iter : every workgroup regroups its values (slot = bucket offset + rank); the last one iterates.
struct FpgHeader {          // first 256 bytes of the workspace
  float range;              // bucket range R = max|v| of this call
  unsigned ticket0, ticket1, ticket2;
  double tot_abs;
};
constexpr int FPG_T = 256;
constexpr int FPG_MAXB = 65536;
constexpr int FPG_EPT = 8;        // values per thread of the streaming kernels
constexpr int FPG_MAXBLK = 4096;  // >= 2^23 / (FPG_T * FPG_EPT)

struct FpgWs {
  FpgHeader* hdr;
  unsigned* cnt;              // [B]   zero between calls
  unsigned long long* isum;   // [B]   zero between calls
  unsigned* off;              // [B]   exclusive prefix inside the bucket's segment of 1024
  double* spre;               // [B]
  unsigned* segc;             // [S+1] exclusive prefix over the segment totals
  double* segs;               // [S+1]
  unsigned* segc_tot;         // [S]
  double* segs_tot;           // [S]
  double* partials;           // [2 * FPG_MAXBLK]
  unsigned* rank;             // [n]
  float* grouped;             // [n]
};
static size_t fpg_ws_bytes(size_t n) {
  return 256 + sizeof(unsigned) * FPG_MAXB + sizeof(unsigned long long) * FPG_MAXB + sizeof(unsigned) * (FPG_MAXB + 2) +
         sizeof(double) * (FPG_MAXB + 2) + sizeof(double) * 2 * FPG_MAXBLK + 4096 +
         (sizeof(unsigned) + sizeof(float)) * (n + 64);
}
static FpgWs fpg_carve(void* ws, size_t n) {
  FpgWs w;
  char* p = reinterpret_cast<char*>(ws);
  w.hdr = reinterpret_cast<FpgHeader*>(p);
  p += 256;
  w.isum = reinterpret_cast<unsigned long long*>(p);
  p += sizeof(unsigned long long) * FPG_MAXB;
  w.spre = reinterpret_cast<double*>(p);
  p += sizeof(double) * (FPG_MAXB + 2);
  w.partials = reinterpret_cast<double*>(p);
  p += sizeof(double) * 2 * FPG_MAXBLK;
  w.segs = reinterpret_cast<double*>(p);          // 4096 bytes: 4 arrays of at most 65 entries
  p += sizeof(double) * 128;
  w.segs_tot = reinterpret_cast<double*>(p);
  p += sizeof(double) * 128;
  w.segc = reinterpret_cast<unsigned*>(p);
  p += sizeof(unsigned) * 128;
  w.segc_tot = reinterpret_cast<unsigned*>(p);
  p += 4096 - 2 * sizeof(double) * 128 - sizeof(unsigned) * 128;
  w.cnt = reinterpret_cast<unsigned*>(p);
  p += sizeof(unsigned) * FPG_MAXB;
  w.off = reinterpret_cast<unsigned*>(p);
  p += sizeof(unsigned) * (FPG_MAXB + 2);
  w.rank = reinterpret_cast<unsigned*>(p);
  p += sizeof(unsigned) * (n + 64);
  w.grouped = reinterpret_cast<float*>(p);
  return w;
}

__global__ __launch_bounds__(FPG_T) void k_fpg_sum(const float* __restrict__ a, const float* __restrict__ b2,
                                                   float* __restrict__ v_out, size_t n, FpgWs w) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams

  __shared__ double s_red[16];
  __shared__ float s_mx[FPG_T / 64];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  double sabs = 0.0;
  float mx = 0.0f;
  float vv[FPG_EPT];
  const size_t base = (size_t)blockIdx.x * (FPG_T * FPG_EPT) + tid;
#pragma unroll
  for (int e = 0; e < FPG_EPT; ++e) {            // loads first: all in flight together
    const size_t i = base + (size_t)e * FPG_T;
    const size_t ic = (i < n) ? i : (n - 1);
    vv[e] = (b2 != nullptr) ? (a[ic] + b2[ic]) : a[ic];
  }
#pragma unroll
  for (int e = 0; e < FPG_EPT; ++e) {
    const size_t i = base + (size_t)e * FPG_T;
    if (i < n) {
      if (v_out != nullptr) v_out[i] = vv[e];
      sabs += fabs((double)vv[e]);
      mx = fmaxf(mx, fabsf(vv[e]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
  if (lane == 0) s_mx[tid >> 6] = mx;
  double acc[1] = {sabs};
  block_sum<1>(acc, s_red);
  // hand-off to the last block: release / ticket / acquire (cdna_hip_programming.md Guideline 16)
  if (tid == 0) {
    float bm = 0.0f;
#pragma unroll
    for (int wv = 0; wv < FPG_T / 64; ++wv) bm = fmaxf(bm, s_mx[wv]);
    w.partials[2 * blockIdx.x] = acc[0];
    w.partials[2 * blockIdx.x + 1] = (double)bm;       // (no atomicMax: thousands of waves on one word serialise)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(&w.hdr->ticket0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  double t = 0.0;                                  // sum|v| in block order (deterministic), max|v|
  float gm = 0.0f;
  for (unsigned b = tid; b < gridDim.x; b += FPG_T) {
    t += __hip_atomic_load(&w.partials[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gm = fmaxf(gm, (float)__hip_atomic_load(&w.partials[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_down(gm, o, 64));
  __syncthreads();
  if (lane == 0) s_mx[tid >> 6] = gm;
  double tv[1] = {t};
  block_sum<1>(tv, s_red);
  if (tid == 0) {
    float m = 0.0f;
#pragma unroll
    for (int wv = 0; wv < FPG_T / 64; ++wv) m = fmaxf(m, s_mx[wv]);
    w.hdr->tot_abs = tv[0];
    w.hdr->range = m;
    __hip_atomic_store(&w.hdr->ticket0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Counting with workgroup-private histograms: inside ADMM the values cluster around the few quantisation levels, so
// thousands of them share a bucket and global atomics on those few addresses serialise (154 us per pass in situ with
// one global atomic per value).  Each workgroup counts its slice of FPG_SLICE values in LDS (count + integer sum per
// bucket, LDS atomics; the returned count is the value's rank among the slice's values of that bucket), then adds its
// non-empty buckets to the global tables with ONE atomic each (consecutive addresses: coalesced wave instructions);
// the value returned by that add is where the slice's values start inside the global bucket.
constexpr int FPG_B = 8192;                   // buckets of the multi-workgroup path (count + sum tables: 96 KB of LDS)
constexpr int FPG_CT = 1024;                  // threads of the count kernel
constexpr int FPG_SLICE = FPG_CT * FPG_EPT;   // values per workgroup
__global__ __launch_bounds__(FPG_CT) void k_fpg_count(const float* __restrict__ src, size_t n, FpgWs w) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* isum_l = reinterpret_cast<unsigned long long*>(smem_raw);     // [FPG_B]
  unsigned* cnt_l = reinterpret_cast<unsigned*>(isum_l + FPG_B);                     // [FPG_B]
  const int tid = threadIdx.x;
  const FpbGeo g = fpb_geo(w.hdr->range, FPG_B);
  float vv[FPG_EPT];
  unsigned rk[FPG_EPT];
  int bk[FPG_EPT];
  const size_t base = (size_t)blockIdx.x * FPG_SLICE + tid;
#pragma unroll
  for (int e = 0; e < FPG_EPT; ++e) {
    const size_t i = base + (size_t)e * FPG_CT;
    vv[e] = src[(i < n) ? i : (n - 1)];
  }
  for (int b = tid; b < FPG_B; b += FPG_CT) {
    cnt_l[b] = 0u;
    isum_l[b] = 0ull;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < FPG_EPT; ++e) {
    const size_t i = base + (size_t)e * FPG_CT;
    bk[e] = fpb_bucket(vv[e], g);
    rk[e] = 0u;
    if (i < n) {
      rk[e] = atomicAdd(&cnt_l[bk[e]], 1u);
      atomicAdd(&isum_l[bk[e]], (unsigned long long)fpb_units(vv[e], bk[e], g));
    }
  }
  __syncthreads();
  for (int b = tid; b < FPG_B; b += FPG_CT) {
    const unsigned c = cnt_l[b];
    if (c != 0u) {
      cnt_l[b] = atomicAdd(&w.cnt[b], c);       // start of this slice's values inside the global bucket
      atomicAdd(&w.isum[b], isum_l[b]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < FPG_EPT; ++e) {
    const size_t i = base + (size_t)e * FPG_CT;
    if (i < n) w.rank[i] = cnt_l[bk[e]] + rk[e];
  }
}

// exclusive scan of (count, sum) over the buckets: one workgroup per segment of 1024 buckets scans its segment
// (coalesced loads, one bucket per thread) and clears the counters for the next call; the last workgroup to finish
// scans the segment totals.
constexpr int FPG_SEG = 1024;
__global__ __launch_bounds__(FPG_SEG) void k_fpg_scan(int B, FpgWs w) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams

  __shared__ unsigned s_wcnt[FPG_SEG / 64];
  __shared__ double s_wsum[FPG_SEG / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const FpbGeo g = fpb_geo(w.hdr->range, B);
  const int b = blockIdx.x * FPG_SEG + tid;
  const unsigned c = __hip_atomic_load(&w.cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long q = __hip_atomic_load(&w.isum[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double sv = (double)c * ((double)(b - 1) * g.w - g.R64) + g.q * (double)(long long)q;
  w.cnt[b] = 0u;
  w.isum[b] = 0ull;
  unsigned c_inc = c;
  double s_inc = sv;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned cu = __shfl_up(c_inc, o, 64);
    const double su = __shfl_up(s_inc, o, 64);
    if (lane >= o) {
      c_inc += cu;
      s_inc += su;
    }
  }
  if (lane == 63) {
    s_wcnt[wid] = c_inc;
    s_wsum[wid] = s_inc;
  }
  __syncthreads();
  unsigned c_run = c_inc - c;
  double s_run = s_inc - sv;
  for (int wv = 0; wv < wid; ++wv) {
    c_run += s_wcnt[wv];
    s_run += s_wsum[wv];
  }
  w.off[b] = c_run;
  w.spre[b] = s_run;
  if (tid == FPG_SEG - 1) {
    w.segc_tot[blockIdx.x] = c_run + c;
    w.segs_tot[blockIdx.x] = s_run + sv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(&w.hdr->ticket1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned cc = 0;
      double ss = 0.0;
      for (unsigned sgi = 0; sgi < gridDim.x; ++sgi) {      // <= 64 segments, in order
        w.segc[sgi] = cc;
        w.segs[sgi] = ss;
        cc += __hip_atomic_load(&w.segc_tot[sgi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ss += __hip_atomic_load(&w.segs_tot[sgi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      w.segc[gridDim.x] = cc;
      w.segs[gridDim.x] = ss;
      __hip_atomic_store(&w.hdr->ticket1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(FPG_T) void k_fpg_scatter_iter(const float* __restrict__ src, size_t n, int B, FpgWs w,
                                                            effq_fp_state* st, double lo, double hi, double d, int levels,
                                                            double tol, int max_iter, FptPred* pred) {
  __builtin_amdgcn_s_setprio(3);

  __shared__ FpbShared sh;
  __shared__ int s_last;
  const int tid = threadIdx.x;
  const FpbGeo g = fpb_geo(w.hdr->range, B);
  {
    float vv[FPG_EPT];
    unsigned rk[FPG_EPT], ob[FPG_EPT];
    const size_t base = (size_t)blockIdx.x * (FPG_T * FPG_EPT) + tid;
#pragma unroll
    for (int e = 0; e < FPG_EPT; ++e) {
      const size_t i = base + (size_t)e * FPG_T;
      const size_t ic = (i < n) ? i : (n - 1);
      vv[e] = src[ic];
      rk[e] = w.rank[ic];
    }
#pragma unroll
    for (int e = 0; e < FPG_EPT; ++e) {
      const int b = fpb_bucket(vv[e], g);
      ob[e] = w.off[b] + w.segc[b >> 10];
    }
#pragma unroll
    for (int e = 0; e < FPG_EPT; ++e) {
      const size_t i = base + (size_t)e * FPG_T;
      if (i < n) w.grouped[ob[e] + rk[e]] = vv[e];
    }
  }
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(&w.hdr->ticket2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&w.hdr->ticket2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const double tot = w.hdr->tot_abs;
  TabSeg tab{w.off, w.spre, w.segc, w.segs, B};
  fpb_iterate(w.grouped, tab, g, n, tot, lo, hi, d, levels, tol, max_iter, sh, st, pred);
}

static size_t fps_lds_bytes(size_t n) {
  return sizeof(double) * (FPS2_B + 1) + sizeof(unsigned) * (FPS2_B + 2) + sizeof(float) * ((n + 3) & ~(size_t)3);
}

}  // namespace effq
using namespace effq;

extern "C" {

#ifdef EFFQ_TRACE
int effq_fpb_trace_read(long long* host_out, int n) {
  EFFQ_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fpb_trace), sizeof(long long) * (size_t)n));
  return EFFQ_OK;
}
#endif

size_t effq_fp_bucket_max(void) { return (size_t)1 << 23; }

size_t effq_fp_bucket_ws_bytes(size_t n) { return n <= (size_t)FPS2_MAXN ? 256 : fpg_ws_bytes(n); }

int effq_fixed_point_bucket_fused(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                                  double tol, int max_iter, effq_fp_state* state_dev, void* ws, size_t ws_bytes,
                                  void* pred_dev, const ProjFused* pf_in, int* fused_out, void* stream);

int effq_fixed_point_bucket_rec(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                                double tol, int max_iter, effq_fp_state* state_dev, void* ws, size_t ws_bytes,
                                void* pred_dev, void* stream) {
  return effq_fixed_point_bucket_fused(a, b, v_out, n, levels, lo, hi, tol, max_iter, state_dev, ws, ws_bytes, pred_dev,
                                       nullptr, nullptr, stream);
}

// internal (admm_run.hip): pf_in != NULL and n <= 32768 (one workgroup): the projection of the ADMM iteration runs as the
// kernel's epilogue and *fused_out = 1; larger tensors ignore pf_in (*fused_out = 0: the caller launches the projection)
int effq_fixed_point_bucket_fused(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                                  double tol, int max_iter, effq_fp_state* state_dev, void* ws, size_t ws_bytes,
                                  void* pred_dev, const ProjFused* pf_in, int* fused_out, void* stream) {
  FptPred* pred = reinterpret_cast<FptPred*>(pred_dev);
  ProjFused pf;
  memset(&pf, 0, sizeof(pf));
  if (fused_out != nullptr) *fused_out = 0;
  if (pf_in != nullptr && n <= (size_t)FPS2_MAXN && v_out != nullptr) {
    pf = *pf_in;
    if (fused_out != nullptr) *fused_out = 1;
  }
  EFFQ_CHECK_ARG(a && state_dev && n > 0 && levels >= 2 && levels <= 256 && hi > lo && max_iter > 0);
  EFFQ_CHECK_ARG(n <= effq_fp_bucket_max());
  EFFQ_CHECK_ARG(b == nullptr || v_out != nullptr);
  const double d = (hi - lo) / (double)(levels - 1);
  hipStream_t st = as_stream(stream);
  if (n > (size_t)FPS2_MAXN) {
    EFFQ_CHECK_ARG(ws != nullptr);
    if (ws_bytes < fpg_ws_bytes(n)) {
      set_error("fixed_point_bucket: workspace %zu < %zu bytes", ws_bytes, fpg_ws_bytes(n));
      return EFFQ_ERR_WORKSPACE;
    }
    const FpgWs w = fpg_carve(ws, n);
    const int B = FPG_B;
    const size_t blocks = (n + (size_t)FPG_T * FPG_EPT - 1) / ((size_t)FPG_T * FPG_EPT);
    EFFQ_CHECK_ARG(blocks <= FPG_MAXBLK);
    // (without b the values are already in place: the sum pass then only reads)
    const float* src = (v_out != nullptr) ? v_out : a;
    hipLaunchKernelGGL(k_fpg_sum, dim3((unsigned)blocks), dim3(FPG_T), 0, st, a, b, v_out, n, w);
    {
      static bool attr_set2 = false;
      const int lds = (int)((sizeof(unsigned long long) + sizeof(unsigned)) * FPG_B);
      if (!attr_set2) {
        EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fpg_count), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set2 = true;
      }
      const size_t cblocks = (n + FPG_SLICE - 1) / FPG_SLICE;
      hipLaunchKernelGGL(k_fpg_count, dim3((unsigned)cblocks), dim3(FPG_CT), lds, st, src, n, w);
    }
    hipLaunchKernelGGL(k_fpg_scan, dim3((unsigned)(B / FPG_SEG)), dim3(FPG_SEG), 0, st, B, w);
    hipLaunchKernelGGL(k_fpg_scatter_iter, dim3((unsigned)blocks), dim3(FPG_T), 0, st, src, n, B, w, state_dev, lo, hi, d,
                       levels, tol, max_iter, pred);
    EFFQ_LAUNCH_CHECK();
    return EFFQ_OK;
  }
  static bool attr_set = false;
  if (!attr_set) {
    const int lim = (int)fps_lds_bytes(FPS2_MAXN);
    EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fps<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fps<16>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fps<32>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    attr_set = true;
  }
  const size_t lds = fps_lds_bytes(n);
  const int per = (int)((n + FPS2_T - 1) / FPS2_T);
  if (per <= 8)
    hipLaunchKernelGGL(k_fps<8>, dim3(1), dim3(FPS2_T), lds, st, a, b, v_out, n, state_dev, lo, hi, d, levels, tol, max_iter, pred, pf);
  else if (per <= 16)
    hipLaunchKernelGGL(k_fps<16>, dim3(1), dim3(FPS2_T), lds, st, a, b, v_out, n, state_dev, lo, hi, d, levels, tol, max_iter, pred, pf);
  else
    hipLaunchKernelGGL(k_fps<32>, dim3(1), dim3(FPS2_T), lds, st, a, b, v_out, n, state_dev, lo, hi, d, levels, tol, max_iter, pred, pf);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_fixed_point_bucket(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                            double tol, int max_iter, effq_fp_state* state_dev, void* ws, size_t ws_bytes,
                            void* stream) {
  return effq_fixed_point_bucket_rec(a, b, v_out, n, levels, lo, hi, tol, max_iter, state_dev, ws, ws_bytes, nullptr,
                                     stream);
}

}  // extern "C"
