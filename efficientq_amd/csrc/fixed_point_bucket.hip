// project_by_iter (layer_helper.py:40-70) on a value-bucketed copy of the tensor: ONE workgroup, one launch.
//
// The fixed point  a <- sum(b*v)/sum(b*b),  b = discretize(v/a)  only ever asks, per iteration, how many values -
// and which sum of values - lie below each of the L-1 level boundaries.  Instead of re-classifying all n values in
// fp64 on every iteration (what k_fp_small / k_fp_coop do: 7 us per iteration at 27 k values on one CU, or a grid
// barrier per iteration on many), the kernel
//   1. forms v = a + b2, sum|v| and max|v|                                        (one pass),
//   2. counts the values into B equal-width buckets over [-max|v|, max|v|], with an exact integer sum per bucket,
//      prefix-scans counts and sums, and regroups the values by bucket            (two passes, LDS atomics),
//   3. iterates on that structure: a level boundary falls into one bucket (rarely two); every bucket below it is
//      "below" as a whole (prefix tables), only the values of the boundary bucket are classified one by one - with
//      the reference's own arithmetic (disc64: IEEE fp64 divisions, round-half-even), so the classification of
//      every value is bit-identical to discretize(v/a).
// Which buckets are "the boundary" is not taken on trust: the level function is monotone in v, and each boundary is
// bracketed by two fp32 values whose levels are CHECKED with disc64 (k-1 or less below, k or more above); the
// bracket is widened until the check holds (it does so at once except for the boundary at v = 0, where
// 1 - |v/a| rounds to 1 for tiny negative v).  So the counts are exactly the reference's; the sums are fp64 sums of
// the same products in another order (integer partial sums, exact and order-independent, hence run-to-run and
// rank-to-rank deterministic), i.e. alpha agrees to ~1e-14 relative and the iteration count is the same.
#include "common.h"

namespace effq {

constexpr int FPB_T = 1024;      // threads of the build phase
constexpr int FPB_TI = 256;      // threads that stay for the iterations (the other waves retire)
constexpr int FPB_SHIFT = 36;    // integer sums carry (v - bucket_lo) in units of bucket_width * 2^-36

__device__ __forceinline__ int fpb_level(double x, double alpha, double lo, double hi, double d) {
  // layer_helper.py:25-37 in fp64, exactly as disc64 in quant_reduce.hip
  double t = x / alpha;
  t = fmin(fmax(t, lo), hi);
  return (int)rint((t - lo) / d);
}

__device__ __forceinline__ unsigned fpb_key(float v) {        // order-preserving map float -> uint32
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fpb_unkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
constexpr unsigned FPB_KEY_MIN = 0x00800000u;   // key(-FLT_MAX)  (= ~0xff7fffff)
constexpr unsigned FPB_KEY_MAX = 0xff7fffffu;   // key(+FLT_MAX)

struct FpbGeo {
  float R, scale;        // bucket(v) = clamp(floor((v + R) * scale), 0, B-1)   (fp32: monotone in v)
  double R64, w, rq, q;  // bucket b starts at b*w - R64; integer unit q = w * 2^-SHIFT, rq = 1/q
};

template <int B>
__device__ __forceinline__ int fpb_bucket(float v, const FpbGeo& g) {
  float f = floorf((v + g.R) * g.scale);
  f = fminf(fmaxf(f, 0.0f), (float)(B - 1));
  return (int)f;
}

template <int B>
__global__ __launch_bounds__(FPB_T) void k_fp_bucket(const float* __restrict__ a, const float* __restrict__ b2,
                                                     float* __restrict__ v_out, float* __restrict__ grouped, size_t n,
                                                     effq_fp_state* st, double lo, double hi, double d, int levels,
                                                     double tol, int max_iter) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // LDS: off[B+1] u32 | cur[B] u32 | acc[B+1] (int64 sums, then fp64 exclusive prefix sums)
  unsigned* off = reinterpret_cast<unsigned*>(smem_raw);
  unsigned* cur = off + (B + 1) + 1;                                  // (+1 keeps the 8-byte alignment below)
  unsigned long long* isum = reinterpret_cast<unsigned long long*>(cur + B);
  double* spre = reinterpret_cast<double*>(isum);
  __shared__ double s_red[2 * 16];
  __shared__ float s_max[16];
  __shared__ unsigned s_wcnt[16];
  __shared__ double s_wsum[16];
  __shared__ double s_thrS[2][257];          // per iteration parity: two barriers per iteration instead of three
  __shared__ unsigned s_thrC[2][257];
  __shared__ double s_part[2][2][FPB_TI / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* src = (v_out != nullptr) ? v_out : a;

  for (int i = tid; i < B; i += FPB_T) {
    cur[i] = 0u;
    isum[i] = 0ull;
  }
  // ---- pass 0: v, sum|v|, max|v| ------------------------------------------------------------------------------
  double sabs = 0.0;
  float mx = 0.0f;
  for (size_t i = tid; i < n; i += FPB_T) {
    const float v = (b2 != nullptr) ? (a[i] + b2[i]) : a[i];
    if (v_out != nullptr) v_out[i] = v;
    sabs += fabs((double)v);
    mx = fmaxf(mx, fabsf(v));
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
  for (int w = 0; w < FPB_T / 64; ++w) {       // every thread adds the wave partials in wave order: same bits everywhere
    tot += s_red[w];
    R = fmaxf(R, s_max[w]);
  }
  if (!(R >= 1e-30f)) R = 1e-30f;
  if (!(R <= 3.0e38f)) R = 3.0e38f;
  FpbGeo g;
  g.R = R;
  g.scale = (float)B / (2.0f * R);
  g.R64 = (double)R;
  g.w = 2.0 * g.R64 / (double)B;
  g.q = ldexp(g.w, -FPB_SHIFT);
  g.rq = 1.0 / g.q;

  // ---- pass 1: bucket counts and integer sums --------------------------------------------------------------------
  for (size_t i = tid; i < n; i += FPB_T) {
    const float v = src[i];
    const int b = fpb_bucket<B>(v, g);
    const long long q = __double2ll_rn(((double)v - ((double)b * g.w - g.R64)) * g.rq);
    atomicAdd(&cur[b], 1u);
    atomicAdd(&isum[b], (unsigned long long)q);
  }
  __syncthreads();
  // ---- exclusive scan over the buckets (fixed shape: deterministic) ---------------------------------------------------
  constexpr int PER = B / FPB_T;               // buckets per thread, contiguous
  static_assert(B % FPB_T == 0, "B must be a multiple of the build width");
  unsigned c_loc[PER];
  double s_loc[PER];
  unsigned c_thr = 0;
  double s_thr = 0.0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int b = tid * PER + j;
    c_loc[j] = cur[b];
    s_loc[j] = (double)c_loc[j] * ((double)b * g.w - g.R64) + g.q * (double)(long long)isum[b];
    c_thr += c_loc[j];
    s_thr += s_loc[j];
  }
  unsigned c_inc = c_thr;
  double s_inc = s_thr;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {           // inclusive wave scan
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
  unsigned c_base = 0;
  double s_base = 0.0;
  for (int w = 0; w < wid; ++w) {
    c_base += s_wcnt[w];
    s_base += s_wsum[w];
  }
  unsigned c_run = c_base + (c_inc - c_thr);
  double s_run = s_base + (s_inc - s_thr);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int b = tid * PER + j;
    off[b] = c_run;
    spre[b] = s_run;
    cur[b] = 0u;
    c_run += c_loc[j];
    s_run += s_loc[j];
  }
  if (tid == FPB_T - 1) {
    off[B] = c_run;                            // = n
    spre[B] = s_run;                           // = sum v
  }
  __syncthreads();
  // ---- pass 2: regroup the values by bucket -----------------------------------------------------------------------------
  for (size_t i = tid; i < n; i += FPB_T) {
    const float v = src[i];
    const int b = fpb_bucket<B>(v, g);
    const unsigned slot = off[b] + atomicAdd(&cur[b], 1u);
    grouped[slot] = v;
  }
  __syncthreads();                             // (one workgroup = one CU: its global stores are visible to its own loads)
  if (tid >= FPB_TI) return;                   // the barrier below only counts the surviving waves

  // ---- the fixed point ---------------------------------------------------------------------------------------------------------
  const int nthr = levels - 1;                 // level boundaries k = 1 .. L-1
  int p2 = 1;
  while (p2 < nthr) p2 <<= 1;
  int gs = FPB_TI / p2;                        // lanes per boundary (a power of two, one wave at most)
  if (gs > 64) gs = 64;
  if (gs < 1) gs = 1;
  const int k = tid / gs + 1, gl = tid % gs;
  const bool active = (k <= nthr);
  double alpha = tot / (double)n, alpha_prev = -999.0;
  double last0 = 0.0, last1 = 0.0;
  int it = 0, done = 0;
  if (tid < 2) {
    s_thrC[tid][0] = 0u;
    s_thrS[tid][0] = 0.0;
    s_thrC[tid][levels] = off[B];
    s_thrS[tid][levels] = spre[B];
  }
  while (!done) {
    const int par = it & 1;
    if (!(alpha > 0.0) || !(alpha < 1e300)) {  // NaN / non-positive scale: the reference would spin to its cap
      done = 2;
      break;
    }
    if (active) {
      const double tk = alpha * (lo + ((double)k - 0.5) * d);
      float c0 = (float)tk;
      c0 = fminf(fmaxf(c0, -3.0e38f), 3.0e38f);
      unsigned key0 = fpb_key(c0);
      unsigned klo = (key0 > FPB_KEY_MIN + 2u) ? key0 - 2u : FPB_KEY_MIN;
      unsigned khi = (key0 < FPB_KEY_MAX - 2u) ? key0 + 2u : FPB_KEY_MAX;
      float vlo = fpb_unkey(klo), vhi = fpb_unkey(khi);
      // widen until level(vlo) < k <= level(vhi) (checked with the exact arithmetic); bounded
      float span = fmaxf((float)g.w, fabsf(c0) * 1e-6f);
      for (int t = 0; t < 48 && fpb_level((double)vlo, alpha, lo, hi, d) >= k; ++t) {
        vlo = fmaxf(c0 - span, -3.0e38f);
        span *= 8.0f;
      }
      span = fmaxf((float)g.w, fabsf(c0) * 1e-6f);
      for (int t = 0; t < 48 && fpb_level((double)vhi, alpha, lo, hi, d) < k; ++t) {
        vhi = fminf(c0 + span, 3.0e38f);
        span *= 8.0f;
      }
      const int jlo = fpb_bucket<B>(vlo, g), jhi = fpb_bucket<B>(vhi, g);
      const unsigned seg0 = off[jlo], seg1 = off[jhi + 1];
      const double base_lo = (double)jlo * g.w - g.R64;
      unsigned cnt = 0;
      long long qs = 0;
      for (unsigned i = seg0 + gl; i < seg1; i += gs) {
        const float v = grouped[i];
        bool below;
        if (v <= vlo)
          below = true;
        else if (v >= vhi)
          below = false;
        else
          below = fpb_level((double)v, alpha, lo, hi, d) < k;
        if (below) {
          ++cnt;
          qs += __double2ll_rn(((double)v - base_lo) * g.rq);
        }
      }
      for (int o = gs >> 1; o > 0; o >>= 1) {    // integer sums: exact, any order
        cnt += __shfl_xor(cnt, o, 64);
        qs += __shfl_xor(qs, o, 64);
      }
      if (gl == 0) {
        s_thrC[par][k] = seg0 + cnt;
        s_thrS[par][k] = spre[jlo] + ((double)cnt * base_lo + g.q * (double)qs);
      }
    }
    __syncthreads();
    // level m holds the values between boundaries m and m+1: sum b*v = sum_m b_m S_m, sum b*b = sum_m b_m^2 C_m
    double acc0 = 0.0, acc1 = 0.0;
    if (tid < levels) {
      const double bm = (double)tid * d + lo;
      const double Sm = s_thrS[par][tid + 1] - s_thrS[par][tid];
      const double Cm = (double)(s_thrC[par][tid + 1] - s_thrC[par][tid]);
      acc0 = bm * Sm;
      acc1 = bm * bm * Cm;
    }
    acc0 = wave_sum(acc0);
    acc1 = wave_sum(acc1);
    if (lane == 0) {
      s_part[par][0][wid] = acc0;
      s_part[par][1][wid] = acc1;
    }
    __syncthreads();
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int w = 0; w < FPB_TI / 64; ++w) {    // every thread adds the wave partials in wave order: same bits everywhere
      t0 += s_part[par][0][w];
      t1 += s_part[par][1][w];
    }
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
  }
}

template <int B>
static constexpr size_t fpb_lds_bytes() {
  return sizeof(unsigned) * ((B + 1) + 1 + B) + sizeof(double) * (B + 1);
}

}  // namespace effq
using namespace effq;

extern "C" {

size_t effq_fp_bucket_max(void) { return (size_t)1 << 19; }

size_t effq_fp_bucket_ws_bytes(size_t n) { return (n + 64) * sizeof(float); }

int effq_fixed_point_bucket(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                            double tol, int max_iter, effq_fp_state* state_dev, void* ws, size_t ws_bytes,
                            void* stream) {
  EFFQ_CHECK_ARG(a && state_dev && ws && n > 0 && levels >= 2 && levels <= 256 && hi > lo && max_iter > 0);
  EFFQ_CHECK_ARG(n <= effq_fp_bucket_max());
  EFFQ_CHECK_ARG(b == nullptr || v_out != nullptr);
  if (ws_bytes < effq_fp_bucket_ws_bytes(n)) {
    set_error("fixed_point_bucket: workspace %zu < %zu bytes", ws_bytes, effq_fp_bucket_ws_bytes(n));
    return EFFQ_ERR_WORKSPACE;
  }
  const double d = (hi - lo) / (double)(levels - 1);
  static bool attr_set = false;
  if (!attr_set) {
    EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fp_bucket<4096>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)fpb_lds_bytes<4096>()));
    attr_set = true;
  }
  hipStream_t st = as_stream(stream);
  float* grouped = reinterpret_cast<float*>(ws);
  if (n <= 32768)
    hipLaunchKernelGGL(k_fp_bucket<2048>, dim3(1), dim3(FPB_T), fpb_lds_bytes<2048>(), st, a, b, v_out, grouped, n,
                       state_dev, lo, hi, d, levels, tol, max_iter);
  else
    hipLaunchKernelGGL(k_fp_bucket<4096>, dim3(1), dim3(FPB_T), fpb_lds_bytes<4096>(), st, a, b, v_out, grouped, n,
                       state_dev, lo, hi, d, levels, tol, max_iter);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
