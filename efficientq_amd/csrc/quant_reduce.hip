// Quantiser primitives and fp64 statistics (HBM-bound streaming kernels).
// Reference: layer_helper.py:25-70 (discretize, project_by_iter), PTQConv.py:114-116.
// All arithmetic follows the reference's operation order with IEEE divisions and no FMA
// contraction (the library is built with -ffp-contract=off).
#include <stdarg.h>
#include <stdlib.h>
#include "common.h"
#include "fp_level.h"
#include "project_dual.h"

namespace effq {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

constexpr int TPB = 256;

static inline int stream_grid(size_t n_vec) {
  size_t b = (n_vec + TPB - 1) / TPB;
  if (b < 1) b = 1;
  if (b > RED_MAX_BLOCKS) b = RED_MAX_BLOCKS;
  return (int)b;
}

// ---- scalar quantiser bodies ----------------------------------------------------------
__device__ __forceinline__ float qd32(float x, float alpha, float lo, float hi, float d, float* idx) {
  float t = x / alpha;
  t = fminf(fmaxf(t, lo), hi);
  // torch.clamp propagates NaN; fmaxf/fminf would drop it
  t = (x != x) ? x : t;
  float r = rintf((t - lo) / d);
  *idx = r;
  return (r * d + lo) * alpha;
}

// Same level index as disc64 (bit-exact), without the two IEEE fp64 divisions on the common path: the
// quotient is formed with reciprocals (a few ulp off) and accepted only when it is provably on the same
// side of every rounding boundary as the exact one; otherwise the exact divisions are redone.
__device__ __forceinline__ double disc64_fast(double x, double alpha, double ralpha, double lo, double hi, double d,
                                              double rd, double* idx) {
  double t = fmin(fmax(x * ralpha, lo), hi);
  const double u = (t - lo) * rd;
  const double fr = u - floor(u);
  // |u_exact - u| <= ~8 ulp(u) + the clamp edges; 1e-9 is far above that and far below any real margin
  // (a few-ulp change of t at a clamp edge moves u by a few ulp next to an INTEGER, which rint absorbs)
  const bool safe = fabs(fr - 0.5) > 1e-9 * (1.0 + u);
  if (safe) {
    const double r = rint(u);
    *idx = r;
    return r * d + lo;
  }
  return disc64(x, alpha, lo, hi, d, idx);
}

// Statistics of one fixed-point pass without per-value fp64 arithmetic beyond one multiply-add: b = r d + lo depends on
// the level index r alone, so  sum b x = d sum(r x) + lo sum(x)  and  sum b^2 = d^2 sum(r^2) + 2 d lo sum(r) + lo^2 n,
// with sum(r), sum(r^2) exact integers and r x exact in fp64.  r comes from an fp32 evaluation of u = (x/alpha - lo)/d
// (error <= 3e-5 at 256 levels), accepted unless u lies within 2e-4 of a rounding boundary, where the reference's own
// fp64 arithmetic (disc64) decides: the level indices are exactly the reference's.
// four doubles of scratch in the tail of the reduction workspace (after the ticket and the cooperative kernel's two
// counter words at +64 / +68): the raw totals of a level-statistics pass before level_finish
__device__ __forceinline__ double* level_scratch(double* partials) {
  return reinterpret_cast<double*>(reinterpret_cast<char*>(partials) + sizeof(double) * RED_MAX_BLOCKS * RED_SLOTS + 128);
}
struct LevelStats {
  double arx, sx;          // sum r x, sum x (sx only when lo != 0)
  long long sr, sr2;       // sum r, sum r^2
};
__device__ __forceinline__ void level_accum(float xf, const LevelConsts& c, LevelStats& a) {
  float u = __builtin_fmaf(xf, c.c1, c.c0);
  u = fminf(fmaxf(u, 0.0f), c.lmax);
  float rf = rintf(u);
  if (!(fabsf(u - rf) < 0.4998f)) {
    double r;
    disc64((double)xf, c.alpha, c.lo, c.hi, c.d, &r);
    rf = (float)r;
  }
  const int ri = (int)rf;
  a.sr += ri;
  a.sr2 += ri * ri;
  a.arx = __builtin_fma((double)rf, (double)xf, a.arx);
  if (c.need_sx) a.sx += (double)xf;
}
// [sum r x, sum r, sum r^2, sum x] over n values -> [sum b x, sum b b]
__device__ __forceinline__ void level_finish(const double* t4, size_t n, double lo, double d, double* out2) {
  out2[0] = d * t4[0] + lo * t4[3];
  out2[1] = (d * d * t4[2] + 2.0 * d * lo * t4[1]) + lo * lo * (double)n;
}

__global__ __launch_bounds__(TPB) void k_quant_dequant_f32(const float* __restrict__ x,
                                                           const float* __restrict__ alpha_dev, float lo,
                                                           float hi, float d, float* __restrict__ y,
                                                           uint8_t* __restrict__ idx, size_t n) {
  const float alpha = *alpha_dev;
  const size_t nv = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    float r0, r1, r2, r3;
    float4 o;
    o.x = qd32(v.x, alpha, lo, hi, d, &r0);
    o.y = qd32(v.y, alpha, lo, hi, d, &r1);
    o.z = qd32(v.z, alpha, lo, hi, d, &r2);
    o.w = qd32(v.w, alpha, lo, hi, d, &r3);
    if (y) reinterpret_cast<float4*>(y)[i] = o;
    if (idx) {
      uchar4 u = make_uchar4((unsigned char)r0, (unsigned char)r1, (unsigned char)r2, (unsigned char)r3);
      reinterpret_cast<uchar4*>(idx)[i] = u;
    }
  }
  // ragged tail
  for (size_t i = nv * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float r;
    float o = qd32(x[i], alpha, lo, hi, d, &r);
    if (y) y[i] = o;
    if (idx) idx[i] = (unsigned char)r;
  }
}

__global__ __launch_bounds__(TPB) void k_quant_dequant_f64path(const float* __restrict__ x,
                                                               const double* __restrict__ alpha_dev, double lo,
                                                               double hi, double d, float* __restrict__ y,
                                                               float* __restrict__ bout,
                                                               uint8_t* __restrict__ idx, size_t n) {
  const double alpha = *alpha_dev;
  const float alpha32 = (float)alpha;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t nv = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    double r0, r1, r2, r3;
    float4 b;
    b.x = (float)disc64((double)v.x, alpha, lo, hi, d, &r0);
    b.y = (float)disc64((double)v.y, alpha, lo, hi, d, &r1);
    b.z = (float)disc64((double)v.z, alpha, lo, hi, d, &r2);
    b.w = (float)disc64((double)v.w, alpha, lo, hi, d, &r3);
    if (bout) reinterpret_cast<float4*>(bout)[i] = b;
    if (y) {
      float4 o = make_float4(alpha32 * b.x, alpha32 * b.y, alpha32 * b.z, alpha32 * b.w);
      reinterpret_cast<float4*>(y)[i] = o;
    }
    if (idx)
      reinterpret_cast<uchar4*>(idx)[i] =
          make_uchar4((unsigned char)r0, (unsigned char)r1, (unsigned char)r2, (unsigned char)r3);
  }
  for (size_t i = nv * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double r;
    float b = (float)disc64((double)x[i], alpha, lo, hi, d, &r);
    if (bout) bout[i] = b;
    if (y) y[i] = alpha32 * b;
    if (idx) idx[i] = (unsigned char)r;
  }
}

// ---- fp64 reductions ---------------------------------------------------------------------
// MODE 0: sum|x|, n     MODE 1: sum x, sum x^2, n     MODE 2: sum b*x, sum b*b  (b=discretize(x/alpha))
template <int MODE>
__global__ __launch_bounds__(TPB) void k_reduce(const float* __restrict__ x, size_t n,
                                                const double* __restrict__ alpha_dev, double lo, double hi,
                                                double d, const int32_t* __restrict__ done_flag,
                                                double* partials, unsigned int* ticket, double* out) {
  constexpr int NS = (MODE == 1) ? 3 : (MODE == 2) ? 4 : 2;
  __shared__ double smem[NS * 16];
  __shared__ int s_last;
  if (MODE == 2 && done_flag != nullptr && *done_flag != 0) return;  // uniform across the grid
  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0.0;
  LevelStats ls = {0.0, 0.0, 0, 0};
  LevelConsts lc = level_consts((MODE == 2) ? *alpha_dev : 1.0, lo, hi, (MODE == 2) ? d : 1.0);
  const size_t nv = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  auto body = [&](float xf) {
    if (MODE == 0) {
      acc[0] += fabs((double)xf);
    } else if (MODE == 1) {
      const double v = (double)xf;
      acc[0] += v;
      acc[1] += v * v;
    } else {
      level_accum(xf, lc, ls);
    }
  };
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    body(v.x);
    body(v.y);
    body(v.z);
    body(v.w);
  }
  for (size_t i = nv * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) body(x[i]);
  if (MODE != 2) {
    if (blockIdx.x == 0 && threadIdx.x == 0) acc[NS - 1] = (double)n;
    grid_sum_finish<NS>(acc, partials, ticket, out, smem, &s_last);
  } else {
    acc[0] = ls.arx; acc[1] = (double)ls.sr; acc[2] = (double)ls.sr2; acc[NS - 1] = ls.sx;
    double* t4 = level_scratch(partials);
    grid_sum_finish<NS>(acc, partials, ticket, t4, smem, &s_last);
    if (s_last && threadIdx.x == 0) level_finish(t4, n, lo, d, out);
  }
}


// ---- fused fixed-point iteration: statistics pass whose last-arriving block also performs the scalar
// update (layer_helper.py:55-60), so one launch = one iteration.  Used where no all-reduce sits between
// the two (replicated weights; single-GPU activations).
__global__ __launch_bounds__(TPB) void k_fp_iter(const float* __restrict__ x, size_t n, effq_fp_state* st, double lo,
                                                 double hi, double d, double tol, int max_iter, double* partials,
                                                 unsigned int* ticket) {
  __shared__ double smem[4 * 16];
  __shared__ int s_last;
  if (st->done != 0) return;  // uniform across the grid
  const double alpha = st->alpha;
  const LevelConsts lc = level_consts(alpha, lo, hi, d);
  LevelStats ls = {0.0, 0.0, 0, 0};
  const size_t nv = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    level_accum(v.x, lc, ls);
    level_accum(v.y, lc, ls);
    level_accum(v.z, lc, ls);
    level_accum(v.w, lc, ls);
  }
  for (size_t i = nv * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) level_accum(x[i], lc, ls);
  double acc[4] = {ls.arx, (double)ls.sr, (double)ls.sr2, ls.sx};
  double* t4 = level_scratch(partials);
  grid_sum_finish<4>(acc, partials, ticket, t4, smem, &s_last);
  if (s_last && threadIdx.x == 0) level_finish(t4, n, lo, d, st->sums);
  // the finishing block has written the sums into st->sums; it then applies the update
  if (s_last && threadIdx.x == 0) {
    const double a_new = st->sums[0] / st->sums[1];
    st->alpha_prev = alpha;
    st->alpha = a_new;
    const int it = st->iters + 1;
    st->iters = it;
    if (it >= max_iter)
      st->done = 2;
    else if (!(fabs(a_new - alpha) > tol))
      st->done = 1;
  }
}

// ---- whole project_by_iter in ONE launch for small tensors (weights of most layers): a single
// 1024-thread workgroup computes mean|v|, then iterates statistics + update until convergence or the
// cap, all on chip.  v = a + b2 (b2 may be NULL) is formed on the fly and optionally stored to v_out.
constexpr int FPS_T = 1024;
// One barrier per iteration: every wave publishes its two partial sums into a parity-double-buffered LDS table,
// and EVERY thread adds the table in wave order and takes the division itself (same bits everywhere), so there is
// no serial thread-0 section and no broadcast barrier.  At 256 levels the weight fixed point of the first conv
// runs ~300 iterations per ADMM iteration: the per-iteration latency (2.2 us with three barriers) is what counts.
// PER = register slots per thread (compile time, so the element loop is branch-free and the fp64 chains of the
// slots interleave).
// Arithmetic per value: b = r d + lo is a function of the level index r alone, so
//   sum b v = d sum(r v) + lo sum(v),   sum b^2 = d^2 sum(r^2) + 2 d lo sum(r) + lo^2 n
// with sum(r), sum(r^2) exact integers and r v exact in fp64: per value ONE fp64 multiply-add instead of the ~16 fp64
// operations of disc64_fast + two accumulations (fp64 min/max/rint/floor issue at a fraction of the fp32 rate; at 256
// levels the first conv's weight scale takes ~290 iterations per ADMM iteration).  r comes from an fp32 evaluation of
// u = (v/alpha - lo)/d (error <= 3e-5 at 256 levels) and is accepted when u is not within 2e-4 of a rounding boundary;
// otherwise (2e-4 of the values) the reference's own fp64 arithmetic (disc64) decides: the level indices are exact.
template <int T, int PER>
__global__ __launch_bounds__(T) void k_fp_small(const float* __restrict__ a, const float* b2, float* v_out, size_t n,
                                                effq_fp_state* st, double lo, double hi, double d, double tol,
                                                int max_iter, ProjFused pf) {
  constexpr int NW = T / 64;
  __shared__ double part[2][3][NW];
  // one workgroup on the critical path of the ADMM chain, sharing its CU with the waves of the loss conv of the previous
  // iterate: ask the issue arbiter for priority
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kmax = (int)((n + T - 1) / T);   // live register slots (uniform)
  float vr[PER];
  unsigned live = 0;
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const size_t i = (size_t)tid + (size_t)k * T;
    float v = 0.0f;
    if (i < n) {
      v = (b2 != nullptr) ? (a[i] + b2[i]) : a[i];
      if (v_out != nullptr) v_out[i] = v;
      live |= 1u << k;
    }
    vr[k] = v;
    acc0 += fabs((double)v);
    acc1 += (double)v;
  }
  acc0 = wave_sum_f64_dpp(acc0);
  acc1 = wave_sum_f64_dpp(acc1);
  if (lane == 0) {
    part[0][0][wid] = acc0;
    part[0][1][wid] = acc1;
  }
  lds_barrier();
  double tot = 0.0, sv = 0.0;                // sum |v|, sum v
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    tot += part[0][0][w];
    sv += part[0][1][w];
  }
  double alpha = tot / (double)n, alpha_prev = -999.0;
  double ralpha = (double)n / tot;           // a reciprocal good to a few ulp is all the fast path needs
  double last0 = 0.0, last1 = 0.0;
  int it = 0, done = 0;
  const double rd = 1.0 / d;
  const float c0 = (float)(-lo * rd), lmax = (float)rint((hi - lo) * rd);
  const double lo_sv = lo * sv, lo2n = lo * lo * (double)n, d2 = d * d, dlo2 = 2.0 * d * lo;
  while (!done) {
    const int par = (it + 1) & 1;            // parity 0 carried the prologue sums
    const float c1 = (float)(ralpha * rd);
    double arv = 0.0;
    int sr = 0, sr2 = 0;                     // <= 32 slots x 255^2 per thread
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      // 1024 threads leave 128 VGPRs: interleaving the chains of all 32 slots spills there, so that variant keeps a
      // (uniform) branch per slot; (a branch-free common path with the exact fallback hoisted out measured slower)
      if (T < 1024 || k < kmax) {
        const float vf = vr[k];
        float u = __builtin_fmaf(vf, c1, c0);
        u = fminf(fmaxf(u, 0.0f), lmax);
        float rf = rintf(u);
        if (!(fabsf(u - rf) < 0.4998f)) {    // within 2e-4 of a rounding boundary (or NaN): exact arithmetic decides
          double r;
          disc64((double)vf, alpha, lo, hi, d, &r);
          rf = (float)r;
        }
        const int ri = ((live >> k) & 1u) ? (int)rf : 0;      // dead slots hold v = 0: they must not count
        sr += ri;
        sr2 += ri * ri;
        arv = __builtin_fma((double)rf, (double)vf, arv);     // r v is exact in fp64 (8 + 24 bits)
      }
    }
    arv = wave_sum_f64_dpp(arv);
    const unsigned wr = group_sum_u32((unsigned)sr, 64), wr2 = group_sum_u32((unsigned)sr2, 64);
    if (lane == 0) {
      part[par][0][wid] = arv;
      part[par][1][wid] = (double)wr;
      part[par][2][wid] = (double)wr2;
    }
    lds_barrier();
    double trv = 0.0, tr = 0.0, tr2 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      trv += part[par][0][w];
      tr += part[par][1][w];
      tr2 += part[par][2][w];
    }
    const double t0 = d * trv + lo_sv;                         // sum b v
    const double t1 = (d2 * tr2 + dlo2 * tr) + lo2n;           // sum b^2
    const double a_new = t0 / t1;
    const double ra_new = t1 / t0;           // independent of the division above (pipelines with it)
    ++it;
    if (it >= max_iter)
      done = 2;
    else if (!(fabs(a_new - alpha) > tol))
      done = 1;
    alpha_prev = alpha;
    alpha = a_new;
    ralpha = ra_new;
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
  if (pf.G != nullptr) {                       // the projection + dual update of this ADMM iteration, same launch
    __syncthreads();                           // v_out of every thread is in place
    proj_fused_epilogue(pf, v_out, alpha, done, tid, T);
  }
}

// ---- cooperative whole-fixed-point kernel for larger tensors --------------------------------------
// G <= 256 workgroups of 1024 threads, one per CU, each owning a contiguous slice of v that stays in LDS for
// all iterations.  Per iteration every workgroup publishes its two partial sums, all meet at a grid
// barrier, and EVERY workgroup adds the G partials in workgroup order (identical alpha everywhere, run-to-
// run and rank-to-rank deterministic -- replicated data-parallel ranks must stay bit-identical).
// Grid barrier: monotonic agent-scope counter, release fence before the arrive, relaxed polling with
// s_sleep, acquire fence after (cdna_hip_programming.md Guideline 16 / microarch "barrier-counter").
// Every spin is bounded: on time-out the state is marked done=3 and all workgroups leave.
constexpr int FPC_T = 1024;
constexpr int FPC_SLICE = 27648;          // floats per workgroup kept in LDS (108 KiB)
constexpr int FPC_MAXG = 256;          // one workgroup per CU at most: 7.08 M values = 512 x 512 x 27 weights
constexpr unsigned FPC_SPIN_LIMIT = 1u << 24;
static unsigned g_fpc_spin_limit = FPC_SPIN_LIMIT;     // effq_fp_coop_set_spin_limit (test hook)

// counter[0] = arrivals, counter[1] = check-outs, counter[2] = POISON: set by the first workgroup whose barrier times
// out.  A poisoned workspace turns every later launch into a no-op that reports done = 3 (the launches of the following
// ADMM iterations are already enqueued when a time-out happens, and the arrival counter is left non-zero by the early
// exits: they must not run on it); the host clears the workspace when it sees the error (qconv.ptq).
// INVARIANT of the exchange through fpc_barrier: whatever workgroups hand to each other across it is WRITTEN with
// fpc_publish (agent-scope atomic store: write-through, no stale line left in the writer's L2) before the barrier and
// READ with agent-scope atomic loads after it - never with plain loads: the barrier issues a release fence but NO acquire
// fence (see below), so a plain load could be served from this CU's L1.  The lock-step of data-parallel replicas rests
// on this (tests/test_configs_gpu.py: test_config2_at_its_stated_size_is_deterministic, in the default GPU selection).
__device__ __forceinline__ void fpc_publish(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool fpc_barrier(unsigned int* counter, unsigned target, int* s_fail, unsigned spin_limit) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > spin_limit ||
          __hip_atomic_load(counter + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(counter + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_fail = 1;
        break;
      }
    }
    // No acquire fence here: everything the workgroups exchange through this barrier (the partial sums) is read with
    // agent-scope atomic loads, which are served by L2, so the L1 invalidation an acquire fence performs (buffer_inv sc1:
    // ~1.7 us per barrier, 14 barriers per call) would only protect data nobody reads; the poll above has completed
    // (its value was consumed) before any of those loads is issued.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return *s_fail == 0;
}

template <int T>
__global__ __launch_bounds__(T) void k_fp_coop(const float* __restrict__ a, const float* __restrict__ b2,
                                                   float* __restrict__ v_out, size_t n, effq_fp_state* st, double lo,
                                                   double hi, double d, double tol, int max_iter, double* partials,
                                                   unsigned int* counter, unsigned spin_limit, FptPred* pred,
                                                   int levels) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams
  // a workspace poisoned by an earlier time-out: report and leave, touching nothing (uniform across the grid: the
  // poison word only ever goes 0 -> 1 before this launch started, or during it - then the barrier below catches it)
  if (__hip_atomic_load(counter + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st->done = 3;
    return;
  }

  extern __shared__ __attribute__((aligned(16))) float vs[];      // this workgroup's slice of v
  constexpr int NW = T / 64;
  __shared__ double s_wave[3][NW];
  __shared__ double s_tot[3];
  __shared__ int s_fail;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, G = gridDim.x, wg = blockIdx.x;
  const size_t per = (n + G - 1) / G;
  const size_t s0 = (size_t)wg * per, s1 = (s0 + per < n) ? s0 + per : n;
  const int cnt = (s1 > s0) ? (int)(s1 - s0) : 0;
  if (tid == 0) s_fail = 0;
  // workgroup sums of three doubles -> thread 0 (DPP wave sums, one LDS exchange; fixed tree: deterministic)
  auto wg_sum3 = [&](double& x0, double& x1, double& x2) {
    x0 = wave_sum_f64_dpp(x0);
    x1 = wave_sum_f64_dpp(x1);
    x2 = wave_sum_f64_dpp(x2);
    if (lane == 0) {
      s_wave[0][wid] = x0;
      s_wave[1][wid] = x1;
      s_wave[2][wid] = x2;
    }
    __syncthreads();
    if (tid == 0) {
      double u0 = 0.0, u1 = 0.0, u2 = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        u0 += s_wave[0][w];
        u1 += s_wave[1][w];
        u2 += s_wave[2][w];
      }
      x0 = u0;
      x1 = u1;
      x2 = u2;
    }
  };
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  for (int i = tid; i < cnt; i += T) {
    const float v = (b2 != nullptr) ? (a[s0 + i] + b2[s0 + i]) : a[s0 + i];
    if (v_out != nullptr) v_out[s0 + i] = v;
    vs[i] = v;
    acc0 += fabs((double)v);
    acc1 += (double)v;
  }
  wg_sum3(acc0, acc1, acc2);
  unsigned epoch = 0;
  // partials layout: [parity][wg][3]
  if (tid == 0) {
    fpc_publish(&partials[(0 * FPC_MAXG + wg) * 3 + 0], acc0);
    fpc_publish(&partials[(0 * FPC_MAXG + wg) * 3 + 1], acc1);
    fpc_publish(&partials[(0 * FPC_MAXG + wg) * 3 + 2], 0.0);
  }
  if (!fpc_barrier(counter, (++epoch) * (unsigned)G, &s_fail, spin_limit)) {
    if (tid == 0) st->done = 3;          // (any workgroup: workgroup 0 may have left through the poison check)
    return;
  }
  // the G partials are fetched by the lanes of wave 0 (agent-scope loads, lane l takes workgroups l, l + 64, ...) and
  // added by a fixed DPP tree: the same bits in every workgroup and run to run; the totals go to all threads through LDS
  auto combine = [&](int par, double& t0, double& t1, double& t2) {
    if (wid == 0) {
      double u0 = 0.0, u1 = 0.0, u2 = 0.0;
      for (int g = lane; g < G; g += 64) {
        u0 += __hip_atomic_load(&partials[(par * FPC_MAXG + g) * 3 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u1 += __hip_atomic_load(&partials[(par * FPC_MAXG + g) * 3 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u2 += __hip_atomic_load(&partials[(par * FPC_MAXG + g) * 3 + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      u0 = wave_sum_f64_dpp(u0);
      u1 = wave_sum_f64_dpp(u1);
      u2 = wave_sum_f64_dpp(u2);
      if (lane == 0) {
        s_tot[0] = u0;
        s_tot[1] = u1;
        s_tot[2] = u2;
      }
    }
    __syncthreads();
    t0 = s_tot[0];
    t1 = s_tot[1];
    t2 = s_tot[2];
    __syncthreads();
  };
  double tot = 0.0, sv = 0.0, tdummy = 0.0;
  combine(0, tot, sv, tdummy);
  double alpha = tot / (double)n, alpha_prev = -999.0;
  int it = 0, done = 0;
  double last0 = 0.0, last1 = 0.0;
  // per value: level index r from an fp32 evaluation (exact fp64 arithmetic within 2e-4 of a rounding boundary), then
  // sum b v = d sum(r v) + lo sum(v), sum b^2 = d^2 sum(r^2) + 2 d lo sum(r) + lo^2 n (see k_fp_small)
  const double rd = 1.0 / d;
  const float c0 = (float)(-lo * rd), lmax = (float)rint((hi - lo) * rd);
  const double lo_sv = lo * sv, lo2n = lo * lo * (double)n, d2 = d * d, dlo2 = 2.0 * d * lo;
  while (!done) {
    const int par = (it + 1) & 1;          // parity 0 was used by the abs-sum epoch
    if (wg == 0 && tid == 0) fpt_note(pred, it, alpha);      // (seeds the next call's predictions: fixed_point_traj.hip)
    const float c1 = (float)((1.0 / alpha) * rd);
    double arv = 0.0;
    long long sr = 0, sr2 = 0;
    for (int i = tid; i < cnt; i += T) {
      const float vf = vs[i];
      float u = __builtin_fmaf(vf, c1, c0);
      u = fminf(fmaxf(u, 0.0f), lmax);
      float rf = rintf(u);
      if (!(fabsf(u - rf) < 0.4998f)) {
        double r;
        disc64((double)vf, alpha, lo, hi, d, &r);
        rf = (float)r;
      }
      const int ri = (int)rf;
      sr += ri;
      sr2 += ri * ri;
      arv = __builtin_fma((double)rf, (double)vf, arv);
    }
    double dr = (double)sr, dr2 = (double)sr2;
    wg_sum3(arv, dr, dr2);
    if (tid == 0) {
      fpc_publish(&partials[(par * FPC_MAXG + wg) * 3 + 0], arv);
      fpc_publish(&partials[(par * FPC_MAXG + wg) * 3 + 1], dr);
      fpc_publish(&partials[(par * FPC_MAXG + wg) * 3 + 2], dr2);
    }
    if (!fpc_barrier(counter, (++epoch) * (unsigned)G, &s_fail, spin_limit)) {
      if (tid == 0) st->done = 3;
      return;
    }
    double trv = 0.0, tr = 0.0, tr2 = 0.0;
    combine(par, trv, tr, tr2);
    const double t0 = d * trv + lo_sv;                         // sum b v
    const double t1 = (d2 * tr2 + dlo2 * tr) + lo2n;           // sum b^2
    const double a_new = t0 / t1;
    alpha_prev = alpha;
    ++it;
    if (it >= max_iter)
      done = 2;
    else if (!(fabs(a_new - alpha) > tol))
      done = 1;
    alpha = a_new;
    last0 = t0;
    last1 = t1;
  }
  if (wg == 0 && tid == 0) {
    st->alpha = alpha;
    st->alpha_prev = alpha_prev;
    st->sums[0] = last0;
    st->sums[1] = last1;
    st->iters = it;
    st->done = done;
    if (pred != nullptr) {
      for (int j = 0; j < FPT_SLOTS; ++j) fpt_finish_slot(pred, j, it, alpha);
      fpt_finish_head(pred, it, tot, levels);
    }
  }
  // leave the barrier counter at zero for the next launch: every workgroup is past its last poll when it gets
  // here, so the last one to check out (counter[1]) resets both words - no memset command per call.  (After a
  // barrier time-out the early returns above skip this; the host then sees done = 3 and raises.)
  if (tid == 0) {
    const unsigned left = __hip_atomic_fetch_add(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == (unsigned)G - 1) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void k_check_state(const effq_fp_state* st, int32_t* err_flag) {
  if (st->done != 1) *err_flag = (st->done == 2) ? 2 : 3;
}

__global__ void k_fp_init(effq_fp_state* st, const double* abs_sums) {
  st->alpha = abs_sums[0] / abs_sums[1];
  st->alpha_prev = -999.0;
  st->sums[0] = 0.0;
  st->sums[1] = 0.0;
  st->iters = 0;
  st->done = 0;
}

__global__ void k_fp_update(effq_fp_state* st, double tol, int max_iter) {
  if (st->done) return;
  // loop head of layer_helper.py:55: while abs(a - a_prev) > 1e-5 and c < max_iter
  double a_new = st->sums[0] / st->sums[1];
  st->alpha_prev = st->alpha;
  st->alpha = a_new;
  st->iters += 1;
  // the reference raises whenever c == max_iter, even if that last step converged (:62-64)
  if (st->iters >= max_iter)
    st->done = 2;
  else if (!(fabs(st->alpha - st->alpha_prev) > tol))
    st->done = 1;
}

__global__ __launch_bounds__(TPB) void k_presum(const float* __restrict__ a, const float* __restrict__ b,
                                                float* __restrict__ o, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) o[i] = a[i] + b[i];
}

__global__ __launch_bounds__(TPB) void k_project_dual(const float* __restrict__ v, const float* __restrict__ wstar,
                                                      const effq_fp_state* __restrict__ st, double d,
                                                      float* __restrict__ G, float* __restrict__ dual,
                                                      float dual_div, int8_t* __restrict__ Gq, int lm1, size_t n,
                                                      int32_t* __restrict__ err_flag, ProjNext nx) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams

  // (optional) the convergence check of the fixed point that produced `st`, folded in to save a launch
  if (err_flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && st->done != 1) *err_flag = (st->done == 2) ? 2 : 3;
  const double alpha = st->alpha;
  const float alpha32 = (float)alpha;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double r;
    float b = (float)disc64((double)v[i], alpha, -1.0, 1.0, d, &r);
    float g = alpha32 * b;
    G[i] = g;
    // int8 operand of the exact-integer convs: the signed numerator j' = 2*level - (L-1), or, beyond 128
    // levels where that no longer fits, level - 128 (conv3d_i8s.hip rebuilds j' = 2*(level-128) + 1)
    if (Gq != nullptr) Gq[i] = (lm1 >= 128) ? (int8_t)((int)r - 128) : (int8_t)(2 * (int)r - lm1);
    float du = (wstar[i] - g) + dual[i];        // EfficientQConv.py:111
    if (dual_div != 1.0f) du = du / dual_div;   // "dual /= 2" or "dual /= rho_m/rho" (:131-136)
    dual[i] = du;
    if (nx.Bm != nullptr) {                     // right-hand side of the NEXT prox solve (k_build_b4's arithmetic)
      const size_t r = i / (size_t)nx.nwrow, k = i - r * (size_t)nx.nwrow;
      float bv = nx.B0[r * (size_t)nx.n + k] + nx.eta * nx.W0[i];
      bv = bv + nx.rho * (g - du);
      nx.Bm[r * (size_t)nx.ldb + k] = bv;
    }
  }
}

// Four consecutive weights per thread (weight rows that are a multiple of 4 long: every layer of the shipped nets): 16-byte
// accesses, one (row, column) split per thread with 32-bit arithmetic, and the level index from the fp32 evaluation of
// level_accum (the reference's fp64 arithmetic decides within 2e-4 of a rounding boundary: indices are exact).
__global__ __launch_bounds__(TPB) void k_project_dual4(const float* __restrict__ v, const float* __restrict__ wstar,
                                                       const effq_fp_state* __restrict__ st, double d,
                                                       float* __restrict__ G, float* __restrict__ dual,
                                                       float dual_div, int8_t* __restrict__ Gq, int lm1, unsigned n4,
                                                       int32_t* __restrict__ err_flag, ProjNext nx) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams
  if (err_flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && st->done != 1) *err_flag = (st->done == 2) ? 2 : 3;
  const double alpha = st->alpha;
  const float alpha32 = (float)alpha;
  const LevelConsts lc = level_consts(alpha, -1.0, 1.0, d);
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride)
    proj4_apply(q, v, wstar, alpha, alpha32, lc, d, G, dual, dual_div, Gq, lm1, nx);
}

// ---- backward of PTQConv._quantize_act with the straight-through estimator (row f3) ------------------------------------
// q = discretize(x / alpha, L, 0, 1) * alpha (PTQConv.py:114-116), round with identity gradient (layer_helper.py:13-22),
// clamp with torch's gradient mask (1 where lo <= u <= hi, bounds included).  With u = x / alpha, r = discretize(u):
//   dq/dx = mask,    dq/dalpha = r - mask * u          =>   gx = gq * mask,   galpha = sum gq * (r - mask * u)
// (the chain rule autograd applies to the reference's five elementwise ops, collected into one pass).
__global__ __launch_bounds__(TPB) void k_act_quant_bwd(const float* __restrict__ x, const float* __restrict__ alpha_dev,
                                                       float lo, float hi, float d, const float* __restrict__ gq,
                                                       float* __restrict__ gx, size_t n, double* partials,
                                                       unsigned int* ticket, double* galpha_out) {
  __shared__ double smem[16];
  __shared__ int s_last;
  const float alpha = *alpha_dev;
  double acc[1] = {0.0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float u = x[i] / alpha;
    const float c = fminf(fmaxf(u, lo), hi);
    const float r = rintf((c - lo) / d) * d + lo;
    const float m = (u >= lo && u <= hi) ? 1.0f : 0.0f;
    const float g = gq[i];
    if (gx != nullptr) gx[i] = g * m;
    acc[0] += (double)(g * (r - m * u));
  }
  grid_sum_finish<1>(acc, partials, ticket, galpha_out, smem, &s_last);
}

__global__ __launch_bounds__(TPB) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, float lr, float b1,
                                              float b2, float eps, float bc1, float bc2, size_t n) {
  // torch.optim.Adam (no weight decay, no amsgrad): ptqer.py:255 Adam(opt_param, lr=5e-4)
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gi = g[i];
    float mi = m[i] + (1.0f - b1) * (gi - m[i]);          // lerp form used by torch
    float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
  }
}

// ---- bit-packed storage of level ids (row f2: the reference stores one uint8 per weight, PTQConv.py:125-152) ----
// element i occupies bits [i*bits, (i+1)*bits) of the little-endian bit stream; bits in {1, 2, 4, 8}
__global__ __launch_bounds__(TPB) void k_pack_levels(const uint8_t* __restrict__ idx, size_t n, int bits,
                                                     uint8_t* __restrict__ packed, size_t nbytes) {
  const int per = 8 / bits;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < nbytes; b += stride) {
    unsigned v = 0;
    for (int k = 0; k < per; ++k) {
      const size_t i = b * per + k;
      if (i < n) v |= ((unsigned)idx[i] & ((1u << bits) - 1u)) << (k * bits);
    }
    packed[b] = (uint8_t)v;
  }
}

__global__ __launch_bounds__(TPB) void k_unpack_levels(const uint8_t* __restrict__ packed, size_t n, int bits,
                                                       uint8_t* __restrict__ idx) {
  const int per = 8 / bits;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    idx[i] = (uint8_t)((packed[i / per] >> ((i % per) * bits)) & ((1u << bits) - 1u));
}

}  // namespace effq

using namespace effq;

extern "C" {

const char* effq_last_error(void) { return g_err; }
int effq_version(void) { return 100; }

int effq_device_count(int* count) {
  EFFQ_CHECK_ARG(count != nullptr);
  EFFQ_HIP(hipGetDeviceCount(count));
  return EFFQ_OK;
}

size_t effq_reduce_ws_bytes(void) { return RED_WS_BYTES; }

int effq_quant_dequant_f32(const float* x, const float* alpha_dev, float lo, float hi, int levels, float* y_out,
                           uint8_t* idx_out, size_t n, void* stream) {
  if (n == 0) return EFFQ_OK;  /* empty tensors are legal (and carry null pointers) */
  EFFQ_CHECK_ARG(x && alpha_dev && levels >= 2 && hi > lo);
  EFFQ_CHECK_ARG(idx_out == nullptr || levels <= 256);
  const float d = (float)(((double)hi - (double)lo) / (double)(levels - 1));
  hipLaunchKernelGGL(k_quant_dequant_f32, dim3(stream_grid((n + 3) / 4)), dim3(TPB), 0, as_stream(stream), x,
                     alpha_dev, lo, hi, d, y_out, idx_out, n);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_quant_dequant_f64path(const float* x, const double* alpha_dev, double lo, double hi, int levels,
                               float* y_out, float* b_out, uint8_t* idx_out, size_t n, void* stream) {
  if (n == 0) return EFFQ_OK;  /* empty tensors are legal (and carry null pointers) */
  EFFQ_CHECK_ARG(x && alpha_dev && levels >= 2 && hi > lo);
  EFFQ_CHECK_ARG(idx_out == nullptr || levels <= 256);
  const double d = (hi - lo) / (double)(levels - 1);
  hipLaunchKernelGGL(k_quant_dequant_f64path, dim3(stream_grid((n + 3) / 4)), dim3(TPB), 0, as_stream(stream), x,
                     alpha_dev, lo, hi, d, y_out, b_out, idx_out, n);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_abs_sum_f64(const float* x, size_t n, double* sums_out, void* ws, void* stream) {
  EFFQ_CHECK_ARG(x && sums_out && ws && n > 0);
  RedWs r = red_ws(ws);
  hipLaunchKernelGGL(k_reduce<0>, dim3(stream_grid((n + 3) / 4)), dim3(TPB), 0, as_stream(stream), x, n,
                     (const double*)nullptr, 0.0, 0.0, 0.0, (const int32_t*)nullptr, r.partials, r.ticket,
                     sums_out);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_moments_f64(const float* x, size_t n, double* sums_out, void* ws, void* stream) {
  EFFQ_CHECK_ARG(x && sums_out && ws && n > 0);
  RedWs r = red_ws(ws);
  hipLaunchKernelGGL(k_reduce<1>, dim3(stream_grid((n + 3) / 4)), dim3(TPB), 0, as_stream(stream), x, n,
                     (const double*)nullptr, 0.0, 0.0, 0.0, (const int32_t*)nullptr, r.partials, r.ticket,
                     sums_out);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_alpha_stats_f64(const float* x, const double* alpha_dev, double lo, double hi, int levels, size_t n,
                         double* sums_out, const int32_t* done_flag_dev, void* ws, void* stream) {
  EFFQ_CHECK_ARG(x && alpha_dev && sums_out && ws && n > 0 && levels >= 2 && hi > lo);
  RedWs r = red_ws(ws);
  const double d = (hi - lo) / (double)(levels - 1);
  hipLaunchKernelGGL(k_reduce<2>, dim3(stream_grid((n + 3) / 4)), dim3(TPB), 0, as_stream(stream), x, n, alpha_dev,
                     lo, hi, d, done_flag_dev, r.partials, r.ticket, sums_out);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_fp_init(effq_fp_state* state_dev, const double* abs_sums_dev, void* stream) {
  EFFQ_CHECK_ARG(state_dev && abs_sums_dev);
  hipLaunchKernelGGL(k_fp_init, dim3(1), dim3(1), 0, as_stream(stream), state_dev, abs_sums_dev);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_fp_update(effq_fp_state* state_dev, double tol, int max_iter, void* stream) {
  EFFQ_CHECK_ARG(state_dev && max_iter > 0);
  hipLaunchKernelGGL(k_fp_update, dim3(1), dim3(1), 0, as_stream(stream), state_dev, tol, max_iter);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_alpha_fixed_point(const float* x, size_t n, int levels, double lo, double hi, double tol, int max_iter,
                           int n_iters, effq_fp_state* state_dev, void* ws, void* stream) {
  EFFQ_CHECK_ARG(x && state_dev && ws && n > 0 && n_iters >= 0 && levels >= 2 && hi > lo);
  RedWs r = red_ws(ws);
  const double d = (hi - lo) / (double)(levels - 1);
  const int grid = stream_grid((n + 3) / 4);
  for (int i = 0; i < n_iters; ++i)
    hipLaunchKernelGGL(k_fp_iter, dim3(grid), dim3(TPB), 0, as_stream(stream), x, n, state_dev, lo, hi, d, tol,
                       max_iter, r.partials, r.ticket);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

size_t effq_fp_small_max(void) { return (size_t)1 << 15; }
int effq_fixed_point_small_fused(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                                 double tol, int max_iter, effq_fp_state* state_dev, const ProjFused* pf_in, void* stream);

int effq_fixed_point_small(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                           double tol, int max_iter, effq_fp_state* state_dev, void* stream) {
  return effq_fixed_point_small_fused(a, b, v_out, n, levels, lo, hi, tol, max_iter, state_dev, nullptr, stream);
}

// internal (admm_run.hip): pf != NULL runs the projection of the ADMM iteration as the kernel's epilogue
int effq_fixed_point_small_fused(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                                 double tol, int max_iter, effq_fp_state* state_dev, const ProjFused* pf_in, void* stream) {
  ProjFused pf;
  memset(&pf, 0, sizeof(pf));
  if (pf_in != nullptr) pf = *pf_in;
  EFFQ_CHECK_ARG(pf.G == nullptr || v_out != nullptr);
  EFFQ_CHECK_ARG(a && state_dev && n > 0 && levels >= 2 && hi > lo && max_iter > 0);
  EFFQ_CHECK_ARG(n <= effq_fp_small_max());
  EFFQ_CHECK_ARG(b == nullptr || v_out != nullptr);
  const double d = (hi - lo) / (double)(levels - 1);
  {
    // threads: 256 up to 2048 elements, 512 up to 16384 (few waves: the barrier is cheap and the element loop stays
    // short; measured best on MI355X, scripts/exp_fp256.py), else 1024; slots per thread rounded up to a power of 2.  EFFQ_FPS_T overrides (tuning aid).
    static const int force_t = getenv("EFFQ_FPS_T") ? atoi(getenv("EFFQ_FPS_T")) : 0;
    int T = (n <= 2048) ? 256 : (n <= 16384) ? 512 : FPS_T;
    if (force_t == 256 && n <= 8192) T = 256;
    if (force_t == 512 && n <= 16384) T = 512;
    if (force_t == 1024) T = 1024;
    int per = (int)((n + T - 1) / T), pp = 1;
    while (pp < per) pp <<= 1;
    hipStream_t st = as_stream(stream);
#define EFFQ_FPS(TT, PP)                                                                                          \
  hipLaunchKernelGGL((k_fp_small<TT, PP>), dim3(1), dim3(TT), 0, st, a, b, v_out, n, state_dev, lo, hi, d, tol, max_iter, pf)
    if (T == 256) {
      switch (pp) {
        case 1: EFFQ_FPS(256, 1); break;
        case 2: EFFQ_FPS(256, 2); break;
        case 4: EFFQ_FPS(256, 4); break;
        case 8: EFFQ_FPS(256, 8); break;
        case 16: EFFQ_FPS(256, 16); break;
        default: EFFQ_FPS(256, 32); break;
      }
    } else if (T == 512) {
      switch (pp) {
        case 1: case 2: case 4: EFFQ_FPS(512, 4); break;
        case 8: EFFQ_FPS(512, 8); break;
        case 16: EFFQ_FPS(512, 16); break;
        default: EFFQ_FPS(512, 32); break;
      }
    } else {
      if (pp <= 16) EFFQ_FPS(1024, 16); else EFFQ_FPS(1024, 32);
    }
#undef EFFQ_FPS
  }
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

size_t effq_fp_coop_max(void) { return (size_t)FPC_SLICE * FPC_MAXG; }

int effq_fp_coop_set_spin_limit(unsigned int polls) {
  g_fpc_spin_limit = polls ? polls : FPC_SPIN_LIMIT;
  return EFFQ_OK;
}

int effq_fixed_point_coop(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                          double tol, int max_iter, effq_fp_state* state_dev, void* ws, void* stream) {
  return effq_fixed_point_coop_rec(a, b, v_out, n, levels, lo, hi, tol, max_iter, state_dev, ws, nullptr, stream);
}

int effq_fixed_point_coop_rec(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                              double tol, int max_iter, effq_fp_state* state_dev, void* ws, void* pred_dev,
                              void* stream) {
  FptPred* pred = reinterpret_cast<FptPred*>(pred_dev);
  EFFQ_CHECK_ARG(a && state_dev && ws && n > 0 && levels >= 2 && hi > lo && max_iter > 0);
  EFFQ_CHECK_ARG(n <= effq_fp_coop_max());
  EFFQ_CHECK_ARG(b == nullptr || v_out != nullptr);
  EFFQ_CHECK_ARG(v_out == nullptr || (v_out != a && v_out != b));      // k_fp_coop's operands are __restrict__
  const double d = (hi - lo) / (double)(levels - 1);
  int G = (int)((n + FPC_SLICE - 1) / FPC_SLICE);
  if (G < 1) G = 1;
  EFFQ_CHECK_ARG(G <= FPC_MAXG);
  const size_t per = (n + G - 1) / G;
  const size_t lds = per * sizeof(float);
  // workspace: reuse the reduction workspace: partials [2][FPC_MAXG][3] doubles at its start
  double* partials = reinterpret_cast<double*>(ws);
  // the counter words sit in the tail of the reduction workspace (after the ticket), where no reduction kernel
  // writes partial sums: they must still be zero from the previous launch (the kernel leaves them at zero; the
  // reduction workspace is zero-filled at creation)
  unsigned int* counter = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(ws) +
                                                          sizeof(double) * RED_MAX_BLOCKS * RED_SLOTS + 64);
  hipStream_t st = as_stream(stream);
  int dev = 0;
  EFFQ_HIP(hipGetDevice(&dev));
  EFFQ_CHECK_ARG(dev >= 0 && dev < 64);
  // per DEVICE: the LDS attribute of the kernel and the number of workgroups that can be resident at once
  static bool known[64] = {};
  static int resident_max[64];
  if (!known[dev]) {
    EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fp_coop<FPC_T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(FPC_SLICE * sizeof(float))));
    int ncu = 0, per_cu = 0;
    EFFQ_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    EFFQ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fp_coop<FPC_T>, FPC_T, FPC_SLICE * sizeof(float)));
    resident_max[dev] = ncu * per_cu;
    known[dev] = true;
  }
  // The grid barrier needs every workgroup resident at once.  That holds on a whole MI355X (G <= 256 = its CU count,
  // one workgroup per CU by LDS; workgroups of other streams only delay a late arrival: they retire, they never wait
  // for this kernel); on a partitioned device (CPX / DPX) or a smaller part it may not: then the fixed point runs as one
  // launch per iteration (each a no-op once converged) - slower, never stuck.
  if (G > resident_max[dev]) {
    RedWs r = red_ws(ws);
    const float* src = a;
    if (b != nullptr) {
      hipLaunchKernelGGL(k_presum, dim3(stream_grid(n)), dim3(TPB), 0, st, a, b, v_out, n);
      src = v_out;
    }
    double* s0 = r.partials + (size_t)RED_MAX_BLOCKS * (RED_SLOTS - 1);          // two spare doubles of the workspace
    hipLaunchKernelGGL(k_reduce<0>, dim3(stream_grid((n + 3) / 4)), dim3(TPB), 0, st, src, n, (const double*)nullptr, 0.0,
                       0.0, 0.0, (const int32_t*)nullptr, r.partials, r.ticket, s0);
    hipLaunchKernelGGL(k_fp_init, dim3(1), dim3(1), 0, st, state_dev, s0);
    const int grid = stream_grid((n + 3) / 4);
    for (int i = 0; i < max_iter; ++i)
      hipLaunchKernelGGL(k_fp_iter, dim3(grid), dim3(TPB), 0, st, src, n, state_dev, lo, hi, d, tol, max_iter, r.partials,
                         r.ticket);
    EFFQ_LAUNCH_CHECK();
    return EFFQ_OK;
  }
  hipLaunchKernelGGL(k_fp_coop<FPC_T>, dim3(G), dim3(FPC_T), lds, st, a, b, v_out, n, state_dev, lo, hi, d, tol,
                     max_iter, partials, counter, g_fpc_spin_limit, pred, levels);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_fp_check(const effq_fp_state* state_dev, int32_t* err_flag_dev, void* stream) {
  EFFQ_CHECK_ARG(state_dev && err_flag_dev);
  hipLaunchKernelGGL(k_check_state, dim3(1), dim3(1), 0, as_stream(stream), state_dev, err_flag_dev);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_admm_presum(const float* wstar, const float* dual, float* v, size_t n, void* stream) {
  EFFQ_CHECK_ARG(wstar && dual && v);
  if (n == 0) return EFFQ_OK;
  hipLaunchKernelGGL(k_presum, dim3(stream_grid(n)), dim3(TPB), 0, as_stream(stream), wstar, dual, v, n);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

static bool proj_aligned(const void* a, const void* b, const void* c, const void* d, const void* q) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) & 15) == 0 && (reinterpret_cast<uintptr_t>(q) & 3) == 0;
}

int effq_project_dual_checked(const float* v, const float* wstar, const effq_fp_state* state_dev, int levels, float* G,
                              float* dual, float dual_div, int8_t* Gq_out, size_t n, int32_t* err_flag_dev,
                              void* stream) {
  EFFQ_CHECK_ARG(v && wstar && state_dev && G && dual && levels >= 2 && dual_div > 0.0f);
  EFFQ_CHECK_ARG(Gq_out == nullptr || levels <= 256);
  if (n == 0) return EFFQ_OK;
  const double d = 2.0 / (double)(levels - 1);
  ProjNext nx;
  memset(&nx, 0, sizeof(nx));
  if ((n % 4) == 0 && n < ((size_t)1 << 32) && proj_aligned(v, wstar, G, dual, Gq_out))
    hipLaunchKernelGGL(k_project_dual4, dim3(stream_grid(n / 4)), dim3(TPB), 0, as_stream(stream), v, wstar, state_dev, d,
                       G, dual, dual_div, Gq_out, levels - 1, (unsigned)(n / 4), err_flag_dev, nx);
  else
    hipLaunchKernelGGL(k_project_dual, dim3(stream_grid(n)), dim3(TPB), 0, as_stream(stream), v, wstar, state_dev, d,
                       G, dual, dual_div, Gq_out, levels - 1, n, err_flag_dev, nx);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

// internal (admm_run.hip): the projection that also leaves the right-hand side of the next prox solve in Bm
int effq_project_dual_next(const float* v, const float* wstar, const effq_fp_state* state_dev, int levels, float* G,
                           float* dual, float dual_div, int8_t* Gq_out, size_t n, int32_t* err_flag_dev, float* Bm,
                           const float* B0, const float* W0, int nwrow, int nb0, int ldb, double rho_next, double eta,
                           void* stream) {
  EFFQ_CHECK_ARG(v && wstar && state_dev && G && dual && levels >= 2 && dual_div > 0.0f);
  EFFQ_CHECK_ARG(Gq_out == nullptr || levels <= 256);
  EFFQ_CHECK_ARG(Bm && B0 && W0 && nwrow > 0 && nb0 >= nwrow && ldb >= nb0 && (n % (size_t)nwrow) == 0);
  if (n == 0) return EFFQ_OK;
  const double d = 2.0 / (double)(levels - 1);
  ProjNext nx;
  nx.Bm = Bm; nx.B0 = B0; nx.W0 = W0; nx.nwrow = nwrow; nx.n = nb0; nx.ldb = ldb;
  nx.rho = (float)rho_next; nx.eta = (float)eta;
  if ((nwrow % 4) == 0 && (ldb % 4) == 0 && n < ((size_t)1 << 32) && proj_aligned(v, wstar, G, dual, Gq_out) &&
      (reinterpret_cast<uintptr_t>(W0) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bm) & 15) == 0)
    hipLaunchKernelGGL(k_project_dual4, dim3(stream_grid(n / 4)), dim3(TPB), 0, as_stream(stream), v, wstar, state_dev, d,
                       G, dual, dual_div, Gq_out, levels - 1, (unsigned)(n / 4), err_flag_dev, nx);
  else
    hipLaunchKernelGGL(k_project_dual, dim3(stream_grid(n)), dim3(TPB), 0, as_stream(stream), v, wstar, state_dev, d,
                       G, dual, dual_div, Gq_out, levels - 1, n, err_flag_dev, nx);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_admm_project_dual(const float* v, const float* wstar, const effq_fp_state* state_dev, int levels, float* G,
                           float* dual, float dual_div, int8_t* Gq_out, size_t n, void* stream) {
  return effq_project_dual_checked(v, wstar, state_dev, levels, G, dual, dual_div, Gq_out, n, nullptr, stream);
}

int effq_act_quant_backward(const float* x, const float* alpha_dev, int levels, const float* gq, float* gx_out,
                            double* galpha_out, size_t n, void* ws, void* stream) {
  EFFQ_CHECK_ARG(x && alpha_dev && gq && galpha_out && ws && n > 0 && levels >= 2);
  RedWs r = red_ws(ws);
  const float d = (float)(1.0 / (double)(levels - 1));
  hipLaunchKernelGGL(k_act_quant_bwd, dim3(stream_grid(n)), dim3(TPB), 0, as_stream(stream), x, alpha_dev, 0.0f, 1.0f, d,
                     gq, gx_out, n, r.partials, r.ticket, galpha_out);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_adam_step(float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps, int t,
                   size_t n, void* stream) {
  EFFQ_CHECK_ARG(p && g && m && v && t >= 1);
  if (n == 0) return EFFQ_OK;
  const float bc1 = 1.0f - powf(b1, (float)t), bc2 = 1.0f - powf(b2, (float)t);
  hipLaunchKernelGGL(k_adam, dim3(stream_grid(n)), dim3(TPB), 0, as_stream(stream), p, g, m, v, lr, b1, b2, eps, bc1,
                     bc2, n);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

size_t effq_packed_bytes(size_t n, int bits) {
  if (!(bits == 1 || bits == 2 || bits == 4 || bits == 8)) return 0;
  return (n * (size_t)bits + 7) / 8;
}

int effq_pack_levels(const uint8_t* idx, size_t n, int bits, uint8_t* packed, void* stream) {
  EFFQ_CHECK_ARG(bits == 1 || bits == 2 || bits == 4 || bits == 8);
  if (n == 0) return EFFQ_OK;
  EFFQ_CHECK_ARG(idx && packed);
  const size_t nbytes = effq_packed_bytes(n, bits);
  hipLaunchKernelGGL(k_pack_levels, dim3(stream_grid(nbytes)), dim3(TPB), 0, as_stream(stream), idx, n, bits, packed,
                     nbytes);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_unpack_levels(const uint8_t* packed, size_t n, int bits, uint8_t* idx, void* stream) {
  EFFQ_CHECK_ARG(bits == 1 || bits == 2 || bits == 4 || bits == 8);
  if (n == 0) return EFFQ_OK;
  EFFQ_CHECK_ARG(idx && packed);
  hipLaunchKernelGGL(k_unpack_levels, dim3(stream_grid(n)), dim3(TPB), 0, as_stream(stream), packed, n, bits, idx);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
