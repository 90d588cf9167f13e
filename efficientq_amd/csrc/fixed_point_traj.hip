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

namespace effq {

constexpr int FPT_T = 512;
constexpr int FPT_EPT = 8;
constexpr int FPT_WGV = FPT_T * FPT_EPT;          // values per workgroup
// tallies per workgroup: (sum l*u, packed sum l | sum l^2) of the values settled over the HULL of all brackets (0, 1),
// then the same for (wide_j, ring_j) at 2 + 4 j
constexpr int FPT_NQ = 2 + 4 * FPT_SLOTS;
constexpr int FPT_NE = 4 * FPT_SLOTS;             // bracket ends: wide lo, narrow lo, narrow hi, wide hi per slot
constexpr int FPT_CR = 8;                         // narrow-list entries a thread of the last workgroup keeps in registers
constexpr int FPT_FC = 16;                        // ... and a lane of its first wave for the one-wave iterations
constexpr int FPT_PF = 8;                         // loads in flight per thread in the streaming loops
constexpr size_t FPT_MAXN = (size_t)1 << 23;

struct FptHdr {
  unsigned ticket, n_narrow, n_ring;
};
struct FptWs {
  FptHdr* hdr;
  double* dpart;               // [G][2]  sum|v|, sum v of workgroup g
  long long* part;             // [G][FPT_NQ]
  unsigned long long* narrow;  // [n]  float bits | narrow mask << 32 | ring mask << 40
  unsigned long long* ring;    // [n]
};
static size_t fpt_groups(size_t n) { return (n + FPT_WGV - 1) / FPT_WGV; }
static size_t fpt_ws_bytes(size_t n) {
  const size_t G = fpt_groups(n);
  return 256 + sizeof(double) * 2 * G + sizeof(long long) * FPT_NQ * G + 2 * sizeof(unsigned long long) * (n + 8);
}
static FptWs fpt_carve(void* ws, size_t n) {
  const size_t G = fpt_groups(n);
  FptWs w;
  char* p = reinterpret_cast<char*>(ws);
  w.hdr = reinterpret_cast<FptHdr*>(p);
  p += 256;
  w.dpart = reinterpret_cast<double*>(p);
  p += sizeof(double) * 2 * G;
  w.part = reinterpret_cast<long long*>(p);
  p += sizeof(long long) * FPT_NQ * G;
  w.narrow = reinterpret_cast<unsigned long long*>(p);
  p += sizeof(unsigned long long) * (n + 8);
  w.ring = reinterpret_cast<unsigned long long*>(p);
  return w;
}

// 64-lane sum of a 64-bit integer (mod 2^64: signed totals come out right) on the DPP network: three limbs of 21 / 21 / 22
// bits keep every partial sum below 2^32; every lane ends with the total
__device__ __forceinline__ long long fpt_wave_sum(long long v) {
  const unsigned long long u = (unsigned long long)v;
  const unsigned l0 = group_sum_u32((unsigned)(u & 0x1fffffu), 64), l1 = group_sum_u32((unsigned)((u >> 21) & 0x1fffffu), 64),
                 l2 = group_sum_u32((unsigned)(u >> 42), 64);
  return (long long)(((unsigned long long)l2 << 42) + ((unsigned long long)l1 << 21) + (unsigned long long)l0);
}
__device__ __forceinline__ long long fpt_pack(int r) { return (long long)r + ((long long)(r * r) << 32); }
// acc + level * u mod 2^64 (the unit of u is a PREDICTION: a tensor that grew by more than ~4 x since the last call
// can wrap the sum, which phase 2 detects and discards - but it must not be signed overflow)
__device__ __forceinline__ long long fpt_add_wrap(long long acc, int level, long long u) {
  return (long long)((unsigned long long)acc + (unsigned long long)(long long)level * (unsigned long long)u);
}
// Slot of a wave-aggregated push: ONE LDS atomic per wave (same-address LDS atomics serialise: a few hundred pushes of a
// workgroup onto one counter took longer than the classification itself).  Every lane of the wave must call it.
__device__ __forceinline__ unsigned fpt_push_slot(bool take, unsigned* counter) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
  unsigned base = 0u;
  if (m != 0ull) {
    const int first = __builtin_ctzll(m);
    if ((int)(threadIdx.x & 63) == first) base = atomicAdd(counter, (unsigned)__builtin_popcountll(m));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, first);
  }
  return base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__global__ __launch_bounds__(FPT_T) void k_fpt(const float* __restrict__ a, const float* __restrict__ b2,
                                               float* __restrict__ v_out, size_t n, FptWs w, FptPred* pred,
                                               effq_fp_state* st, double lo, double hi, double d, int levels, double tol,
                                               int max_iter) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams
  __shared__ float s_c1[FPT_NE + 2];                   // bracket ends: wide lo, narrow lo, narrow hi, wide hi of slot j at 4 j; hull at NE
  __shared__ double s_end[FPT_NE + 2];
  __shared__ unsigned long long s_list[FPT_WGV];      // narrow entries from the front, ring entries from the back
  __shared__ float s_work[FPT_WGV];                    // values whose level changes somewhere inside the hull of the brackets
  __shared__ unsigned s_cn, s_cr, s_bn, s_br, s_nw;
  __shared__ int s_last;
  __shared__ long long s_red[FPT_NQ][FPT_T / 64];
  __shared__ double s_dred[2][FPT_T / 64];
  __shared__ long long s_T[FPT_NQ];
  __shared__ long long s_it[2][2][FPT_T / 64];
  __shared__ double s_tot[2];
  __shared__ double s_fd[4];                           // hand-over from the one-wave iterations: alpha, alpha_prev, last0, last1
  __shared__ int s_fi[4];                              //   it, done, n_warm
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wg = blockIdx.x;
  constexpr int NW = FPT_T / 64;
  const long long tr0 = wall_clock64();
  // the operands first: their loads are in flight while the predictions are fetched and the bracket ends set up
  float va[FPT_EPT], vb[FPT_EPT];
  const size_t base = (size_t)wg * FPT_WGV + tid;
#pragma unroll
  for (int e = 0; e < FPT_EPT; ++e) {
    const size_t i = base + (size_t)e * FPT_T;
    const size_t ic = (i < n) ? i : (n - 1);
    va[e] = a[ic];
    vb[e] = (b2 != nullptr) ? b2[ic] : 0.0f;
  }
  // the predictions in ONE round trip: every field is requested before the first is looked at (slots beyond K hold
  // whatever the last calls left there: loaded, never used)
  const int K_raw = pred->K, e_valid = pred->e_valid, e_units = pred->e;
  double p_lo = 1.0, p_hi = 1.0, p_eps = 0.0, p_epsn = 0.0;
  if (tid < FPT_NE) {
    const int j = tid >> 2;
    p_lo = pred->lo[j];
    p_hi = pred->hi[j];
    p_eps = pred->eps[j];
    p_epsn = pred->eps_n[j];
  }
  // (a prediction buffer the caller did not zero-fill must not index past the slots)
  const int K = min(max(K_raw, 0), FPT_SLOTS);
  const bool warm = K > 0 && e_valid != 0;
  const double inv_q = ldexp(1.0, e_units);
  const double rd = 1.0 / d;
  const float c0 = (float)(-lo * rd), lmax = (float)rint((hi - lo) * rd);
  if (warm && tid < 4 * K) {
    const int c = tid & 3;
    const double ew = p_eps, en = fmin(p_epsn, ew);
    const double end = (c == 0) ? p_lo * (1.0 - ew) : (c == 1) ? p_lo * (1.0 - en)
                     : (c == 2) ? p_hi * (1.0 + en) : p_hi * (1.0 + ew);
    s_end[tid] = end;
    s_c1[tid] = (float)((1.0 / end) * rd);
  }
  if (tid == 0) {
    s_cn = 0u;
    s_cr = 0u;
    s_nw = 0u;
  }
  __syncthreads();
  if (warm && tid < 2) {                         // the hull of every bracket (the level of a value is monotone in the scale)
    double h = s_end[tid == 0 ? 0 : 3];
    for (int j = 1; j < K; ++j) h = (tid == 0) ? fmin(h, s_end[4 * j]) : fmax(h, s_end[4 * j + 3]);
    s_end[FPT_NE + tid] = h;
    s_c1[FPT_NE + tid] = (float)((1.0 / h) * rd);
  }
  auto level_end = [&](float v, int g) -> int {
    float u = __builtin_fmaf(v, s_c1[g], c0);
    u = fminf(fmaxf(u, 0.0f), lmax);
    const float rf = rintf(u);
    if (!(fabsf(u - rf) < 0.4998f)) return fp_level_exact(v, s_end[g], lo, hi, d);
    return (int)rf;
  };

  // ---- phase 1 ------------------------------------------------------------------------------------------------------
  float vv[FPT_EPT];
#pragma unroll
  for (int e = 0; e < FPT_EPT; ++e) vv[e] = (b2 != nullptr) ? (va[e] + vb[e]) : va[e];
  double sabs = 0.0, sv = 0.0;
#pragma unroll
  for (int e = 0; e < FPT_EPT; ++e) {
    const size_t i = base + (size_t)e * FPT_T;
    if (i < n) {
      const float v = vv[e];
      if (v_out != nullptr) v_out[i] = v;
      sabs += fabs((double)v);
      sv += (double)v;
    }
  }
  // (a) every value at the two ends of the hull: equal levels (nine values in ten at 4 levels) settle it for EVERY slot -
  // one common tally; the others go to a worklist in LDS
  long long tc_ru = 0, tc_ct = 0;
  if (warm) {
    __syncthreads();                             // the hull ends
#pragma unroll
    for (int e = 0; e < FPT_EPT; ++e) {
      const size_t i = base + (size_t)e * FPT_T;
      const float v = vv[e];
      bool cross = false;
      if (i < n) {
        const int ra = level_end(v, FPT_NE), rb = level_end(v, FPT_NE + 1);
        if (ra == rb) {
          tc_ru = fpt_add_wrap(tc_ru, ra, __double2ll_rn((double)v * inv_q));
          tc_ct += fpt_pack(ra);
        } else {
          cross = true;
        }
      }
      const unsigned slot = fpt_push_slot(cross, &s_nw);
      if (cross) s_work[slot] = v;
    }
    __syncthreads();
  }
  // (b) the worklist, one value per thread and round: per slot, the two ends of the wide bracket, then of the narrow one
  long long tw_ru[FPT_SLOTS], tw_ct[FPT_SLOTS], tr_ru[FPT_SLOTS], tr_ct[FPT_SLOTS];
#pragma unroll
  for (int j = 0; j < FPT_SLOTS; ++j) {
    tw_ru[j] = 0;
    tw_ct[j] = 0;
    tr_ru[j] = 0;
    tr_ct[j] = 0;
  }
  const unsigned n_work = warm ? s_nw : 0u;
  const bool wave_works = (unsigned)(wid * 64) < n_work;          // (wave-uniform)
  if (wave_works) {
    for (unsigned i0 = (unsigned)(wid * 64); i0 < n_work; i0 += FPT_T) {
      const unsigned idx = i0 + (unsigned)lane;
      const bool have = idx < n_work;
      const float v = have ? s_work[idx] : 0.0f;
      unsigned nmask = 0u, rmask = 0u;
      if (have) {
        const long long u = __double2ll_rn((double)v * inv_q);
#pragma unroll
        for (int j = 0; j < FPT_SLOTS; ++j) {
          if (j < K) {
            const int ra = level_end(v, 4 * j), rb = level_end(v, 4 * j + 3);
            if (ra == rb) {
              tw_ru[j] = fpt_add_wrap(tw_ru[j], ra, u);
              tw_ct[j] += fpt_pack(ra);
            } else {
              const int na = level_end(v, 4 * j + 1), nb = level_end(v, 4 * j + 2);
              if (na == nb) {
                tr_ru[j] = fpt_add_wrap(tr_ru[j], na, u);
                tr_ct[j] += fpt_pack(na);
                rmask |= 1u << j;
              } else {
                nmask |= 1u << j;
              }
            }
          }
        }
      }
      const unsigned long long en = (unsigned long long)__float_as_uint(v) | ((unsigned long long)nmask << 32) |
                                    ((unsigned long long)rmask << 40);
      const bool to_n = nmask != 0u, to_r = nmask == 0u && rmask != 0u;
      const unsigned sn = fpt_push_slot(to_n, &s_cn), sr = fpt_push_slot(to_r, &s_cr);
      if (to_n) s_list[sn] = en;
      if (to_r) s_list[FPT_WGV - 1 - sr] = en;
    }
  }
  __syncthreads();                               // the lists and their lengths
  // the list offsets (a device-wide atomic each: a round trip) travel while the tallies are summed; the last wave issues
  // them - it seldom has worklist entries of its own
  if (tid == FPT_T - 1) {
    const unsigned c = s_cn;
    s_bn = (c != 0u) ? __hip_atomic_fetch_add(&w.hdr->n_narrow, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  }
  if (tid == FPT_T - 2) {
    const unsigned c = s_cr;
    s_br = (c != 0u) ? __hip_atomic_fetch_add(&w.hdr->n_ring, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  }
  // workgroup sums: doubles by a fixed tree (deterministic), integers in any order
  sabs = wave_sum_f64_dpp(sabs);
  sv = wave_sum_f64_dpp(sv);
  if (lane == 0) {
    s_dred[0][wid] = sabs;
    s_dred[1][wid] = sv;
  }
  if (warm) {
    const long long c0s = fpt_wave_sum(tc_ru), c1s = fpt_wave_sum(tc_ct);
    if (lane == 0) {
      s_red[0][wid] = c0s;
      s_red[1][wid] = c1s;
    }
#pragma unroll
    for (int j = 0; j < FPT_SLOTS; ++j) {
      if (j < K) {                               // (uniform)
        long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        if (wave_works) {
          q0 = fpt_wave_sum(tw_ru[j]);
          q1 = fpt_wave_sum(tw_ct[j]);
          // ring tallies are empty for most waves: skip their sums then (wave-uniform)
          if (__builtin_amdgcn_ballot_w64(tr_ct[j] != 0) != 0ull) {
            q2 = fpt_wave_sum(tr_ru[j]);
            q3 = fpt_wave_sum(tr_ct[j]);
          }
        }
        if (lane == 0) {
          s_red[2 + 4 * j + 0][wid] = q0;
          s_red[2 + 4 * j + 1][wid] = q1;
          s_red[2 + 4 * j + 2][wid] = q2;
          s_red[2 + 4 * j + 3][wid] = q3;
        }
      }
    }
  }
  __syncthreads();
  if (tid < 2) {
    double t = 0.0;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) t += s_dred[tid][wv];
    w.dpart[2 * wg + tid] = t;
  }
  if (warm && tid >= 64 && tid < 64 + 2 + 4 * K) {
    const int s2 = tid - 64;
    long long t = 0;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) t += s_red[s2][wv];
    w.part[(size_t)wg * FPT_NQ + s2] = t;
  }
  {
    const unsigned cn = s_cn, cr = s_cr, bn = s_bn, br = s_br;
    for (unsigned i = tid; i < cn; i += FPT_T) w.narrow[bn + i] = s_list[i];
    for (unsigned i = tid; i < cr; i += FPT_T) w.ring[br + i] = s_list[FPT_WGV - 1 - i];
  }
  __syncthreads();
  // hand-off to the last workgroup: release / ticket / acquire (cdna_hip_programming.md Guideline 16)
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(&w.hdr->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  const long long tr1 = wall_clock64();
  __builtin_amdgcn_s_setprio(3);
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  // ---- phase 2 ------------------------------------------------------------------------------------------------------
  const int G = gridDim.x;
  const unsigned m_n = __hip_atomic_load(&w.hdr->n_narrow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned m_r = __hip_atomic_load(&w.hdr->n_ring, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the narrow list (what nearly every iterate scans) in registers: FPT_CR entries per thread, the rest is streamed;
  // the first wave holds the first 64 FPT_FC entries once more, lane-major: a list that short it scans ALONE
  unsigned long long creg[FPT_CR], fcr[FPT_FC];
#pragma unroll
  for (int c = 0; c < FPT_CR; ++c) {
    const unsigned i = (unsigned)tid + (unsigned)c * FPT_T;
    creg[c] = (i < m_n) ? w.narrow[i] : 0ull;
  }
#pragma unroll
  for (int c = 0; c < FPT_FC; ++c) {
    const unsigned i = (unsigned)lane + (unsigned)c * 64u;
    fcr[c] = (wid == 0 && i < m_n) ? w.narrow[i] : 0ull;
  }
  {
    double t0 = 0.0, t1 = 0.0;                   // partials in workgroup order, fixed tree: deterministic
    for (int g = tid; g < G; g += FPT_T) {
      t0 += __hip_atomic_load(&w.dpart[2 * g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t1 += __hip_atomic_load(&w.dpart[2 * g + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    t0 = wave_sum_f64_dpp(t0);
    t1 = wave_sum_f64_dpp(t1);
    if (lane == 0) {
      s_dred[0][wid] = t0;
      s_dred[1][wid] = t1;
    }
    if (tid < FPT_NQ) s_T[tid] = 0;
  }
  __syncthreads();
  if (tid < 2) {
    double t = 0.0;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) t += s_dred[tid][wv];
    s_tot[tid] = t;
  }
  if (warm) {                                    // tallies: thread = (quantity, subset of the workgroups); integers, any order
    const int s2 = tid & 63, sub = tid >> 6;     // 64 quantities x 8 subsets; plain loads: ordered by the acquire above
    if (s2 < 2 + 4 * K) {
      long long t = 0;
      for (int g0 = sub; g0 < G; g0 += NW * FPT_PF) {
        long long x[FPT_PF];
#pragma unroll
        for (int u = 0; u < FPT_PF; ++u) {
          const int g = g0 + u * NW;
          x[u] = (g < G) ? w.part[(size_t)g * FPT_NQ + s2] : 0ll;
        }
#pragma unroll
        for (int u = 0; u < FPT_PF; ++u) t += x[u];
      }
      atomicAdd(reinterpret_cast<unsigned long long*>(&s_T[s2]), (unsigned long long)t);
    }
  }
  __syncthreads();
  if (tid == 0) {                                // counters back to zero for the next call
    __hip_atomic_store(&w.hdr->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&w.hdr->n_narrow, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&w.hdr->n_ring, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const double tot_abs = s_tot[0], tot_v = s_tot[1];
  // unit of the full passes (always valid for this tensor); the tallies' unit must not overflow either
  int e_full = 0;
  {
    const double bound = (double)(levels - 1) * tot_abs;
    if (bound > 0.0 && bound < 1e300) e_full = 58 - ilogb(bound);
  }
  const bool tallies_ok = warm && ((double)(levels - 1) * tot_abs * inv_q < 2.0e18);
  const double q_tally = ldexp(1.0, -e_units), inv_qf = ldexp(1.0, e_full), q_full = ldexp(1.0, -e_full);
  const bool vec_ok = (v_out != nullptr) && ((reinterpret_cast<uintptr_t>(v_out) & 15u) == 0);
  const float* vsrc = (v_out != nullptr) ? v_out : a;

  const long long tr2 = wall_clock64();
  double alpha = tot_abs / (double)n, alpha_prev = -999.0, last0 = 0.0, last1 = 0.0;
  int it = 0, done = 0, n_warm = 0, n_ring = 0, n_full = 0;
  // One-wave iterations: while the iterate stays inside its NARROW bracket and the narrow list fits the first wave's
  // registers, that wave iterates alone - no LDS exchange, no workgroup barrier (0.6 against 2.2 us per iterate); the first
  // iterate that needs more (the ring list, a full pass) goes to the loop of the whole workgroup below.
  if (wid == 0) {
    if (tallies_ok && m_n <= (unsigned)(64 * FPT_FC)) {
      const int nfc = (int)((m_n + 63u) / 64u);
      while (!done) {
        if (!(alpha > 0.0) || !(alpha < 1e300)) break;
        const int j = (it < K) ? it : K - 1;
        if (!(alpha >= s_end[4 * j + 1] && alpha <= s_end[4 * j + 2])) break;
        if (lane == 0) fpt_note(pred, it, alpha);
        const FpLevel lc = fp_level_consts(alpha, lo, hi, d);
        const unsigned long long want = 1ull << (32 + j);
        long long ru = 0, ct = 0;
#pragma unroll
        for (int c = 0; c < FPT_FC; ++c) {
          if (c < nfc && (fcr[c] & want)) {
            const float v = __uint_as_float((unsigned)fcr[c]);
            const int r = fp_level(v, lc, lo, hi, d);
            ru += (long long)r * __double2ll_rn((double)v * inv_q);
            ct += fpt_pack(r);
          }
        }
        const long long Sru = fpt_wave_sum(ru) + s_T[0] + s_T[2 + 4 * j] + s_T[2 + 4 * j + 2];
        const long long Sct = fpt_wave_sum(ct) + s_T[1] + s_T[2 + 4 * j + 1] + s_T[2 + 4 * j + 3];
        const double Sr = (double)(Sct & 0xffffffffll), Sr2 = (double)(Sct >> 32);
        const double t0 = d * (q_tally * (double)Sru) + lo * tot_v;                          // sum b v
        const double t1 = (d * d * Sr2 + 2.0 * d * lo * Sr) + lo * lo * (double)n;            // sum b^2
        const double a_new = t0 / t1;
        ++it;
        ++n_warm;
        if (it >= max_iter)
          done = 2;
        else if (!(fabs(a_new - alpha) > tol))
          done = 1;
        alpha_prev = alpha;
        alpha = a_new;
        last0 = t0;
        last1 = t1;
      }
    }
    if (lane == 0) {
      s_fd[0] = alpha;
      s_fd[1] = alpha_prev;
      s_fd[2] = last0;
      s_fd[3] = last1;
      s_fi[0] = it;
      s_fi[1] = done;
      s_fi[2] = n_warm;
    }
  }
  __syncthreads();
  alpha = s_fd[0];
  alpha_prev = s_fd[1];
  last0 = s_fd[2];
  last1 = s_fd[3];
  it = s_fi[0];
  done = s_fi[1];
  n_warm = s_fi[2];
  while (!done) {
    if (!(alpha > 0.0) || !(alpha < 1e300)) {    // NaN / non-positive scale: the reference would spin to its cap
      done = 2;
      break;
    }
    if (tid == 0) fpt_note(pred, it, alpha);
    const int par = it & 1;
    const int j = (it < K) ? it : K - 1;
    const bool in_w = tallies_ok && alpha >= s_end[4 * j] && alpha <= s_end[4 * j + 3];
    const bool in_n = in_w && alpha >= s_end[4 * j + 1] && alpha <= s_end[4 * j + 2];
    const FpLevel lc = fp_level_consts(alpha, lo, hi, d);
    long long ru = 0, ct = 0;
    if (in_w) {
      // bits of an entry that matter: narrow bit j always; ring bit j too when the iterate missed the narrow bracket
      const unsigned long long want = (1ull << (32 + j)) | (in_n ? 0ull : (1ull << (40 + j)));
      auto entry = [&](unsigned long long en) {
        if (en & want) {
          const float v = __uint_as_float((unsigned)en);
          const int r = fp_level(v, lc, lo, hi, d);
          ru += (long long)r * __double2ll_rn((double)v * inv_q);
          ct += fpt_pack(r);
        }
      };
#pragma unroll
      for (int c = 0; c < FPT_CR; ++c) entry(creg[c]);
      auto stream = [&](const unsigned long long* __restrict__ list, unsigned first, unsigned count) {
        for (unsigned b0 = first + (unsigned)tid; b0 < count; b0 += FPT_T * FPT_PF) {
          unsigned long long en[FPT_PF];
#pragma unroll
          for (int u = 0; u < FPT_PF; ++u) {
            const unsigned i = b0 + (unsigned)u * FPT_T;
            en[u] = (i < count) ? list[i] : 0ull;
          }
#pragma unroll
          for (int u = 0; u < FPT_PF; ++u) entry(en[u]);
        }
      };
      if (m_n > (unsigned)(FPT_CR * FPT_T)) stream(w.narrow, FPT_CR * FPT_T, m_n);
      if (!in_n) {
        stream(w.ring, 0u, m_r);
        ++n_ring;
      } else {
        ++n_warm;
      }
    } else {
      auto one = [&](float v) {
        const int r = fp_level(v, lc, lo, hi, d);
        ru += (long long)r * __double2ll_rn((double)v * inv_qf);
        ct += fpt_pack(r);
      };
      const size_t nv = vec_ok ? n / 4 : 0;
      for (size_t b0 = tid; b0 < nv; b0 += (size_t)FPT_T * FPT_PF) {
        float4 x4[FPT_PF];
#pragma unroll
        for (int u = 0; u < FPT_PF; ++u) {
          const size_t i = b0 + (size_t)u * FPT_T;
          x4[u] = (i < nv) ? reinterpret_cast<const float4*>(vsrc)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < FPT_PF; ++u) {
          if (b0 + (size_t)u * FPT_T < nv) {
            one(x4[u].x);
            one(x4[u].y);
            one(x4[u].z);
            one(x4[u].w);
          }
        }
      }
      for (size_t i = nv * 4 + tid; i < n; i += FPT_T) one(vsrc[i]);
      ++n_full;
    }
    ru = fpt_wave_sum(ru);
    ct = fpt_wave_sum(ct);
    if (lane == 0) {
      s_it[par][0][wid] = ru;
      s_it[par][1][wid] = ct;
    }
    __syncthreads();
    long long Sru = 0, Sct = 0;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) {
      Sru += s_it[par][0][wv];
      Sct += s_it[par][1][wv];
    }
    double q_used = q_full;
    if (in_w) {
      Sru += s_T[0] + s_T[2 + 4 * j];            // settled over the hull of all brackets + settled by the wide bracket of slot j
      Sct += s_T[1] + s_T[2 + 4 * j + 1];
      if (in_n) {                                // values settled by the narrow bracket only: their tally stands
        Sru += s_T[2 + 4 * j + 2];
        Sct += s_T[2 + 4 * j + 3];
      }
      q_used = q_tally;
    }
    const double Sr = (double)(Sct & 0xffffffffll), Sr2 = (double)(Sct >> 32);
    const double t0 = d * (q_used * (double)Sru) + lo * tot_v;                            // sum b v
    const double t1 = (d * d * Sr2 + 2.0 * d * lo * Sr) + lo * lo * (double)n;              // sum b^2
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
  const long long tr3 = wall_clock64();
  if (tid == 0) {
    st->alpha = alpha;
    st->alpha_prev = alpha_prev;
    st->sums[0] = last0;
    st->sums[1] = last1;
    st->iters = it;
    st->done = done;
  }
  if (tid < FPT_SLOTS) fpt_finish_slot(pred, tid, it, alpha);
  __syncthreads();
  if (tid == 0) {
    fpt_finish_head(pred, it, tot_abs, levels);
    pred->warm_iters += n_warm;
    pred->ring_iters += n_ring;
    pred->full_iters += n_full;
    pred->listed += m_n;
    pred->ring_listed += m_r;
    if ((long long)(m_n + m_r) > pred->list_max) pred->list_max = m_n + m_r;
    pred->trace[0] = tr0;                        // (last call)
    pred->trace[1] = tr1;
    pred->trace[2] = tr2;
    pred->trace[3] = tr3;
    pred->trace[4] = wall_clock64();
    pred->trace[5] += tr1 - tr0;                 // (sums over the calls: phase 1 of the last workgroup, set-up, iterations)
    pred->trace[6] += tr2 - tr1;
    pred->trace[7] += tr3 - tr2;
  }
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
  hipLaunchKernelGGL(k_fpt, dim3((unsigned)fpt_groups(n)), dim3(FPT_T), 0, as_stream(stream), a, b, v_out, n,
                     fpt_carve(ws, n), reinterpret_cast<FptPred*>(pred_dev), state_dev, lo, hi, d, levels, tol, max_iter);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
