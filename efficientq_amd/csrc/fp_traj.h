// Device side of the weight projection from the previous iteration's iterates (fixed_point_traj.hip has the description):
// phase 1 (every workgroup: classification of its values against the predicted iterates) and phase 2 (one workgroup: the
// fixed point from the tallies and lists) as device functions of the values a thread holds, so that a kernel that has just
// COMPUTED the values can run them without a round trip through memory (k_fpt loads them; a cooperative whole-iteration
// kernel for the narrow layers was built on these and measured: DESIGN.md section 6, round 4).
#pragma once
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
static inline size_t fpt_groups(size_t n) { return (n + FPT_WGV - 1) / FPT_WGV; }
static inline size_t fpt_ws_bytes(size_t n) {
  const size_t G = fpt_groups(n);
  return 256 + sizeof(double) * 2 * G + sizeof(long long) * FPT_NQ * G + 2 * sizeof(unsigned long long) * (n + 8);
}
static inline FptWs fpt_carve(void* ws, size_t n) {
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

// LDS of one workgroup (52 KB)
struct FptSmem {
  float c1[FPT_NE + 2];                    // bracket ends: wide lo, narrow lo, narrow hi, wide hi of slot j at 4 j; hull at NE
  double end[FPT_NE + 2];
  unsigned long long list[FPT_WGV];        // narrow entries from the front, ring entries from the back
  float work[FPT_WGV];                     // values whose level changes somewhere inside the hull of the brackets
  unsigned cn, cr, bn, br, nw;
  int last;
  long long red[FPT_NQ][FPT_T / 64];
  double dred[2][FPT_T / 64];
  long long T[FPT_NQ];
  long long it[2][2][FPT_T / 64];
  double tot[2];
  double fd[4];                            // hand-over from the one-wave iterations: alpha, alpha_prev, last0, last1
  int fi[4];                               //   it, done, n_warm
};
// what phase 1 learnt from the predictions and phase 2 needs again
struct FptCtx {
  int K, e_units;
  bool warm;
  double inv_q, lo, hi, d;
  long long tr0, tr1;
};

// ---- phase 1: the EPT values of this thread (ok[e]: the value exists) of workgroup `wg` of `G`.  Every thread of the
// workgroup (FPT_T) calls it; on return the workgroup's tallies, sums and lists are in the workspace (NOT yet released
// to other workgroups: the caller's ticket or grid barrier does that).
template <int EPT>
__device__ __forceinline__ void fpt_phase1(FptSmem& sm, const float (&vv)[EPT], const bool (&ok)[EPT], int wg, const FptWs& w,
                                           const FptPred* pred, double lo, double hi, double d, FptCtx& cx) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NW = FPT_T / 64;
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
    sm.end[tid] = end;
    sm.c1[tid] = (float)((1.0 / end) * rd);
  }
  if (tid == 0) {
    sm.cn = 0u;
    sm.cr = 0u;
    sm.nw = 0u;
  }
  __syncthreads();
  if (warm && tid < 2) {                         // the hull of every bracket (the level of a value is monotone in the scale)
    double h = sm.end[tid == 0 ? 0 : 3];
    for (int j = 1; j < K; ++j) h = (tid == 0) ? fmin(h, sm.end[4 * j]) : fmax(h, sm.end[4 * j + 3]);
    sm.end[FPT_NE + tid] = h;
    sm.c1[FPT_NE + tid] = (float)((1.0 / h) * rd);
  }
  auto level_end = [&](float v, int g) -> int {
    float u = __builtin_fmaf(v, sm.c1[g], c0);
    u = fminf(fmaxf(u, 0.0f), lmax);
    const float rf = rintf(u);
    if (!(fabsf(u - rf) < 0.4998f)) return fp_level_exact(v, sm.end[g], lo, hi, d);
    return (int)rf;
  };

  double sabs = 0.0, sv = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    if (ok[e]) {
      sabs += fabs((double)vv[e]);
      sv += (double)vv[e];
    }
  }
  // (a) every value at the two ends of the hull: equal levels (nine values in ten at 4 levels) settle it for EVERY slot -
  // one common tally; the others go to a worklist in LDS
  long long tc_ru = 0, tc_ct = 0;
  if (warm) {
    __syncthreads();                             // the hull ends
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const float v = vv[e];
      bool cross = false;
      if (ok[e]) {
        const int ra = level_end(v, FPT_NE), rb = level_end(v, FPT_NE + 1);
        if (ra == rb) {
          tc_ru = fpt_add_wrap(tc_ru, ra, __double2ll_rn((double)v * inv_q));
          tc_ct += fpt_pack(ra);
        } else {
          cross = true;
        }
      }
      const unsigned slot = fpt_push_slot(cross, &sm.nw);
      if (cross) sm.work[slot] = v;
    }
    __syncthreads();
  }
  // (b) the worklist, one value per lane and round: per slot, the two ends of the wide bracket, then of the narrow one.
  // (Running sums per thread and slot, reduced once at the end: summing each round over the wave at once and keeping the
  // totals of slot j in lane j saves 66 VGPRs - 232 -> 166, still one workgroup per CU - and measured 4 us SLOWER per call
  // inside the calibration: the DPP chains then sit between the classifications of consecutive slots.)
  long long tw_ru[FPT_SLOTS], tw_ct[FPT_SLOTS], tr_ru[FPT_SLOTS], tr_ct[FPT_SLOTS];
#pragma unroll
  for (int j = 0; j < FPT_SLOTS; ++j) {
    tw_ru[j] = 0;
    tw_ct[j] = 0;
    tr_ru[j] = 0;
    tr_ct[j] = 0;
  }
  const unsigned n_work = warm ? sm.nw : 0u;
  const bool wave_works = (unsigned)(wid * 64) < n_work;          // (wave-uniform)
  if (wave_works) {
    for (unsigned i0 = (unsigned)(wid * 64); i0 < n_work; i0 += FPT_T) {
      const unsigned idx = i0 + (unsigned)lane;
      const bool have = idx < n_work;
      const float v = have ? sm.work[idx] : 0.0f;
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
      const unsigned sn = fpt_push_slot(to_n, &sm.cn), sr = fpt_push_slot(to_r, &sm.cr);
      if (to_n) sm.list[sn] = en;
      if (to_r) sm.list[FPT_WGV - 1 - sr] = en;
    }
  }
  __syncthreads();                               // the lists and their lengths
  // the list offsets (a device-wide atomic each: a round trip) travel while the tallies are summed; the last wave issues
  // them - it seldom has worklist entries of its own
  if (tid == FPT_T - 1) {
    const unsigned c = sm.cn;
    sm.bn = (c != 0u) ? __hip_atomic_fetch_add(&w.hdr->n_narrow, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  }
  if (tid == FPT_T - 2) {
    const unsigned c = sm.cr;
    sm.br = (c != 0u) ? __hip_atomic_fetch_add(&w.hdr->n_ring, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  }
  // workgroup sums: doubles by a fixed tree (deterministic), integers in any order
  sabs = wave_sum_f64_dpp(sabs);
  sv = wave_sum_f64_dpp(sv);
  if (lane == 0) {
    sm.dred[0][wid] = sabs;
    sm.dred[1][wid] = sv;
  }
  if (warm) {
    const long long c0s = fpt_wave_sum(tc_ru), c1s = fpt_wave_sum(tc_ct);
    if (lane == 0) {
      sm.red[0][wid] = c0s;
      sm.red[1][wid] = c1s;
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
          sm.red[2 + 4 * j + 0][wid] = q0;
          sm.red[2 + 4 * j + 1][wid] = q1;
          sm.red[2 + 4 * j + 2][wid] = q2;
          sm.red[2 + 4 * j + 3][wid] = q3;
        }
      }
    }
  }
  __syncthreads();
  if (tid < 2) {
    double t = 0.0;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) t += sm.dred[tid][wv];
    w.dpart[2 * wg + tid] = t;
  }
  if (warm && tid >= 64 && tid < 64 + 2 + 4 * K) {
    const int s2 = tid - 64;
    long long t = 0;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) t += sm.red[s2][wv];
    w.part[(size_t)wg * FPT_NQ + s2] = t;
  }
  {
    const unsigned cn = sm.cn, cr = sm.cr, bn = sm.bn, br = sm.br;
    for (unsigned i = tid; i < cn; i += FPT_T) w.narrow[bn + i] = sm.list[i];
    for (unsigned i = tid; i < cr; i += FPT_T) w.ring[br + i] = sm.list[FPT_WGV - 1 - i];
  }
  __syncthreads();
  cx.K = K;
  cx.e_units = e_units;
  cx.warm = warm;
  cx.inv_q = inv_q;
  cx.lo = lo;
  cx.hi = hi;
  cx.d = d;
}

// ---- phase 2: ONE workgroup, after every workgroup's phase 1 has been released to it (acquire done by the caller):
// the fixed point itself.  n values in all, readable at vsrc (the full passes); G workgroups took part in phase 1.
template <int CR, int FC, int PF>
__device__ __forceinline__ void fpt_phase2(FptSmem& sm, const FptCtx& cx, const FptWs& w, FptPred* pred, effq_fp_state* st,
                                           const float* __restrict__ vsrc, bool vec_ok, size_t n, int G, int levels,
                                           double tol, int max_iter, double* alpha_out, int* done_out) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NW = FPT_T / 64;
  const int K = cx.K, e_units = cx.e_units;
  const bool warm = cx.warm;
  const double inv_q = cx.inv_q, lo = cx.lo, hi = cx.hi, d = cx.d;
  const long long tr0 = cx.tr0, tr1 = cx.tr1;
  // ---- phase 2 ------------------------------------------------------------------------------------------------------
  const unsigned m_n = __hip_atomic_load(&w.hdr->n_narrow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned m_r = __hip_atomic_load(&w.hdr->n_ring, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the narrow list (what nearly every iterate scans) in registers: CR entries per thread, the rest is streamed;
  // the first wave holds the first 64 FC entries once more, lane-major: a list that short it scans ALONE
  unsigned long long creg[CR], fcr[FC];
#pragma unroll
  for (int c = 0; c < CR; ++c) {
    const unsigned i = (unsigned)tid + (unsigned)c * FPT_T;
    creg[c] = (i < m_n) ? w.narrow[i] : 0ull;
  }
#pragma unroll
  for (int c = 0; c < FC; ++c) {
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
      sm.dred[0][wid] = t0;
      sm.dred[1][wid] = t1;
    }
    if (tid < FPT_NQ) sm.T[tid] = 0;
  }
  __syncthreads();
  if (tid < 2) {
    double t = 0.0;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) t += sm.dred[tid][wv];
    sm.tot[tid] = t;
  }
  if (warm) {                                    // tallies: thread = (quantity, subset of the workgroups); integers, any order
    const int s2 = tid & 63, sub = tid >> 6;     // 64 quantities x 8 subsets; plain loads: ordered by the acquire above
    if (s2 < 2 + 4 * K) {
      long long t = 0;
      for (int g0 = sub; g0 < G; g0 += NW * PF) {
        long long x[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int g = g0 + u * NW;
          x[u] = (g < G) ? w.part[(size_t)g * FPT_NQ + s2] : 0ll;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) t += x[u];
      }
      atomicAdd(reinterpret_cast<unsigned long long*>(&sm.T[s2]), (unsigned long long)t);
    }
  }
  __syncthreads();
  if (tid == 0) {                                // counters back to zero for the next call
    __hip_atomic_store(&w.hdr->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&w.hdr->n_narrow, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&w.hdr->n_ring, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const double tot_abs = sm.tot[0], tot_v = sm.tot[1];
  // unit of the full passes (always valid for this tensor); the tallies' unit must not overflow either
  int e_full = 0;
  {
    const double bound = (double)(levels - 1) * tot_abs;
    if (bound > 0.0 && bound < 1e300) e_full = 58 - ilogb(bound);
  }
  const bool tallies_ok = warm && ((double)(levels - 1) * tot_abs * inv_q < 2.0e18);
  const double q_tally = ldexp(1.0, -e_units), inv_qf = ldexp(1.0, e_full), q_full = ldexp(1.0, -e_full);

  const long long tr2 = wall_clock64();
  double alpha = tot_abs / (double)n, alpha_prev = -999.0, last0 = 0.0, last1 = 0.0;
  int it = 0, done = 0, n_warm = 0, n_ring = 0, n_full = 0;
  // One-wave iterations: while the iterate stays inside its NARROW bracket and the narrow list fits the first wave's
  // registers, that wave iterates alone - no LDS exchange, no workgroup barrier (0.6 against 2.2 us per iterate); the first
  // iterate that needs more (the ring list, a full pass) goes to the loop of the whole workgroup below.
  if (wid == 0) {
    if (tallies_ok && m_n <= (unsigned)(64 * FC)) {
      const int nfc = (int)((m_n + 63u) / 64u);
      while (!done) {
        if (!(alpha > 0.0) || !(alpha < 1e300)) break;
        const int j = (it < K) ? it : K - 1;
        if (!(alpha >= sm.end[4 * j + 1] && alpha <= sm.end[4 * j + 2])) break;
        if (lane == 0) fpt_note(pred, it, alpha);
        const FpLevel lc = fp_level_consts(alpha, lo, hi, d);
        const unsigned long long want = 1ull << (32 + j);
        long long ru = 0, ct = 0;
#pragma unroll
        for (int c = 0; c < FC; ++c) {
          if (c < nfc && (fcr[c] & want)) {
            const float v = __uint_as_float((unsigned)fcr[c]);
            const int r = fp_level(v, lc, lo, hi, d);
            ru += (long long)r * __double2ll_rn((double)v * inv_q);
            ct += fpt_pack(r);
          }
        }
        const long long Sru = fpt_wave_sum(ru) + sm.T[0] + sm.T[2 + 4 * j] + sm.T[2 + 4 * j + 2];
        const long long Sct = fpt_wave_sum(ct) + sm.T[1] + sm.T[2 + 4 * j + 1] + sm.T[2 + 4 * j + 3];
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
      sm.fd[0] = alpha;
      sm.fd[1] = alpha_prev;
      sm.fd[2] = last0;
      sm.fd[3] = last1;
      sm.fi[0] = it;
      sm.fi[1] = done;
      sm.fi[2] = n_warm;
    }
  }
  __syncthreads();
  alpha = sm.fd[0];
  alpha_prev = sm.fd[1];
  last0 = sm.fd[2];
  last1 = sm.fd[3];
  it = sm.fi[0];
  done = sm.fi[1];
  n_warm = sm.fi[2];
  while (!done) {
    if (!(alpha > 0.0) || !(alpha < 1e300)) {    // NaN / non-positive scale: the reference would spin to its cap
      done = 2;
      break;
    }
    if (tid == 0) fpt_note(pred, it, alpha);
    const int par = it & 1;
    const int j = (it < K) ? it : K - 1;
    const bool in_w = tallies_ok && alpha >= sm.end[4 * j] && alpha <= sm.end[4 * j + 3];
    const bool in_n = in_w && alpha >= sm.end[4 * j + 1] && alpha <= sm.end[4 * j + 2];
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
      for (int c = 0; c < CR; ++c) entry(creg[c]);
      auto stream = [&](const unsigned long long* __restrict__ list, unsigned first, unsigned count) {
        for (unsigned b0 = first + (unsigned)tid; b0 < count; b0 += FPT_T * PF) {
          unsigned long long en[PF];
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            const unsigned i = b0 + (unsigned)u * FPT_T;
            en[u] = (i < count) ? list[i] : 0ull;
          }
#pragma unroll
          for (int u = 0; u < PF; ++u) entry(en[u]);
        }
      };
      if (m_n > (unsigned)(CR * FPT_T)) stream(w.narrow, CR * FPT_T, m_n);
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
      for (size_t b0 = tid; b0 < nv; b0 += (size_t)FPT_T * PF) {
        float4 x4[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const size_t i = b0 + (size_t)u * FPT_T;
          x4[u] = (i < nv) ? reinterpret_cast<const float4*>(vsrc)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
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
      sm.it[par][0][wid] = ru;
      sm.it[par][1][wid] = ct;
    }
    __syncthreads();
    long long Sru = 0, Sct = 0;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) {
      Sru += sm.it[par][0][wv];
      Sct += sm.it[par][1][wv];
    }
    double q_used = q_full;
    if (in_w) {
      Sru += sm.T[0] + sm.T[2 + 4 * j];            // settled over the hull of all brackets + settled by the wide bracket of slot j
      Sct += sm.T[1] + sm.T[2 + 4 * j + 1];
      if (in_n) {                                // values settled by the narrow bracket only: their tally stands
        Sru += sm.T[2 + 4 * j + 2];
        Sct += sm.T[2 + 4 * j + 3];
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
  if (alpha_out != nullptr) *alpha_out = alpha;      // (every thread holds the same values)
  if (done_out != nullptr) *done_out = done;
}

}  // namespace effq
