// project_by_iter (layer_helper.py:40-70) on tensors far too large to keep on chip: the activations of a layer
// (up to 2^27 values per rank, fitted once per layer: PTQConv.py:74-78, EfficientQConv.py:64-72).
//
// The reference re-classifies every value on every iteration (~40 iterations at 4 levels, ~200 at 16): as many passes
// over HBM.  But the level of a value, l(v, a) = rint((clamp(v / a, lo, hi) - lo) / d), is MONOTONE in the scale a (every
// step of the reference's fp64 arithmetic - correctly rounded division, clamp, subtraction, division, rint - is
// monotone), so a value whose level is the same at both ends of an interval [a_lo, a_hi] keeps that level for every
// scale inside it.  The coming iterates are confined to a short interval (predicted from the last iterates; the
// prediction only decides what is worth doing, never what is correct):
//   * a NARROWING pass classifies each value at the current scale AND at both ends of the predicted bracket; values
//     with equal end levels are "decided" - their contribution (level, level^2, level * value) moves into running
//     tallies - the others are appended to a compact list;
//   * the following iterations read only the list (and narrow it further whenever the predicted bracket has shrunk);
//   * an iterate that leaves the bracket invalidates list and tallies: the next pass starts from the BASE again - the
//     tensor itself or, for post-ReLU tensors, the list of its non-zeros that the very first pass leaves behind (an
//     exact zero has the same level at every scale).
// Two kinds of bracket: up to the predicted LIMIT of the sequence (geometric extrapolation with a safety factor: the
// endgame, and everything at few levels), or - while the sequence still converges slowly (many levels: step ratios of
// 0.95) and that bracket would keep most values undecided - only as far as the next K iterations are expected to get,
// with K balancing the list length against the cost of the rebuild from the base that follows (a planned escape).
// The levels are exactly the reference's in every iteration (same fp32 screening + fp64 fallback as k_fp_iter).  The sums
// are integers: every value enters as u = rint(v * 2^e), with e chosen from sum|v| so that no partial sum of level * u
// can overflow 63 bits; integer sums do not depend on the order or on which pass a value was tallied in, so the result is
// run-to-run deterministic and the same for every split of the tensor.  Rounding v to the unit 2^-e (>= 2^-34 of the
// mean |v| for 16 levels and 2^27 values) perturbs the two sums by ~1e-14 relative: alpha agrees with the fp64 kernels to
// ~1e-13, far inside the 1e-11 bar of the parity tests, with the same iteration count.
//
// Launch shape: G <= 1024 persistent workgroups; workgroup w owns slice w of the tensor and segment w of every list, so
// compaction needs no global atomic (one LDS counter per workgroup) and the per-workgroup tallies live in private slots
// that a one-workgroup finish kernel adds up (integers: any order).  Per iteration: k_fbr_iter + k_fbr_finish (which
// also applies the scalar update and plans the next pass); with data-parallel ranks the all-reduce of the two sums sits
// between the two halves of the finish step (effq_fp_bracket_stats / effq_fp_bracket_update).
#include <math.h>
#include "common.h"

namespace effq {

constexpr int FBR_T = 256;
constexpr int FBR_MAXG = 1024;
constexpr int FBR_FT = 1024;                 // threads of the finish kernel: one tally slot each
constexpr size_t FBR_MIN_PER = 8192;      // values per workgroup below which fewer workgroups are used
enum { FBR_X = 0, FBR_A = 1, FBR_B = 2, FBR_Z = 3 };      // sources: the tensor, the two working lists, the base list

// header of the workspace; the first sixteen 8-byte words are read from Python (tests, diagnostics)
struct FbrHdr {
  double blo, bhi;              // bracket under which the current source list and the decided tallies are valid
  double nlo, nhi;              // plan of the next pass: narrowed bracket
  double alpha_pp;              // the iterate before alpha_prev
  double inv_q, q;              // integer unit q = 2^-e
  long long src, dst, narrow;   // source / destination of the next pass
  long long G, per;             // workgroups, values per slice
  long long escapes, narrowings;
  long long visited;            // values read so far by the iteration passes (diagnostics)
  long long list_total;         // length of the current source list
  // ---- planning state ----
  double r_prev;                // ratio of the last two steps as of the previous iteration (0 = none yet)
  double density;               // undecided values per unit of bracket width, from the last narrowing (0 = unknown)
  long long mono;               // consecutive iterations with a positive step ratio
  long long planned;            // the current bracket's far end is not a prediction of the limit (horizon / nesting)
  long long widen;              // doublings of the safety factor: escapes from brackets that WERE such predictions
  long long base;               // FBR_X, or FBR_Z once the first pass has listed the non-zeros
  long long z_total;            // length of the base list
  // ---- a fit that was IMPORTED (data-parallel ranks: effq_fp_bracket_import): its base is the gathered list of the values
  // that were undecided under [guard_lo, guard_hi] and DZ holds the tallies of everything decided under that bracket, so
  // the fit is only valid while the iterates stay inside it: an iterate outside ends it with state.done = 4
  double guard_lo, guard_hi;    // (0, inf) for an ordinary fit
  long long ext[4];             // tallies of everything the OTHER ranks and this one had decided before the exchange (0)
};
static_assert(sizeof(FbrHdr) <= 256, "FbrHdr layout");

struct FbrWs {
  FbrHdr* hdr;
  unsigned* segcnt;          // [3][FBR_MAXG]  length of segment w of list A / B / Z
  long long* D;              // [FBR_MAXG][4]  decided since the base: sum l*u, sum l, sum l^2, sum u
  long long* U;              // [FBR_MAXG][4]  the same sums over the undecided values of the last pass
  long long* DZ;             // [FBR_MAXG][4]  decided between the tensor and the base list
  float* L[3];               // A, B, Z: [G * per] each
};
static void fbr_shape(size_t n, size_t* G, size_t* per) {
  size_t g = (n + FBR_MIN_PER - 1) / FBR_MIN_PER;
  if (g < 1) g = 1;
  if (g > (size_t)FBR_MAXG) g = FBR_MAXG;
  size_t p = (n + g - 1) / g;
  p = (p + 3) & ~(size_t)3;
  *G = g;
  *per = p;
}
static size_t fbr_ws_bytes(size_t n) {
  size_t G, per;
  fbr_shape(n, &G, &per);
  return 256 + sizeof(unsigned) * 4 * FBR_MAXG + sizeof(long long) * 12 * FBR_MAXG + 3 * sizeof(float) * (G * per + 64);
}
static FbrWs fbr_carve(void* ws, size_t n) {
  size_t G, per;
  fbr_shape(n, &G, &per);
  FbrWs w;
  char* p = reinterpret_cast<char*>(ws);
  w.hdr = reinterpret_cast<FbrHdr*>(p);
  p += 256;
  w.segcnt = reinterpret_cast<unsigned*>(p);
  p += sizeof(unsigned) * 4 * FBR_MAXG;
  w.D = reinterpret_cast<long long*>(p);
  p += sizeof(long long) * 4 * FBR_MAXG;
  w.U = reinterpret_cast<long long*>(p);
  p += sizeof(long long) * 4 * FBR_MAXG;
  w.DZ = reinterpret_cast<long long*>(p);
  p += sizeof(long long) * 4 * FBR_MAXG;
  for (int i = 0; i < 3; ++i) {
    w.L[i] = reinterpret_cast<float*>(p);
    p += sizeof(float) * (G * per + 64);
  }
  return w;
}

// level index of the reference (layer_helper.py:25-37 in fp64), screened in fp32 exactly as level_accum of
// quant_reduce.hip: u = (v / a - lo) / d evaluated in fp32 is off by <= 3e-5 at 256 levels and accepted unless it lies
// within 2e-4 of a rounding boundary, where the reference's own arithmetic decides
struct FbrLevel {
  float c1, c0, lmax;
  double a;
};
__device__ __forceinline__ FbrLevel fbr_level_consts(double a, double lo, double hi, double d) {
  FbrLevel c;
  const double rd = 1.0 / d;
  c.c1 = (float)((1.0 / a) * rd);
  c.c0 = (float)(-lo * rd);
  c.lmax = (float)rint((hi - lo) * rd);
  c.a = a;
  return c;
}
__device__ __forceinline__ int fbr_level(float v, const FbrLevel& c, double lo, double hi, double d) {
  float u = __builtin_fmaf(v, c.c1, c.c0);
  u = fminf(fmaxf(u, 0.0f), c.lmax);
  float rf = rintf(u);
  if (!(fabsf(u - rf) < 0.4998f)) {
    double t = (double)v / c.a;
    t = fmin(fmax(t, lo), hi);
    rf = (float)rint((t - lo) / d);
  }
  return (int)rf;
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ __launch_bounds__(FBR_T) void k_fbr_iter(const float* __restrict__ x, size_t n, FbrWs w,
                                                    const effq_fp_state* __restrict__ st, double lo, double hi,
                                                    double d, int vec_ok) {
  if (st->done != 0) return;          // uniform across the grid
  __shared__ unsigned s_out;
  __shared__ long long s_red[8][FBR_T / 64];
  const FbrHdr h = *w.hdr;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wg = blockIdx.x;
  const size_t per = (size_t)h.per;
  const float* src;
  size_t cnt;
  if (h.src == FBR_X) {
    const size_t s0 = (size_t)wg * per;
    cnt = (s0 < n) ? ((n - s0 < per) ? n - s0 : per) : 0;
    src = x + s0;
  } else {
    src = w.L[h.src - 1] + (size_t)wg * per;
    cnt = w.segcnt[(h.src - 1) * FBR_MAXG + wg];
  }
  const bool narrow = h.narrow != 0;
  float* __restrict__ dst = w.L[narrow ? h.dst - 1 : 0] + (size_t)wg * per;
  const FbrLevel la = fbr_level_consts(st->alpha, lo, hi, d);
  const FbrLevel ll = fbr_level_consts(narrow ? h.nlo : st->alpha, lo, hi, d);
  const FbrLevel lh = fbr_level_consts(narrow ? h.nhi : st->alpha, lo, hi, d);
  const double inv_q = h.inv_q;
  if (tid == 0) s_out = 0u;
  __syncthreads();

  long long D0 = 0, D1 = 0, D2 = 0, D3 = 0, U0 = 0, U1 = 0, U2 = 0, U3 = 0;
  // one value: tallies; returns whether it stays undecided under the new bracket
  auto tally = [&](float v, bool valid) -> bool {
    const int r = fbr_level(v, la, lo, hi, d);
    const long long u = __double2ll_rn((double)v * inv_q);
    bool und = true;
    if (narrow) und = fbr_level(v, ll, lo, hi, d) != fbr_level(v, lh, lo, hi, d);
    const long long ru = (long long)r * u, r1 = valid ? (long long)r : 0ll, r2 = valid ? (long long)(r * r) : 0ll;
    const long long ruv = valid ? ru : 0ll, uv = valid ? u : 0ll;
    if (und) {
      U0 += ruv; U1 += r1; U2 += r2; U3 += uv;
    } else {
      D0 += ruv; D1 += r1; D2 += r2; D3 += uv;
    }
    return und && valid;
  };
  // the undecided values of one wave step are appended behind one LDS counter bump
  auto append4 = [&](const float (&v)[4], const bool (&keep)[4]) {
    unsigned long long m[4];
    unsigned tot = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      m[c] = __builtin_amdgcn_ballot_w64(keep[c]);
      tot += (unsigned)__builtin_popcountll(m[c]);
    }
    if (tot == 0u) return;            // wave-uniform
    unsigned base = 0u;
    if (lane == 0) base = atomicAdd(&s_out, tot);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(m[c] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[c], 0u));
      if (keep[c]) dst[base + below] = v[c];
      base += (unsigned)__builtin_popcountll(m[c]);
    }
  };

  const size_t nvec = vec_ok ? cnt / 4 : 0;
  const size_t nvec_pad = (nvec + 63) & ~(size_t)63;
  // wave-uniform trip count (ballots inside); two 16-byte loads in flight per lane
  for (size_t j0 = (size_t)wid * 64; j0 < nvec_pad; j0 += 2 * FBR_T) {
    const size_t ja = j0 + lane, jb = ja + FBR_T;
    const bool va = ja < nvec, vb = jb < nvec;
    float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa;
    if (va) qa = reinterpret_cast<const float4*>(src)[ja];
    if (vb) qb = reinterpret_cast<const float4*>(src)[jb];
    {
      const float v[4] = {qa.x, qa.y, qa.z, qa.w};
      bool keep[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) keep[c] = tally(v[c], va);
      if (narrow) append4(v, keep);
    }
    if (j0 + FBR_T < nvec_pad) {       // wave-uniform
      const float v[4] = {qb.x, qb.y, qb.z, qb.w};
      bool keep[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) keep[c] = tally(v[c], vb);
      if (narrow) append4(v, keep);
    }
  }
  // ragged tail (and the whole slice when the tensor is not 16-byte aligned): one value per lane and step
  const size_t t0 = nvec * 4;
  const size_t ntail_pad = (cnt - t0 + 63) & ~(size_t)63;
  for (size_t j0 = (size_t)wid * 64; j0 < ntail_pad; j0 += FBR_T) {
    const size_t j = t0 + j0 + lane;
    const bool valid = j < cnt;
    const float v[4] = {valid ? src[j] : 0.f, 0.f, 0.f, 0.f};
    bool keep[4] = {tally(v[0], valid), false, false, false};
    if (narrow) append4(v, keep);
  }

  long long acc[8] = {D0, D1, D2, D3, U0, U1, U2, U3};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    acc[s] = wave_sum_i64(acc[s]);
    if (lane == 0) s_red[s][wid] = acc[s];
  }
  __syncthreads();
  if (tid < 8) {
    long long t = 0;
#pragma unroll
    for (int wv = 0; wv < FBR_T / 64; ++wv) t += s_red[tid][wv];
    if (tid < 4) {
      if (narrow && h.dst == FBR_Z) {           // the pass that builds the base list
        w.DZ[wg * 4 + tid] = t;
        w.D[wg * 4 + tid] = 0;
      } else if (h.src == FBR_X || h.src == FBR_Z) {   // a pass over the base restarts the decided tallies
        w.D[wg * 4 + tid] = t;
      } else {
        w.D[wg * 4 + tid] += t;
      }
    } else {
      w.U[wg * 4 + (tid - 4)] = t;
    }
  }
  if (tid == 0 && narrow) w.segcnt[(h.dst - 1) * FBR_MAXG + wg] = s_out;
}

// The plan of the pass that classifies at a_new (thread 0 of the finish kernel).
__device__ void fbr_plan(FbrHdr* h, double a_new, double alpha, double a_pp, int it, size_t n) {
  const bool base_z = h->base == FBR_Z;
  if (h->src != FBR_X && !(a_new >= h->blo && a_new <= h->bhi)) {     // the iterate left the bracket (or is NaN)
    h->src = h->base;
    h->blo = fmax(base_z ? 1e-300 : 0.0, h->guard_lo);
    h->bhi = fmin(base_z ? 1e300 : INFINITY, h->guard_hi);
    h->list_total = base_z ? h->z_total : 0;
    h->escapes += 1;
    // a far end set by a horizon or by an older bracket says nothing about the prediction of the limit; only an escape
    // from a bracket that WAS that prediction widens the next ones
    if (h->planned == 0 && h->widen < 4) h->widen += 1;
    h->planned = 0;
  }
  h->narrow = 0;
  if (!(it >= 2 && a_new > 0.0 && a_new < 1e300)) return;
  const double d1 = a_new - alpha, d0 = alpha - a_pp;
  const double safety = 3.0 * (double)(1 << (int)h->widen);
  const bool at_base = (h->src == h->base);
  double nlo, nhi;
  bool planned = false;
  const double r = d1 / d0;
  if (r > 0.0 && r < 1e6) {                      // monotone so far: geometric extrapolation of what is left
    // the step ratio typically still GROWS towards its asymptotic value while the first brackets are chosen (many
    // levels: 0.56, 0.67, 0.73 ... 0.96): lean on its trend, and never trust a ratio below the last one
    const double rp = h->r_prev;
    double rc = r;
    if (rp > 0.0 && r > rp) rc = r + 2.0 * (r - rp);
    if (rp > rc) rc = rp;
    if (rc > 0.985) rc = 0.985;
    const double T = d1 * rc / (1.0 - rc);       // signed remaining travel
    const double sg = (T < 0.0) ? -1.0 : 1.0, ad = fabs(d1);
    double ahead = safety * fabs(T) + 0.25 * ad;
    // behind the iterate: a monotone sequence does not come back; until that has shown, half the travel
    const double behind = (h->mono >= 2) ? 0.1 * ad : 0.5 * fabs(T) + 0.25 * ad;
    if (h->mono >= 2 && ad > 0.0) {
      // Horizon instead of limit?  Per iteration a bracket of width W keeps ~density * W values undecided (and the
      // list thins out as the iterate crosses it: ~0.6 of that on average); a horizon of K steps costs ~density * K * d
      // per iteration plus the rebuild from the base every K iterations: least at K = sqrt(base / (density * d)).
      const double R = base_z ? (double)h->z_total : (double)n;
      const double dens = (h->density > 0.0) ? h->density : R / a_new;
      const double cost_limit = 0.6 * dens * ahead, cost_horizon = 2.0 * sqrt(R * dens * ad);
      if (cost_horizon < 0.8 * cost_limit) {
        double K = sqrt(R / (dens * ad));
        K = fmin(fmax(K, 2.0), 64.0);
        const double TK = ad * rc * (1.0 - pow(rc, K)) / (1.0 - rc);
        ahead = 1.3 * TK + 0.25 * ad;
        planned = true;
      }
    }
    const double fwd = a_new + sg * ahead, back = a_new - sg * behind;
    nlo = fmin(fwd, back);
    nhi = fmax(fwd, back);
    h->r_prev = r;
    h->mono += 1;
  } else {                                       // oscillating (or the very first differences)
    const double wdt = safety * fabs(d1);
    nlo = a_new - wdt;
    nhi = a_new + wdt;
    h->r_prev = 0.0;
    h->mono = 0;
  }
  if ((nhi > h->bhi && d1 > 0.0) || (nlo < h->blo && d1 < 0.0)) planned = true;      // cut short by the nesting
  nlo = fmax(fmax(nlo, h->blo), a_new * 1e-3);
  nhi = fmin(nhi, h->bhi);
  if (nlo <= a_new && a_new <= nhi && nlo < nhi) {
    const bool worth = at_base || ((nhi - nlo) < 0.7 * (h->bhi - h->blo));
    if (worth) {
      h->narrow = 1;
      h->dst = (h->src == FBR_A) ? FBR_B : FBR_A;
      h->nlo = nlo;
      h->nhi = nhi;
      h->planned = planned ? 1 : 0;
    }
  }
}

// mode bit 0: add the per-workgroup tallies -> st->sums = [sum b*x, sum b*b] of this rank
// mode bit 1: scalar update from st->sums (layer_helper.py:55-60) and the plan of the next pass
__global__ __launch_bounds__(FBR_FT) void k_fbr_finish(FbrWs w, effq_fp_state* st, size_t n, double lo, double d,
                                                      double tol, int max_iter, int mode) {
  if (st->done != 0) return;
  __shared__ long long s_red[5][FBR_FT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  FbrHdr* h = w.hdr;
  const int G = (int)h->G;
  if (mode & 1) {
    long long a[5] = {0, 0, 0, 0, 0};
    for (int g = tid; g < G; g += FBR_FT) {
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] += w.DZ[g * 4 + s] + w.D[g * 4 + s] + w.U[g * 4 + s];
      if (h->narrow != 0) a[4] += (long long)w.segcnt[(h->dst - 1) * FBR_MAXG + g];
    }
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      a[s] = wave_sum_i64(a[s]);
      if (lane == 0) s_red[s][wid] = a[s];
    }
    __syncthreads();
    if (tid == 0) {
      long long t[5];
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        t[s] = (s < 4) ? h->ext[s] : 0;
        for (int wv = 0; wv < FBR_FT / 64; ++wv) t[s] += s_red[s][wv];
      }
      const double q = h->q;
      const double srx = q * (double)t[0], sx = q * (double)t[3];
      st->sums[0] = d * srx + lo * sx;
      st->sums[1] = (d * d * (double)t[2] + 2.0 * d * lo * (double)t[1]) + lo * lo * (double)n;
      // bookkeeping of the pass that has just run: commit its narrowing
      h->visited += (h->src == FBR_X) ? (long long)n : h->list_total;
      if (h->narrow != 0) {
        if (h->dst == FBR_Z) {
          h->base = FBR_Z;
          h->z_total = t[4];
        } else if (h->nhi > h->nlo) {
          h->density = (double)t[4] / (h->nhi - h->nlo);
        }
        h->src = h->dst;
        h->blo = h->nlo;
        h->bhi = h->nhi;
        h->list_total = t[4];
        h->narrowings += 1;
        h->narrow = 0;
      }
    }
  }
  if ((mode & 2) && tid == 0) {
    const double alpha = st->alpha;
    const double a_new = st->sums[0] / st->sums[1];
    const double a_pp = st->alpha_prev;
    h->alpha_pp = a_pp;
    st->alpha_prev = alpha;
    st->alpha = a_new;
    const int it = st->iters + 1;
    st->iters = it;
    int done = 0;
    if (it >= max_iter)
      done = 2;
    else if (!(fabs(a_new - alpha) > tol))
      done = 1;
    else if (!(a_new >= h->guard_lo && a_new <= h->guard_hi))
      done = 4;                                  // left the bracket an imported fit is valid under: the caller falls back
    st->done = done;
    if (done == 0) fbr_plan(h, a_new, alpha, a_pp, it, n);
  }
}

__global__ __launch_bounds__(FBR_T) void k_fbr_init(FbrWs w, effq_fp_state* st, const double* abs_sums, long long G,
                                                    long long per, int levels, int list_first) {
  for (int i = threadIdx.x; i < 4 * FBR_MAXG; i += FBR_T) {
    w.DZ[i] = 0;
    w.D[i] = 0;
    w.U[i] = 0;
  }
  if (threadIdx.x != 0) return;
  const double S = abs_sums[0], cnt = abs_sums[1];
  st->alpha = S / cnt;
  st->alpha_prev = -999.0;
  st->sums[0] = S;
  st->sums[1] = cnt;
  st->iters = 0;
  st->done = 0;
  // integer unit: (levels - 1) * (sum|v| / q + n / 2) < 2^61
  int e = 0;
  const double bound = (double)(levels - 1) * S;
  if (bound > 0.0 && bound < 1e300) e = 60 - ilogb(bound);
  if (e > 1000) e = 1000;
  if (e < -1000) e = -1000;
  FbrHdr* h = w.hdr;
  h->blo = 0.0;
  h->bhi = INFINITY;
  h->nlo = 0.0;
  h->nhi = INFINITY;
  h->alpha_pp = -999.0;
  h->inv_q = ldexp(1.0, e);
  h->q = ldexp(1.0, -e);
  h->src = FBR_X;
  h->dst = FBR_A;
  // list_first: the very first pass already leaves out what is settled for EVERY scale (exact zeros: half or more of
  // a post-ReLU tensor) and its output becomes the base that rebuilds start from
  h->narrow = list_first ? 1 : 0;
  if (list_first) {
    h->dst = FBR_Z;
    h->nlo = 1e-300;
    h->nhi = 1e300;
  }
  h->G = G;
  h->per = per;
  h->escapes = 0;
  h->narrowings = 0;
  h->visited = 0;
  h->list_total = 0;
  h->r_prev = 0.0;
  h->density = 0.0;
  h->mono = 0;
  h->planned = 0;
  h->widen = 0;
  h->base = FBR_X;
  h->z_total = 0;
  h->guard_lo = 0.0;
  h->guard_hi = INFINITY;
  for (int i = 0; i < 4; ++i) h->ext[i] = 0;
}

// ---- data-parallel ranks: gather once, finish everywhere -------------------------------------------------------------
// A fit over volumes sharded across ranks needs the two sums of EVERY iteration all-reduced: ~40 collectives per layer at
// 4 levels, each between two dependent kernels.  But once the bracket [blo, bhi] is narrow, all that is left of a rank's
// shard is (a) four integers - the tallies of the values decided under the bracket - and (b) the short list of the
// undecided ones.  effq_fp_bracket_export hands both out; the caller all-reduces (a), all-gathers (b), and
// effq_fp_bracket_import sets up a fit over the gathered list with the summed tallies as its constant part, which every
// rank then runs to the end ON ITS OWN - same integers, same order, same kernels: bit-identical iterates on every rank, no
// further collective.  The imported fit is valid while its iterates stay inside [blo, bhi] (state.done = 4 otherwise).
// out: [0..3] tallies, [4] list length (-1: no list yet, -2: horizon bracket, see below), [5], [6] bit patterns of the
// bracket, [8 .. 8 + G] prefix offsets of the list segments (scratch for k_fbr_export_copy)
__global__ __launch_bounds__(FBR_FT) void k_fbr_export(FbrWs w, long long* __restrict__ out) {
  __shared__ long long s_red[4][FBR_FT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const FbrHdr h = *w.hdr;
  const int G = (int)h.G;
  long long a[4] = {0, 0, 0, 0};
  for (int g = tid; g < G; g += FBR_FT)
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] += w.DZ[g * 4 + s] + w.D[g * 4 + s];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    a[s] = wave_sum_i64(a[s]);
    if (lane == 0) s_red[s][wid] = a[s];
  }
  __syncthreads();
  if (tid < 4) {
    long long t = h.ext[tid];
    for (int wv = 0; wv < FBR_FT / 64; ++wv) t += s_red[tid][wv];
    out[tid] = t;
  }
  if (tid == 0) {
    long long run = 0;
    for (int g = 0; g < G; ++g) {
      out[8 + g] = run;
      run += (h.src == FBR_X) ? 0 : (long long)w.segcnt[(h.src - 1) * FBR_MAXG + g];
    }
    out[8 + G] = run;
    // -1: no list yet; -2: the list belongs to a HORIZON bracket (many levels: it only reaches as far as the next few
    // iterations are expected to get, and the iterates leave it by design): not worth exchanging
    out[4] = (h.src == FBR_X) ? -1 : ((h.planned != 0) ? -2 : run);
    out[5] = __double_as_longlong(h.blo);      // the bracket these tallies and this list are valid under: ranks plan on
    out[6] = __double_as_longlong(h.bhi);      // their own shard's density, so their brackets may differ
  }
}

// workgroup g copies segment g of the list behind the segments before it; everything beyond the list - or the whole
// buffer if there is no list to hand out or it does not fit - is zero-filled (zeros are harmless to the receiver)
__global__ __launch_bounds__(FBR_T) void k_fbr_export_copy(FbrWs w, const long long* __restrict__ out,
                                                          float* __restrict__ list_out, size_t cap) {
  const FbrHdr h = *w.hdr;
  const int G = (int)h.G, g = blockIdx.x, tid = threadIdx.x;
  const long long count = out[4];
  const bool ok = count >= 0 && (size_t)count <= cap;
  const size_t used = ok ? (size_t)count : 0;
  if (ok && h.src != FBR_X) {
    const float* L = w.L[h.src - 1] + (size_t)g * (size_t)h.per;
    const size_t o = (size_t)out[8 + g], c = (size_t)(out[8 + g + 1] - out[8 + g]);
    for (size_t i = tid; i < c; i += FBR_T) list_out[o + i] = L[i];
  }
  const size_t tail = cap - used, per = (tail + G - 1) / G;
  const size_t z0 = used + (size_t)g * per, z1 = (z0 + per < cap) ? z0 + per : cap;
  for (size_t i = z0 + tid; i < z1; i += FBR_T) list_out[i] = 0.0f;
}

// back from an imported fit that left its bracket: the rank's own list and tallies belong to a bracket the iterates have
// moved on from since - the next pass starts from the base, exactly as after an escape
__global__ void k_fbr_rebase(FbrWs w, effq_fp_state* st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  FbrHdr* h = w.hdr;
  const bool base_z = h->base == FBR_Z;
  h->src = h->base;
  h->blo = fmax(base_z ? 1e-300 : 0.0, h->guard_lo);
  h->bhi = fmin(base_z ? 1e300 : INFINITY, h->guard_hi);
  h->list_total = base_z ? h->z_total : 0;
  h->narrow = 0;
  h->planned = 0;
  h->r_prev = 0.0;
  h->mono = 0;
  if (st->done == 4) st->done = 0;
}

// pack (all-reduced over the ranks): [0..3] summed tallies, then per rank {list length, bits of blo, bits of bhi}.
// Decided ON THE DEVICE (the host has not seen the lengths): the exchange is usable if every rank handed out a list that
// fitted its slot and the current iterate lies inside the intersection of the ranks' brackets; otherwise state.done = 4 and
// the launches of the imported fit that follow are no-ops.
__global__ __launch_bounds__(FBR_T) void k_fbr_import(FbrWs dst, const FbrHdr* __restrict__ src_hdr,
                                                      const long long* __restrict__ pack, int world, long long cap,
                                                      long long G, long long per, effq_fp_state* st) {
  for (int i = threadIdx.x; i < 4 * FBR_MAXG; i += FBR_T) {
    dst.DZ[i] = 0;
    dst.D[i] = 0;
    dst.U[i] = 0;
  }
  if (threadIdx.x != 0) return;
  double glo = 0.0, ghi = INFINITY;
  bool ok = st->done == 0;
  for (int r = 0; r < world; ++r) {
    const long long len = pack[4 + 3 * r];
    if (len < 0 || len > cap) ok = false;
    glo = fmax(glo, __longlong_as_double(pack[5 + 3 * r]));
    ghi = fmin(ghi, __longlong_as_double(pack[6 + 3 * r]));
  }
  if (!(glo <= st->alpha && st->alpha <= ghi)) ok = false;
  FbrHdr h = *src_hdr;              // iterates and integer unit: carried over (the same on every rank)
  // valid under the INTERSECTION of the ranks' brackets; the planning state starts afresh, so that every rank plans alike
  // from here (it only ever decides what is worth doing - the iterates do not depend on it)
  h.guard_lo = glo;
  h.guard_hi = ghi;
  h.blo = glo;
  h.bhi = ghi;
  for (int i = 0; i < 4; ++i) h.ext[i] = pack[i];
  h.r_prev = 0.0;
  h.density = 0.0;
  h.mono = 0;
  h.planned = 0;
  h.widen = 0;
  h.escapes = 0;
  h.narrowings = 0;
  h.visited = 0;
  h.src = FBR_X;                    // the gathered lists ARE this fit's tensor; its first pass drops their zero padding
  h.dst = FBR_Z;                    // (and whatever the narrower common bracket decides) into the base list
  h.narrow = 1;
  h.nlo = fmax(glo, 1e-300);
  h.nhi = fmin(ghi, 1e300);
  h.base = FBR_X;
  h.z_total = 0;
  h.list_total = 0;
  h.G = G;
  h.per = per;
  *dst.hdr = h;
  if (!ok && st->done == 0) st->done = 4;
}

}  // namespace effq
using namespace effq;

extern "C" {

size_t effq_fp_bracket_ws_bytes(size_t n) { return fbr_ws_bytes(n); }

int effq_fp_bracket_init(effq_fp_state* state_dev, const double* abs_sums_dev, size_t n, int levels, int list_first,
                         void* ws, size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG(state_dev && abs_sums_dev && ws && n > 0 && levels >= 2 && levels <= 256);
  if (ws_bytes < fbr_ws_bytes(n)) {
    set_error("fp_bracket: workspace %zu < %zu bytes", ws_bytes, fbr_ws_bytes(n));
    return EFFQ_ERR_WORKSPACE;
  }
  size_t G, per;
  fbr_shape(n, &G, &per);
  hipLaunchKernelGGL(k_fbr_init, dim3(1), dim3(FBR_T), 0, as_stream(stream), fbr_carve(ws, n), state_dev, abs_sums_dev,
                     (long long)G, (long long)per, levels, list_first);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

static int fbr_launch_iter(const float* x, size_t n, const FbrWs& w, effq_fp_state* st, double lo, double hi, double d,
                           hipStream_t s) {
  size_t G, per;
  fbr_shape(n, &G, &per);
  const int vec_ok = ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) ? 1 : 0;
  hipLaunchKernelGGL(k_fbr_iter, dim3((unsigned)G), dim3(FBR_T), 0, s, x, n, w, st, lo, hi, d, vec_ok);
  return 0;
}

int effq_fp_bracket_run(const float* x, size_t n, int levels, double lo, double hi, double tol, int max_iter, int n_iters,
                        effq_fp_state* state_dev, void* ws, void* stream) {
  EFFQ_CHECK_ARG(x && state_dev && ws && n > 0 && n_iters >= 0 && levels >= 2 && levels <= 256 && hi > lo);
  const double d = (hi - lo) / (double)(levels - 1);
  const FbrWs w = fbr_carve(ws, n);
  hipStream_t s = as_stream(stream);
  for (int i = 0; i < n_iters; ++i) {
    fbr_launch_iter(x, n, w, state_dev, lo, hi, d, s);
    hipLaunchKernelGGL(k_fbr_finish, dim3(1), dim3(FBR_FT), 0, s, w, state_dev, n, lo, d, tol, max_iter, 3);
  }
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_fp_bracket_stats(const float* x, size_t n, int levels, double lo, double hi, effq_fp_state* state_dev, void* ws,
                          void* stream) {
  EFFQ_CHECK_ARG(x && state_dev && ws && n > 0 && levels >= 2 && levels <= 256 && hi > lo);
  const double d = (hi - lo) / (double)(levels - 1);
  const FbrWs w = fbr_carve(ws, n);
  hipStream_t s = as_stream(stream);
  fbr_launch_iter(x, n, w, state_dev, lo, hi, d, s);
  hipLaunchKernelGGL(k_fbr_finish, dim3(1), dim3(FBR_FT), 0, s, w, state_dev, n, lo, d, 0.0, 1, 1);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

size_t effq_fp_bracket_export_words(void) { return 8 + FBR_MAXG + 1; }

int effq_fp_bracket_export(const void* ws, size_t n, long long* out, float* list_out, size_t list_cap, void* stream) {
  EFFQ_CHECK_ARG(ws && out && list_out && n > 0 && list_cap > 0);
  size_t G, per;
  fbr_shape(n, &G, &per);
  const FbrWs w = fbr_carve(const_cast<void*>(ws), n);
  hipLaunchKernelGGL(k_fbr_export, dim3(1), dim3(FBR_FT), 0, as_stream(stream), w, out);
  hipLaunchKernelGGL(k_fbr_export_copy, dim3((unsigned)G), dim3(FBR_T), 0, as_stream(stream), w, out, list_out, list_cap);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_fp_bracket_import(const void* ws_src, size_t n_src, const long long* pack_dev, int world, size_t list_cap,
                           effq_fp_state* state_dev, void* ws_dst, size_t ws_dst_bytes, void* stream) {
  EFFQ_CHECK_ARG(ws_src && pack_dev && state_dev && ws_dst && n_src > 0 && world > 0 && list_cap > 0 && ws_src != ws_dst);
  const size_t m = (size_t)world * list_cap;
  if (ws_dst_bytes < fbr_ws_bytes(m)) {
    set_error("fp_bracket_import: workspace %zu < %zu bytes", ws_dst_bytes, fbr_ws_bytes(m));
    return EFFQ_ERR_WORKSPACE;
  }
  size_t G, per;
  fbr_shape(m, &G, &per);
  hipLaunchKernelGGL(k_fbr_import, dim3(1), dim3(FBR_T), 0, as_stream(stream), fbr_carve(ws_dst, m),
                     fbr_carve(const_cast<void*>(ws_src), n_src).hdr, pack_dev, world, (long long)list_cap, (long long)G,
                     (long long)per, state_dev);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_fp_bracket_rebase(effq_fp_state* state_dev, void* ws, size_t n, void* stream) {
  EFFQ_CHECK_ARG(state_dev && ws && n > 0);
  hipLaunchKernelGGL(k_fbr_rebase, dim3(1), dim3(64), 0, as_stream(stream), fbr_carve(ws, n), state_dev);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_fp_bracket_update(size_t n, int levels, double lo, double hi, double tol, int max_iter,
                           effq_fp_state* state_dev, void* ws, void* stream) {
  EFFQ_CHECK_ARG(state_dev && ws && n > 0 && levels >= 2 && levels <= 256 && hi > lo && max_iter > 0);
  const double d = (hi - lo) / (double)(levels - 1);
  hipLaunchKernelGGL(k_fbr_finish, dim3(1), dim3(FBR_FT), 0, as_stream(stream), fbr_carve(ws, n), state_dev, n, lo, d,
                     tol, max_iter, 2);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
