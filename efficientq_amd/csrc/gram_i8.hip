// effq_gram_accum_i8: the attention-weighted Gram system of a layer whose input is already quantised, on the
// i8 matrix cores with EXACT integer accumulation.
// Reference: solver.py:86-111 (im2col_loop), :253-257 (ones row), :282-314 (getA0B0) - same quantity as
// gram.hip, evaluated without rounding:
//   xhat_v = s * u_v,  s = alpha_act / (La - 1),  u_v = level ids of the receptive field (u8, tap-major)
//   A0 = 2 s^2 sum_v a_v u_v u_v^T   (+ the ones row/column for the bias)
//   B0 = 2 s q sum_v a_v Y_v u_v^T,  Y_v = rint(y_v / q) as a 32-bit fixed-point integer, q = 2^e, split into
//        four balanced base-256 digits so that it rides the same i8 MFMAs (the fixed point keeps 30 bits
//        below max|y|: its error is far below the fp32 rounding of the reference's own product).
// Extended integer rows:  [ x rows, r = tap*C1 + c | ones cell (16 rows, first = 1) | y digit rows d*C2P + c ].
// K dimension = voxels.  The attention weight a_v is NOT multiplied in: the host sorts the voxels by weight
// value (the reference's masks hold a handful of distinct class weights, ptqer.py:141-167) into a voxel list
// whose 128-voxel chunks are weight-uniform; integer partial sums are flushed per class (int64 atomics: exact,
// hence order independent and deterministic) and the finish kernel applies w_cls in fp64 in class order.
// Operands are gathered from NDHWC memory as 16 channels x 4 voxels units, byte-transposed in registers
// (v_perm_b32) and staged in LDS as [row][voxel] so that a lane's 16 K-bytes are one conflict-free ds_read_b128.
#include <stdint.h>
#include <stdlib.h>
#include "common.h"

namespace effq {

typedef int gi_v4i __attribute__((ext_vector_type(4)));
typedef int gi_v16i __attribute__((ext_vector_type(16)));

constexpr int GI_KC = 128;                 // voxels per chunk (4 MFMA K steps)
constexpr int GI_MB = 128;                 // rows per macro block
constexpr int GI_RS = GI_KC + 16;          // LDS row stride (bytes): 144 keeps ds_read_b128 conflict free
constexpr int GI_T = 256;                  // threads: 4 waves, each a 64x64 sub-block
constexpr int GI_PANEL = GI_MB * GI_RS;    // bytes per staged panel
constexpr int GI_LDS = 4 * GI_PANEL + 2 * GI_KC * 16;
constexpr int GI_MAXCLS = 16;
constexpr int GI_MAX_CPS = 1000;           // chunks per split: 128 * 1000 * 128*127 < 2^31 (int32 accumulators)

struct GramI8Params {
  const uint8_t* x;
  const int8_t* yd;
  const int* vox_list;
  const int* chunk_cls;   // int32 so that the (uniform) reads are scalar loads, off the vmcnt queue
  const int* vtab;        // [nchunks*128][4] {xbase, tap mask, vout, -} per list slot, built once per launch
  int N, C1, C2, C2P, D, H, W, OD, OH, OW;
  int KD, KH, KW, SD, SH, SW, PD, PH, PW;
  int T, RX, YR0, E, NB, NBX, npairs;
  long long V;
  int nchunks, nsplit, cps, ncls;
  long long* slabs;   // [ncls][npairs][128*128]
  int debug;          // profiling ablations (EFFQ_GI8_DEBUG): 1 no MFMA, 2 no global loads, 3 no LDS staging
};

__device__ __forceinline__ void gi_transpose4(int r0, int r1, int r2, int r3, int (&o)[4]) {
  // bytes of r_q = channels c..c+3 of voxel q  ->  o[b] = channel c+b of voxels 0..3
  const int t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400);
  const int t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602);
  const int t2 = __builtin_amdgcn_perm(r3, r2, 0x05010400);
  const int t3 = __builtin_amdgcn_perm(r3, r2, 0x07030602);
  o[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100);
  o[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302);
  o[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100);
  o[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302);
}

// per-voxel gather entry: input-space corner of the receptive field, validity mask of the taps (bit 31 = voxel
// valid) and the output voxel index.  Built ONCE per launch for every list slot (k_gi_voxtable); the main kernel
// used to rebuild it per (block pair, chunk) - three integer divisions and a 27-step mask loop on half of its waves.
__device__ __forceinline__ gi_v4i gi_build_entry(const GramI8Params& p, int v) {
  gi_v4i e = {0, 0, 0, 0};
  if (v < 0) return e;
  int t = v;
  const int ow = t % p.OW;
  t /= p.OW;
  const int oh = t % p.OH;
  t /= p.OH;
  const int od = t % p.OD;
  const int n = t / p.OD;
  const int id0 = od * p.SD - p.PD, ih0 = oh * p.SH - p.PH, iw0 = ow * p.SW - p.PW;
  unsigned m = 0x80000000u;
  int tap = 0;
  for (int kd = 0; kd < p.KD; ++kd)
    for (int kh = 0; kh < p.KH; ++kh)
      for (int kw = 0; kw < p.KW; ++kw, ++tap) {
        const int id = id0 + kd, ih = ih0 + kh, iw = iw0 + kw;
        const bool ok = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        m |= (ok ? 1u : 0u) << tap;
      }
  e[0] = ((n * p.D + id0) * p.H + ih0) * p.W + iw0;
  e[1] = (int)m;
  e[2] = v;
  return e;
}

__global__ __launch_bounds__(256) void k_gi_voxtable(GramI8Params p, gi_v4i* __restrict__ table) {
  const long long nslots = (long long)p.nchunks * GI_KC;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += stride) {
    const int v = (p.vox_list != nullptr) ? p.vox_list[s] : ((s < p.V) ? (int)s : -1);
    table[s] = gi_build_entry(p, v);
  }
}

__global__ __launch_bounds__(GI_T, 2) void k_gram_i8(GramI8Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gi_smem[];
  unsigned char* const panI = gi_smem;                      // [2][GI_PANEL]
  unsigned char* const panJ = gi_smem + 2 * GI_PANEL;       // [2][GI_PANEL]
  gi_v4i* const tbl = reinterpret_cast<gi_v4i*>(gi_smem + 4 * GI_PANEL);   // [2][GI_KC] {xbase, mask, vout, -}

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wr = wid >> 1, wc = wid & 1;

  int I = 0, rem = blockIdx.x;
  while (rem >= p.NB - I) {
    rem -= p.NB - I;
    ++I;
  }
  const int J = I + rem;
  const bool diag = (I == J);

  const int c_begin = blockIdx.y * p.cps;
  const int c_end = min(c_begin + p.cps, p.nchunks);
  if (c_begin >= c_end) return;   // uniform over the workgroup

  // staging role: 16 rows (16*rg ..) x 4 voxels of each panel.  The 4 voxels are list slots vg, vg+32, vg+64,
  // vg+96 of the chunk, so that one load instruction walks 32 CONSECUTIVE voxels (whole cache lines); they
  // land in K positions 4*vg+q of the LDS row - the K order is arbitrary as long as both panels share it.
  const int vg = tid & 31, rg = tid >> 5;
  // row-group descriptor: kind 0 x cell, 1 y digit cell, 2 ones cell, 3 padding
  int kindI, tapI, offI, kindJ, tapJ, offJ;
  auto make_rg = [&](int r0, int& kind, int& tap, int& off) {
    kind = 3;
    tap = 0;
    off = 0;
    if (r0 < p.RX) {
      kind = 0;
      tap = r0 / p.C1;
      const int c0 = r0 - tap * p.C1;
      const int kw = tap % p.KW, t2 = tap / p.KW;
      const int kh = t2 % p.KH, kd = t2 / p.KH;
      off = ((kd * p.H + kh) * p.W + kw) * p.C1 + c0;
    } else if (r0 == p.RX) {
      kind = 2;
    } else if (r0 >= p.YR0 && r0 < p.E) {
      kind = 1;
      off = r0 - p.YR0;     // d*C2P + c0
    }
  };
  make_rg(I * GI_MB + 16 * rg, kindI, tapI, offI);
  make_rg(J * GI_MB + 16 * rg, kindJ, tapJ, offJ);
  const unsigned char* const srcI = (kindI == 1) ? reinterpret_cast<const unsigned char*>(p.yd) : p.x;
  const unsigned char* const srcJ = (kindJ == 1) ? reinterpret_cast<const unsigned char*>(p.yd) : p.x;
  const int ystride = 4 * p.C2P;

  const gi_v4i* const vtab = reinterpret_cast<const gi_v4i*>(p.vtab);

  // one unit = 4 unconditional 16-byte loads (clamped addresses); validity bits are applied when the unit is
  // staged, so nothing touches the loaded registers before the MFMAs (keeps the prefetch asynchronous)
  gi_v4i vI[4], vJ[4];
  unsigned okI = 0, okJ = 0;
#define GI_FETCH(val, okb, kind, tap, off, src, slot)                                            \
  {                                                                                              \
    okb = 0;                                                                                     \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                              \
      const gi_v4i e = tbl[(slot) * GI_KC + vg + 32 * q];                                        \
      const unsigned m = (unsigned)e[1];                                                         \
      const unsigned okx = (m >> (tap)) & 1u, okv = m >> 31;                                     \
      const unsigned ok = (kind == 0) ? okx : ((kind <= 2) ? okv : 0u);                          \
      const int ax = okx ? (e[0] * p.C1 + (off)) : 0;                                            \
      const int ay = e[2] * ystride + (off);                                                     \
      const int addr = (kind == 0) ? ax : ((kind == 1) ? ay : 0);                                \
      val[q] = *reinterpret_cast<const gi_v4i*>(src + ((EFFQ_DBG(p) == 2) ? 0 : addr));             \
      okb |= ok << q;                                                                            \
    }                                                                                            \
  }
  auto stage = [&](unsigned char* P, gi_v4i (&val)[4], unsigned okb, int kind) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = (okb >> q) & 1u;
      gi_v4i z = {0, 0, 0, 0};
      if (kind == 2) z[0] = 1;
      if (!ok || kind >= 2) val[q] = ok ? z : gi_v4i{0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int o[4];
      gi_transpose4(val[0][k], val[1][k], val[2][k], val[3][k], o);
#pragma unroll
      for (int b = 0; b < 4; ++b) *reinterpret_cast<int*>(P + (16 * rg + 4 * k + b) * GI_RS + 4 * vg) = o[b];
    }
  };

  gi_v16i acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0;
  const bool live = (I * GI_MB + wr * 64 < p.E) && (J * GI_MB + wc * 64 < p.E);

  auto flush = [&](int cls) {
    long long* S = p.slabs + ((size_t)cls * p.npairs + blockIdx.x) * (size_t)(GI_MB * GI_MB);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int col = wc * 64 + n * 32 + li;
          const int v = acc[m][n][r];
          if (v != 0)
            atomicAdd(reinterpret_cast<unsigned long long*>(&S[row * GI_MB + col]), (unsigned long long)(long long)v);
          acc[m][n][r] = 0;
        }
  };

  // prologue: tables of the first two chunks, first chunk staged
  {
    const int k = tid & (GI_KC - 1);
    const int cc = (tid < GI_KC) ? c_begin : min(c_begin + 1, c_end - 1);
    tbl[(tid < GI_KC ? 0 : 1) * GI_KC + k] = vtab[(size_t)cc * GI_KC + k];
  }
  __syncthreads();
  GI_FETCH(vI, okI, kindI, tapI, offI, srcI, 0)
  if (!diag) GI_FETCH(vJ, okJ, kindJ, tapJ, offJ, srcJ, 0)
  stage(panI, vI, okI, kindI);
  if (!diag) stage(panJ, vJ, okJ, kindJ);
  __syncthreads();

  // class ids of chunk c / c+1 ride in registers, loaded two iterations ahead (an in-loop load issued after
  // the prefetch would make its wait drain the whole prefetch: vmcnt is in order)
  const bool has_cls = p.chunk_cls != nullptr;
  const int* const clsp = has_cls ? p.chunk_cls : reinterpret_cast<const int*>(p.x);
  int cls_cur = has_cls ? clsp[c_begin] : 0;
  int cls_next = (c_begin + 1 < c_end) ? (has_cls ? clsp[c_begin + 1] : 0) : -1;
  for (int c = c_begin; c < c_end; ++c) {
    const int b = (c - c_begin) & 1;
    const int c2 = min(c + 2, c_end - 1);
    const int cls_raw = clsp[has_cls ? c2 : 0];
    const gi_v4i traw = vtab[(size_t)c2 * GI_KC + (tid & (GI_KC - 1))];   // lands under the MFMAs
    GI_FETCH(vI, okI, kindI, tapI, offI, srcI, b ^ 1)          // chunk c+1 (its table was built last iteration)
    if (!diag) GI_FETCH(vJ, okJ, kindJ, tapJ, offJ, srcJ, b ^ 1)
    __builtin_amdgcn_sched_barrier(0);
    if (live && EFFQ_DBG(p) != 1) {
      const unsigned char* PI = panI + b * GI_PANEL + (wr * 64 + li) * GI_RS + lh * 16;
      const unsigned char* PJ = (diag ? panI : panJ) + b * GI_PANEL + (wc * 64 + li) * GI_RS + lh * 16;
#pragma unroll
      for (int s = 0; s < GI_KC / 32; ++s) {
        gi_v4i a[2], bb[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const gi_v4i*>(PI + m * 32 * GI_RS + s * 32);
#pragma unroll
        for (int n = 0; n < 2; ++n) bb[n] = *reinterpret_cast<const gi_v4i*>(PJ + n * 32 * GI_RS + s * 32);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], bb[n], acc[m][n], 0, 0, 0);
      }
    }
    if (cls_next != cls_cur && live) flush(cls_cur);
    if (tid < GI_KC) tbl[b * GI_KC + tid] = traw;              // table of chunk c+2 replaces chunk c's
    if (EFFQ_DBG(p) != 3) {
      stage(panI + (b ^ 1) * GI_PANEL, vI, okI, kindI);
      if (!diag) stage(panJ + (b ^ 1) * GI_PANEL, vJ, okJ, kindJ);
    }
    cls_cur = cls_next;
    cls_next = (c + 2 < c_end) ? (has_cls ? cls_raw : 0) : -1;
    __syncthreads();
  }
#undef GI_FETCH
}

// max |y| as the bit pattern of a non-negative float (atomicMax on uint is order independent)
__global__ __launch_bounds__(256) void k_gi_absmax(const float* __restrict__ y, size_t n, unsigned* __restrict__ out) {
  float m = 0.0f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) m = fmaxf(m, fabsf(y[e]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

__device__ __forceinline__ int gi_y_exponent(unsigned ymax_bits) {
  // 2^ex > max|y| ; ex = 0 for an all-zero (or non-finite) tensor
  const float ymax = __uint_as_float(ymax_bits);
  int ex = 0;
  if (ymax > 0.0f && ymax < 3.0e38f) (void)frexpf(ymax, &ex);
  return ex < -90 ? -90 : ex;
}

// Y = rint(y * 2^(30-ex)) in four balanced base-256 digits, layout [voxel][digit][C2P] (pad channels zero)
__global__ __launch_bounds__(256) void k_gi_ydigits(const float* __restrict__ y, long long V, int C2, int C2P,
                                                    const unsigned* __restrict__ ymax_bits, int8_t* __restrict__ yd) {
  const int ex = gi_y_exponent(*ymax_bits);
  const float scale = ldexpf(1.0f, 30 - ex);
  const size_t tot = (size_t)V * C2P, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) {
    const size_t v = e / C2P;
    const int c = (int)(e - v * C2P);
    int d[4] = {0, 0, 0, 0};
    if (c < C2) {
      float f = y[v * C2 + c] * scale;                       // exact (power of two), |f| < 2^30
      if (!(fabsf(f) < 1.1e9f)) f = 0.0f;                    // non-finite input: dropped
      int t = (int)rintf(f);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        d[k] = ((t + 128) & 255) - 128;
        t = (t - d[k]) >> 8;
      }
      d[3] = t;
    }
    int8_t* dst = yd + v * (size_t)(4 * C2P) + c;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[(size_t)k * C2P] = (int8_t)d[k];
  }
}

// A0 / B0 in the reference's row order from the integer class slabs (fp64, classes in fixed order)
__global__ __launch_bounds__(256) void k_gram_i8_finish(GramI8Params p, int n, int hb, const float* __restrict__ alpha,
                                                        int act_levels, const unsigned* __restrict__ ymax_bits,
                                                        const float* __restrict__ cls_w, float* __restrict__ A0,
                                                        float* __restrict__ B0, int accumulate, double* __restrict__ Au,
                                                        double* __restrict__ Bu) {
  const double s = (double)alpha[0] / (double)(act_levels - 1);
  const double q = ldexp(1.0, gi_y_exponent(*ymax_bits) - 30);
  const size_t nA = (size_t)n * n, nB = (size_t)p.C2 * n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t blk = (size_t)(GI_MB * GI_MB);
  double unw = 0.0;
  auto S = [&](int ri, int rj) -> double {
    const int I = ri / GI_MB, J = rj / GI_MB;
    const size_t pidx = (size_t)I * p.NB - (size_t)I * (I - 1) / 2 + (J - I);
    const size_t off = pidx * blk + (size_t)(ri - I * GI_MB) * GI_MB + (rj - J * GI_MB);
    double t = 0.0;
    long long u = 0;
    for (int c = 0; c < p.ncls; ++c) {
      const long long sl = p.slabs[(size_t)c * p.npairs * blk + off];
      t += (cls_w ? (double)cls_w[c] : 1.0) * (double)sl;
      u += sl;
    }
    unw = (double)u;           // the same sum without the attention weights (exact integer)
    return t;
  };
  auto to_int = [&](int qq, bool& isx) {
    if (hb && qq == n - 1) {
      isx = false;
      return p.RX;
    }
    isx = true;
    const int c = qq / p.T, tap = qq - c * p.T;
    return tap * p.C1 + c;
  };
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nA + nB; e += stride) {
    double val;
    float* dst;
    if (e < nA) {
      const int a = (int)(e / n), b = (int)(e % n);
      bool xa, xb;
      int ri = to_int(a, xa), rj = to_int(b, xb);
      if (ri > rj) {
        const int t = ri;
        ri = rj;
        rj = t;
      }
      const double sc = (xa ? s : 1.0) * (xb ? s : 1.0);
      val = 2.0 * sc * S(ri, rj);
      dst = A0 + e;
      // unweighted system of the same pass, in fp64 and without the factor 2: Au = sum_v xhat xhat^T (ones row included),
      // Bu = sum_v y xhat^T - what the loss of an iterate needs (effq_gram_loss)
      if (Au != nullptr) Au[e] = accumulate ? Au[e] + sc * unw : sc * unw;
    } else {
      const size_t f = e - nA;
      const int c2 = (int)(f / n), b = (int)(f % n);
      bool xb;
      const int ri = to_int(b, xb);
      double t = 0.0, tu = 0.0, w = 1.0;
      for (int d = 0; d < 4; ++d, w *= 256.0) {
        t += w * S(ri, p.YR0 + d * p.C2P + c2);
        tu += w * unw;
      }
      val = 2.0 * q * (xb ? s : 1.0) * t;
      dst = B0 + f;
      if (Bu != nullptr) {
        const double vu = q * (xb ? s : 1.0) * tu;
        Bu[f] = accumulate ? Bu[f] + vu : vu;
      }
    }
    *dst = accumulate ? (*dst + (float)val) : (float)val;
  }
}

static int gram_i8_plan(const effq_geom* g, int ncls, long long n_list, GramI8Params* pp) {
  EFFQ_CHECK_ARG(g != nullptr);
  EFFQ_CHECK_ARG(g->N > 0 && g->C1 > 0 && g->C2 > 0 && g->D > 0 && g->H > 0 && g->W > 0);
  EFFQ_CHECK_ARG(g->KD >= 1 && g->KH >= 1 && g->KW >= 1 && g->SD >= 1 && g->SH >= 1 && g->SW >= 1);
  EFFQ_CHECK_ARG((g->C1 % 16) == 0 && g->KD * g->KH * g->KW <= 31);
  EFFQ_CHECK_ARG(ncls >= 1 && ncls <= GI_MAXCLS);
  GramI8Params& p = *pp;
  memset(&p, 0, sizeof(p));
  p.N = g->N; p.C1 = g->C1; p.C2 = g->C2; p.D = g->D; p.H = g->H; p.W = g->W;
  p.KD = g->KD; p.KH = g->KH; p.KW = g->KW; p.SD = g->SD; p.SH = g->SH; p.SW = g->SW;
  p.PD = g->PD; p.PH = g->PH; p.PW = g->PW;
  p.OD = (g->D + 2 * g->PD - g->KD) / g->SD + 1;
  p.OH = (g->H + 2 * g->PH - g->KH) / g->SH + 1;
  p.OW = (g->W + 2 * g->PW - g->KW) / g->SW + 1;
  EFFQ_CHECK_ARG(p.OD > 0 && p.OH > 0 && p.OW > 0);
  p.C2P = (g->C2 + 15) / 16 * 16;
  p.T = p.KD * p.KH * p.KW;
  p.RX = p.T * p.C1;
  p.YR0 = p.RX + 16;
  p.E = p.YR0 + 4 * p.C2P;
  p.NB = (p.E + GI_MB - 1) / GI_MB;
  p.NBX = (p.YR0 + GI_MB - 1) / GI_MB;
  p.npairs = p.NBX * p.NB - p.NBX * (p.NBX - 1) / 2;
  p.V = (long long)p.N * p.OD * p.OH * p.OW;
  // 32-bit byte offsets inside the kernel
  EFFQ_CHECK_ARG((long long)g->N * g->D * g->H * g->W * g->C1 < (1ll << 31));
  EFFQ_CHECK_ARG(p.V * 4 * p.C2P < (1ll << 31));
  const long long nchunks = (n_list > 0) ? n_list / GI_KC : (p.V + GI_KC - 1) / GI_KC;
  EFFQ_CHECK_ARG(nchunks > 0 && nchunks < (1ll << 24));
  p.nchunks = (int)nchunks;
  p.ncls = ncls;
  // splits: ~1536 workgroups, at least 4 chunks each, at most GI_MAX_CPS (int32 accumulator range)
  static const int wgs = getenv("EFFQ_GI8_WGS") ? atoi(getenv("EFFQ_GI8_WGS")) : 3072;   // tuning aid
  long long want = (wgs + p.npairs - 1) / p.npairs;
  if (want > nchunks / 4) want = nchunks / 4;
  if (want < 1) want = 1;
  long long cps = (nchunks + want - 1) / want;
  if (cps > GI_MAX_CPS) cps = GI_MAX_CPS;
  p.cps = (int)cps;
  p.nsplit = (int)((nchunks + cps - 1) / cps);
  EFFQ_CHECK_ARG(p.nsplit <= 65535);
  return EFFQ_OK;
}

// upper bound of the list length: every class run is padded to a multiple of 128 slots
static size_t gram_i8_table_bytes(long long V, int ncls) { return (size_t)(V + (long long)GI_KC * (ncls + 1)) * 16; }

static size_t gram_i8_slab_bytes(const GramI8Params& p) {
  return (size_t)p.ncls * p.npairs * (size_t)(GI_MB * GI_MB) * sizeof(long long);
}

}  // namespace effq

using namespace effq;

extern "C" {

int effq_gram_i8_supported(const effq_geom* g, int act_levels) {
  if (g == nullptr) return 0;
  if ((g->C1 % 16) != 0 || g->KD * g->KH * g->KW > 31 || g->KD < 1 || g->KH < 1 || g->KW < 1) return 0;
  if (act_levels < 2 || act_levels > 128) return 0;
  GramI8Params p;
  if (gram_i8_plan(g, 1, 0, &p) != EFFQ_OK) return 0;
  return 1;
}

size_t effq_gram_i8_ws_bytes(const effq_geom* g, int ncls) {
  GramI8Params p;
  if (gram_i8_plan(g, ncls, 0, &p) != EFFQ_OK) return 0;
  return 256 + gram_i8_slab_bytes(p) + (size_t)p.V * 4 * p.C2P + 256 + gram_i8_table_bytes(p.V, ncls) + 256;
}

int effq_gram_accum_i8_unw(const uint8_t* xidx_ndhwc, const float* y_ndhwc, const effq_geom* g, int has_bias,
                           const float* act_alpha_dev, int act_levels, const int32_t* vox_list, const int32_t* chunk_cls,
                           const float* cls_w_dev, int ncls, long long n_list, float* A0, float* B0, int accumulate,
                           double* Au, double* Bu, void* ws, size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG((Au == nullptr) == (Bu == nullptr));
  EFFQ_CHECK_ARG(xidx_ndhwc && y_ndhwc && g && act_alpha_dev && A0 && B0 && ws);
  EFFQ_CHECK_ARG(effq_gram_i8_supported(g, act_levels));
  if (vox_list == nullptr) {
    EFFQ_CHECK_ARG(ncls == 1 && chunk_cls == nullptr && n_list == 0);
  } else {
    EFFQ_CHECK_ARG(chunk_cls != nullptr && cls_w_dev != nullptr && n_list > 0 && (n_list % GI_KC) == 0);
  }
  GramI8Params p;
  int rc = gram_i8_plan(g, ncls, n_list, &p);
  if (rc != EFFQ_OK) return rc;
  const size_t slab_bytes = gram_i8_slab_bytes(p);
  const size_t need = 256 + slab_bytes + (size_t)p.V * 4 * p.C2P + 256 + gram_i8_table_bytes(p.V, ncls) + 256;
  EFFQ_CHECK_ARG((size_t)p.nchunks * GI_KC * 16 <= gram_i8_table_bytes(p.V, ncls));
  if (ws_bytes < need) {
    set_error("gram_i8: workspace %zu < required %zu", ws_bytes, need);
    return EFFQ_ERR_WORKSPACE;
  }
  char* base = reinterpret_cast<char*>(ws);
  unsigned* ymax = reinterpret_cast<unsigned*>(base);
  p.slabs = reinterpret_cast<long long*>(base + 256);
  int8_t* yd = reinterpret_cast<int8_t*>(base + 256 + slab_bytes);
  gi_v4i* vtab = reinterpret_cast<gi_v4i*>((reinterpret_cast<uintptr_t>(base + 256 + slab_bytes + (size_t)p.V * 4 * p.C2P) + 255) & ~(uintptr_t)255);
  p.x = xidx_ndhwc;
  p.yd = yd;
  p.vox_list = vox_list;
  p.chunk_cls = chunk_cls;
  p.vtab = reinterpret_cast<const int*>(vtab);
  p.debug = effq_ablate_env("EFFQ_GI8_DEBUG");
  hipStream_t st = as_stream(stream);
  EFFQ_HIP(hipMemsetAsync(base, 0, 256 + slab_bytes, st));
  const size_t ny = (size_t)p.V * p.C2;
  {
    size_t nb = (ny + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_gi_absmax, dim3((unsigned)nb), dim3(256), 0, st, y_ndhwc, ny, ymax);
    size_t nd = ((size_t)p.V * p.C2P + 255) / 256;
    if (nd > 8192) nd = 8192;
    hipLaunchKernelGGL(k_gi_ydigits, dim3((unsigned)nd), dim3(256), 0, st, y_ndhwc, p.V, p.C2, p.C2P, ymax, yd);
    EFFQ_LAUNCH_CHECK();
  }
  {
    size_t nt = ((size_t)p.nchunks * GI_KC + 255) / 256;
    if (nt > 4096) nt = 4096;
    hipLaunchKernelGGL(k_gi_voxtable, dim3((unsigned)nt), dim3(256), 0, st, p, vtab);
    EFFQ_LAUNCH_CHECK();
  }
  static bool attr_set = false;
  if (!attr_set) {
    EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gram_i8), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 GI_LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_gram_i8, dim3((unsigned)p.npairs, (unsigned)p.nsplit), dim3(GI_T), GI_LDS, st, p);
  EFFQ_LAUNCH_CHECK();
  const int n = p.RX + (has_bias ? 1 : 0);
  size_t tot = (size_t)n * n + (size_t)p.C2 * n;
  size_t nb = (tot + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(k_gram_i8_finish, dim3((unsigned)nb), dim3(256), 0, st, p, n, has_bias ? 1 : 0, act_alpha_dev,
                     act_levels, ymax, cls_w_dev, A0, B0, accumulate, Au, Bu);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_gram_accum_i8(const uint8_t* xidx_ndhwc, const float* y_ndhwc, const effq_geom* g, int has_bias,
                       const float* act_alpha_dev, int act_levels, const int32_t* vox_list, const int32_t* chunk_cls,
                       const float* cls_w_dev, int ncls, long long n_list, float* A0, float* B0, int accumulate,
                       void* ws, size_t ws_bytes, void* stream) {
  return effq_gram_accum_i8_unw(xidx_ndhwc, y_ndhwc, g, has_bias, act_alpha_dev, act_levels, vox_list, chunk_cls,
                                cls_w_dev, ncls, n_list, A0, B0, accumulate, nullptr, nullptr, ws, ws_bytes, stream);
}

}  // extern "C"

// ---- voxel list sorted by attention weight (the input format of effq_gram_accum_i8) ---------------------------------------
// The reference's masks hold a handful of integer class weights (quirk Q1).  Three small launches replace a library
// sort: (1) distinct values + counts through workgroup-local tables merged into a 32-slot global table, (2) one thread
// orders the classes by value and lays out their 128-padded segments, (3) every workgroup reserves its share of each
// segment with one atomic per class and writes its voxel indices.  The order of the voxels inside a class depends on
// the scheduling - the Gram sums built on the list are exact integer sums, so they do not.
namespace effq {
constexpr int CLS_SLOTS = 32;      // table size; more than GI_MAX_CLS distinct values -> overflow (caller uses the fp32 Gram)
constexpr int CLS_MAX = 16;
struct ClsTable {
  unsigned key[CLS_SLOTS];         // float bit pattern, 0xffffffff = empty
  unsigned cnt[CLS_SLOTS];
  unsigned cursor[CLS_MAX];        // next free entry of each class segment (scatter phase)
  int overflow;
  int ncls;
  int n_list;                      // total length of the padded list
  int slot_cls[CLS_SLOTS];         // table slot -> class index (by ascending value)
  unsigned seg_start[CLS_MAX];
  float cls_w[CLS_MAX];
};

__device__ __forceinline__ int cls_find_or_insert(unsigned* keys, unsigned k) {
  unsigned h = (k * 2654435761u) >> 27;        // 5 bits
  for (int probe = 0; probe < CLS_SLOTS; ++probe) {
    const unsigned cur = atomicCAS(&keys[h], 0xffffffffu, k);
    if (cur == 0xffffffffu || cur == k) return (int)h;
    h = (h + 1) & (CLS_SLOTS - 1);
  }
  return -1;
}

__global__ __launch_bounds__(256) void k_cls_count(const float* __restrict__ att, long long V, ClsTable* t) {
  __shared__ unsigned lkey[CLS_SLOTS], lcnt[CLS_SLOTS];
  __shared__ int lover;
  if (threadIdx.x < CLS_SLOTS) {
    lkey[threadIdx.x] = 0xffffffffu;
    lcnt[threadIdx.x] = 0u;
  }
  if (threadIdx.x == 0) lover = 0;
  __syncthreads();
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += stride) {
    const unsigned k = __float_as_uint(att[i]);
    const int sidx = (k == 0xffffffffu) ? -1 : cls_find_or_insert(lkey, k);
    if (sidx < 0)
      lover = 1;
    else
      atomicAdd(&lcnt[sidx], 1u);
  }
  __syncthreads();
  if (threadIdx.x < CLS_SLOTS && lkey[threadIdx.x] != 0xffffffffu) {
    const int g = cls_find_or_insert(t->key, lkey[threadIdx.x]);
    if (g < 0)
      t->overflow = 1;
    else
      atomicAdd(&t->cnt[g], lcnt[threadIdx.x]);
  }
  if (threadIdx.x == 0 && lover) t->overflow = 1;
}

__global__ void k_cls_layout(ClsTable* t, int* __restrict__ chunk_cls, int max_chunks) {
  // one thread: order the (at most 32) used slots by value, give every class a segment padded to a multiple of 128
  int used[CLS_SLOTS], nu = 0;
  for (int sidx = 0; sidx < CLS_SLOTS; ++sidx)
    if (t->key[sidx] != 0xffffffffu) used[nu++] = sidx;
  if (nu > CLS_MAX) t->overflow = 1;
  if (t->overflow) {
    t->ncls = 0;
    t->n_list = 0;
    return;
  }
  for (int i = 1; i < nu; ++i) {               // insertion sort by float value
    const int sidx = used[i];
    const float v = __uint_as_float(t->key[sidx]);
    int j = i - 1;
    while (j >= 0 && __uint_as_float(t->key[used[j]]) > v) {
      used[j + 1] = used[j];
      --j;
    }
    used[j + 1] = sidx;
  }
  unsigned start = 0;
  int chunk = 0;
  for (int c = 0; c < nu; ++c) {
    const int sidx = used[c];
    t->slot_cls[sidx] = c;
    t->cls_w[c] = __uint_as_float(t->key[sidx]);
    t->seg_start[c] = start;
    t->cursor[c] = start;
    const unsigned padded = (t->cnt[sidx] + 127u) / 128u * 128u;
    for (unsigned q = 0; q < padded / 128u && chunk < max_chunks; ++q) chunk_cls[chunk++] = c;
    start += padded;
  }
  t->ncls = nu;
  t->n_list = (int)start;
}

__global__ __launch_bounds__(256) void k_cls_scatter(const float* __restrict__ att, long long V, ClsTable* t,
                                                     int* __restrict__ vox_list) {
  __shared__ unsigned lcnt[CLS_MAX], lbase[CLS_MAX];
  if (t->overflow) return;
  if (threadIdx.x < CLS_MAX) lcnt[threadIdx.x] = 0u;
  __syncthreads();
  // contiguous slice per workgroup, two sweeps: count, reserve, write
  const long long per = (V + gridDim.x - 1) / gridDim.x;
  const long long i0 = (long long)blockIdx.x * per, i1 = (i0 + per < V) ? i0 + per : V;
  for (long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const unsigned k = __float_as_uint(att[i]);
    unsigned h = (k * 2654435761u) >> 27;
    while (t->key[h] != k) h = (h + 1) & (CLS_SLOTS - 1);
    atomicAdd(&lcnt[t->slot_cls[h]], 1u);
  }
  __syncthreads();
  if (threadIdx.x < CLS_MAX) {
    lbase[threadIdx.x] = (lcnt[threadIdx.x] != 0u) ? atomicAdd(&t->cursor[threadIdx.x], lcnt[threadIdx.x]) : 0u;
    lcnt[threadIdx.x] = 0u;
  }
  __syncthreads();
  for (long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const unsigned k = __float_as_uint(att[i]);
    unsigned h = (k * 2654435761u) >> 27;
    while (t->key[h] != k) h = (h + 1) & (CLS_SLOTS - 1);
    const int c = t->slot_cls[h];
    vox_list[lbase[c] + atomicAdd(&lcnt[c], 1u)] = (int)i;
  }
}
}  // namespace effq

extern "C" {

size_t effq_att_classes_ws_bytes(void) { return sizeof(ClsTable) + 64; }

/* Voxel list of an attention mask for effq_gram_accum_i8.  vox_list: V + 128*16 int32 (filled with -1 here, then the class
 * segments), chunk_cls: (V / 128 + 16) int32, cls_w_dev: 16 floats.  info_host_out[3] = {ncls, n_list, overflow}: the call
 * SYNCHRONISES the stream to return them (once per mask, not per layer: several layers share a pyramid level).
 * overflow != 0: more than 16 distinct weights, the lists are not filled. */
int effq_att_classes(const float* att, long long V, int32_t* vox_list, int32_t* chunk_cls, float* cls_w_dev,
                     int32_t* info_host_out, void* ws, void* stream) {
  EFFQ_CHECK_ARG(att && vox_list && chunk_cls && cls_w_dev && info_host_out && ws && V > 0 && V < (1ll << 31) - 4096);
  hipStream_t st = as_stream(stream);
  ClsTable* t = reinterpret_cast<ClsTable*>(ws);
  EFFQ_HIP(hipMemsetAsync(t, 0, sizeof(ClsTable), st));
  EFFQ_HIP(hipMemsetAsync(t->key, 0xff, sizeof(t->key), st));
  const long long cap = V + 128 * CLS_MAX;
  EFFQ_HIP(hipMemsetAsync(vox_list, 0xff, sizeof(int32_t) * (size_t)cap, st));     // -1 = padding
  int blocks = (int)((V + 256 * 16 - 1) / (256 * 16));
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_cls_count, dim3(blocks), dim3(256), 0, st, att, V, t);
  hipLaunchKernelGGL(k_cls_layout, dim3(1), dim3(1), 0, st, t, chunk_cls, (int)(V / 128 + CLS_MAX));
  hipLaunchKernelGGL(k_cls_scatter, dim3(blocks), dim3(256), 0, st, att, V, t, vox_list);
  EFFQ_LAUNCH_CHECK();
  ClsTable host;
  EFFQ_HIP(hipMemcpyAsync(&host, t, sizeof(ClsTable), hipMemcpyDeviceToHost, st));
  EFFQ_HIP(hipStreamSynchronize(st));
  info_host_out[0] = host.ncls;
  info_host_out[1] = host.n_list;
  info_host_out[2] = host.overflow;
  if (!host.overflow) EFFQ_HIP(hipMemcpyAsync(cls_w_dev, t->cls_w, sizeof(float) * CLS_MAX, hipMemcpyDeviceToDevice, st));
  return EFFQ_OK;
}

}  // extern "C"
