// effq_gram_accum: attention-weighted Gram system of one layer without materialising im2col.
// Reference: solver.py:86-111 (im2col_loop), :253-257 (ones row), :282-314 (getA0B0).
//
// The patch matrix is never built.  Rows of the extended operand E are
//   [ x rows (tap-major: r = tap*C1 + c) | y rows (C2) | ones row (bias) ]
// and one symmetric product S = sum_v att_v E(:,v) E(:,v)^T gives everything:
//   A0 = 2*S[x|1 , x|1],  B0 = 2*S[y , x|1].
// Work split: 128x128 macro blocks of the upper triangle of S  x  voxel splits (split-K).
// Each workgroup streams its voxel range in chunks of KC voxels: the two 128-row panels are
// gathered (coalesced along channels) into LDS as [voxel][row], and the 4 waves run
// v_mfma_f32_32x32x2_f32 with K = voxels.  Partial blocks go to per-split slabs; a second
// kernel adds the (fp64) slabs in a fixed order (deterministic), mirrors, scales by 2 and
// scatters into the reference's (c,kd,kh,kw)+bias row order.
#include <stdlib.h>
#include "common.h"

namespace effq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;      // voxels per staged chunk
constexpr int MB = 128;     // macro block rows
constexpr int PS = MB + 4;  // LDS row stride (floats)

struct GramParams {
  const float* x;
  const float* att;
  const float* y;
  int N, C1, C2, D, H, W, OD, OH, OW;
  int KD, KH, KW, SD, SH, SW, PD, PH, PW;
  int T, RX, hb, E, NB, npairs;
  long long V;
  int nsplit;
  long long vox_per_split;
  int fold;
  int debug;   // profiling ablations (EFFQ_GRAM_DEBUG): 1 no MFMA, 2 no global loads, 3 no LDS stores
  double* slabs;
};

struct VoxInfo {
  int n, id0, ih0, iw0;  // input-space origin of the receptive field; n < 0 => voxel out of range
};

// One staged cell = 4 consecutive extended rows of one voxel (16 bytes).  Extended row order:
//   [ x rows, tap-major r = tap*C1 + c | y rows (C2) | ones row ]
// so that with C1 % 4 == 0 and C2 % 4 == 0 every cell is either 4 channels of one tap (one 16-byte
// load of x), 4 channels of y, or the ones/padding cell.
struct RowGroup {
  int kind;        // 0: x cell (kd,kh,kw,c0)  1: y cell (c0)  2: mixed, decode per row  3: padding  4: ones cell
  int kd, kh, kw, c0;
  int r0;
};

__device__ __forceinline__ float gram_fetch_row(const GramParams& p, int r, const VoxInfo& vi, unsigned v) {
  if (r < p.RX) {
    const int tap = r / p.C1, c = r - tap * p.C1;
    const int kw = tap % p.KW, t2 = tap / p.KW;
    const int kh = t2 % p.KH, kd = t2 / p.KH;
    const int id = vi.id0 + kd, ih = vi.ih0 + kh, iw = vi.iw0 + kw;
    if (id < 0 || id >= p.D || ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) return 0.0f;
    return p.x[((((size_t)vi.n * p.D + id) * p.H + ih) * p.W + iw) * p.C1 + c];
  }
  if (r < p.RX + p.C2) return p.y[(size_t)v * p.C2 + (r - p.RX)];
  if (p.hb && r == p.RX + p.C2) return 1.0f;
  return 0.0f;
}

__device__ __forceinline__ float4 gram_fetch_cell(const GramParams& p, const RowGroup& g, const VoxInfo& vi,
                                                  unsigned v) {
  float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vi.n < 0 || g.kind == 3) return val;
  if (g.kind == 0) {
    const int id = vi.id0 + g.kd, ih = vi.ih0 + g.kh, iw = vi.iw0 + g.kw;
    if (id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
      val = *reinterpret_cast<const float4*>(p.x + ((((size_t)vi.n * p.D + id) * p.H + ih) * p.W + iw) * p.C1 + g.c0);
  } else if (g.kind == 1) {
    val = *reinterpret_cast<const float4*>(p.y + (size_t)v * p.C2 + g.c0);
  } else {
    val.x = gram_fetch_row(p, g.r0 + 0, vi, v);
    val.y = gram_fetch_row(p, g.r0 + 1, vi, v);
    val.z = gram_fetch_row(p, g.r0 + 2, vi, v);
    val.w = gram_fetch_row(p, g.r0 + 3, vi, v);
  }
  return val;
}

constexpr int GT = 512;   // threads: 8 waves = 4 row sub-tiles x 2 column halves

template <bool VEC>
__global__ __launch_bounds__(GT) void k_gram(GramParams p) {
  // double-buffered panels: chunk c+1 is written while chunk c is still being consumed (one barrier per chunk)
  __shared__ __attribute__((aligned(16))) float panI[2][KC * PS];
  __shared__ __attribute__((aligned(16))) float panJ[2][KC * PS];
  __shared__ float att_s[2][KC];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wi = wid & 3, wjh = wid >> 2;

  int I = 0, rem = blockIdx.x;
  while (rem >= p.NB - I) {
    rem -= p.NB - I;
    ++I;
  }
  const int J = I + rem;
  const bool diag = (I == J);

  // staging role: cell column g (rows 4g..4g+3 of each 128-row panel), voxels vsub and vsub+16 of a chunk
  const int g = tid & 31, vsub = tid >> 5;
  constexpr bool vec = VEC;
  auto make_group = [&](int r0) {
    RowGroup q;
    q.r0 = r0;
    q.kd = q.kh = q.kw = q.c0 = 0;
    if (r0 >= p.E) {
      q.kind = 3;
    } else if (vec && r0 + 3 < p.RX) {
      q.kind = 0;
      const int tap = r0 / p.C1;
      q.c0 = r0 - tap * p.C1;
      q.kw = tap % p.KW;
      const int t2 = tap / p.KW;
      q.kh = t2 % p.KH;
      q.kd = t2 / p.KH;
    } else if (vec && r0 >= p.RX && r0 + 3 < p.RX + p.C2) {
      q.kind = 1;
      q.c0 = r0 - p.RX;
    } else if (vec && p.hb && r0 == p.RX + p.C2) {
      q.kind = 4;
    } else if (vec && r0 > p.RX + p.C2) {
      q.kind = 3;
    } else {
      q.kind = 2;
    }
    return q;
  };
  const RowGroup gI = make_group(I * MB + 4 * g), gJ = make_group(J * MB + 4 * g);

  const unsigned v_begin = (unsigned)((long long)blockIdx.y * p.vox_per_split);
  unsigned v_end = v_begin + (unsigned)p.vox_per_split;
  if (v_end > (unsigned)p.V) v_end = (unsigned)p.V;

  // One staged cell, BRANCH-FREE on the vectorised path: the 16-byte load is unconditional on a clamped (always
  // valid) address and masked afterwards.  A load under a runtime branch makes hipcc wait for it inside the
  // branch (vmcnt(0)), which serialised the whole prefetch in front of the MFMA loop (cdna_hip_programming.md,
  // projection-GEMM trap 4c); with straight-line code the loads fly under the MFMAs.
  // returns the RAW loaded cell; `code` says what to do with it when it is stored to LDS (0 zero, 1 keep,
  // 2 ones cell): nothing touches the loaded registers before the MFMA loop, so no wait is placed there
  auto fetch_cell = [&](const RowGroup& q, unsigned v, int& code) -> float4 {
    const bool vvalid = v < v_end;
    const unsigned cv = vvalid ? v : v_begin;
    unsigned t = cv;
    const int ow = (int)(t % (unsigned)p.OW);
    t /= (unsigned)p.OW;
    const int oh = (int)(t % (unsigned)p.OH);
    t /= (unsigned)p.OH;
    const int od = (int)(t % (unsigned)p.OD);
    const int n = (int)(t / (unsigned)p.OD);
    const int id = od * p.SD - p.PD + q.kd, ih = oh * p.SH - p.PH + q.kh, iw = ow * p.SW - p.PW + q.kw;
    const bool inb = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
    if (VEC) {
      const int cd = min(max(id, 0), p.D - 1), ch = min(max(ih, 0), p.H - 1), cw = min(max(iw, 0), p.W - 1);
      const size_t xoff = ((((size_t)n * p.D + cd) * p.H + ch) * p.W + cw) * p.C1 + q.c0;
      const size_t yoff = (size_t)cv * p.C2 + ((q.kind == 1) ? q.c0 : 0);
      const float* src = (q.kind == 0) ? (p.x + xoff) : (p.y + yoff);
      const float4 val = *reinterpret_cast<const float4*>(src);
      const bool ok = vvalid && ((q.kind == 0 && inb) || q.kind == 1);
      code = ok ? 1 : ((q.kind == 4 && vvalid) ? 2 : 0);
      return val;
    } else {
      code = 1;
      VoxInfo vi;
      vi.n = vvalid ? n : -1;
      vi.id0 = od * p.SD - p.PD;
      vi.ih0 = oh * p.SH - p.PH;
      vi.iw0 = ow * p.SW - p.PW;
      return gram_fetch_cell(p, q, vi, v);
    }
  };

  float4 rI[2], rJ[2];
  int cI[2], cJ[2];
  float ratt = 1.0f;
  bool ratt_ok = false;
  auto prefetch = [&](unsigned v0) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const unsigned v = v0 + vsub + 16 * ps;
      rI[ps] = fetch_cell(gI, v, cI[ps]);
      if (!diag) rJ[ps] = fetch_cell(gJ, v, cJ[ps]);
    }
    {
      const unsigned v = v0 + (tid & (KC - 1));
      ratt_ok = v < v_end;                               // masked at the LDS store, like the cells
      if (p.att != nullptr) ratt = p.att[ratt_ok ? v : v_begin];
    }
  };

  // fp32 MFMA accumulation runs over `fold` chunks (KC voxels each) and is then folded into fp64
  // accumulators: the normal equations are ill-conditioned (cond ~1e3-1e6, the bias column is nearly
  // collinear with the non-negative activations); a 2e-7 relative error in A0 moved the ADMM losses by
  // 1e-3.  fold = 1 for small problems, 4 when the voxel count itself averages the rounding down.
  double acc64[2][16];
  f32x16 acc[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc64[jt][r] = 0.0;
      acc[jt][r] = 0.0f;
    }
  const int fold = p.fold;
  int since = 0;
  // sub-tiles made only of padding rows (>= E) are skipped: wave-uniform predicates
  const bool rows_live = I * MB + wi * 32 < p.E;
  const bool col_live[2] = {J * MB + (2 * wjh + 0) * 32 < p.E, J * MB + (2 * wjh + 1) * 32 < p.E};
  auto masked = [](const float4& raw, int code) {
    float4 o = raw;
    if (code == 0) o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (code == 2) o = make_float4(1.f, 0.f, 0.f, 0.f);
    return o;
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int vv = vsub + 16 * ps;
      *reinterpret_cast<float4*>(&panI[buf][vv * PS + 4 * g]) = masked(rI[ps], cI[ps]);
      if (!diag) *reinterpret_cast<float4*>(&panJ[buf][vv * PS + 4 * g]) = masked(rJ[ps], cJ[ps]);
    }
    if (tid < KC) att_s[buf][tid] = ratt_ok ? ratt : 0.0f;
  };

  int buf = 0;
  if (v_begin < v_end) {
    prefetch(v_begin);
    stage(0);
  }
  lds_barrier();
  for (unsigned v0 = v_begin; v0 < v_end; v0 += KC, buf ^= 1) {
    const bool more = v0 + KC < v_end;
    if (EFFQ_DBG(p) != 2) prefetch(more ? v0 + KC : v0);   // unconditional (the last chunk re-reads itself): lands
                                                        // in registers under the MFMAs below
    if (rows_live && col_live[0] && EFFQ_DBG(p) != 1) {
      const float* pi = panI[buf];
      const float* pj = diag ? panI[buf] : panJ[buf];
      if (col_live[1]) {
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
          const int vv = 2 * s + lh;
          const float a = pi[vv * PS + wi * 32 + li] * att_s[buf][vv];
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            const float b = pj[vv * PS + (2 * wjh + jt) * 32 + li];
            acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[jt], 0, 0, 0);
          }
        }
      } else {      // second column sub-tile is padding only
#pragma unroll 4
        for (int s = 0; s < KC / 2; ++s) {
          const int vv = 2 * s + lh;
          const float a = pi[vv * PS + wi * 32 + li] * att_s[buf][vv];
          const float b = pj[vv * PS + (2 * wjh) * 32 + li];
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
        }
      }
      if (++since == fold) {
        since = 0;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc64[jt][r] += (double)acc[jt][r];
            acc[jt][r] = 0.0f;
          }
      }
    }
    if (more && EFFQ_DBG(p) != 3) stage(buf ^ 1);
    lds_barrier();
  }
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc64[jt][r] += (double)acc[jt][r];

  double* dst = p.slabs + ((size_t)blockIdx.y * p.npairs + blockIdx.x) * (size_t)(MB * MB);
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      dst[(size_t)(wi * 32 + i) * MB + (2 * wjh + jt) * 32 + li] = acc64[jt][r];
    }
}

// Sum the split slabs (fp64, fixed order), scale by 2, mirror, scatter to reference order.
__global__ __launch_bounds__(256) void k_gram_finish(GramParams p, int n, float* __restrict__ A0,
                                                     float* __restrict__ B0, int accumulate) {
  const size_t nA = (size_t)n * n, nB = (size_t)p.C2 * n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nA + nB; e += stride) {
    int ri, rj;
    float* dst;
    if (e < nA) {
      const int a = (int)(e / n), b = (int)(e % n);
      // reference row a = c*T + tap  ->  internal tap*C1 + c ; bias row -> RX
      auto to_int = [&](int q) {
        if (p.hb && q == n - 1) return p.RX + p.C2;
        const int c = q / p.T, tap = q - c * p.T;
        return tap * p.C1 + c;
      };
      ri = to_int(a);
      rj = to_int(b);
      if (ri > rj) {
        const int t = ri;
        ri = rj;
        rj = t;
      }
      dst = A0 + e;
    } else {
      const size_t f = e - nA;
      const int c2 = (int)(f / n), b = (int)(f % n);
      if (p.hb && b == n - 1)
        ri = p.RX + p.C2;
      else {
        const int c = b / p.T, tap = b - c * p.T;
        ri = tap * p.C1 + c;
      }
      rj = p.RX + c2;
      if (ri > rj) {   // (y, ones): the ones row sits after the y rows
        const int t = ri;
        ri = rj;
        rj = t;
      }
      dst = B0 + f;
    }
    const int I = ri / MB, J = rj / MB;
    const size_t pidx = (size_t)I * p.NB - (size_t)I * (I - 1) / 2 + (J - I);
    const size_t off = pidx * (size_t)(MB * MB) + (size_t)(ri - I * MB) * MB + (rj - J * MB);
    double s = 0.0;
    for (int k = 0; k < p.nsplit; ++k) s += p.slabs[(size_t)k * p.npairs * (size_t)(MB * MB) + off];
    const float val = (float)(2.0 * s);
    *dst = accumulate ? (*dst + val) : val;
  }
}

static int gram_plan(const effq_geom* g, int has_bias, GramParams* pp) {
  EFFQ_CHECK_ARG(g != nullptr);
  EFFQ_CHECK_ARG(g->N > 0 && g->C1 > 0 && g->C2 > 0 && g->D > 0 && g->H > 0 && g->W > 0);
  EFFQ_CHECK_ARG(g->KD >= 1 && g->KH >= 1 && g->KW >= 1 && g->SD >= 1 && g->SH >= 1 && g->SW >= 1);
  GramParams& p = *pp;
  memset(&p, 0, sizeof(p));
  p.N = g->N; p.C1 = g->C1; p.C2 = g->C2; p.D = g->D; p.H = g->H; p.W = g->W;
  p.KD = g->KD; p.KH = g->KH; p.KW = g->KW; p.SD = g->SD; p.SH = g->SH; p.SW = g->SW;
  p.PD = g->PD; p.PH = g->PH; p.PW = g->PW;
  p.OD = (g->D + 2 * g->PD - g->KD) / g->SD + 1;
  p.OH = (g->H + 2 * g->PH - g->KH) / g->SH + 1;
  p.OW = (g->W + 2 * g->PW - g->KW) / g->SW + 1;
  EFFQ_CHECK_ARG(p.OD > 0 && p.OH > 0 && p.OW > 0);
  p.T = p.KD * p.KH * p.KW;
  p.RX = p.T * p.C1;
  p.hb = has_bias ? 1 : 0;
  p.E = p.RX + p.hb + p.C2;
  p.NB = (p.E + MB - 1) / MB;
  p.npairs = p.NB * (p.NB + 1) / 2;
  p.V = (long long)p.N * p.OD * p.OH * p.OW;
  // split-K: aim at ~4096 workgroups, at least 8 chunks per split, slabs <= 1 GiB
  long long chunks = (p.V + KC - 1) / KC;
  long long want = (4096 + p.npairs - 1) / p.npairs;
  if (want < 1) want = 1;
  long long max_by_work = chunks / 8;
  if (max_by_work < 1) max_by_work = 1;
  long long max_by_mem = ((long long)1 << 30) / ((long long)p.npairs * MB * MB * 8);
  if (max_by_mem < 1) max_by_mem = 1;
  long long ns = want;
  if (ns > max_by_work) ns = max_by_work;
  if (ns > max_by_mem) ns = max_by_mem;
  if (ns > 65535) ns = 65535;
  long long cps = (chunks + ns - 1) / ns;
  ns = (chunks + cps - 1) / cps;
  p.nsplit = (int)ns;
  p.vox_per_split = cps * KC;
  p.fold = (p.V >= (1ll << 20)) ? 4 : 1;
  EFFQ_CHECK_ARG(p.V < (1ll << 31));
  return EFFQ_OK;
}

}  // namespace effq

using namespace effq;

extern "C" {

size_t effq_gram_ws_bytes(const effq_geom* g, int has_bias) {
  GramParams p;
  if (gram_plan(g, has_bias, &p) != EFFQ_OK) return 0;
  return (size_t)p.nsplit * p.npairs * (size_t)(MB * MB) * sizeof(double) + 256;
}

int effq_gram_accum(const float* x_ndhwc, const float* att, const float* y_ndhwc, const effq_geom* g, int has_bias,
                    float* A0, float* B0, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG(x_ndhwc && y_ndhwc && A0 && B0 && ws);
  GramParams p;
  int rc = gram_plan(g, has_bias, &p);
  if (rc != EFFQ_OK) return rc;
  const size_t need = (size_t)p.nsplit * p.npairs * (size_t)(MB * MB) * sizeof(double);
  if (ws_bytes < need) {
    set_error("gram: workspace %zu < required %zu", ws_bytes, need);
    return EFFQ_ERR_WORKSPACE;
  }
  p.debug = effq_ablate_env("EFFQ_GRAM_DEBUG");
  p.x = x_ndhwc;
  p.att = att;
  p.y = y_ndhwc;
  p.slabs = reinterpret_cast<double*>(ws);
  hipStream_t st = as_stream(stream);
  if ((p.C1 & 3) == 0 && (p.C2 & 3) == 0)
    hipLaunchKernelGGL(k_gram<true>, dim3((unsigned)p.npairs, (unsigned)p.nsplit), dim3(GT), 0, st, p);
  else
    hipLaunchKernelGGL(k_gram<false>, dim3((unsigned)p.npairs, (unsigned)p.nsplit), dim3(GT), 0, st, p);
  EFFQ_LAUNCH_CHECK();
  const int n = p.RX + p.hb;
  size_t tot = (size_t)n * n + (size_t)p.C2 * n;
  size_t nb = (tot + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(k_gram_finish, dim3((unsigned)nb), dim3(256), 0, st, p, n, A0, B0, accumulate);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
