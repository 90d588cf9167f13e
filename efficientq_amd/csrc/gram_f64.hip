// effq_gram_f64: the UNWEIGHTED Gram system of a layer with a full-precision input, in fp64 on the matrix cores:
//     Au = sum_v xhat_v xhat_v^T  (n x n),   Bu = sum_v y_v xhat_v^T  (c2 x n),   xhat_v = [im2col patch of x; 1].
// These are the operands of effq_gram_loss (gram_loss.hip): with them the 200 per-iteration losses of a layer
// (F.mse_loss(F.conv3d(x, G, b*), y), EfficientQConv.py:118-122) cost c2 n^2 multiply-adds each instead of a pass over
// all calibration voxels.  The layers with QUANTISED input get Au / Bu from the exact-integer Gram pass
// (effq_gram_accum_i8_unw); the first conv and the classifier (q_first = q_last = "256,-1": full-precision activations,
// quirk Q14) have no level ids, so their system is accumulated here - fp32 values widened to fp64, products exact
// (24 + 24 bits), fp64 accumulation on v_mfma_f64_16x16x4_f64.  n <= 128 (first conv 4 -> 32, 3^3: n = 109; classifier
// 32 -> 3, 1^3: n = 33; LiTS first conv: n = 28), c2 <= 64.
// Patch order = the reference's im2col row order (c1, kd, kh, kw) (solver.py:104-108), voxel order (n, d, h, w).
// Persistent workgroups; every workgroup keeps its accumulator tiles in registers over all its voxel chunks and writes
// ONE partial slab at the end; a second kernel adds the slabs in workgroup order (deterministic) and mirrors Au.
#include "common.h"

namespace effq {

typedef double f64x4g __attribute__((ext_vector_type(4)));

constexpr int GF_VC = 32;            // voxels per LDS chunk
constexpr int GF_T = 256;
constexpr int GF_MAXN = 128, GF_MAXC2 = 64;
constexpr int GF_MAXTILES = 72;      // 36 (upper triangle at 8 tiles per edge) + 32 (4 x 8) rounded up

struct GramF64Params {
  const float* x;                    // NDHWC
  const float* y;                    // NDHWC (output voxels x c2)
  effq_geom g;
  int n, has_bias, np, c2p, ld;      // np = n rounded up to 16, row = [np patch entries | c2p target entries | pad]
  int od, oh, ow;
  long long V;                       // output voxels
  int nchunk, ntiles, te, tc;        // te = np / 16, tc = c2p / 16
  double* slab;                      // [gridDim.x][ntiles][256]
};

template <int TPW>
__global__ __launch_bounds__(GF_T, (TPW <= 6) ? 4 : 2) void k_gram_f64(GramF64Params p) {
  extern __shared__ __attribute__((aligned(16))) double rows[];      // [GF_VC][ld]
  __shared__ int tab_a[GF_MAXTILES], tab_b[GF_MAXTILES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  // tile table: Au tiles (ti <= tj) then Bu tiles; entry = column offset of the A / B operand inside a row
  if (tid == 0) {
    int t = 0;
    for (int i = 0; i < p.te; ++i)
      for (int j = i; j < p.te; ++j) {
        tab_a[t] = 16 * i;
        tab_b[t] = 16 * j;
        ++t;
      }
    for (int c = 0; c < p.tc; ++c)
      for (int j = 0; j < p.te; ++j) {
        tab_a[t] = p.np + 16 * c;
        tab_b[t] = 16 * j;
        ++t;
      }
  }
  __syncthreads();
  int aoff[TPW], boff[TPW];
  f64x4g acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = wid + 4 * t;
    const bool ok = tile < p.ntiles;
    aoff[t] = ok ? tab_a[tile] : 0;
    boff[t] = ok ? tab_b[tile] : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
  }
  const int nw = p.n - p.has_bias;
  // patch-entry table (c, kd, kh, kw) of the reference's im2col row order, decoded once
  __shared__ int qtab[GF_MAXN];
  __shared__ int vrow[GF_VC][4];                       // per chunk row: batch index and input origin (id0, ih0, iw0)
  {
    const int k3 = p.g.KD * p.g.KH * p.g.KW, k2 = p.g.KH * p.g.KW;
    for (int q = tid; q < nw; q += GF_T) {
      const int c = q / k3, kr = q - c * k3;
      const int kd = kr / k2, kr2 = kr - kd * k2;
      const int kh = kr2 / p.g.KW, kw = kr2 - kh * p.g.KW;
      qtab[q] = c | (kd << 16) | (kh << 20) | (kw << 24);
    }
  }
  const int rowlen = p.np + p.c2p;
  const int V = (int)p.V;
  for (int ch = blockIdx.x; ch < p.nchunk; ch += gridDim.x) {
    __syncthreads();                                   // the previous chunk has been consumed (and qtab is written)
    const int v0 = ch * GF_VC;
    if (tid < GF_VC) {
      const int v = v0 + tid;
      int t = v;
      const int ow_ = t % p.ow;
      t /= p.ow;
      const int oh_ = t % p.oh;
      t /= p.oh;
      const int od_ = t % p.od;
      vrow[tid][0] = (v < V) ? t / p.od : -1;
      vrow[tid][1] = od_ * p.g.SD - p.g.PD;
      vrow[tid][2] = oh_ * p.g.SH - p.g.PH;
      vrow[tid][3] = ow_ * p.g.SW - p.g.PW;
    }
    __syncthreads();
    for (int e = tid; e < GF_VC * rowlen; e += GF_T) {
      const int vl = e / rowlen, q = e - vl * rowlen;
      const int nn = vrow[vl][0];
      double val = 0.0;
      if (nn >= 0) {
        if (q < nw) {
          const int pk = qtab[q];
          const int c = pk & 0xffff;
          const int id = vrow[vl][1] + ((pk >> 16) & 15), ih = vrow[vl][2] + ((pk >> 20) & 15),
                    iw = vrow[vl][3] + ((pk >> 24) & 15);
          if (id >= 0 && id < p.g.D && ih >= 0 && ih < p.g.H && iw >= 0 && iw < p.g.W)
            val = (double)p.x[((((size_t)nn * p.g.D + id) * p.g.H + ih) * p.g.W + iw) * p.g.C1 + c];
        } else if (q < p.n) {
          val = 1.0;                                   // the ones row of the bias column (solver.py:256)
        } else if (q >= p.np && q - p.np < p.g.C2) {
          val = (double)p.y[(size_t)(v0 + vl) * p.g.C2 + (q - p.np)];
        }
      }
      rows[vl * p.ld + q] = val;
    }
    __syncthreads();
#pragma unroll 2
    for (int ks = 0; ks < GF_VC / 4; ++ks) {
      const double* r0 = rows + (ks * 4 + lk) * p.ld + lr;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const double a = r0[aoff[t]], b = r0[boff[t]];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = wid + 4 * t;
    if (tile < p.ntiles) {
      double* dst = p.slab + ((size_t)blockIdx.x * p.ntiles + tile) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r * 64 + lane] = acc[t][r];
    }
  }
}

// Au / Bu = sum of the slabs in workgroup order; element (r, lane) of a tile is (row = lane / 16 + 4 r, col = lane % 16)
__global__ __launch_bounds__(256) void k_gram_f64_reduce(GramF64Params p, int nslab, double* __restrict__ Au,
                                                         double* __restrict__ Bu) {
  const int tile = blockIdx.x, e = threadIdx.x;
  double s = 0.0;
  for (int w = 0; w < nslab; ++w) s += p.slab[((size_t)w * p.ntiles + tile) * 256 + e];
  const int r = e >> 6, lane = e & 63;
  const int row = (lane >> 4) + 4 * r, col = lane & 15;
  const int nau = p.te * (p.te + 1) / 2;
  if (tile < nau) {
    int i = 0, rem = tile;
    while (rem >= p.te - i) {
      rem -= p.te - i;
      ++i;
    }
    const int j = i + rem;
    const int gi = 16 * i + row, gj = 16 * j + col;
    if (gi < p.n && gj < p.n) {
      Au[(size_t)gi * p.n + gj] = s;
      if (i != j) Au[(size_t)gj * p.n + gi] = s;     // mirror of an off-diagonal tile (diagonal tiles hold both halves)
    }
  } else {
    const int t = tile - nau;
    const int c = t / p.te, j = t - c * p.te;
    const int gc = 16 * c + row, gj = 16 * j + col;
    if (gc < p.g.C2 && gj < p.n) Bu[(size_t)gc * p.n + gj] = s;
  }
}

static bool gram_f64_plan(const effq_geom* g, int has_bias, GramF64Params* p) {
  if (g == nullptr) return false;
  const long long nw = (long long)g->C1 * g->KD * g->KH * g->KW;
  const long long n = nw + (has_bias ? 1 : 0);
  if (n < 1 || n > GF_MAXN || g->C2 < 1 || g->C2 > GF_MAXC2) return false;
  const int od = (g->D + 2 * g->PD - g->KD) / g->SD + 1, oh = (g->H + 2 * g->PH - g->KH) / g->SH + 1,
            ow = (g->W + 2 * g->PW - g->KW) / g->SW + 1;
  if (od <= 0 || oh <= 0 || ow <= 0 || g->N <= 0) return false;
  p->g = *g;
  p->n = (int)n;
  p->has_bias = has_bias ? 1 : 0;
  p->np = ((int)n + 15) / 16 * 16;
  p->c2p = (g->C2 + 15) / 16 * 16;
  p->ld = p->np + p->c2p + 2;                       // (+2 doubles: rows of 16-double operand reads land on shifted banks)
  p->od = od; p->oh = oh; p->ow = ow;
  p->V = (long long)g->N * od * oh * ow;
  p->nchunk = (int)((p->V + GF_VC - 1) / GF_VC);
  p->te = p->np / 16;
  p->tc = p->c2p / 16;
  p->ntiles = p->te * (p->te + 1) / 2 + p->tc * p->te;
  return p->ntiles <= GF_MAXTILES && p->V < ((long long)1 << 31) - GF_VC && g->KD <= 15 && g->KH <= 15 && g->KW <= 15;
}

// persistent grid: 4 workgroups per CU (37 KB of LDS and <= 128 registers each): the gather of the patches is a stream of
// 4-byte loads that only many waves in flight keep busy (one workgroup per CU: 10.4 ms for the first conv of the BraTS
// net, 11 % of the fp64 matrix peak)
static int gram_f64_grid(const GramF64Params& p) { return p.nchunk < 1024 ? p.nchunk : 1024; }

}  // namespace effq
using namespace effq;

extern "C" {

int effq_gram_f64_supported(const effq_geom* g, int has_bias) {
  GramF64Params p;
  return gram_f64_plan(g, has_bias, &p) ? 1 : 0;
}

size_t effq_gram_f64_ws_bytes(const effq_geom* g, int has_bias) {
  GramF64Params p;
  if (!gram_f64_plan(g, has_bias, &p)) return 0;
  return (size_t)gram_f64_grid(p) * p.ntiles * 256 * sizeof(double);
}

int effq_gram_f64(const float* x_ndhwc, const float* y_ndhwc, const effq_geom* g, int has_bias, double* Au, double* Bu,
                  void* ws, size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG(x_ndhwc && y_ndhwc && g && Au && Bu && ws);
  GramF64Params p;
  if (!gram_f64_plan(g, has_bias, &p)) {
    set_error("gram_f64: unsupported geometry (n = C1*k^3 + bias <= %d, C2 <= %d)", GF_MAXN, GF_MAXC2);
    return EFFQ_ERR_ARG;
  }
  const size_t need = effq_gram_f64_ws_bytes(g, has_bias);
  if (ws_bytes < need) {
    set_error("gram_f64: workspace %zu < required %zu", ws_bytes, need);
    return EFFQ_ERR_WORKSPACE;
  }
  p.x = x_ndhwc;
  p.y = y_ndhwc;
  p.slab = reinterpret_cast<double*>(ws);
  const int grid = gram_f64_grid(p);
  const size_t lds = (size_t)GF_VC * p.ld * sizeof(double);
  const int tpw = (p.ntiles + 3) / 4;
  hipStream_t st = as_stream(stream);
#define EFFQ_GF(TPW_) hipLaunchKernelGGL((k_gram_f64<TPW_>), dim3(grid), dim3(GF_T), lds, st, p)
  if (tpw <= 3) EFFQ_GF(3); else if (tpw <= 6) EFFQ_GF(6); else if (tpw <= 11) EFFQ_GF(11); else EFFQ_GF(18);
#undef EFFQ_GF
  EFFQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gram_f64_reduce, dim3(p.ntiles), dim3(256), 0, st, p, grid, Au, Bu);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
