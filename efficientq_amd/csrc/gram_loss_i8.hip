// effq_gram_loss_i8: the losses of a GROUP of ADMM iterates of a wide layer from the layer's unweighted Gram system, with
// the quadratic form on the i8 matrix cores - exact integer arithmetic.
// Reference: EfficientQConv.py:118-122 evaluates F.mse_loss(F.conv3d(Qx, G, b*), y) once per iteration; gram_loss.hip
// explains the identity  sum (out - y)^2 = sum_c g_c^T Au g_c - 2 sum_c g_c . Bu_c + sum y^2.  Its fp64 evaluation costs
// 2 c2 n^2 flop per iterate - 24.5 GFLOP at c2 = 256, n = 6913: more than the conv pass it would replace, which is why the
// 128- and 256-channel layers kept the exact-integer conv kernels (k_conv3d_i8g<4/8>: 0.10 / 0.20 ms per iterate in situ,
// 8 - 12 % of the i8 matrix peak, 0.11 s of loss-stream time per calibration).  But on those layers BOTH factors of the
// quadratic form are small integers:
//   w_c = s_w J_c,  J = 2 level - (L_w - 1)  (the int8 ring the projection already writes for the integer conv),
//   Aww = s_a^2 K,  K = sum_v k k^T over the level ids k of the quantised input (exact integers <= (L_a - 1)^2 V),
// so  sum_c w_c^T Aww w_c = s_w^2 s_a^2 Q,  Q = sum_c J_c^T K J_c = <K, J^T J>  is an INTEGER, computed here as an i8
// GEMM: K split once per layer into balanced base-256 digit planes D_p (K = sum_p 256^p D_p, |D_p| <= 128), and per group
//   T_p = D_p . [J_1^T | J_2^T | ...]   (v_mfma_i32_32x32x32_i8, int32 accumulators: |T| <= 128 * 63 * n < 2^31),
//   Q_j = sum_p 256^p sum_{r,c} J_j[c][r] T_p[r][c]   (int64, integer atomics: exact, order-independent).
// K is symmetric, so only the chunks k >= the row tile's first row are visited, the ones beyond the diagonal tile
// doubled.  The bias row / column of Au and the cross term with Bu are c2 n fp64 multiply-adds (k_gl8_finish).
// One launch evaluates the whole group the loss stream picks up (8 iterates): the planes are read once per group.
// Cost at c2 = 256, n = 6913, 3 planes: 3 x 12.2 GOP per iterate on the i8 cores instead of a 29 GOP conv pass at 8 % of
// their peak; at c2 = 128, n = 3457: 3 x 1.5 GOP.
#include "common.h"

namespace effq {

typedef int g8_v4i __attribute__((ext_vector_type(4)));
typedef int g8_v16i __attribute__((ext_vector_type(16)));

constexpr int G8_TM = 256, G8_TN = 256;    // workgroup tile (rows of K x columns of the stacked iterates)
constexpr int G8_KC = 64;                  // bytes of K per chunk (two MFMA K steps)
constexpr int G8_RS = 80;                  // LDS row pitch (bytes): ds_read_b128 operand fetches conflict-free
constexpr int G8_T = 512;                  // 8 waves, each 64 x 128
constexpr int G8_TILE = G8_TM * G8_RS;     // bytes per staged operand tile
constexpr int G8_MAXP = 6;
constexpr int G8_MAXGROUP = 16;
constexpr int G8_FIN_WG = 32;

struct Gl8Params {
  const int8_t* planes;      // [P][nwp][nw]
  const int8_t* Gq;          // [ncols][nw]: the int8 iterates of the group, stacked (ring slots are contiguous)
  unsigned long long* Qacc;  // [count]
  unsigned int* ticket;      // next tile (zero before the launch; k_gl8_finish leaves it at zero again)
  int P, nwp, nw, ncols, c2, mt, nt, ntiles;
};

// One tile of the product per call; the launch is a PERSISTENT grid of far fewer workgroups than CUs (EFFQ_GL8_WGS,
// default 64) that draw tiles from a ticket: the losses only rank iterates that the chain has long left behind (a group of
// 8 iterates has 1 - 3 ms of chain time to finish in), and a grid that fills the chip keeps the 256-row prox GEMM of the
// chain (one workgroup per CU, 147 KB of LDS) waiting for CUs: 0.20 -> 0.42 ms per solve with 648 resident-at-will
// workgroups.
__device__ __forceinline__ void gl8_tile(const Gl8Params& p, int tile, unsigned char* g8_smem) {
  unsigned char* const As = g8_smem;                    // [2][G8_TILE]
  unsigned char* const Bs = g8_smem + 2 * G8_TILE;      // [2][G8_TILE]
  // tiles in order of descending work (ascending row tile), the N tiles of one (row tile, plane) next to each other
  const int nti = tile % p.nt, mp = tile / p.nt;
  const int mtile = mp / p.P, plane = mp % p.P;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int r0 = mtile * G8_TM, n0 = nti * G8_TN;
  const int nchunks = p.nw / G8_KC;
  const int cd0 = mtile * (G8_TM / G8_KC);                                  // first chunk of the diagonal tile
  int n_diag = nchunks - cd0;
  if (n_diag > G8_TM / G8_KC) n_diag = G8_TM / G8_KC;
  const int n_off = nchunks - cd0 - n_diag;                                 // chunks beyond the diagonal tile
  const int n_all = n_off + n_diag;

  g8_v16i acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0;

  // staging: 1024 pieces of 16 B per operand tile, two per thread
  const int8_t* Ag[2];
  const int8_t* Bg[2];
  bool bok[2];
  int soff[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int q = tid + G8_T * v, row = q >> 2, c16 = q & 3;
    Ag[v] = p.planes + ((size_t)plane * p.nwp + (size_t)(r0 + row)) * p.nw + c16 * 16;
    bok[v] = (n0 + row) < p.ncols;
    Bg[v] = p.Gq + (size_t)(bok[v] ? (n0 + row) : 0) * p.nw + c16 * 16;
    soff[v] = row * G8_RS + c16 * 16;
  }
  auto chunk_k = [&](int c) -> size_t {
    const int ch = (c < n_off) ? (cd0 + n_diag + c) : (cd0 + (c - n_off));
    return (size_t)ch * G8_KC;
  };
  g8_v4i ra[2], rb[2];
  const g8_v4i zero4 = {0, 0, 0, 0};
  if (n_all > 0) {
    const size_t k0 = chunk_k(0);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      ra[v] = *reinterpret_cast<const g8_v4i*>(Ag[v] + k0);
      rb[v] = bok[v] ? *reinterpret_cast<const g8_v4i*>(Bg[v] + k0) : zero4;
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      *reinterpret_cast<g8_v4i*>(As + soff[v]) = ra[v];
      *reinterpret_cast<g8_v4i*>(Bs + soff[v]) = rb[v];
    }
  }
  lds_barrier();
  for (int c = 0; c < n_all; ++c) {
    const int buf = c & 1;
    if (c + 1 < n_all) {
      const size_t k1 = chunk_k(c + 1);
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        ra[v] = *reinterpret_cast<const g8_v4i*>(Ag[v] + k1);
        rb[v] = bok[v] ? *reinterpret_cast<const g8_v4i*>(Bg[v] + k1) : zero4;
      }
    }
    if (c == n_off && n_off > 0) {         // everything so far lies beyond the diagonal tile: its mirror image is not visited
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= 2;
    }
    const unsigned char* Ab = As + buf * G8_TILE + (wr * 64 + lr) * G8_RS + lh * 16;
    const unsigned char* Bb = Bs + buf * G8_TILE + (wc * 128 + lr) * G8_RS + lh * 16;
#pragma unroll
    for (int ks = 0; ks < G8_KC / 32; ++ks) {
      g8_v4i a[2], bq[4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const g8_v4i*>(Ab + mi * 32 * G8_RS + ks * 32);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bq[ni] = *reinterpret_cast<const g8_v4i*>(Bb + ni * 32 * G8_RS + ks * 32);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], bq[ni], acc[mi][ni], 0, 0, 0);
    }
    if (c + 1 < n_all) {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        *reinterpret_cast<g8_v4i*>(As + (buf ^ 1) * G8_TILE + soff[v]) = ra[v];
        *reinterpret_cast<g8_v4i*>(Bs + (buf ^ 1) * G8_TILE + soff[v]) = rb[v];
      }
    }
    lds_barrier();
  }
  // ---- epilogue: sum_{r, c} J[c][r] T[r][c] per iterate, int64.  Accumulator element reg of lane (lr, lh) of a 32 x 32
  // tile: row (reg & 3) + 8 (reg >> 2) + 4 lh, column lr; the 32 J bytes of that column are two 16-byte loads.
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int ncol = n0 + wc * 128 + ni * 32 + lr;
    long long sum = 0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int row0 = r0 + wr * 64 + mi * 32;
      if (row0 < p.nw && ncol < p.ncols) {
        const int8_t* jp = p.Gq + (size_t)ncol * p.nw + row0;
        const g8_v4i j0 = *reinterpret_cast<const g8_v4i*>(jp), j1 = *reinterpret_cast<const g8_v4i*>(jp + 16);
        const int w[8] = {j0[0], j0[1], j0[2], j0[3], j1[0], j1[1], j1[2], j1[3]};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int word = lh ? w[2 * (r >> 2) + 1] : w[2 * (r >> 2)];
          const int jv = (int)(signed char)((unsigned)word >> (8 * (r & 3)));
          sum += (long long)acc[mi][ni][r] * (long long)jv;   // (|T| <= 2 * 128 * 63 * n < 2^31 for n < 2^16)
        }
      }
    }
    // the 32 columns of a tile belong to ONE iterate (c2 is a multiple of 32): wave total -> one atomic
    unsigned long long u = (unsigned long long)sum;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned lo = (unsigned)__shfl_down((int)(unsigned)u, off, 64), hi = (unsigned)__shfl_down((int)(unsigned)(u >> 32), off, 64);
      u += ((unsigned long long)hi << 32) | lo;
    }
    const int ncol0 = n0 + wc * 128 + ni * 32;
    if (lane == 0 && ncol0 < p.ncols) atomicAdd(p.Qacc + ncol0 / p.c2, u << (8 * plane));
  }
}

__global__ __launch_bounds__(G8_T) void k_gl8(const Gl8Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char g8_smem[];
  __shared__ int s_tile;
  for (;;) {
    if (threadIdx.x == 0) s_tile = (int)atomicAdd(p.ticket, 1u);
    __syncthreads();
    const int tile = s_tile;
    if (tile >= p.ntiles) break;                         // (every workgroup draws exactly one ticket >= ntiles)
    gl8_tile(p, tile, g8_smem);
    __syncthreads();                                     // LDS and s_tile are reused
  }
}

// K = rint(Au / s_a^2) on the weight rows, split into P balanced base-256 digits; rows nw .. nwp-1 of every plane are zero
__global__ __launch_bounds__(256) void k_gl8_planes(const double* __restrict__ Au, int n, int nw, int nwp,
                                                    const float* __restrict__ alpha, int act_levels, int P,
                                                    int8_t* __restrict__ planes, int32_t* __restrict__ err) {
  const double s = (double)alpha[0] / (double)(act_levels - 1);
  const double inv = 1.0 / (s * s);
  const size_t tot = (size_t)nwp * nw, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) {
    const int r = (int)(e / nw), k = (int)(e % nw);
    long long t = 0;
    if (r < nw) {
      const double x = Au[(size_t)r * n + k] * inv;
      const double xr = rint(x);
      if (!(fabs(x - xr) < 1e-3) || !(fabs(xr) < 1e15)) *err = 1;      // Au is not the integer system it should be
      t = (long long)xr;
    }
    for (int q = 0; q < P; ++q) {
      const int d = (int)(((t + 128) & 255) - 128);
      planes[(size_t)q * tot + e] = (int8_t)d;
      t = (t - d) >> 8;
    }
    if (t != 0) *err = 2;                                               // more digits than planes
  }
}

struct Gl8Fin {
  const double* Au;           // [n][n]: only its last row / column (bias) is read here
  const double* Bu;           // [c2][n]
  const double* syy;
  const int8_t* Gq;           // [count][c2][nw]
  const float* b;             // [count][c2] or null
  const effq_fp_state* states;
  const float* alpha;
  unsigned long long* Qacc;
  unsigned int* tile_ticket;  // k_gl8's ticket: reset here
  double* partials;           // [count][G8_FIN_WG]
  unsigned int* tickets;      // [count]
  double* hist;               // [count][2]
  int c2, n, nw, has_bias, act_levels, w_levels;
};

__global__ __launch_bounds__(256) void k_gl8_finish(const Gl8Fin p) {
  __shared__ double red_smem[16];
  __shared__ int s_last;
  const int j = (int)blockIdx.y, tid = threadIdx.x;
  const double sw = (double)(float)p.states[j].alpha / (double)(p.w_levels - 1);
  const int8_t* J = p.Gq + (size_t)j * p.c2 * p.nw;
  const float* bj = p.b ? p.b + (size_t)j * p.c2 : nullptr;
  const double* arow = p.Au + (size_t)(p.n - 1) * p.n;         // bias row of Au (has_bias): sum_v xhat
  const size_t tot = (size_t)p.c2 * p.n;
  double acc = 0.0;
  for (size_t e = (size_t)blockIdx.x * 256 + tid; e < tot; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e / p.n), k = (int)(e % p.n);
    if (k < p.nw) {
      const double g = sw * (double)J[(size_t)c * p.nw + k];
      double t = -2.0 * p.Bu[e];
      if (bj != nullptr) t += 2.0 * (double)bj[c] * arow[k];
      acc = __builtin_fma(g, t, acc);
    } else if (bj != nullptr) {
      const double g = (double)bj[c];
      acc = __builtin_fma(g, -2.0 * p.Bu[e] + g * arow[k], acc);
    }
  }
  double v[1] = {acc};
  double out[1];
  grid_sum_finish<1>(v, p.partials + (size_t)j * G8_FIN_WG, p.tickets + j, out, red_smem, &s_last, blockIdx.x, gridDim.x);
  if (s_last && tid == 0) {
    const double sa = (double)p.alpha[0] / (double)(p.act_levels - 1);
    const long long q = (long long)__hip_atomic_load(p.Qacc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double loss = (sw * sw) * (sa * sa) * (double)q + out[0] + *p.syy;
    p.hist[2 * j] = loss;
    p.hist[2 * j + 1] = loss;
    __hip_atomic_store(p.Qacc + j, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left at zero for the next group
    if (j == 0) __hip_atomic_store(p.tile_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static inline int g8_round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace effq
using namespace effq;

extern "C" {

int effq_gram_loss_i8_supported(int c2, int n, int has_bias, int w_levels) {
  const int nw = n - (has_bias ? 1 : 0);
  return (c2 > 0 && nw > 0 && (c2 % 32) == 0 && (nw % G8_KC) == 0 && nw < 65536 && w_levels >= 2 && w_levels <= 64) ? 1 : 0;
}

int effq_gram_loss_i8_num_planes(long long kmax) {
  // balanced digits: P planes hold |K| < 128 * 256^(P-1) + ... >= 2^(8P - 1) - 1
  int P = 1;
  while (P < G8_MAXP && kmax > ((1ll << (8 * P - 1)) - 1)) ++P;
  return (kmax <= ((1ll << (8 * P - 1)) - 1)) ? P : -1;
}

size_t effq_gram_loss_i8_planes_bytes(int n, int has_bias, int nplanes) {
  const int nw = n - (has_bias ? 1 : 0);
  if (nw <= 0 || nplanes <= 0) return 0;
  return (size_t)nplanes * (size_t)g8_round_up(nw, G8_TM) * (size_t)nw;
}

int effq_gram_loss_i8_prepare(const double* Au, int n, int has_bias, const float* act_alpha_dev, int act_levels,
                              int nplanes, int8_t* planes, int32_t* err_flag_dev, void* stream) {
  EFFQ_CHECK_ARG(Au && act_alpha_dev && planes && err_flag_dev && n > 1 && act_levels >= 2);
  EFFQ_CHECK_ARG(nplanes >= 1 && nplanes <= G8_MAXP);
  const int nw = n - (has_bias ? 1 : 0), nwp = g8_round_up(nw, G8_TM);
  size_t nb = ((size_t)nwp * nw + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(k_gl8_planes, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), Au, n, nw, nwp, act_alpha_dev,
                     act_levels, nplanes, planes, err_flag_dev);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

size_t effq_gram_loss_i8_ws_bytes(void) {
  return 256 + sizeof(unsigned long long) * G8_MAXGROUP + sizeof(unsigned int) * G8_MAXGROUP +
         sizeof(double) * G8_MAXGROUP * G8_FIN_WG + 256;
}

int effq_gram_loss_i8(const int8_t* planes, int nplanes, const double* Au, const double* Bu, const double* syy_dev,
                      const int8_t* Gq, const float* b, const effq_fp_state* states, const float* act_alpha_dev,
                      int act_levels, int w_levels, int c2, int n, int has_bias, int count, double* hist_out, void* ws,
                      size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG(planes && Au && Bu && syy_dev && Gq && states && act_alpha_dev && hist_out && ws);
  EFFQ_CHECK_ARG(count >= 1 && count <= G8_MAXGROUP && nplanes >= 1 && nplanes <= G8_MAXP && act_levels >= 2);
  EFFQ_CHECK_ARG(effq_gram_loss_i8_supported(c2, n, has_bias, w_levels));
  EFFQ_CHECK_ARG(!has_bias || b != nullptr);
  if (ws_bytes < effq_gram_loss_i8_ws_bytes()) {
    set_error("gram_loss_i8: workspace %zu < required %zu", ws_bytes, effq_gram_loss_i8_ws_bytes());
    return EFFQ_ERR_WORKSPACE;
  }
  const int nw = n - (has_bias ? 1 : 0);
  char* base = reinterpret_cast<char*>(ws);                     // zero-filled once by the caller, left at zero by the kernels
  unsigned long long* Qacc = reinterpret_cast<unsigned long long*>(base);
  unsigned int* tickets = reinterpret_cast<unsigned int*>(base + sizeof(unsigned long long) * G8_MAXGROUP);
  unsigned int* tile_ticket = tickets + G8_MAXGROUP;
  double* partials = reinterpret_cast<double*>(base + 256 + sizeof(unsigned long long) * G8_MAXGROUP);
  Gl8Params p;
  p.planes = planes; p.Gq = Gq; p.Qacc = Qacc;
  p.P = nplanes; p.nw = nw; p.nwp = g8_round_up(nw, G8_TM); p.ncols = count * c2; p.c2 = c2;
  p.mt = p.nwp / G8_TM; p.nt = (p.ncols + G8_TN - 1) / G8_TN;
  p.ntiles = p.mt * p.P * p.nt;
  p.ticket = tile_ticket;
  static const int wgs_env = getenv("EFFQ_GL8_WGS") ? atoi(getenv("EFFQ_GL8_WGS")) : 64;
  int wgs = wgs_env < 1 ? 1 : wgs_env;
  if (wgs > p.ntiles) wgs = p.ntiles;
  static bool attr_set = false;
  const int lds = 4 * G8_TILE;
  if (!attr_set) {
    EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gl8), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_gl8, dim3((unsigned)wgs), dim3(G8_T), lds, st, p);
  EFFQ_LAUNCH_CHECK();
  Gl8Fin f;
  f.Au = Au; f.Bu = Bu; f.syy = syy_dev; f.Gq = Gq; f.b = has_bias ? b : nullptr; f.states = states; f.alpha = act_alpha_dev;
  f.Qacc = Qacc; f.tile_ticket = tile_ticket; f.partials = partials; f.tickets = tickets; f.hist = hist_out;
  f.c2 = c2; f.n = n; f.nw = nw; f.has_bias = has_bias ? 1 : 0; f.act_levels = act_levels; f.w_levels = w_levels;
  hipLaunchKernelGGL(k_gl8_finish, dim3(G8_FIN_WG, count), dim3(256), 0, st, f);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
