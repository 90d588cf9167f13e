// effq_gram_loss: the loss of one ADMM iterate from the layer's unweighted Gram system instead of a pass over the voxels.
// Reference: EfficientQConv.py:118-122 evaluates F.mse_loss(F.conv3d(Qx, G, b*), y) 200x per layer, each a full pass over
// the calibration volumes.  The conv output is linear in the iterate: out_v,c = g_c . xhat_v with g_c = [G[c,:], b*_c] and
// xhat_v = [im2col patch of the quantised input; 1], so
//   sum_v,c (out - y)^2 = sum_c g_c^T Au g_c - 2 sum_c g_c . Bu_c + sum y^2,     Au = sum_v xhat xhat^T,  Bu = sum_v y xhat^T
// - the objective of the (unweighted, quirk Q5) least-squares problem itself.  Au / Bu come out of the SAME exact-integer
// pass that builds the attention-weighted A0 / B0 (effq_gram_accum_i8_unw: integer class slabs summed without the weights,
// scaled in fp64), so an evaluation costs c2 n^2 multiply-adds on an n x n matrix - 6 MB for a 32 -> 32 3^3 layer against
// 0.67 GB of targets and level ids per pass - whenever the voxel count is far above n = 27 c1 + 1 (the 32- and 64-channel
// layers and the 1^3 convs; the 128/256-channel layers on 16^3 / 8^3 volumes keep the conv, which is cheaper there).
// Arithmetic: fp64 throughout.  g . Au . g is evaluated as <Au, M>, M = sum_c g_c g_c^T (products of fp32 values: exact in
// fp64), on the upper block triangle (Au is symmetric).  The three terms are each of the size of sum y^2 while the loss
// may be 1e-2 of it: fp64 leaves ~1e-13 of sum y^2, i.e. <= 1e-11 of the loss - tighter than the fp32 epilogue of the conv
// kernels (1e-6).  Bu carries y in 32-bit fixed point (30 bits below max|y|, as B0 does): <= 5e-9 of sum y^2 worst case.
#include "common.h"

namespace effq {

constexpr int GL_TB = 64;            // tile edge
constexpr int GL_T = 256;
constexpr int GL_CC = 32;            // output channels per LDS chunk

struct GramLossParams {
  const double* Au;                  // [n][n]
  const double* Bu;                  // [c2][n]
  const double* syy;                 // device scalar: sum y^2 over this rank's voxels
  const float* G;                    // [c2][n - has_bias]
  const float* b;                    // [c2] or null
  int c2, n, has_bias, nt;           // nt = tiles per edge
  int ntile, ncross;                 // upper-triangle tiles, cross-term workgroups
  double* partials;
  unsigned int* ticket;
  double* out;                       // [2]
};

__device__ __forceinline__ float gl_g(const GramLossParams& p, int c, int k) {
  const int nw = p.n - p.has_bias;
  if (k < nw) return p.G[(size_t)c * nw + k];
  return (k < p.n && p.b != nullptr) ? p.b[c] : 0.0f;
}

__global__ __launch_bounds__(GL_T) void k_gram_loss(GramLossParams p) {
  __shared__ float gk[GL_CC][GL_TB + 4], gj[GL_CC][GL_TB + 4];
  __shared__ double red_smem[16];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  double part = 0.0;
  if ((int)blockIdx.x < p.ntile) {
    // upper-triangle tile (K, J), J >= K, from the linear index
    int K = 0, rem = (int)blockIdx.x;
    while (rem >= p.nt - K) {
      rem -= p.nt - K;
      ++K;
    }
    const int J = K + rem;
    const int k0 = K * GL_TB, j0 = J * GL_TB;
    const int tk = (tid >> 4) * 4, tj = (tid & 15) * 4;          // 4 x 4 micro-tile of this thread
    double m[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) m[a][b] = 0.0;
    for (int c0 = 0; c0 < p.c2; c0 += GL_CC) {
      __syncthreads();
      for (int e = tid; e < GL_CC * GL_TB; e += GL_T) {
        const int c = e / GL_TB, q = e % GL_TB;
        const bool cok = c0 + c < p.c2;
        gk[c][q] = cok ? gl_g(p, c0 + c, k0 + q) : 0.0f;
        gj[c][q] = cok ? gl_g(p, c0 + c, j0 + q) : 0.0f;
      }
      __syncthreads();
#pragma unroll 4
      for (int c = 0; c < GL_CC; ++c) {
        double a4[4], b4[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          a4[a] = (double)gk[c][tk + a];
          b4[a] = (double)gj[c][tj + a];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) m[a][b] = __builtin_fma(a4[a], b4[b], m[a][b]);
      }
    }
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int k = k0 + tk + a;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = j0 + tj + b;
        if (k < p.n && j < p.n) acc = __builtin_fma(p.Au[(size_t)k * p.n + j], m[a][b], acc);
      }
    }
    part = (J > K) ? 2.0 * acc : acc;                             // Au and M are symmetric: the mirror tile is equal
  } else {
    // cross term -2 sum g . Bu, strided over the c2 x n elements
    const int w = (int)blockIdx.x - p.ntile;
    const size_t tot = (size_t)p.c2 * p.n;
    double acc = 0.0;
    for (size_t e = (size_t)w * GL_T + tid; e < tot; e += (size_t)p.ncross * GL_T) {
      const int c = (int)(e / p.n), k = (int)(e % p.n);
      acc = __builtin_fma((double)gl_g(p, c, k), p.Bu[e], acc);
    }
    part = -2.0 * acc;
  }
  double v[1] = {part};
  grid_sum_finish<1>(v, p.partials, p.ticket, p.out, red_smem, &s_last);
  if (s_last && tid == 0) {
    const double loss = p.out[0] + *p.syy;
    p.out[0] = loss;
    p.out[1] = loss;
  }
}

}  // namespace effq
using namespace effq;

extern "C" {

size_t effq_gram_loss_ws_bytes(int n) {
  if (n <= 0) return 0;
  const size_t nt = ((size_t)n + GL_TB - 1) / GL_TB;
  return 256 + (nt * (nt + 1) / 2 + 64) * sizeof(double) + 256;
}

int effq_gram_loss(const double* Au, const double* Bu, const double* syy_dev, const float* G, const float* b, int c2, int n,
                   int has_bias, double* sqerr_out, void* ws, size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG(Au && Bu && syy_dev && G && sqerr_out && ws && c2 > 0 && n > 0);
  EFFQ_CHECK_ARG(!has_bias || b != nullptr);
  if (ws_bytes < effq_gram_loss_ws_bytes(n)) {
    set_error("gram_loss: workspace %zu < required %zu", ws_bytes, effq_gram_loss_ws_bytes(n));
    return EFFQ_ERR_WORKSPACE;
  }
  GramLossParams p;
  p.Au = Au; p.Bu = Bu; p.syy = syy_dev; p.G = G; p.b = has_bias ? b : nullptr;
  p.c2 = c2; p.n = n; p.has_bias = has_bias ? 1 : 0;
  p.nt = (n + GL_TB - 1) / GL_TB;
  p.ntile = p.nt * (p.nt + 1) / 2;
  size_t nc = ((size_t)c2 * n + (size_t)GL_T * 8 - 1) / ((size_t)GL_T * 8);
  if (nc < 1) nc = 1;
  if (nc > 64) nc = 64;
  p.ncross = (int)nc;
  EFFQ_CHECK_ARG(p.ntile + p.ncross <= RED_MAX_BLOCKS * 16);
  char* base = reinterpret_cast<char*>(ws);
  p.ticket = reinterpret_cast<unsigned int*>(base);          // zero-filled once by the caller, left at zero by the kernel
  p.partials = reinterpret_cast<double*>(base + 256);
  p.out = sqerr_out;
  hipLaunchKernelGGL(k_gram_loss, dim3((unsigned)(p.ntile + p.ncross)), dim3(GL_T), 0, as_stream(stream), p);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
