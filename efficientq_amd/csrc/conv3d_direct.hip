// Direct-gather fp32 loss convs for the two layers the tiled kernels serve badly (conv3d_quant_calib_step picks
// them when only the loss is wanted: no output tensor, no attention weights, input already quantised or raw):
//   * k_conv3d_c4: C1 == 4, 3x3x3 taps, any stride/padding, C2 == 32 - the first conv of the 3D-UNets
//     (4 MRI modalities -> 32 channels, stride 2: config/brats_ptq.yaml; EfficientQConv.py:118-122).  K = 108 is
//     too short for channel-slab LDS tiling; here one wave owns 32 consecutive output voxels, its 54 B operands
//     (27 taps x 2 K steps of v_mfma_f32_32x32x2_f32) sit in registers for the whole kernel and each lane gathers
//     the two channels it feeds (8 bytes per tap) straight from L2 - no LDS, latency covered by occupancy.
//   * k_conv1_mfma: 1x1x1 convs onto at most 4 output channels - the classifier (32 -> 3).  HBM-bound streaming; the dot
//     products run on v_mfma_f32_16x16x4_f32 (no cross-lane reduction).
// Both end in the deterministic last-block reduction of common.h and write [sum d^2, sum d^2] like the tiled path.
#include <stdint.h>
#include <stdlib.h>
#include "common.h"
#include "conv_direct.h"

namespace effq {

typedef float d_f32x16 __attribute__((ext_vector_type(16)));
typedef float d_f32x2 __attribute__((ext_vector_type(2)));
typedef float d_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_conv3d_c4(DirectParams p) {
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int T = 27;

  // B operands: K index of MFMA (tap, e) on lane half lh is channel 2*lh + e of that tap
  float breg[T][2];
#pragma unroll
  for (int tap = 0; tap < T; ++tap)
#pragma unroll
    for (int e = 0; e < 2; ++e) breg[tap][e] = p.G[((size_t)li * 4 + 2 * lh + e) * T + tap];   // G [C2][C1][T]
  const float bv = (p.bias != nullptr) ? p.bias[li] : 0.0f;

  double l0 = 0.0;
  const int wave_global = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;
  for (int tile = wave_global; tile < p.ntiles; tile += nwaves) {
    const long long v = (long long)tile * 32 + li;
    const bool vvalid = v < p.V;
    int t = (int)(vvalid ? v : 0);
    const int ow = t % p.OW;
    t /= p.OW;
    const int oh = t % p.OH;
    t /= p.OH;
    const int od = t % p.OD;
    const int n = t / p.OD;
    const int id0 = od * p.SD - p.PD, ih0 = oh * p.SH - p.PH, iw0 = ow * p.SW - p.PW;
    unsigned wm = 0, hm = 0, dm = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      wm |= (unsigned)(iw0 + k >= 0 && iw0 + k < p.W) << k;
      hm |= (unsigned)(ih0 + k >= 0 && ih0 + k < p.H) << k;
      dm |= (unsigned)(id0 + k >= 0 && id0 + k < p.D) << k;
    }
    if (!vvalid) dm = 0;
    const int xbase = (((n * p.D + id0) * p.H + ih0) * p.W + iw0) * 4 + 2 * lh;

    // targets first (independent of everything else), then the 27 gathers, then the MFMA chain
    const long long v0 = (long long)tile * 32;
    float yv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long vr = v0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      yv[r] = p.y[(vr < p.V ? vr : 0) * 32 + li];
    }
    d_f32x2 xv[T];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const bool ok = ((dm >> kd) & (hm >> kh) & (wm >> kw) & 1u) != 0;
          const int addr = ok ? (xbase + ((kd * p.H + kh) * p.W + kw) * 4) : 0;
          const d_f32x2 raw = *reinterpret_cast<const d_f32x2*>(p.x + addr);
          xv[(kd * 3 + kh) * 3 + kw] = ok ? raw : d_f32x2{0.0f, 0.0f};
        }
    d_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[tap][0], breg[tap][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[tap][1], breg[tap][1], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long vr = v0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (vr < p.V) {
        const float d = (acc[r] + bv) - yv[r];
        l0 += (double)d * (double)d;
      }
    }
  }
  double vsum[2] = {l0, l0};
  grid_sum_finish<2>(vsum, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.x, gridDim.x);
}

// 1x1x1 conv onto C2 <= 16 channels on the f32 matrix cores (v_mfma_f32_16x16x4_f32; the classifier uses 3 of the 16
// columns - the matrix cores are idle anyway and the dot products need no cross-lane reduction this way).  One wave-tile
// = 16 voxels: lane (row = l & 15, kq = l >> 4) loads the C1/4 consecutive channels [kq C1/4, (kq+1) C1/4) of voxel `row`
// as float4s - the 64 lanes of a load instruction together cover whole cache lines of the 16 x C1 floats - and MFMA j
// contracts channel kq C1/4 + j of every quarter (the B operand holds the weights in the same order).  U tiles are in
// flight per wave (all loads of a body are issued before its first MFMA): HBM-bound streaming of 4 C1 + 4 C2 bytes/voxel.
template <int C1, int U>
__global__ __launch_bounds__(256) void k_conv1_mfma(DirectParams p) {
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;
  constexpr int Q = C1 / 4, NV = Q / 4;          // floats / float4s per lane and tile
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int row = lane & 15, kq = lane >> 4;
  float breg[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) breg[j] = (row < p.C2) ? p.G[(size_t)row * C1 + kq * Q + j] : 0.0f;   // column = lane & 15
  const float bv = (p.bias != nullptr && row < p.C2) ? p.bias[row] : 0.0f;
  const long long ntile = (p.V + 15) / 16, nbody = (ntile + U - 1) / U;
  const long long nwaves = (long long)gridDim.x * 4;
  double l0 = 0.0;
  for (long long body = (long long)blockIdx.x * 4 + wid; body < nbody; body += nwaves) {
    d_f32x4 xv[U][NV];
    float yv[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long v0 = (body * U + u) * 16;
      const long long vx = (v0 + row < p.V) ? v0 + row : p.V - 1;
#pragma unroll
      for (int q = 0; q < NV; ++q) xv[u][q] = *reinterpret_cast<const d_f32x4*>(p.x + vx * C1 + kq * Q + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long vy = v0 + 4 * kq + i;        // output row 4 kq + i of the tile, column `row`
        yv[u][i] = (vy < p.V && row < p.C2) ? p.y[vy * p.C2 + row] : 0.0f;
      }
    }
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      d_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < Q; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][j >> 2][j & 3], breg[j], acc, 0, 0, 0);
      const long long v0 = (body * U + u) * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = (v0 + 4 * kq + i < p.V) && row < p.C2;
        const float d = (acc[i] + bv) - yv[u][i];
        s = ok ? __builtin_fmaf(d, d, s) : s;
      }
    }
    l0 += (double)s;
  }
  double vsum[2] = {l0, l0};
  grid_sum_finish<2>(vsum, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.x, gridDim.x);
}

int conv_direct_kind(const effq_geom* g) {
  if (g->C1 == 4 && g->C2 == 32 && g->KD == 3 && g->KH == 3 && g->KW == 3) return 1;
  if (g->KD == 1 && g->KH == 1 && g->KW == 1 && g->SD == 1 && g->SH == 1 && g->SW == 1 && g->PD == 0 && g->PH == 0 &&
      g->PW == 0 && g->C2 <= 4 && (g->C1 == 32 || g->C1 == 64 || g->C1 == 128 || g->C1 == 256))
    return 2;
  return 0;
}

int conv_direct_launch(int kind, DirectParams& p, size_t max_blocks, hipStream_t st) {
  if ((long long)p.N * p.D * p.H * p.W * p.C1 >= (1ll << 31) || p.V >= (1ll << 31)) return EFFQ_ERR_ARG;
  p.ntiles = (int)((p.V + 31) / 32);
  if (kind == 1) {
    size_t grid = ((size_t)p.ntiles + 3) / 4;
    static const size_t cap = getenv("EFFQ_C4_GRID") ? (size_t)atoi(getenv("EFFQ_C4_GRID")) : 512;   // tuning aid
    // 2 workgroups per CU (3 are 5 % faster alone): the 164-VGPR waves then leave room for the scale fixed point
    // of the next ADMM iteration to run beside this kernel instead of queueing behind it (first-conv layer 135 -> 122 ms)
    if (grid > cap) grid = cap;
    if (grid > max_blocks) grid = max_blocks;
    hipLaunchKernelGGL(k_conv3d_c4, dim3((unsigned)grid), dim3(256), 0, st, p);
    return EFFQ_OK;
  }
  static const size_t cap1 = getenv("EFFQ_C1_GRID") ? (size_t)atoi(getenv("EFFQ_C1_GRID")) : 1024;   // tuning aid
  const size_t nbody4 = (size_t)((p.V + 63) / 64);
  size_t grid = (nbody4 + 3) / 4;
  if (grid > cap1) grid = cap1;
  if (grid > max_blocks) grid = max_blocks;
  if (grid < 1) grid = 1;
  switch (p.C1) {
    case 32: hipLaunchKernelGGL((k_conv1_mfma<32, 4>), dim3((unsigned)grid), dim3(256), 0, st, p); break;
    case 64: hipLaunchKernelGGL((k_conv1_mfma<64, 4>), dim3((unsigned)grid), dim3(256), 0, st, p); break;
    case 128: hipLaunchKernelGGL((k_conv1_mfma<128, 2>), dim3((unsigned)grid), dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((k_conv1_mfma<256, 1>), dim3((unsigned)grid), dim3(256), 0, st, p); break;
  }
  return EFFQ_OK;
}

}  // namespace effq
