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

// ---- the same conv with the input staged through LDS -------------------------------------------------------------
// k_conv3d_c4 gathers every tap straight from L1/L2: 27 gathers of 8 bytes per lane and tile, 16 cache lines each at
// stride 2 - the texture path, not HBM or the matrix cores, set its 0.52 ms (26 % of 8 TB/s).  Here a workgroup owns
// 4 x 4 x 8 output voxels (wave w = d-plane w), stages their ((4-1)S+3) x ((4-1)S+3) x ((8-1)S+3) input voxels (16 bytes
// each) once through LDS - coalesced 16-byte loads, prefetched into registers under the MFMAs of the previous tile - and
// every tap is one ds_read_b64 per lane.  Measured (16 x 4 x 128^3, stride 2): 0.44 ms against 0.52; with the MFMA chain
// removed 0.25 ms (1.26 GB incl. halo overlap = 5.1 TB/s), with the loads removed 0.28 ms (K = 108 on
// v_mfma_f32_32x32x2_f32 is 54 instructions of 64 cycles per 32 x 32 outputs: 0.18 ms at full issue rate), with both
// removed 0.07 ms: the two phases still add up instead of overlapping at 2 workgroups per CU (254 VGPRs).
// S = stride (1 or 2, isotropic); persistent workgroups walk contiguous runs of tiles.
constexpr int C4_TD = 4, C4_TH = 4, C4_TW = 8;
template <int S>
__global__ __launch_bounds__(256, 2) void k_conv3d_c4h(DirectParams p, int tiles_d, int tiles_h, int tiles_w, int ntiles) {
  constexpr int HD = (C4_TD - 1) * S + 3, HH = (C4_TH - 1) * S + 3, HW = (C4_TW - 1) * S + 3, NHV = HD * HH * HW;
  constexpr int NHL = (NHV + 255) / 256;
  constexpr int T = 27;
  __shared__ __attribute__((aligned(16))) float halo[NHV * 4];
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  float breg[T][2];
#pragma unroll
  for (int tap = 0; tap < T; ++tap)
#pragma unroll
    for (int e = 0; e < 2; ++e) breg[tap][e] = p.G[((size_t)li * 4 + 2 * lh + e) * T + tap];   // G [C2][C1][T]
  const float bv = (p.bias != nullptr) ? p.bias[li] : 0.0f;

  // per-thread halo voxels (relative coordinates and offsets fixed at start)
  int hcd[NHL], hch[NHL], hcw[NHL];
  long long hrel[NHL];
#pragma unroll
  for (int k = 0; k < NHL; ++k) {
    const int u = tid + k * 256;
    const int v = (u < NHV) ? u : 0;
    hcw[k] = v % HW;
    const int t2 = v / HW;
    hch[k] = t2 % HH;
    const int hd = t2 / HH;
    hcd[k] = (u < NHV) ? hd : (1 << 20);            // dead slot: never in range
    hrel[k] = (((long long)hd * p.H + hch[k]) * p.W + hcw[k]) * 4;
  }
  int yrel[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;                // voxel of the plane: h = i >> 3, w = i & 7
    yrel[r] = ((i >> 3) * p.OW + (i & 7)) * 32 + li;
  }
  // A fragment of this lane: output voxel (plane wid, h = li >> 3, w = li & 7), channels 2 lh, 2 lh + 1
  const int abase = (((wid * S) * HH + (li >> 3) * S) * HW + (li & 7) * S) * 4 + 2 * lh;

  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = (t_begin + per < ntiles) ? t_begin + per : ntiles;

  d_f32x4 hreg[NHL];
  float ynext[16], ycur[16];
  unsigned hmask = 0, ymask_next = 0, ymask_cur = 0;
  auto fetch = [&](int tile) {
    int t = tile;
    const int ow0 = (t % tiles_w) * C4_TW;
    t /= tiles_w;
    const int oh0 = (t % tiles_h) * C4_TH;
    t /= tiles_h;
    const int od0 = (t % tiles_d) * C4_TD;
    const int n = t / tiles_d;
    const int id0 = od0 * S - p.PD, ih0 = oh0 * S - p.PH, iw0 = ow0 * S - p.PW;
    // tiles that touch no face of the input or output volume (uniform test): one base address + per-lane offsets
    const bool interior = id0 >= 0 && id0 + HD <= p.D && ih0 >= 0 && ih0 + HH <= p.H && iw0 >= 0 && iw0 + HW <= p.W &&
                          od0 + C4_TD <= p.OD && oh0 + C4_TH <= p.OH && ow0 + C4_TW <= p.OW;
    if (interior) {
      const float* hb = p.x + ((((long long)n * p.D + id0) * p.H + ih0) * p.W + iw0) * 4;
#pragma unroll
      for (int k = 0; k < NHL; ++k) hreg[k] = *reinterpret_cast<const d_f32x4*>(hb + ((hcd[k] < HD) ? hrel[k] : 0));
      hmask = 0xffffffffu;
      const float* yb = p.y + ((((long long)n * p.OD + od0 + wid) * p.OH + oh0) * p.OW + ow0) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) ynext[r] = yb[yrel[r]];
      ymask_next = 0xffffu;
      return;
    }
    unsigned hm = 0;
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int id = id0 + hcd[k], ih = ih0 + hch[k], iw = iw0 + hcw[k];
      const bool ok = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      const int cd = min(max(id, 0), p.D - 1), chh = min(max(ih, 0), p.H - 1), cw = min(max(iw, 0), p.W - 1);
      hm |= (ok ? 1u : 0u) << k;
      hreg[k] = *reinterpret_cast<const d_f32x4*>(p.x + ((((size_t)n * p.D + cd) * p.H + chh) * p.W + cw) * 4);
    }
    hmask = hm;
    const int od = od0 + wid;
    unsigned ym = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int oh = oh0 + (i >> 3), ow = ow0 + (i & 7);
      const bool ok = od < p.OD && oh < p.OH && ow < p.OW;
      ym |= (ok ? 1u : 0u) << r;
      ynext[r] = p.y[((((size_t)n * p.OD + min(od, p.OD - 1)) * p.OH + min(oh, p.OH - 1)) * p.OW + min(ow, p.OW - 1)) * 32 + li];
    }
    ymask_next = ym;
  };

  double l0 = 0.0;
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    lds_barrier();                                         // the previous tile's fragment reads are done
    if (hmask == 0xffffffffu) {
#pragma unroll
      for (int k = 0; k < NHL; ++k)
        if (tid + k * 256 < NHV) *reinterpret_cast<d_f32x4*>(&halo[(tid + k * 256) * 4]) = hreg[k];
    } else {
#pragma unroll
      for (int k = 0; k < NHL; ++k) {
        const int u = tid + k * 256;
        const d_f32x4 val = ((hmask >> k) & 1u) ? hreg[k] : d_f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // zero padding
        if (u < NHV) *reinterpret_cast<d_f32x4*>(&halo[u * 4]) = val;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ycur[r] = ynext[r];
    ymask_cur = ymask_next;
    lds_barrier();
    fetch((tile + 1 < t_end) ? tile + 1 : tile);           // the last tile harmlessly re-reads itself
    __builtin_amdgcn_sched_barrier(0);
    d_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const d_f32x2 a = *reinterpret_cast<const d_f32x2*>(&halo[abase + ((kd * HH + kh) * HW + kw) * 4]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], breg[tap][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], breg[tap][1], acc, 0, 0, 0);
    }
    float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ymask_cur == 0xffffu) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = (acc[r] + bv) - ycur[r];
        s4[r & 3] = __builtin_fmaf(d, d, s4[r & 3]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = (acc[r] + bv) - ycur[r];
        s4[r & 3] = ((ymask_cur >> r) & 1u) ? __builtin_fmaf(d, d, s4[r & 3]) : s4[r & 3];
      }
    }
    l0 += (double)((s4[0] + s4[1]) + (s4[2] + s4[3]));
  }
  double vsum[2] = {l0, l0};
  grid_sum_finish<2>(vsum, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.x, gridDim.x);
}

// ---- single-channel first conv (1 -> 32, 3^3: the LiTS nets, CT volumes) -----------------------------------------------------
// The generic tiled kernel pads K = 27 to its 32-channel slabs and took 2.1 ms per evaluation on 8 x 160^3 volumes
// (0.66 GB of input and targets).  Same scheme as k_conv3d_c4h: the input tile is staged through LDS (4 bytes per voxel),
// K = 27 (+ 1 zero tap) runs as 14 v_mfma_f32_32x32x2_f32 steps - lane half h of step j contracts tap 2 j + h - and the
// targets are read in MFMA layout.
template <int SD, int SH, int SW>
__global__ __launch_bounds__(256, 2) void k_conv3d_c1h(DirectParams p, int tiles_d, int tiles_h, int tiles_w, int ntiles) {
  constexpr int HD = (C4_TD - 1) * SD + 3, HH = (C4_TH - 1) * SH + 3, HW = (C4_TW - 1) * SW + 3, NHV = HD * HH * HW;
  constexpr int NHL = (NHV + 255) / 256;
  constexpr int NJ = 14;                             // K steps of 2 taps
  __shared__ float halo[NHV + 4];
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  float breg[NJ];
  int toff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int tap = 2 * j + lh;
    breg[j] = (tap < 27) ? p.G[(size_t)li * 27 + tap] : 0.0f;              // G [C2][1][27]
    const int tc = (tap < 27) ? tap : 0;                                    // (the zero weight makes the 28th tap harmless)
    const int kd = tc / 9, kh = (tc / 3) % 3, kw = tc % 3;
    toff[j] = (kd * HH + kh) * HW + kw;
  }
  const float bv = (p.bias != nullptr) ? p.bias[li] : 0.0f;

  int hcd[NHL], hch[NHL], hcw[NHL];
  long long hrel[NHL];
#pragma unroll
  for (int k = 0; k < NHL; ++k) {
    const int u = tid + k * 256;
    const int v = (u < NHV) ? u : 0;
    hcw[k] = v % HW;
    const int t2 = v / HW;
    hch[k] = t2 % HH;
    const int hd = t2 / HH;
    hcd[k] = (u < NHV) ? hd : (1 << 20);
    hrel[k] = ((long long)hd * p.H + hch[k]) * p.W + hcw[k];
  }
  int yrel[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
    yrel[r] = ((i >> 3) * p.OW + (i & 7)) * 32 + li;
  }
  const int abase = ((wid * SD) * HH + (li >> 3) * SH) * HW + (li & 7) * SW;

  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = (t_begin + per < ntiles) ? t_begin + per : ntiles;

  float hreg[NHL];
  float ynext[16], ycur[16];
  unsigned hmask = 0, ymask_next = 0, ymask_cur = 0;
  auto fetch = [&](int tile) {
    int t = tile;
    const int ow0 = (t % tiles_w) * C4_TW;
    t /= tiles_w;
    const int oh0 = (t % tiles_h) * C4_TH;
    t /= tiles_h;
    const int od0 = (t % tiles_d) * C4_TD;
    const int n = t / tiles_d;
    const int id0 = od0 * SD - p.PD, ih0 = oh0 * SH - p.PH, iw0 = ow0 * SW - p.PW;
    const bool interior = id0 >= 0 && id0 + HD <= p.D && ih0 >= 0 && ih0 + HH <= p.H && iw0 >= 0 && iw0 + HW <= p.W &&
                          od0 + C4_TD <= p.OD && oh0 + C4_TH <= p.OH && ow0 + C4_TW <= p.OW;
    if (interior) {
      const float* hb = p.x + (((long long)n * p.D + id0) * p.H + ih0) * p.W + iw0;
#pragma unroll
      for (int k = 0; k < NHL; ++k) hreg[k] = hb[(hcd[k] < HD) ? hrel[k] : 0];
      hmask = 0xffffffffu;
      const float* yb = p.y + ((((long long)n * p.OD + od0 + wid) * p.OH + oh0) * p.OW + ow0) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) ynext[r] = yb[yrel[r]];
      ymask_next = 0xffffu;
      return;
    }
    unsigned hm = 0;
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int id = id0 + hcd[k], ih = ih0 + hch[k], iw = iw0 + hcw[k];
      const bool ok = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      const int cd = min(max(id, 0), p.D - 1), chh = min(max(ih, 0), p.H - 1), cw = min(max(iw, 0), p.W - 1);
      hm |= (ok ? 1u : 0u) << k;
      hreg[k] = p.x[(((size_t)n * p.D + cd) * p.H + chh) * p.W + cw];
    }
    hmask = hm;
    const int od = od0 + wid;
    unsigned ym = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int oh = oh0 + (i >> 3), ow = ow0 + (i & 7);
      const bool ok = od < p.OD && oh < p.OH && ow < p.OW;
      ym |= (ok ? 1u : 0u) << r;
      ynext[r] = p.y[((((size_t)n * p.OD + min(od, p.OD - 1)) * p.OH + min(oh, p.OH - 1)) * p.OW + min(ow, p.OW - 1)) * 32 + li];
    }
    ymask_next = ym;
  };

  double l0 = 0.0;
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    lds_barrier();
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int u = tid + k * 256;
      if (u < NHV) halo[u] = ((hmask >> k) & 1u) ? hreg[k] : 0.0f;       // zero padding
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ycur[r] = ynext[r];
    ymask_cur = ymask_next;
    lds_barrier();
    fetch((tile + 1 < t_end) ? tile + 1 : tile);
    __builtin_amdgcn_sched_barrier(0);
    d_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(halo[abase + toff[j]], breg[j], acc, 0, 0, 0);
    float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = (acc[r] + bv) - ycur[r];
      s4[r & 3] = ((ymask_cur >> r) & 1u) ? __builtin_fmaf(d, d, s4[r & 3]) : s4[r & 3];
    }
    l0 += (double)((s4[0] + s4[1]) + (s4[2] + s4[3]));
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
  if (g->C1 == 1 && g->C2 == 32 && g->KD == 3 && g->KH == 3 && g->KW == 3 &&
      ((g->SD == 1 && g->SH == 1 && g->SW == 1) || (g->SD == 2 && g->SH == 2 && (g->SW == 1 || g->SW == 2))))
    return 3;
  if (g->KD == 1 && g->KH == 1 && g->KW == 1 && g->SD == 1 && g->SH == 1 && g->SW == 1 && g->PD == 0 && g->PH == 0 &&
      g->PW == 0 && g->C2 <= 4 && (g->C1 == 32 || g->C1 == 64 || g->C1 == 128 || g->C1 == 256))
    return 2;
  return 0;
}

int conv_direct_launch(int kind, DirectParams& p, size_t max_blocks, hipStream_t st) {
  if ((long long)p.N * p.D * p.H * p.W * p.C1 >= (1ll << 31) || p.V >= (1ll << 31)) return EFFQ_ERR_ARG;
  p.ntiles = (int)((p.V + 31) / 32);
  if (kind == 3) {
    const int td = (p.OD + C4_TD - 1) / C4_TD, th = (p.OH + C4_TH - 1) / C4_TH, tw = (p.OW + C4_TW - 1) / C4_TW;
    const long long nt = (long long)p.N * td * th * tw;
    if (nt >= (1ll << 30)) return EFFQ_ERR_ARG;
    size_t grid = (size_t)nt < 512 ? (size_t)nt : 512;
    if (grid > max_blocks) grid = max_blocks;
    if (p.SD == 2 && p.SW == 2)
      hipLaunchKernelGGL((k_conv3d_c1h<2, 2, 2>), dim3((unsigned)grid), dim3(256), 0, st, p, td, th, tw, (int)nt);
    else if (p.SD == 2)
      hipLaunchKernelGGL((k_conv3d_c1h<2, 2, 1>), dim3((unsigned)grid), dim3(256), 0, st, p, td, th, tw, (int)nt);   // LiTS
    else
      hipLaunchKernelGGL((k_conv3d_c1h<1, 1, 1>), dim3((unsigned)grid), dim3(256), 0, st, p, td, th, tw, (int)nt);
    return EFFQ_OK;
  }
  if (kind == 1 && p.SD == p.SH && p.SH == p.SW && (p.SD == 1 || p.SD == 2)) {
    static const int use_lds = getenv("EFFQ_C4_LDS") ? atoi(getenv("EFFQ_C4_LDS")) : 1;      // tuning aid
    if (use_lds) {
      const int td = (p.OD + C4_TD - 1) / C4_TD, th = (p.OH + C4_TH - 1) / C4_TH, tw = (p.OW + C4_TW - 1) / C4_TW;
      const long long nt = (long long)p.N * td * th * tw;
      if (nt < (1ll << 30)) {
        static const size_t cap = getenv("EFFQ_C4_GRID") ? (size_t)atoi(getenv("EFFQ_C4_GRID")) : 512;   // 2 workgroups per CU
        size_t grid = (size_t)nt < cap ? (size_t)nt : cap;
        if (grid > max_blocks) grid = max_blocks;
        if (p.SD == 2)
          hipLaunchKernelGGL(k_conv3d_c4h<2>, dim3((unsigned)grid), dim3(256), 0, st, p, td, th, tw, (int)nt);
        else
          hipLaunchKernelGGL(k_conv3d_c4h<1>, dim3((unsigned)grid), dim3(256), 0, st, p, td, th, tw, (int)nt);
        return EFFQ_OK;
      }
    }
  }
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
