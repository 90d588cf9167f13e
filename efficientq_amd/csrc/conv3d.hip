// conv3d_quant_calib_step: NDHWC implicit-GEMM Conv3d on the f32 matrix cores with an LDS-staged
// halo tile and a fused squared-error epilogue.
// Reference: EfficientQConv.py:118-122,161-165 (conv3d + mse per ADMM iteration),
//            PTQConv.py:154-167 (fp / quantised forward).
//
// Mapping (gfx950, wave64): one workgroup = 4 waves = one 4x4x8 block of output voxels (M=128) x
// (32*NT) output channels.  Wave w owns d-plane w of the block: its 32x32 MFMA tile rows are the
// 4x8 (h,w) voxels of that plane, columns are 32 output channels.  K runs over (tap, channel):
// the halo tile of one <=32-channel slab lives in LDS as [voxel][slab+4 pad] floats, so that a
// ds_read_b128 of 4 channels per lane is bank-conflict free; the lane halves (k index of
// v_mfma_f32_32x32x2_f32) take channels {0-3} and {4-7} of an 8-channel chunk, giving 4 MFMAs per
// LDS read.  Weights are repacked once per call into [tap][c/4][c2][4] so that a wave's B operand
// is one coalesced 16-byte load per lane straight from L2 (shared by every workgroup).
#include <stdlib.h>
#include "common.h"
#include "conv_direct.h"

namespace effq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TD = 4, TH = 4, TW = 8;

struct ConvParams {
  const float* x;
  const float* wp;
  const float* bias;
  const float* y;
  const float* att;
  float* out;
  const float* act_alpha;
  float act_d;
  int act_on;
  int N, C1, C2, D, H, W, OD, OH, OW;
  int KD, KH, KW, SD, SH, SW, PD, PH, PW;
  int c1p, c2p, cslab, nslab, CS;
  int HD, HH, HW, nhalo;
  int tiles_d, tiles_h, tiles_w, ntiles;
  double* partials;
  unsigned int* ticket;
  double* sqerr;
  int debug;   // profiling ablations only (EFFQ_CONV_DEBUG): 1 = no MFMA loop, 2 = no halo staging, 3 = no epilogue loads
};

__device__ __forceinline__ float act_qd(float x, float alpha, float d) {
  // PTQConv._quantize_act, fp32 (PTQConv.py:114-116): discretize(x/alpha, L, 0, 1) * alpha
  float t = x / alpha;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  float r = rintf(t / d);
  return (r * d) * alpha;
}

__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ G, float* __restrict__ wp, int C1,
                                                     int C2, int T, int c1p, int c2p) {
  const size_t total = (size_t)T * c1p * c2p;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    int c_lo = (int)(e & 3);
    size_t r = e >> 2;
    int j = (int)(r % c2p);
    r /= c2p;
    int cq = (int)(r % (c1p / 4));
    int tap = (int)(r / (c1p / 4));
    int c = cq * 4 + c_lo;
    float v = 0.0f;
    if (c < C1 && j < C2) v = G[((size_t)j * C1 + c) * T + tap];
    wp[e] = v;
  }
}

// Generic geometry (any kernel / stride / channel count), persistent: each workgroup walks a contiguous
// run of tiles; the halo slab of stage k+1 and the targets of the next tile are fetched into registers
// right before the MFMA loop of stage k, so the memory phases of these (mostly bandwidth-bound) layers
// overlap the arithmetic.  Weights come straight from L2 (one 16-byte load per lane per 4 MFMAs).
constexpr int G_MAXH = 12;   // halo 16-byte loads per thread per slab (plan guarantees nhalo*cslab/4 <= 3072)

template <int NT, bool VEC, bool HAS_Y>
__global__ __launch_bounds__(256, 2) void k_conv3d(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ch0 = blockIdx.y * 32 * NT;
  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = (t_begin + per < p.ntiles) ? t_begin + per : p.ntiles;
  const int nstage = (t_end > t_begin) ? (t_end - t_begin) * p.nslab : 0;

  struct Tile {
    int n, od0, oh0, ow0;
  };
  auto decode = [&](int tile) {
    Tile r;
    int t = tile;
    r.ow0 = (t % p.tiles_w) * TW;
    t /= p.tiles_w;
    r.oh0 = (t % p.tiles_h) * TH;
    t /= p.tiles_h;
    r.od0 = (t % p.tiles_d) * TD;
    r.n = t / p.tiles_d;
    return r;
  };
  const int HH = p.HH, HW = p.HW, CS = p.CS;
  const int upv = p.cslab >> 2;
  const int nh4 = p.nhalo * upv;
  // Prefetch loads are UNCONDITIONAL on clamped (always valid) addresses; validity bits are applied when the
  // registers are stored to LDS.  A load under a runtime branch makes hipcc wait for it inside the branch
  // (vmcnt(0)), which serialised the whole prefetch in front of the MFMA loop.
  auto load_halo = [&](const Tile& tl, int slab, float4(&hreg)[G_MAXH], unsigned& hmask) {
    const int id0 = tl.od0 * p.SD - p.PD, ih0 = tl.oh0 * p.SH - p.PH, iw0 = tl.ow0 * p.SW - p.PW;
    hmask = 0u;
#pragma unroll
    for (int k = 0; k < G_MAXH; ++k) {
      int u = tid + k * 256;
      const bool live = u < nh4;
      u = live ? u : 0;
      const int vox = u / upv, c4 = u - vox * upv;
      const int hw = vox % HW;
      const int t2 = vox / HW;
      const int hh = t2 % HH, hd = t2 / HH;
      const int id = id0 + hd, ih = ih0 + hh, iw = iw0 + hw;
      const int c = slab * p.cslab + c4 * 4;
      const bool ok = live && id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && c < p.C1;
      const int cd = min(max(id, 0), p.D - 1), chh = min(max(ih, 0), p.H - 1), cw = min(max(iw, 0), p.W - 1);
      const float* vp = p.x + ((((size_t)tl.n * p.D + cd) * p.H + chh) * p.W + cw) * p.C1;
      if (VEC) {
        hreg[k] = *reinterpret_cast<const float4*>(vp + min(c, p.C1 - 4));
      } else {
        hreg[k].x = vp[min(c + 0, p.C1 - 1)];
        hreg[k].y = vp[min(c + 1, p.C1 - 1)];
        hreg[k].z = vp[min(c + 2, p.C1 - 1)];
        hreg[k].w = vp[min(c + 3, p.C1 - 1)];
      }
      hmask |= (ok ? 1u : 0u) << k;
    }
  };
  // targets in accumulator layout: lane (li,lh), register r <-> voxel (r&3)+8*(r>>2)+4*lh of d-plane wid
  auto load_y = [&](const Tile& tl, float(&yv)[NT][16]) {
    const int od = min(tl.od0 + wid, p.OD - 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = min(ch0 + nt * 32 + li, p.C2 - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int oh = min(tl.oh0 + (i >> 3), p.OH - 1), ow = min(tl.ow0 + (i & 7), p.OW - 1);
        yv[nt][r] = p.y[((((size_t)tl.n * p.OD + od) * p.OH + oh) * p.OW + ow) * p.C2 + ch];   // masked in the epilogue
      }
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;

  const int hv = ((wid * p.SD) * HH + (li >> 3) * p.SH) * HW + (li & 7) * p.SW;
  float alpha = 1.0f;
  if (p.act_on) alpha = *p.act_alpha;
  constexpr bool has_y = HAS_Y;
  double l0 = 0.0, l1 = 0.0;

  float4 hreg[G_MAXH];
  unsigned hmask = 0u;
  float ynext[NT][16], ycur[NT][16];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ycur[nt][r] = ynext[nt][r] = 0.0f;
  if (nstage > 0) {
    const Tile t0 = decode(t_begin);
    load_halo(t0, 0, hreg, hmask);
    if (HAS_Y) load_y(t0, ynext);
  }

  int tile = t_begin, slab = 0;
  for (int k = 0; k < nstage; ++k) {
    lds_barrier();                             // LDS free
#pragma unroll
    for (int q = 0; q < G_MAXH; ++q) {
      const int u = tid + q * 256;
      if (u < nh4) {
        float4 v = ((hmask >> q) & 1u) ? hreg[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (!VEC) {      // components beyond C1 of a partial channel quad
          const int c = slab * p.cslab + (u % upv) * 4;
          if (c + 1 >= p.C1) v.y = 0.f;
          if (c + 2 >= p.C1) v.z = 0.f;
          if (c + 3 >= p.C1) v.w = 0.f;
        }
        if (p.act_on) {
          v.x = act_qd(v.x, alpha, p.act_d);
          v.y = act_qd(v.y, alpha, p.act_d);
          v.z = act_qd(v.z, alpha, p.act_d);
          v.w = act_qd(v.w, alpha, p.act_d);
        }
        const int vox = u / upv, c4 = u - vox * upv;
        *reinterpret_cast<float4*>(&lds[vox * CS + c4 * 4]) = v;
      }
    }
    if (slab == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ycur[nt][r] = ynext[nt][r];
    }
    lds_barrier();

    int tile_n = tile, slab_n = slab + 1;
    if (slab_n == p.nslab) {
      slab_n = 0;
      tile_n = tile + 1;
    }
    {
      // unconditional: the last stage re-fetches itself (never consumed)
      const bool more = k + 1 < nstage;
      const Tile tn = decode(more ? tile_n : tile);
      const int sl = more ? slab_n : slab;
      load_halo(tn, sl, hreg, hmask);
      if (HAS_Y) load_y(tn, ynext);       // consumed only after a slab-0 stage (ycur = ynext there)
    }

    // flattened (tap, 8-channel chunk) loop with the NEXT step's operands (LDS A fragment, L2 B fragments)
    // fetched before the current step's MFMAs: one step has only 4*NT MFMAs, too few to hide an L2 round trip
    {
      const int nq = p.cslab >> 3;
      const int total = (EFFQ_DBG(p) == 1) ? 0 : p.KD * p.KH * p.KW * nq;
      const float* abase = lds + hv * CS + 4 * lh;
      const float* wbase = p.wp + ((size_t)(((slab * p.cslab) >> 2) + lh) * p.c2p + ch0 + li) * 4;
      const size_t wtap = (size_t)(p.c1p >> 2) * p.c2p * 4;       // floats between taps
      const size_t wq2 = (size_t)2 * p.c2p * 4;                    // floats between 8-channel chunks
      int kd = 0, kh = 0, kw = 0, q = 0, tap = 0;
      float4 a_cur = *reinterpret_cast<const float4*>(abase);
      float4 b_cur[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b_cur[nt] = *reinterpret_cast<const float4*>(wbase + nt * 32 * 4);
      for (int it = 0; it < total; ++it) {
        if (++q == nq) {
          q = 0;
          ++tap;
          if (++kw == p.KW) {
            kw = 0;
            if (++kh == p.KH) {
              kh = 0;
              ++kd;
            }
          }
        }
        float4 a_nxt = a_cur, b_nxt[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = b_cur[nt];
        if (it + 1 < total) {
          a_nxt = *reinterpret_cast<const float4*>(abase + ((kd * HH + kh) * HW + kw) * CS + q * 8);
          const float* wr = wbase + (size_t)tap * wtap + (size_t)q * wq2;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = *reinterpret_cast<const float4*>(wr + nt * 32 * 4);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur[nt].x, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur[nt].y, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur[nt].z, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur[nt].w, acc[nt], 0, 0, 0);
        a_cur = a_nxt;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_cur[nt] = b_nxt[nt];
      }
    }

    if (slab == p.nslab - 1) {
      const Tile tl = decode(tile);
      const int od = tl.od0 + wid;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ch = ch0 + nt * 32 + li;
        const bool chok = ch < p.C2;
        const float bv = (p.bias != nullptr && chok) ? p.bias[ch] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int oh = tl.oh0 + (i >> 3), ow = tl.ow0 + (i & 7);
          if (chok && od < p.OD && oh < p.OH && ow < p.OW) {
            const size_t vo = (((size_t)tl.n * p.OD + od) * p.OH + oh) * p.OW + ow;
            const float o = acc[nt][r] + bv;
            if (p.out != nullptr) p.out[vo * p.C2 + ch] = o;
            if (has_y) {
              const float dlt = o - ycur[nt][r];
              const float sq = dlt * dlt;
              l0 += (double)sq;
              l1 += (p.att != nullptr) ? (double)(p.att[vo] * sq) : (double)sq;
            }
          }
          acc[nt][r] = 0.0f;
        }
      }
    }
    tile = tile_n;
    slab = slab_n;
  }
  if (has_y) {
    double v[2] = {l0, l1};
    grid_sum_finish<2>(v, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.y * gridDim.x + blockIdx.x,
                       gridDim.x * gridDim.y);
  }
}

// ---- specialised path: 3x3x3, stride 1, 32-channel slabs ------------------------------------------
// Same mapping as k_conv3d, plus: compile-time halo geometry (6x6x10 voxels, 36-float stride), the
// per-tap weight block [8 chunks-of-4][32*NT][4] staged once per workgroup through a double-buffered
// LDS ring (global -> registers at the top of a tap, registers -> LDS behind the tap's MFMAs, one
// barrier per tap), and a fully unrolled 4-chunk body so that all LDS operand reads of a tap are in
// flight ahead of its 16*NT MFMAs.
constexpr int F_HD = TD + 2, F_HH = TH + 2, F_HW = TW + 2, F_NH = F_HD * F_HH * F_HW, F_CS = 36;

template <int NT, bool HAS_Y>
__global__ __launch_bounds__(256, 2) void k_conv3d_k3(ConvParams p) {
  // Persistent form: gridDim.x workgroups (<= 2 per CU) each walk a contiguous run of spatial tiles.
  // The global loads of stage k+1 (halo slab, and the targets of the next tile) are issued right
  // before the 27-tap MFMA loop of stage k and land in registers underneath it, so HBM/L2 latency and
  // the epilogue operands are off the critical path; the loss is accumulated per workgroup and only
  // gridDim.x*gridDim.y partials take part in the deterministic grid sum.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;
  float* halo = lds;
  float4* wbuf = reinterpret_cast<float4*>(lds + F_NH * F_CS);   // [2][8][32*NT] float4

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ch0 = blockIdx.y * 32 * NT;
  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = (t_begin + per < p.ntiles) ? t_begin + per : p.ntiles;
  const int nstage = (t_end > t_begin) ? (t_end - t_begin) * p.nslab : 0;

  constexpr int CPV = 8 * NT;                  // 16-byte cells per output voxel
  constexpr int NCELL = 4 * NT;                // cells per thread (128 voxels x CPV / 256)
  constexpr int NHL = (F_NH * 8 + 255) / 256;  // halo 16-byte loads per thread (12)
  constexpr int TS = 32 * NT + 4;              // transpose row stride (floats)

  struct Tile {
    int n, od0, oh0, ow0;
  };
  auto decode = [&](int tile) {
    Tile r;
    int t = tile;
    r.ow0 = (t % p.tiles_w) * TW;
    t /= p.tiles_w;
    r.oh0 = (t % p.tiles_h) * TH;
    t /= p.tiles_h;
    r.od0 = (t % p.tiles_d) * TD;
    r.n = t / p.tiles_d;
    return r;
  };
  // prefetch loads are unconditional on clamped addresses (a load under a runtime branch is waited for inside
  // the branch and would serialise the prefetch); validity is applied at the LDS store / in the epilogue
  auto load_halo = [&](const Tile& tl, int slab, float4(&hreg)[NHL], unsigned& hmask) {
    const int id0 = tl.od0 - p.PD, ih0 = tl.oh0 - p.PH, iw0 = tl.ow0 - p.PW;
    hmask = 0u;
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      int u = tid + k * 256;
      const bool live = u < F_NH * 8;
      u = live ? u : 0;
      const int vox = u >> 3, c4 = u & 7;
      const int hw = vox % F_HW;
      const int t2 = vox / F_HW;
      const int hh = t2 % F_HH, hd = t2 / F_HH;
      const int id = id0 + hd, ih = ih0 + hh, iw = iw0 + hw;
      const bool ok = live && id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      const int cd = min(max(id, 0), p.D - 1), chh = min(max(ih, 0), p.H - 1), cw = min(max(iw, 0), p.W - 1);
      hreg[k] = *reinterpret_cast<const float4*>(p.x + ((((size_t)tl.n * p.D + cd) * p.H + chh) * p.W + cw) * p.C1 +
                                                 slab * 32 + c4 * 4);
      hmask |= (ok ? 1u : 0u) << k;
    }
  };
  auto load_y = [&](const Tile& tl, float4(&yv)[NCELL]) {
#pragma unroll
    for (int k = 0; k < NCELL; ++k) {
      const int u = tid + k * 256;
      const int vox = u / CPV;
      const int od = min(tl.od0 + (vox >> 5), p.OD - 1), oh = min(tl.oh0 + ((vox >> 3) & 3), p.OH - 1),
                ow = min(tl.ow0 + (vox & 7), p.OW - 1);
      yv[k] = *reinterpret_cast<const float4*>(p.y + ((((size_t)tl.n * p.OD + od) * p.OH + oh) * p.OW + ow) * p.C2 +
                                               ch0 + (u % CPV) * 4);
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;

  const int hv = (wid * F_HH + (li >> 3)) * F_HW + (li & 7);
  float alpha = 1.0f;
  if (p.act_on) alpha = *p.act_alpha;
  const int c4q = p.c1p >> 2;
  const int wq = tid >> 5, wj = tid & 31;      // this thread's slot of the per-tap weight block
  const size_t wtap = (size_t)c4q * p.c2p;     // float4 stride between taps
  constexpr bool has_y = HAS_Y;
  double l0 = 0.0, l1 = 0.0;

  float4 hreg[NHL], ynext[NCELL], ycur[NCELL];
  unsigned hmask = 0u;
#pragma unroll
  for (int k = 0; k < NCELL; ++k) ycur[k] = ynext[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nstage > 0) {
    const Tile t0 = decode(t_begin);
    load_halo(t0, 0, hreg, hmask);
    if (HAS_Y) load_y(t0, ynext);
  }

  int tile = t_begin, slab = 0;
  for (int k = 0; k < nstage; ++k) {
    const float4* wsrc = reinterpret_cast<const float4*>(p.wp) + ((size_t)(slab * 8 + wq) * p.c2p + ch0 + wj);
    float4 w0[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) w0[nt] = wsrc[nt * 32];
    lds_barrier();                             // LDS (halo / ring / transpose buffer) is free
#pragma unroll
    for (int q = 0; q < NHL; ++q) {
      const int u = tid + q * 256;
      if (u < F_NH * 8) {
        float4 v = ((hmask >> q) & 1u) ? hreg[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.act_on) {
          v.x = act_qd(v.x, alpha, p.act_d);
          v.y = act_qd(v.y, alpha, p.act_d);
          v.z = act_qd(v.z, alpha, p.act_d);
          v.w = act_qd(v.w, alpha, p.act_d);
        }
        *reinterpret_cast<float4*>(&halo[(u >> 3) * F_CS + (u & 7) * 4]) = v;
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wbuf[wq * (32 * NT) + nt * 32 + wj] = w0[nt];
    if (slab == 0) {
#pragma unroll
      for (int q = 0; q < NCELL; ++q) ycur[q] = ynext[q];
    }
    lds_barrier();

    // prefetch of stage k+1 (next slab of this tile, or slab 0 + targets of the next tile)
    int tile_n = tile, slab_n = slab + 1;
    if (slab_n == p.nslab) {
      slab_n = 0;
      tile_n = tile + 1;
    }
    {
      // unconditional: the last stage re-fetches itself (never consumed)
      const bool more = k + 1 < nstage;
      const Tile tn = decode(more ? tile_n : tile);
      load_halo(tn, more ? slab_n : slab, hreg, hmask);
      if (HAS_Y) load_y(tn, ynext);       // consumed only after a slab-0 stage (ycur = ynext there)
    }

    const int ntap = (EFFQ_DBG(p) == 1) ? 0 : 27;
#pragma unroll 1
    for (int tap = 0; tap < ntap; ++tap) {
      const int buf = tap & 1;
      float4 wnext[NT];
      const int tnx = (tap + 1 < 27) ? tap + 1 : tap;     // last tap re-reads itself (never written back)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wnext[nt] = wsrc[(size_t)tnx * wtap + nt * 32];
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const float* arow = halo + (hv + (kd * F_HH + kh) * F_HW + kw) * F_CS + 4 * lh;
      const float4* brow = wbuf + buf * (8 * 32 * NT) + lh * (32 * NT) + li;
      float4 a[4], b[4][NT];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[q] = *reinterpret_cast<const float4*>(arow + q * 8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[q][nt] = brow[(2 * q) * (32 * NT) + nt * 32];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q][nt].x, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q][nt].y, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q][nt].z, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q][nt].w, acc[nt], 0, 0, 0);
      }
      if (tap + 1 < 27) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wbuf[(buf ^ 1) * (8 * 32 * NT) + wq * (32 * NT) + nt * 32 + wj] = wnext[nt];
      }
      lds_barrier();
    }

    if (slab == p.nslab - 1) {
      // ---- tile epilogue: transpose the accumulators through LDS (halo space is free after the last
      // tap's barrier) so that every thread handles whole 16-byte (voxel, 4-channel) cells.
      const Tile tl = decode(tile);
      float* tb = lds;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = (p.bias != nullptr) ? p.bias[ch0 + nt * 32 + li] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
          tb[(wid * 32 + i) * TS + nt * 32 + li] = acc[nt][r] + bv;
          acc[nt][r] = 0.0f;
        }
      }
      lds_barrier();
#pragma unroll
      for (int q = 0; q < NCELL; ++q) {
        const int u = tid + q * 256;
        const int vox = u / CPV, c4 = u % CPV;
        const int od = tl.od0 + (vox >> 5), oh = tl.oh0 + ((vox >> 3) & 3), ow = tl.ow0 + (vox & 7);
        if (od < p.OD && oh < p.OH && ow < p.OW) {
          const size_t vo = (((size_t)tl.n * p.OD + od) * p.OH + oh) * p.OW + ow;
          const float4 o = *reinterpret_cast<const float4*>(&tb[vox * TS + c4 * 4]);
          if (p.out != nullptr) *reinterpret_cast<float4*>(p.out + vo * p.C2 + ch0 + c4 * 4) = o;
          if (has_y) {
            const float d0 = o.x - ycur[q].x, d1 = o.y - ycur[q].y, d2 = o.z - ycur[q].z, d3 = o.w - ycur[q].w;
            const float s0 = d0 * d0, s1 = d1 * d1, s2 = d2 * d2, s3 = d3 * d3;
            const double sq = ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
            l0 += sq;
            if (p.att != nullptr) {
              const float av = p.att[vo];
              l1 += ((double)(av * s0) + (double)(av * s1)) + ((double)(av * s2) + (double)(av * s3));
            } else {
              l1 += sq;
            }
          }
        }
      }
    }
    tile = tile_n;
    slab = slab_n;
  }
  if (has_y) {
    double v[2] = {l0, l1};
    grid_sum_finish<2>(v, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.y * gridDim.x + blockIdx.x,
                       gridDim.x * gridDim.y);
  }
}

struct ConvPlan {
  ConvParams p;
  int nt;
  bool fast;
  dim3 grid;
  size_t lds_bytes;
  size_t wp_floats;
  size_t nblk;
  int T;
};

static int make_plan(const effq_geom* g, ConvPlan* pl) {
  EFFQ_CHECK_ARG(g != nullptr);
  EFFQ_CHECK_ARG(g->N > 0 && g->C1 > 0 && g->C2 > 0 && g->D > 0 && g->H > 0 && g->W > 0);
  EFFQ_CHECK_ARG(g->KD >= 1 && g->KH >= 1 && g->KW >= 1 && g->KD <= 7 && g->KH <= 7 && g->KW <= 7);
  EFFQ_CHECK_ARG(g->SD >= 1 && g->SH >= 1 && g->SW >= 1 && g->PD >= 0 && g->PH >= 0 && g->PW >= 0);
  ConvParams& p = pl->p;
  memset(&p, 0, sizeof(p));
  p.N = g->N; p.C1 = g->C1; p.C2 = g->C2; p.D = g->D; p.H = g->H; p.W = g->W;
  p.KD = g->KD; p.KH = g->KH; p.KW = g->KW; p.SD = g->SD; p.SH = g->SH; p.SW = g->SW;
  p.PD = g->PD; p.PH = g->PH; p.PW = g->PW;
  p.OD = (g->D + 2 * g->PD - g->KD) / g->SD + 1;
  p.OH = (g->H + 2 * g->PH - g->KH) / g->SH + 1;
  p.OW = (g->W + 2 * g->PW - g->KW) / g->SW + 1;
  EFFQ_CHECK_ARG(p.OD > 0 && p.OH > 0 && p.OW > 0);
  p.c1p = (g->C1 + 7) / 8 * 8;
  p.c2p = (g->C2 + 31) / 32 * 32;
  p.HD = (TD - 1) * p.SD + p.KD;
  p.HH = (TH - 1) * p.SH + p.KH;
  p.HW = (TW - 1) * p.SW + p.KW;
  p.nhalo = p.HD * p.HH * p.HW;
  // channel slab: largest of 32/16/8 that divides c1p and keeps the halo tile within 64 KiB of LDS
  // (two workgroups per CU); fall back to anything that fits the 160 KiB of one CU.
  int cslab = 0;
  const int cands[3] = {32, 16, 8};
  for (int pass = 0; pass < 2 && cslab == 0; ++pass)
    for (int i = 0; i < 3; ++i) {
      const int c = cands[i];
      if (c > p.c1p || p.c1p % c != 0) continue;
      const size_t bytes = (size_t)p.nhalo * (c + 4) * sizeof(float);
      if ((size_t)p.nhalo * (c / 4) > (size_t)G_MAXH * 256) continue;   // register-staged halo loads per thread
      if (bytes <= (pass == 0 ? (size_t)64 * 1024 : (size_t)150 * 1024)) {
        cslab = c;
        break;
      }
    }
  if (cslab == 0) {
    set_error("conv3d: halo tile of %d voxels does not fit LDS", p.nhalo);
    return EFFQ_ERR_ARG;
  }
  p.cslab = cslab;
  p.nslab = p.c1p / cslab;
  p.CS = cslab + 4;
  pl->lds_bytes = (size_t)p.nhalo * p.CS * sizeof(float);
  p.tiles_d = (p.OD + TD - 1) / TD;
  p.tiles_h = (p.OH + TH - 1) / TH;
  p.tiles_w = (p.OW + TW - 1) / TW;
  const long long nt_ll = (long long)p.N * p.tiles_d * p.tiles_h * p.tiles_w;
  EFFQ_CHECK_ARG(nt_ll < (1ll << 30));
  p.ntiles = (int)nt_ll;
  const int nsub = p.c2p / 32;
  int nt = 4;
  while (nsub % nt != 0) nt >>= 1;
  // keep the chip busy on small (deep) layers: prefer more workgroups over wider waves
  while (nt > 1 && (long long)p.ntiles * (nsub / nt) < 1024) nt >>= 1;
  // 3x3x3 / stride 1 / whole 32-channel slabs of 16-byte-aligned channel vectors: specialised kernel
  pl->fast = (p.KD == 3 && p.KH == 3 && p.KW == 3 && p.SD == 1 && p.SH == 1 && p.SW == 1 && cslab == 32 &&
              (p.C1 % 32) == 0 && (p.C2 % 32) == 0);
  if (nt > 2) nt = 2;      // register budget of the persistent kernels (prefetch registers + accumulators)
  if (pl->fast) {
    pl->lds_bytes = (size_t)F_NH * F_CS * sizeof(float) + (size_t)2 * 8 * 32 * nt * sizeof(float4);
  }
  pl->nt = nt;
  pl->grid = dim3((unsigned)p.ntiles, (unsigned)(nsub / nt), 1);
  {
    // persistent: about two workgroups per CU walk contiguous runs of tiles
    const int ny = nsub / nt;
    int gx = (512 + ny - 1) / ny;
    if (gx < 64) gx = 64;
    if (gx > p.ntiles) gx = p.ntiles;
    pl->grid = dim3((unsigned)gx, (unsigned)ny, 1);
  }
  pl->nblk = (size_t)pl->grid.x * pl->grid.y;
  pl->T = p.KD * p.KH * p.KW;
  pl->wp_floats = (size_t)pl->T * p.c1p * p.c2p;
  return EFFQ_OK;
}

constexpr size_t DIRECT_MAX_BLOCKS = 2048;   // partial-sum slots kept for the direct kernels (conv3d_direct.hip)
static size_t conv_partial_slots(const ConvPlan& pl) { return pl.nblk > DIRECT_MAX_BLOCKS ? pl.nblk : DIRECT_MAX_BLOCKS; }
static size_t conv_ws_bytes(const ConvPlan& pl) {
  return 256 + conv_partial_slots(pl) * 2 * sizeof(double) + pl.wp_floats * sizeof(float) + 256;
}

}  // namespace effq

using namespace effq;

extern "C" {

size_t effq_conv_ws_bytes(const effq_geom* g) {
  ConvPlan pl;
  if (make_plan(g, &pl) != EFFQ_OK) return 0;
  return conv_ws_bytes(pl);
}

int conv3d_quant_calib_step(const float* xq_ndhwc, const float* G, const float* bias, const float* y_fp,
                            const float* att, const effq_geom* g, const float* act_alpha_dev, int act_levels,
                            double* sqerr_out, float* out, void* ws, size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG(xq_ndhwc && G && g && ws);
  EFFQ_CHECK_ARG(y_fp != nullptr || out != nullptr);
  EFFQ_CHECK_ARG(y_fp == nullptr || sqerr_out != nullptr);
  EFFQ_CHECK_ARG(act_alpha_dev == nullptr || act_levels >= 2);
  ConvPlan pl;
  int rc = make_plan(g, &pl);
  if (rc != EFFQ_OK) return rc;
  if (ws_bytes < conv_ws_bytes(pl)) {
    set_error("conv3d: workspace %zu < required %zu", ws_bytes, conv_ws_bytes(pl));
    return EFFQ_ERR_WORKSPACE;
  }
  char* base = reinterpret_cast<char*>(ws);
  ConvParams& p = pl.p;
  p.ticket = reinterpret_cast<unsigned int*>(base);
  p.partials = reinterpret_cast<double*>(base + 256);
  float* wp = reinterpret_cast<float*>(base + 256 + conv_partial_slots(pl) * 2 * sizeof(double));
  // 16-byte alignment of the packed weights
  wp = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(wp) + 15) & ~(uintptr_t)15);
  p.x = xq_ndhwc;
  p.wp = wp;
  p.bias = bias;
  p.y = y_fp;
  p.att = att;
  p.out = out;
  p.sqerr = sqerr_out;
  p.act_on = act_alpha_dev != nullptr;
  p.act_alpha = act_alpha_dev;
  p.act_d = p.act_on ? (float)(1.0 / (double)(act_levels - 1)) : 1.0f;
  p.debug = effq_ablate_env("EFFQ_CONV_DEBUG");
  hipStream_t st = as_stream(stream);
  // (the ticket of the last-block reduction is left at zero by the kernel that used it: the caller zero-fills
  //  the workspace once, effq_hip.h)
  // loss-only calls of the short-K layers go to the direct-gather kernels
  const int dkind = (out == nullptr && att == nullptr && !p.act_on && y_fp != nullptr && effq_ablate_env("EFFQ_NO_DIRECT") == 0)
                        ? conv_direct_kind(g) : 0;
  if (dkind != 0) {
    DirectParams dp;
    memset(&dp, 0, sizeof(dp));
    dp.x = xq_ndhwc; dp.G = G; dp.bias = bias; dp.y = y_fp;
    dp.N = p.N; dp.C1 = p.C1; dp.C2 = p.C2; dp.D = p.D; dp.H = p.H; dp.W = p.W;
    dp.OD = p.OD; dp.OH = p.OH; dp.OW = p.OW; dp.SD = p.SD; dp.SH = p.SH; dp.SW = p.SW;
    dp.PD = p.PD; dp.PH = p.PH; dp.PW = p.PW;
    dp.V = (long long)p.N * p.OD * p.OH * p.OW;
    dp.partials = p.partials; dp.ticket = p.ticket; dp.sqerr = sqerr_out;
    rc = conv_direct_launch(dkind, dp, DIRECT_MAX_BLOCKS, st);
    if (rc == EFFQ_OK) {
      EFFQ_LAUNCH_CHECK();
      return EFFQ_OK;
    }
  }
  {
    size_t nb = (pl.wp_floats + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)nb), dim3(256), 0, st, G, wp, p.C1, p.C2, pl.T, p.c1p, p.c2p);
    EFFQ_LAUNCH_CHECK();
  }
  const size_t lds = pl.lds_bytes;
  const bool has_y = y_fp != nullptr, vec = (p.C1 & 3) == 0;
#define EFFQ_LAUNCH_K(KERN)                                                                                    \
  do {                                                                                                         \
    if (lds > 64 * 1024)                                                                                       \
      EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)lds));                                                                 \
    hipLaunchKernelGGL(KERN, pl.grid, dim3(256), lds, st, p);                                                  \
  } while (0)
  if (pl.fast) {
    if (pl.nt == 2) {
      if (has_y) EFFQ_LAUNCH_K((k_conv3d_k3<2, true>)); else EFFQ_LAUNCH_K((k_conv3d_k3<2, false>));
    } else {
      if (has_y) EFFQ_LAUNCH_K((k_conv3d_k3<1, true>)); else EFFQ_LAUNCH_K((k_conv3d_k3<1, false>));
    }
  } else if (pl.nt == 2) {
    if (vec) {
      if (has_y) EFFQ_LAUNCH_K((k_conv3d<2, true, true>)); else EFFQ_LAUNCH_K((k_conv3d<2, true, false>));
    } else {
      if (has_y) EFFQ_LAUNCH_K((k_conv3d<2, false, true>)); else EFFQ_LAUNCH_K((k_conv3d<2, false, false>));
    }
  } else {
    if (vec) {
      if (has_y) EFFQ_LAUNCH_K((k_conv3d<1, true, true>)); else EFFQ_LAUNCH_K((k_conv3d<1, true, false>));
    } else {
      if (has_y) EFFQ_LAUNCH_K((k_conv3d<1, false, true>)); else EFFQ_LAUNCH_K((k_conv3d<1, false, false>));
    }
  }
#undef EFFQ_LAUNCH_K
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
