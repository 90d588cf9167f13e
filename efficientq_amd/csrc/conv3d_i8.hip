// conv3d_calib_step_i8: the "int-simulated" fake-quant forward of one ADMM iteration, exact.
// Reference: EfficientQConv.py:118-122 (conv3d(Qactivation, G, b*) + mse_loss, 200x per layer).
//
// With quantised activations x = alpha_a * k/(La-1) (k = level id >= 0) and projected weights
// G = alpha_w * j'/(Lw-1) (j' = 2*level - (Lw-1)), the conv is  s * sum k*j'  with small integers, so it
// runs on the i8 matrix cores with exact int32 accumulation (v_mfma_i32_32x32x32_i8: K = 32 channels
// per instruction, 32x the f32 MFMA rate) and one fp32 scale + bias in the epilogue.  The kernel is
// then HBM-bound: per output voxel it streams C2*4 bytes of target and C1 bytes of level ids.
//
// Mapping: persistent workgroups (4 waves) walk contiguous runs of 4x4x8 output tiles; wave w owns
// d-plane w (32 voxels) x 32 output channels.  The wave's B operands -- all 27 taps x C1/32 channel
// groups of its 32 output channels -- live in REGISTERS for the whole kernel (27*CG*4 VGPRs), the
// level-id halo tile (6x6x10 voxels x C1 bytes) is staged through LDS with a 16-byte pad per voxel,
// and the next tile's halo + targets are prefetched into registers under the current tile's MFMAs.
#include <stdlib.h>
#include "common.h"

namespace effq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int ITD = 4, ITH = 4, ITW = 8;
constexpr int I_HD = ITD + 2, I_HH = ITH + 2, I_HW = ITW + 2, I_NH = I_HD * I_HH * I_HW;   // 360 voxels
constexpr int I_TS = 36;   // transpose row stride (floats)
// Halo tile in LDS: voxel (row, col) of a row of I_HW voxels sits at row * (I_HW * VS + pad) + col * VS.  A wave reads
// A fragments with lane -> (h = lane >> 3 & 3, w = lane & 7); ds_read_b128 is served in the lane groups {0-3, 12-15,
// 20-27} / {4-11, 16-19, 28-31} (MI355X_MICROARCH.md, LDS): with rows packed back to back (pad 0) three lanes of a
// group shared a bank quad at every channel count (SQ_LDS_BANK_CONFLICT = 56 % of SQ_LDS_IDX_ACTIVE on k_conv3d_i8l2),
// the pads below spread the 16 lanes of each group over all 64 banks.
__host__ __device__ constexpr int halo_row_pad(int cg) { return cg == 1 ? 160 : cg == 2 ? 96 : 224; }

struct ConvI8Params {
  const int8_t* x;
  const int8_t* wq;
  const float* bias;
  const float* y;
  const float* act_alpha;
  const effq_fp_state* wstate;
  double inv_levels;
  int N, C1, C2, c2p, D, H, W, OD, OH, OW, PD, PH, PW;
  int tiles_d, tiles_h, tiles_w, ntiles;
  double* partials;
  unsigned int* ticket;
  double* sqerr;
  int debug;   // profiling ablations (EFFQ_I8_DEBUG): 1 no MFMA loop, 2 no halo loads, 3 no target loads
  // conv3d_quant_forward_i8 (k_conv3d_i8l2e<true>, k_conv3d_i8w<true>): the conv output is also STORED (fp32, NDHWC) and
  // sqerr[1] is the attention-weighted sum (att: one weight per output voxel, or NULL = the plain sum once more)
  float* out;
  const float* att;
};

__global__ __launch_bounds__(256) void k_pack_weight_i8(const int8_t* __restrict__ Gq, int8_t* __restrict__ wq, int C1,
                                                        int C2, int T, int c2p) {
  const size_t total = (size_t)T * c2p * C1;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int c = (int)(e % C1);
    size_t r = e / C1;
    const int j = (int)(r % c2p);
    const int tap = (int)(r / c2p);
    wq[e] = (j < C2) ? Gq[((size_t)j * C1 + c) * T + tap] : (int8_t)0;
  }
}

// Packed layout for the wide-channel kernel: [tap][group g][k-half h][c2p][16 bytes], so that the B operand of one
// MFMA is a 512-byte contiguous run per lane half (coalesced 16-byte loads straight from L2).
template <int RPW>      // output channels per workgroup: 4 = 64-byte store runs, 1 = four times as many workgroups
__global__ __launch_bounds__(256) void k_pack_weight_i8g(const int8_t* __restrict__ Gq, int8_t* __restrict__ wq, int C1,
                                                         int C2, int T, int c2p) {
  // One workgroup = RPW output channels: their rows of Gq ([C1][T] bytes each, contiguous) are read coalesced into LDS,
  // then every thread assembles 16-byte cells (tap, g, h, j) from bytes T apart in LDS and stores them - with RPW = 4 the
  // 4 channels of a (tap, g, h) are 64 contiguous bytes.  (One thread per cell gathering its 16 bytes from global memory,
  // T bytes apart, took 43-69 us per call in situ for 1.77 MB of weights.)
  extern __shared__ __attribute__((aligned(16))) int8_t rows[];          // [RPW][C1 * T]
  const int rowb = C1 * T;
  const int j0 = blockIdx.x * RPW;
  const int rowv = rowb / 16;                      // C1 % 16 == 0: rows are whole 16-byte vectors
  for (int e = threadIdx.x; e < RPW * rowv; e += 256) {
    const int jj = e / rowv, o = e - jj * rowv;
    v4i val = {0, 0, 0, 0};
    if (j0 + jj < C2) val = *reinterpret_cast<const v4i*>(Gq + (size_t)(j0 + jj) * rowb + (size_t)o * 16);
    *reinterpret_cast<v4i*>(rows + (size_t)jj * rowb + (size_t)o * 16) = val;
  }
  __syncthreads();
  const int ncell = T * (C1 / 16);                                       // cells per output channel
  for (int u = threadIdx.x; u < ncell * RPW; u += 256) {
    const int jj = u % RPW, q = u / RPW;
    const int gh = q % (C1 / 16), tap = q / (C1 / 16);
    if (j0 + jj >= c2p) continue;
    const int8_t* src = rows + jj * rowb + (16 * gh) * T + tap;
    v4i out;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned wv = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) wv |= (unsigned)(unsigned char)src[(4 * w + b) * T] << (8 * b);
      out[w] = (int)wv;
    }
    *reinterpret_cast<v4i*>(wq + ((size_t)(tap * (C1 / 16) + gh) * c2p + j0 + jj) * 16) = out;
  }
}

// Wide-channel variant (C1 = 128 / 256): the B operands no longer fit in registers; they stream from L2
// one (tap, group) step ahead of the MFMA that consumes them.  Everything else as in k_conv3d_i8.
template <int CG>
__global__ __launch_bounds__(256, (CG <= 4) ? 2 : 1) void k_conv3d_i8g(ConvI8Params p) {
  constexpr int VS = 32 * CG + 16;
  constexpr int PADB = halo_row_pad(CG), HALOB = I_NH * VS + I_HD * I_HH * PADB;
  constexpr int NHL = (I_NH * 2 * CG + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) int8_t dyn_lds[];
  int8_t* halo = dyn_lds;                                       // HALOB bytes
  float* tb = reinterpret_cast<float*>(dyn_lds + ((HALOB + 15) / 16) * 16);   // 128 * I_TS floats
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ch0 = blockIdx.y * 32;
  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = (t_begin + per < p.ntiles) ? t_begin + per : p.ntiles;
  const float scale = (float)((double)(*p.act_alpha) * (double)(float)p.wstate->alpha * p.inv_levels);
  const float bv = (p.bias != nullptr) ? p.bias[ch0 + li] : 0.0f;
  // B operand of step (tap, g): wq[((tap*CG + g)*2 + lh) * c2p + ch0 + li] (16-byte units)
  const v4i* wbase = reinterpret_cast<const v4i*>(p.wq) + (size_t)lh * p.c2p + ch0 + li;
  const size_t wstep = (size_t)2 * p.c2p;

  struct Tile {
    int n, od0, oh0, ow0;
  };
  auto decode = [&](int tile) {
    Tile r;
    int t = tile;
    r.ow0 = (t % p.tiles_w) * ITW;
    t /= p.tiles_w;
    r.oh0 = (t % p.tiles_h) * ITH;
    t /= p.tiles_h;
    r.od0 = (t % p.tiles_d) * ITD;
    r.n = t / p.tiles_d;
    return r;
  };
  auto load_halo = [&](const Tile& tl, v4i(&hreg)[NHL]) {
    const int id0 = tl.od0 - p.PD, ih0 = tl.oh0 - p.PH, iw0 = tl.ow0 - p.PW;
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int u = tid + k * 256;
      const int vox = u / (2 * CG), part = u % (2 * CG);
      const int hw = vox % I_HW;
      const int t2 = vox / I_HW;
      const int hh = t2 % I_HH, hd = t2 / I_HH;
      const int id = id0 + hd, ih = ih0 + hh, iw = iw0 + hw;
      hreg[k] = v4i{0, 0, 0, 0};
      if (u < I_NH * 2 * CG && id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
        hreg[k] = *reinterpret_cast<const v4i*>(p.x + ((((size_t)tl.n * p.D + id) * p.H + ih) * p.W + iw) * p.C1 +
                                                part * 16);
    }
  };
  auto load_y = [&](const Tile& tl, float4(&yv)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = tid + k * 256;
      const int vox = u >> 3;
      const int od = tl.od0 + (vox >> 5), oh = tl.oh0 + ((vox >> 3) & 3), ow = tl.ow0 + (vox & 7);
      yv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (od < p.OD && oh < p.OH && ow < p.OW)
        yv[k] = *reinterpret_cast<const float4*>(p.y + ((((size_t)tl.n * p.OD + od) * p.OH + oh) * p.OW + ow) * p.C2 +
                                                 ch0 + (u & 7) * 4);
    }
  };

  const int hv = (wid * I_HH + (li >> 3)) * I_HW + (li & 7);
  double l0 = 0.0;
  v4i hreg[NHL];
  float4 ynext[4], ycur[4];
  if (t_begin < t_end) {
    const Tile t0 = decode(t_begin);
    load_halo(t0, hreg);
    load_y(t0, ynext);
  }
  for (int tile = t_begin; tile < t_end; ++tile) {
    lds_barrier();
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int u = tid + k * 256;
      if (u < I_NH * 2 * CG)
        *reinterpret_cast<v4i*>(&halo[(u / (2 * CG)) * VS + ((u / (2 * CG)) / I_HW) * PADB + (u % (2 * CG)) * 16]) = hreg[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) ycur[k] = ynext[k];
    lds_barrier();
    if (tile + 1 < t_end) {
      const Tile tn = decode(tile + 1);
      load_halo(tn, hreg);
      load_y(tn, ynext);
    }
    v16i acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;
    // B operands stream from L2 through a ring of 8 registers = TPA taps: the loads of tap t + TPA are issued right
    // after the MFMAs of tap t (an L2 hit takes ~600 cycles, an MFMA 32: one step of look-ahead left the loop
    // latency-bound).  Loads are unconditional on a clamped step index (a load under a branch is waited for in it).
    constexpr int TPA = 8 / CG, NSTEP = 27 * CG;
    v4i bq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bq[j] = wbase[(size_t)j * wstep];
    auto tap_body = [&](int t, int tj, bool prefetch) {
      const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
      const int8_t* arow = halo + (hv + (kd * I_HH + kh) * I_HW + kw) * VS +
                           (wid * I_HH + (li >> 3) + kd * I_HH + kh) * PADB + 16 * lh;
#pragma unroll
      for (int g = 0; g < CG; ++g) {
        const v4i a = *reinterpret_cast<const v4i*>(arow + 32 * g);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[tj * CG + g], acc, 0, 0, 0);
        if (prefetch) {
          int nxt = (t + TPA) * CG + g;
          nxt = (nxt < NSTEP) ? nxt : NSTEP - 1;
          bq[tj * CG + g] = wbase[(size_t)nxt * wstep];
        }
      }
    };
    int tap = 0;
#pragma unroll 1
    for (; tap + TPA <= 27 - (27 % TPA == 0 ? TPA : 27 % TPA); tap += TPA) {
#pragma unroll
      for (int tj = 0; tj < TPA; ++tj) tap_body(tap + tj, tj, true);
    }
    // the last group(s): their operands are already in the ring
#pragma unroll 1
    for (; tap + TPA <= 27; tap += TPA) {
#pragma unroll
      for (int tj = 0; tj < TPA; ++tj) tap_body(tap + tj, tj, false);
    }
#pragma unroll
    for (int tj = 0; tj < 27 % TPA; ++tj) tap_body(tap + tj, tj, false);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      tb[(wid * 32 + i) * I_TS + li] = (float)acc[r] * scale + bv;
    }
    lds_barrier();
    const Tile tl = decode(tile);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = tid + k * 256;
      const int vox = u >> 3, c4 = u & 7;
      const int od = tl.od0 + (vox >> 5), oh = tl.oh0 + ((vox >> 3) & 3), ow = tl.ow0 + (vox & 7);
      if (od < p.OD && oh < p.OH && ow < p.OW) {
        const float4 o = *reinterpret_cast<const float4*>(&tb[vox * I_TS + c4 * 4]);
        const float d0 = o.x - ycur[k].x, d1 = o.y - ycur[k].y, d2 = o.z - ycur[k].z, d3 = o.w - ycur[k].w;
        l0 += ((double)(d0 * d0) + (double)(d1 * d1)) + ((double)(d2 * d2) + (double)(d3 * d3));
      }
    }
  }
  double v[2] = {l0, l0};
  grid_sum_finish<2>(v, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.y * gridDim.x + blockIdx.x,
                     gridDim.x * gridDim.y);
}

// 512 input channels: the 6 x 6 x 10 halo tile would need 190 KB of LDS, so the tile is 2 x 4 x 8 output voxels (halo
// 4 x 6 x 10 = 127 KB) and the four waves are 2 d-planes x 2 halves of 64 output channels (blockIdx.y counts groups of
// 64).  B operands stream through the same 8-deep register ring, one ring load per MFMA step (tap, group).
constexpr int G2_TD = 2, G2_HD = G2_TD + 2, G2_NH = G2_HD * I_HH * I_HW;    // 240 halo voxels
constexpr int G2_TS = 68;                                                    // transpose row stride (floats): 64 channels + pad
template <int CG>
__global__ __launch_bounds__(256, 1) void k_conv3d_i8g2(ConvI8Params p) {
  static_assert(CG % 8 == 0, "ring blocks of 8 steps must not straddle taps");
  constexpr int VS = 32 * CG + 16;
  constexpr int PADB = halo_row_pad(CG), HALOB = G2_NH * VS + G2_HD * I_HH * PADB;
  constexpr int NHL = (G2_NH * 2 * CG + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) int8_t dyn_lds[];
  int8_t* halo = dyn_lds;
  float* tb = reinterpret_cast<float*>(dyn_lds + ((HALOB + 15) / 16) * 16);   // 64 * G2_TS floats
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int plane = wid & 1, chalf = wid >> 1;
  const int ch0 = blockIdx.y * 64, chw = ch0 + 32 * chalf;
  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = (t_begin + per < p.ntiles) ? t_begin + per : p.ntiles;
  const float scale = (float)((double)(*p.act_alpha) * (double)(float)p.wstate->alpha * p.inv_levels);
  const float bv = (p.bias != nullptr) ? p.bias[chw + li] : 0.0f;
  const v4i* wbase = reinterpret_cast<const v4i*>(p.wq) + (size_t)lh * p.c2p + chw + li;
  const size_t wstep = (size_t)2 * p.c2p;

  struct Tile {
    int n, od0, oh0, ow0;
  };
  auto decode = [&](int tile) {
    Tile r;
    int t = tile;
    r.ow0 = (t % p.tiles_w) * ITW;
    t /= p.tiles_w;
    r.oh0 = (t % p.tiles_h) * ITH;
    t /= p.tiles_h;
    r.od0 = (t % p.tiles_d) * G2_TD;
    r.n = t / p.tiles_d;
    return r;
  };
  auto load_halo = [&](const Tile& tl, v4i(&hreg)[NHL]) {
    const int id0 = tl.od0 - p.PD, ih0 = tl.oh0 - p.PH, iw0 = tl.ow0 - p.PW;
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int u = tid + k * 256;
      const int vox = u / (2 * CG), part = u % (2 * CG);
      const int hw = vox % I_HW;
      const int t2 = vox / I_HW;
      const int hh = t2 % I_HH, hd = t2 / I_HH;
      const int id = id0 + hd, ih = ih0 + hh, iw = iw0 + hw;
      hreg[k] = v4i{0, 0, 0, 0};
      if (u < G2_NH * 2 * CG && id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
        hreg[k] = *reinterpret_cast<const v4i*>(p.x + ((((size_t)tl.n * p.D + id) * p.H + ih) * p.W + iw) * p.C1 +
                                                part * 16);
    }
  };
  // targets: 64 voxels x 64 channels = 1024 float4, 4 per thread: u -> voxel u >> 4, channel quad u & 15
  auto load_y = [&](const Tile& tl, float4(&yv)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = tid + k * 256;
      const int vox = u >> 4;
      const int od = tl.od0 + (vox >> 5), oh = tl.oh0 + ((vox >> 3) & 3), ow = tl.ow0 + (vox & 7);
      yv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (od < p.OD && oh < p.OH && ow < p.OW)
        yv[k] = *reinterpret_cast<const float4*>(p.y + ((((size_t)tl.n * p.OD + od) * p.OH + oh) * p.OW + ow) * p.C2 +
                                                 ch0 + (u & 15) * 4);
    }
  };

  const int hv = (plane * I_HH + (li >> 3)) * I_HW + (li & 7);
  double l0 = 0.0;
  v4i hreg[NHL];
  float4 ynext[4], ycur[4];
  if (t_begin < t_end) {
    const Tile t0 = decode(t_begin);
    load_halo(t0, hreg);
    load_y(t0, ynext);
  }
  for (int tile = t_begin; tile < t_end; ++tile) {
    lds_barrier();
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int u = tid + k * 256;
      if (u < G2_NH * 2 * CG)
        *reinterpret_cast<v4i*>(&halo[(u / (2 * CG)) * VS + ((u / (2 * CG)) / I_HW) * PADB + (u % (2 * CG)) * 16]) = hreg[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) ycur[k] = ynext[k];
    lds_barrier();
    if (tile + 1 < t_end) {
      const Tile tn = decode(tile + 1);
      load_halo(tn, hreg);
      load_y(tn, ynext);
    }
    v16i acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;
    constexpr int NSTEP = 27 * CG;
    v4i bq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bq[j] = wbase[(size_t)j * wstep];
#pragma unroll 1
    for (int s0 = 0; s0 < NSTEP; s0 += 8) {
      const int t = s0 / CG, g0 = s0 % CG;                  // 8 | CG: the tap is constant inside a block of 8 steps
      const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
      const int8_t* arow = halo + (hv + (kd * I_HH + kh) * I_HW + kw) * VS +
                           (plane * I_HH + (li >> 3) + kd * I_HH + kh) * PADB + 16 * lh + 32 * g0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const v4i a = *reinterpret_cast<const v4i*>(arow + 32 * j);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[j], acc, 0, 0, 0);
        int nxt = s0 + 8 + j;
        nxt = (nxt < NSTEP) ? nxt : NSTEP - 1;              // unconditional load on a clamped index
        bq[j] = wbase[(size_t)nxt * wstep];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      tb[(plane * 32 + i) * G2_TS + 32 * chalf + li] = (float)acc[r] * scale + bv;
    }
    lds_barrier();
    const Tile tl = decode(tile);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = tid + k * 256;
      const int vox = u >> 4, c4 = u & 15;
      const int od = tl.od0 + (vox >> 5), oh = tl.oh0 + ((vox >> 3) & 3), ow = tl.ow0 + (vox & 7);
      if (od < p.OD && oh < p.OH && ow < p.OW) {
        const float4 o = *reinterpret_cast<const float4*>(&tb[vox * G2_TS + c4 * 4]);
        const float d0 = o.x - ycur[k].x, d1 = o.y - ycur[k].y, d2 = o.z - ycur[k].z, d3 = o.w - ycur[k].w;
        l0 += ((double)(d0 * d0) + (double)(d1 * d1)) + ((double)(d2 * d2) + (double)(d3 * d3));
      }
    }
  }
  double v[2] = {l0, l0};
  grid_sum_finish<2>(v, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.y * gridDim.x + blockIdx.x,
                     gridDim.x * gridDim.y);
}

template <int CG>
__global__ __launch_bounds__(256, (CG == 1) ? 2 : 1) void k_conv3d_i8(ConvI8Params p) {
  constexpr int VS = 32 * CG + 16;                         // bytes per halo voxel in LDS
  constexpr int PADB = halo_row_pad(CG);
  constexpr int NHL = (I_NH * 2 * CG + 255) / 256;         // halo 16-byte loads per thread
  __shared__ __attribute__((aligned(16))) int8_t halo[I_NH * VS + I_HD * I_HH * PADB];
  __shared__ __attribute__((aligned(16))) float tb[128 * I_TS];
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ch0 = blockIdx.y * 32;
  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = (t_begin + per < p.ntiles) ? t_begin + per : p.ntiles;

  // B operands of this wave's 32 output channels: lane (j=li, k-half lh) holds 16 channels per (tap, group)
  v4i breg[27][CG];
  {
    const int8_t* wrow = p.wq + ((size_t)(ch0 + li) * p.C1 + 16 * lh);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
#pragma unroll
      for (int g = 0; g < CG; ++g)
        breg[tap][g] = *reinterpret_cast<const v4i*>(wrow + (size_t)tap * p.c2p * p.C1 + 32 * g);
  }
  const float scale = (float)((double)(*p.act_alpha) * (double)(float)p.wstate->alpha * p.inv_levels);
  const float bv = (p.bias != nullptr) ? p.bias[ch0 + li] : 0.0f;

  struct Tile {
    int n, od0, oh0, ow0;
  };
  auto decode = [&](int tile) {
    Tile r;
    int t = tile;
    r.ow0 = (t % p.tiles_w) * ITW;
    t /= p.tiles_w;
    r.oh0 = (t % p.tiles_h) * ITH;
    t /= p.tiles_h;
    r.od0 = (t % p.tiles_d) * ITD;
    r.n = t / p.tiles_d;
    return r;
  };
  // per-thread staging constants (independent of the tile); all tile loads are unconditional on clamped
  // coordinates with validity bits applied at the consumer (see k_conv3d_i8l2)
  int hcd[NHL], hch[NHL], hcw[NHL], hpart[NHL];
#pragma unroll
  for (int k = 0; k < NHL; ++k) {
    const int u = tid + k * 256;
    const bool live = u < I_NH * 2 * CG;
    const int vox = live ? u / (2 * CG) : 0;
    hpart[k] = live ? (u % (2 * CG)) * 16 : 0;
    hcw[k] = vox % I_HW;
    const int t2 = vox / I_HW;
    hch[k] = t2 % I_HH;
    hcd[k] = live ? (t2 / I_HH) : (1 << 20);
  }
  typedef float yv4 __attribute__((ext_vector_type(4)));
  auto load_halo = [&](const Tile& tl, v4i(&hreg)[NHL], unsigned& hmask) {
    const int id0 = tl.od0 - p.PD, ih0 = tl.oh0 - p.PH, iw0 = tl.ow0 - p.PW;
    const size_t nbase = (size_t)tl.n * p.D;
    unsigned hm = 0;
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int id = id0 + hcd[k], ih = ih0 + hch[k], iw = iw0 + hcw[k];
      const bool ok = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      const int cd = min(max(id, 0), p.D - 1), chh = min(max(ih, 0), p.H - 1), cw = min(max(iw, 0), p.W - 1);
      hm |= (ok ? 1u : 0u) << k;
      hreg[k] = *reinterpret_cast<const v4i*>(p.x + (((nbase + cd) * p.H + chh) * p.W + cw) * p.C1 + hpart[k]);
    }
    hmask = hm;
  };
  auto load_y = [&](const Tile& tl, yv4(&yv)[4], unsigned& ymask) {
    unsigned ym = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = tid + k * 256;
      const int vox = u >> 3;
      const int od = tl.od0 + (vox >> 5), oh = tl.oh0 + ((vox >> 3) & 3), ow = tl.ow0 + (vox & 7);
      ym |= (unsigned)(od < p.OD && oh < p.OH && ow < p.OW) << k;
      const int cd = min(od, p.OD - 1), chh = min(oh, p.OH - 1), cw = min(ow, p.OW - 1);
      yv[k] = *reinterpret_cast<const yv4*>(p.y + ((((size_t)tl.n * p.OD + cd) * p.OH + chh) * p.OW + cw) * p.C2 +
                                            ch0 + (u & 7) * 4);
    }
    ymask = ym;
  };

  const int hv = (wid * I_HH + (li >> 3)) * I_HW + (li & 7);
  double l0 = 0.0;
  v4i hreg[NHL];
  yv4 ynext[4], ycur[4];
  unsigned hmask = 0, ymask_next = 0, ymask_cur = 0;
#pragma unroll
  for (int k = 0; k < NHL; ++k) hreg[k] = v4i{0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) ynext[k] = yv4{0.f, 0.f, 0.f, 0.f};
  if (t_begin < t_end) {
    const Tile t0 = decode(t_begin);
    load_halo(t0, hreg, hmask);
    load_y(t0, ynext, ymask_next);
  }
  for (int tile = t_begin; tile < t_end; ++tile) {
    lds_barrier();                                       // halo free (previous tile's MFMAs done)
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int u = tid + k * 256;
      const v4i val = ((hmask >> k) & 1u) ? hreg[k] : v4i{0, 0, 0, 0};   // level id 0 == zero padding
      if (u < I_NH * 2 * CG)
        *reinterpret_cast<v4i*>(&halo[(u / (2 * CG)) * VS + ((u / (2 * CG)) / I_HW) * PADB + (u % (2 * CG)) * 16]) = val;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) ycur[k] = ynext[k];
    ymask_cur = ymask_next;
    lds_barrier();
    {
      const Tile tn = decode((tile + 1 < t_end) ? tile + 1 : tile);   // the last tile harmlessly re-reads itself
      load_halo(tn, hreg, hmask);
      load_y(tn, ynext, ymask_next);
    }
    __builtin_amdgcn_sched_barrier(0);
    v16i acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int8_t* arow = halo + (hv + (kd * I_HH + kh) * I_HW + kw) * VS +
                           (wid * I_HH + (li >> 3) + kd * I_HH + kh) * PADB + 16 * lh;
#pragma unroll
      for (int g = 0; g < CG; ++g) {
        const v4i a = *reinterpret_cast<const v4i*>(arow + 32 * g);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, breg[tap][g], acc, 0, 0, 0);
      }
    }
    // epilogue: scale + bias, transpose through LDS, compare whole 16-byte cells with the prefetched targets
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      tb[(wid * 32 + i) * I_TS + li] = (float)acc[r] * scale + bv;
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = tid + k * 256;
      const int vox = u >> 3, c4 = u & 7;
      const yv4 o = *reinterpret_cast<const yv4*>(&tb[vox * I_TS + c4 * 4]);
      const float d0 = o[0] - ycur[k][0], d1 = o[1] - ycur[k][1], d2 = o[2] - ycur[k][2], d3 = o[3] - ycur[k][3];
      const double q = ((double)(d0 * d0) + (double)(d1 * d1)) + ((double)(d2 * d2) + (double)(d3 * d3));
      l0 += ((ymask_cur >> k) & 1u) ? q : 0.0;
    }
  }
  double v[2] = {l0, l0};
  grid_sum_finish<2>(v, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.y * gridDim.x + blockIdx.x,
                     gridDim.x * gridDim.y);
}


// 32-input-channel kernel, tuned for memory-level parallelism (it is HBM-latency bound): B operands (27 KB) live in LDS,
// laid out [tap][k-half][out channel][16 B] (conflict-free ds_read_b128); two register sets alternate roles so that halo
// and targets are fetched TWO tiles ahead; targets are fetched in accumulator layout (16 dword loads per lane, 128-byte
// segments), so the epilogue needs no LDS transpose.  A workgroup tile is 8x4x8 output voxels, wave w owns d-planes w and
// w+4 and feeds BOTH accumulators from one read of the B operand (a one-plane tile was bound by its LDS operand traffic,
// 27 x 2 KB per wave-tile); barriers, tile decoding and halo overlap are amortised over twice the voxels.  2 workgroups
// per CU.  This variant takes ragged volumes and any C2 % 32 == 0; k_conv3d_i8l2e is its fast path.
constexpr int L2_TD = 8;
constexpr int L2_HD = L2_TD + 2, L2_NH = L2_HD * I_HH * I_HW;    // 600 halo voxels
__global__ __launch_bounds__(256, 2) void k_conv3d_i8l2(ConvI8Params p) {
  constexpr int VS = 48, PADB = halo_row_pad(1);
  constexpr int NHL = (L2_NH * 2 + 255) / 256;   // 5
  __shared__ __attribute__((aligned(16))) int8_t wl[27 * 2 * 32 * 16];
  __shared__ __attribute__((aligned(16))) int8_t halo[L2_NH * VS + L2_HD * I_HH * PADB];
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ch0 = blockIdx.y * 32;
  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = (t_begin + per < p.ntiles) ? t_begin + per : p.ntiles;

  for (int u = tid; u < 27 * 2 * 32; u += 256) {
    const int j = u & 31, h = (u >> 5) & 1, tap = u >> 6;
    *reinterpret_cast<v4i*>(&wl[u * 16]) =
        *reinterpret_cast<const v4i*>(p.wq + ((size_t)(tap * p.c2p + ch0 + j) * 32 + 16 * h));
  }
  const float scale = (float)((double)(*p.act_alpha) * (double)(float)p.wstate->alpha * p.inv_levels);
  const float bv = (p.bias != nullptr) ? p.bias[ch0 + li] : 0.0f;

  struct Tile {
    int n, od0, oh0, ow0;
  };
  auto decode = [&](int tile) {           // p.tiles_d counts 8-plane tiles for this kernel
    Tile r;
    int t = tile;
    r.ow0 = (t % p.tiles_w) * ITW;
    t /= p.tiles_w;
    r.oh0 = (t % p.tiles_h) * ITH;
    t /= p.tiles_h;
    r.od0 = (t % p.tiles_d) * L2_TD;
    r.n = t / p.tiles_d;
    return r;
  };
  int hcd[NHL], hch[NHL], hcw[NHL];
#pragma unroll
  for (int k = 0; k < NHL; ++k) {
    const int u = tid + k * 256;
    const int vox = (u < L2_NH * 2) ? (u >> 1) : 0;
    hcw[k] = vox % I_HW;
    const int t2 = vox / I_HW;
    hch[k] = t2 % I_HH;
    hcd[k] = (u < L2_NH * 2) ? (t2 / I_HH) : (1 << 20);
  }
  const int part16 = (tid & 1) * 16;

  struct Regs {
    v4i h[NHL];
    float y[2][16];
    unsigned hmask, ymask[2];
  };
  auto fetch = [&](int tile, Regs& R) {
    const Tile tl = decode(tile);
    const int id0 = tl.od0 - p.PD, ih0 = tl.oh0 - p.PH, iw0 = tl.ow0 - p.PW;
    const size_t nbase = (size_t)tl.n * p.D;
    unsigned hm = 0;
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int id = id0 + hcd[k], ih = ih0 + hch[k], iw = iw0 + hcw[k];
      const bool ok = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      const int cd = min(max(id, 0), p.D - 1), chh = min(max(ih, 0), p.H - 1), cw = min(max(iw, 0), p.W - 1);
      hm |= (ok ? 1u : 0u) << k;
      R.h[k] = *reinterpret_cast<const v4i*>(p.x + (((nbase + cd) * p.H + chh) * p.W + cw) * 32 + part16);
    }
    R.hmask = hm;
    unsigned rowok = 0, colok = 0;
    size_t rowrel[4];
    int coloff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oh = tl.oh0 + q, ow = tl.ow0 + q + 4 * lh;
      rowok |= (unsigned)(oh < p.OH) << q;
      colok |= (unsigned)(ow < p.OW) << q;
      rowrel[q] = (size_t)min(oh, p.OH - 1) * p.OW * p.C2;
      coloff[q] = min(ow, p.OW - 1) * p.C2 + ch0 + li;
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int odr = tl.od0 + wid + 4 * pl;
      const int od = min(odr, p.OD - 1);
      const bool dok = odr < p.OD;
      const size_t ybase = (((size_t)tl.n * p.OD + od) * p.OH) * p.OW * p.C2;
      unsigned ym = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        R.y[pl][r] = p.y[ybase + rowrel[r >> 2] + coloff[r & 3]];
        ym |= (((rowok >> (r >> 2)) & (colok >> (r & 3)) & 1u) & (dok ? 1u : 0u)) << r;
      }
      R.ymask[pl] = ym;
    }
  };

  const int hv0 = (wid * I_HH + (li >> 3)) * I_HW + (li & 7);
  const int hv1 = hv0 + 4 * I_HH * I_HW;
  const int hb0 = hv0 * VS + (wid * I_HH + (li >> 3)) * PADB, hb1 = hv1 * VS + ((wid + 4) * I_HH + (li >> 3)) * PADB;
  double l0 = 0.0;
  auto body = [&](int tile, Regs& X, bool more) {
    lds_barrier();
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int u = tid + k * 256;
      const v4i val = ((X.hmask >> k) & 1u) ? X.h[k] : v4i{0, 0, 0, 0};
      if (u < L2_NH * 2) *reinterpret_cast<v4i*>(&halo[(u >> 1) * VS + ((u >> 1) / I_HW) * PADB + (u & 1) * 16]) = val;
    }
    float ycur[2][16];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int r = 0; r < 16; ++r) ycur[pl][r] = X.y[pl][r];
    const unsigned ym0 = X.ymask[0], ym1 = X.ymask[1];
    lds_barrier();
    fetch(more ? tile + 2 : tile, X);
    __builtin_amdgcn_sched_barrier(0);
    v16i acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int toff = ((kd * I_HH + kh) * I_HW + kw) * VS + (kd * I_HH + kh) * PADB + 16 * lh;
      const v4i b = *reinterpret_cast<const v4i*>(&wl[((tap * 2 + lh) * 32 + li) * 16]);
      const v4i a0 = *reinterpret_cast<const v4i*>(halo + hb0 + toff);
      const v4i a1 = *reinterpret_cast<const v4i*>(halo + hb1 + toff);
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d0 = ((float)acc0[r] * scale + bv) - ycur[0][r];
      const float d1 = ((float)acc1[r] * scale + bv) - ycur[1][r];
      l0 += ((ym0 >> r) & 1u) ? (double)(d0 * d0) : 0.0;
      l0 += ((ym1 >> r) & 1u) ? (double)(d1 * d1) : 0.0;
    }
  };

  Regs A, B;
#pragma unroll
  for (int k = 0; k < NHL; ++k) A.h[k] = B.h[k] = v4i{0, 0, 0, 0};
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int r = 0; r < 16; ++r) A.y[pl][r] = B.y[pl][r] = 0.0f;
  A.hmask = B.hmask = 0;
  A.ymask[0] = A.ymask[1] = B.ymask[0] = B.ymask[1] = 0;
  if (t_begin < t_end) fetch(t_begin, A);
  if (t_begin + 1 < t_end) fetch(t_begin + 1, B);
  for (int tile = t_begin; tile < t_end; tile += 2) {
    body(tile, A, tile + 2 < t_end);
    if (tile + 1 < t_end) body(tile + 1, B, tile + 3 < t_end);
  }
  double v[2] = {l0, l0};
  grid_sum_finish<2>(v, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.y * gridDim.x + blockIdx.x,
                     gridDim.x * gridDim.y);
}

// ---- k_conv3d_i8l2 for volumes the 8 x 4 x 8 tile divides (every shape of the shipped recipes) ---------------------------
// Same tiling and arithmetic as k_conv3d_i8l2, with the per-tile VALU work cut from ~790 to ~300 instructions per wave
// (PMC: 54 MFMAs against 787 VALU instructions per wave-tile made the kernel VALU-issue bound, profiles/r02_pmc_i8l2):
//  * targets are always in range: no validity masks, addresses = one base per tile + per-lane offsets fixed at start;
//  * halo: per-lane relative offsets fixed at start; tiles that touch no volume face (uniform test) skip clamps and masks;
//  * squared errors are accumulated in fp32 within a tile (4 chains of 8 terms per plane) and added to the fp64 sum once
//    per tile, instead of one fp64 conversion + addition per output value.
template <bool OUT>
__global__ __launch_bounds__(256, 2) void k_conv3d_i8l2e(ConvI8Params p) {
  constexpr int VS = 48, PADB = halo_row_pad(1);
  constexpr int NHL = (L2_NH * 2 + 255) / 256;   // 5
  __shared__ __attribute__((aligned(16))) int8_t wl[27 * 2 * 32 * 16];
  __shared__ __attribute__((aligned(16))) int8_t halo[L2_NH * VS + L2_HD * I_HH * PADB];
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = (t_begin + per < p.ntiles) ? t_begin + per : p.ntiles;

  for (int u = tid; u < 27 * 2 * 32; u += 256) {
    const int j = u & 31, h = (u >> 5) & 1, tap = u >> 6;
    *reinterpret_cast<v4i*>(&wl[u * 16]) = *reinterpret_cast<const v4i*>(p.wq + ((size_t)(tap * p.c2p + j) * 32 + 16 * h));
  }
  lds_barrier();
  const float scale = (float)((double)(*p.act_alpha) * (double)(float)p.wstate->alpha * p.inv_levels);
  const float bv = (p.bias != nullptr) ? p.bias[li] : 0.0f;

  // per-lane constants: halo voxel of each of the 5 loads (relative coordinates, LDS offset, global offset relative to
  // the tile's first halo voxel), target offsets of the 2 x 16 outputs relative to the tile's first output voxel
  int hcd[NHL], hch[NHL], hcw[NHL], hlds[NHL];
  long long hrel[NHL];
#pragma unroll
  for (int k = 0; k < NHL; ++k) {
    const int u = tid + k * 256;
    const int vox = (u < L2_NH * 2) ? (u >> 1) : 0;
    hcw[k] = vox % I_HW;
    const int t2 = vox / I_HW;
    hch[k] = t2 % I_HH;
    hcd[k] = t2 / I_HH;
    hlds[k] = (u < L2_NH * 2) ? vox * VS + t2 * PADB + (u & 1) * 16 : -1;
    hrel[k] = (((long long)hcd[k] * p.H + hch[k]) * p.W + hcw[k]) * 32 + (tid & 1) * 16;
  }
  int yrel[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) yrel[r] = (((r >> 2) * p.OW) + (r & 3) + 4 * lh) * p.C2 + li;
  const long long yplane = (long long)p.OH * p.OW * p.C2;

  struct Regs {
    v4i h[NHL];
    float y[2][16];
    unsigned hmask;
    long long yoff;      // (OUT) the tile's first output value of this wave's plane, relative to p.y
  };
  auto fetch = [&](int tile, Regs& R) {
    int t = tile;
    const int ow0 = (t % p.tiles_w) * ITW;
    t /= p.tiles_w;
    const int oh0 = (t % p.tiles_h) * ITH;
    t /= p.tiles_h;
    const int od0 = (t % p.tiles_d) * L2_TD;
    const int n = t / p.tiles_d;
    const int id0 = od0 - p.PD, ih0 = oh0 - p.PH, iw0 = ow0 - p.PW;
    const bool interior = id0 >= 0 && id0 + L2_HD <= p.D && ih0 >= 0 && ih0 + I_HH <= p.H && iw0 >= 0 && iw0 + I_HW <= p.W;
    if (interior) {                       // uniform over the workgroup
      const int8_t* hb = p.x + ((((long long)n * p.D + id0) * p.H + ih0) * p.W + iw0) * 32;
#pragma unroll
      for (int k = 0; k < NHL; ++k) R.h[k] = *reinterpret_cast<const v4i*>(hb + ((hlds[k] >= 0) ? hrel[k] : 0));
      R.hmask = 0xffffffffu;
    } else {
      const size_t nbase = (size_t)n * p.D;
      unsigned hm = 0;
#pragma unroll
      for (int k = 0; k < NHL; ++k) {
        const int id = id0 + hcd[k], ih = ih0 + hch[k], iw = iw0 + hcw[k];
        const bool ok = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        const int cd = min(max(id, 0), p.D - 1), chh = min(max(ih, 0), p.H - 1), cw = min(max(iw, 0), p.W - 1);
        hm |= (ok ? 1u : 0u) << k;
        R.h[k] = *reinterpret_cast<const v4i*>(p.x + (((nbase + cd) * p.H + chh) * p.W + cw) * 32 + (tid & 1) * 16);
      }
      R.hmask = hm;
    }
    const long long yoff = ((((long long)n * p.OD + od0 + wid) * p.OH + oh0) * p.OW + ow0) * p.C2;
    const float* yb = p.y + yoff;
    R.yoff = yoff;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      R.y[0][r] = yb[yrel[r]];
      R.y[1][r] = yb[4 * yplane + yrel[r]];
    }
  };

  const int hv0 = (wid * I_HH + (li >> 3)) * I_HW + (li & 7);
  const int hb0 = hv0 * VS + (wid * I_HH + (li >> 3)) * PADB + 16 * lh;
  const int hb1 = hb0 + 4 * I_HH * (I_HW * VS + PADB);
  double l0 = 0.0, l0w = 0.0;
  auto body = [&](int tile, Regs& X, bool more) {
    const long long yoff_cur = X.yoff;
    lds_barrier();
    if (X.hmask == 0xffffffffu) {
#pragma unroll
      for (int k = 0; k < NHL; ++k)
        if (hlds[k] >= 0) *reinterpret_cast<v4i*>(&halo[hlds[k]]) = X.h[k];
    } else {
#pragma unroll
      for (int k = 0; k < NHL; ++k) {
        const v4i val = ((X.hmask >> k) & 1u) ? X.h[k] : v4i{0, 0, 0, 0};
        if (hlds[k] >= 0) *reinterpret_cast<v4i*>(&halo[hlds[k]]) = val;
      }
    }
    float ycur[2][16];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int r = 0; r < 16; ++r) ycur[pl][r] = X.y[pl][r];
    lds_barrier();
    fetch(more ? tile + 2 : tile, X);
    __builtin_amdgcn_sched_barrier(0);
    v16i acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int toff = ((kd * I_HH + kh) * I_HW + kw) * VS + (kd * I_HH + kh) * PADB;
      const v4i b = *reinterpret_cast<const v4i*>(&wl[((tap * 2 + lh) * 32 + li) * 16]);
      const v4i a0 = *reinterpret_cast<const v4i*>(halo + hb0 + toff);
      const v4i a1 = *reinterpret_cast<const v4i*>(halo + hb1 + toff);
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b, acc1, 0, 0, 0);
    }
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (OUT) {
      float sw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      float* ob = p.out + yoff_cur;
      const float* ab = (p.att != nullptr) ? p.att + yoff_cur / p.C2 : nullptr;
      const long long aplane = (long long)p.OH * p.OW;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int arel = ((r >> 2) * p.OW) + (r & 3) + 4 * lh;
        const float o0 = (float)acc0[r] * scale + bv, o1 = (float)acc1[r] * scale + bv;
        ob[yrel[r]] = o0;
        ob[4 * yplane + yrel[r]] = o1;
        const float d0 = o0 - ycur[0][r], d1 = o1 - ycur[1][r];
        const float w0 = ab ? ab[arel] : 1.0f, w1 = ab ? ab[4 * aplane + arel] : 1.0f;
        s[r & 3] = __builtin_fmaf(d0, d0, s[r & 3]);
        s[r & 3] = __builtin_fmaf(d1, d1, s[r & 3]);
        sw[r & 3] = __builtin_fmaf(w0 * d0, d0, sw[r & 3]);
        sw[r & 3] = __builtin_fmaf(w1 * d1, d1, sw[r & 3]);
      }
      l0w += (double)((sw[0] + sw[1]) + (sw[2] + sw[3]));
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d0 = ((float)acc0[r] * scale + bv) - ycur[0][r];
        const float d1 = ((float)acc1[r] * scale + bv) - ycur[1][r];
        s[r & 3] = __builtin_fmaf(d0, d0, s[r & 3]);
        s[r & 3] = __builtin_fmaf(d1, d1, s[r & 3]);
      }
    }
    l0 += (double)((s[0] + s[1]) + (s[2] + s[3]));
  };

  Regs A, B;
#pragma unroll
  for (int k = 0; k < NHL; ++k) A.h[k] = B.h[k] = v4i{0, 0, 0, 0};
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int r = 0; r < 16; ++r) A.y[pl][r] = B.y[pl][r] = 0.0f;
  A.hmask = B.hmask = 0;
  A.yoff = B.yoff = 0;
  if (t_begin < t_end) fetch(t_begin, A);
  if (t_begin + 1 < t_end) fetch(t_begin + 1, B);
  for (int tile = t_begin; tile < t_end; tile += 2) {
    body(tile, A, tile + 2 < t_end);
    if (tile + 1 < t_end) body(tile + 1, B, tile + 3 < t_end);
  }
  double v[2] = {l0, OUT ? l0w : l0};
  grid_sum_finish<2>(v, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.y * gridDim.x + blockIdx.x,
                     gridDim.x * gridDim.y);
}

// ---- 64 -> 64 channels with the weights in LDS ------------------------------------------------------------------------
// k_conv3d_i8<2> keeps the 54 B operands of a wave in 216 registers: one workgroup of 4 waves per CU, nothing to hide the
// halo staging, the barriers and the transposed epilogue behind (0.122 ms for 16 x 32^3 voxels: 17 % of the HBM stream,
// 5x its MFMA time).  Here ALL weights (64 x 64 x 27 bytes = 108 KB, packed [tap][group][k half][out channel][16 B]) sit in
// LDS beside the 6 x 6 x 10 halo tile (31.5 KB): one workgroup of 8 waves per CU - wave w owns d-plane w & 3 and output
// channels 32 (w >> 2) .. +31 of a 4 x 4 x 8 tile - with ~100 registers per wave, targets read in MFMA layout (no
// transpose through LDS), two tiles of prefetch in flight (register sets A / B) and fp32 per-tile sums as in k_conv3d_i8l2e.
// Tile-divisible volumes only (every shape of the shipped recipes); other shapes keep k_conv3d_i8<2>.
constexpr int W64_WLB = 27 * 2 * 2 * 64 * 16;                                  // 110592 bytes of weights
constexpr int W64_VS = 80, W64_HALOB = I_NH * W64_VS + I_HD * I_HH * halo_row_pad(2);
template <bool OUT>
__global__ __launch_bounds__(512, 1) void k_conv3d_i8w(ConvI8Params p) {
  constexpr int VS = W64_VS, PADB = halo_row_pad(2);
  constexpr int NSLOT = I_NH * 4;                   // 16-byte pieces of the halo tile (64 channels = 4 pieces)
  constexpr int NHL = (NSLOT + 511) / 512;          // 3
  extern __shared__ __attribute__((aligned(16))) int8_t dyn_lds[];
  int8_t* wl = dyn_lds;
  int8_t* halo = dyn_lds + W64_WLB;
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int plane = wid & 3, och = 32 * (wid >> 2);
  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per;
  const int t_end = (t_begin + per < p.ntiles) ? t_begin + per : p.ntiles;

  for (int u = tid; u < W64_WLB / 16; u += 512)
    reinterpret_cast<v4i*>(wl)[u] = reinterpret_cast<const v4i*>(p.wq)[u];
  lds_barrier();
  const float scale = (float)((double)(*p.act_alpha) * (double)(float)p.wstate->alpha * p.inv_levels);
  const float bv = (p.bias != nullptr) ? p.bias[och + li] : 0.0f;

  int hcd[NHL], hch[NHL], hcw[NHL], hlds[NHL];
  long long hrel[NHL];
#pragma unroll
  for (int k = 0; k < NHL; ++k) {
    const int u = tid + k * 512;
    const bool live = u < NSLOT;
    const int vox = live ? (u >> 2) : 0, part = u & 3;
    hcw[k] = vox % I_HW;
    const int t2 = vox / I_HW;
    hch[k] = t2 % I_HH;
    hcd[k] = t2 / I_HH;
    hlds[k] = live ? vox * VS + t2 * PADB + part * 16 : -1;
    hrel[k] = (((long long)hcd[k] * p.H + hch[k]) * p.W + hcw[k]) * 64 + part * 16;
  }
  int yrel[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) yrel[r] = (((r >> 2) * p.OW) + (r & 3) + 4 * lh) * p.C2 + och + li;

  struct Regs {
    v4i h[NHL];
    float y[16];
    unsigned hmask;
    long long yoff;      // (OUT) the tile's first output value of this wave's plane, relative to p.y
  };
  auto fetch = [&](int tile, Regs& R) {
    int t = tile;
    const int ow0 = (t % p.tiles_w) * ITW;
    t /= p.tiles_w;
    const int oh0 = (t % p.tiles_h) * ITH;
    t /= p.tiles_h;
    const int od0 = (t % p.tiles_d) * ITD;
    const int n = t / p.tiles_d;
    const int id0 = od0 - p.PD, ih0 = oh0 - p.PH, iw0 = ow0 - p.PW;
    const bool interior = id0 >= 0 && id0 + I_HD <= p.D && ih0 >= 0 && ih0 + I_HH <= p.H && iw0 >= 0 && iw0 + I_HW <= p.W;
    if (interior) {                       // uniform over the workgroup
      const int8_t* hb = p.x + ((((long long)n * p.D + id0) * p.H + ih0) * p.W + iw0) * 64;
#pragma unroll
      for (int k = 0; k < NHL; ++k) R.h[k] = *reinterpret_cast<const v4i*>(hb + ((hlds[k] >= 0) ? hrel[k] : 0));
      R.hmask = 0xffffffffu;
    } else {
      const size_t nbase = (size_t)n * p.D;
      unsigned hm = 0;
#pragma unroll
      for (int k = 0; k < NHL; ++k) {
        const int id = id0 + hcd[k], ih = ih0 + hch[k], iw = iw0 + hcw[k];
        const bool ok = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        const int cd = min(max(id, 0), p.D - 1), chh = min(max(ih, 0), p.H - 1), cw = min(max(iw, 0), p.W - 1);
        hm |= (ok ? 1u : 0u) << k;
        R.h[k] = *reinterpret_cast<const v4i*>(p.x + (((nbase + cd) * p.H + chh) * p.W + cw) * 64 + ((tid + k * 512) & 3) * 16);
      }
      R.hmask = hm;
    }
    const long long yoff = ((((long long)n * p.OD + od0 + plane) * p.OH + oh0) * p.OW + ow0) * p.C2;
    const float* yb = p.y + yoff;
    R.yoff = yoff;
#pragma unroll
    for (int r = 0; r < 16; ++r) R.y[r] = yb[yrel[r]];
  };

  const int hrow = plane * I_HH + (li >> 3);
  const int hb0 = (hrow * I_HW + (li & 7)) * VS + hrow * PADB + 16 * lh;
  const int8_t* wlane = wl + ((size_t)lh * 64 + och + li) * 16;        // + (tap * 2 + g) * 2 * 64 * 16 per step
  double l0 = 0.0, l0w = 0.0;
  auto body = [&](int tile, Regs& X, bool more) {
    const long long yoff_cur = X.yoff;
    lds_barrier();
    if (X.hmask == 0xffffffffu) {
#pragma unroll
      for (int k = 0; k < NHL; ++k)
        if (hlds[k] >= 0) *reinterpret_cast<v4i*>(&halo[hlds[k]]) = X.h[k];
    } else {
#pragma unroll
      for (int k = 0; k < NHL; ++k) {
        const v4i val = ((X.hmask >> k) & 1u) ? X.h[k] : v4i{0, 0, 0, 0};
        if (hlds[k] >= 0) *reinterpret_cast<v4i*>(&halo[hlds[k]]) = val;
      }
    }
    float ycur[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ycur[r] = X.y[r];
    lds_barrier();
    fetch(more ? tile + 2 : tile, X);
    __builtin_amdgcn_sched_barrier(0);
    v16i acc0, acc1;                       // one chain per channel group
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int toff = ((kd * I_HH + kh) * I_HW + kw) * VS + (kd * I_HH + kh) * PADB;
      const v4i a0 = *reinterpret_cast<const v4i*>(halo + hb0 + toff);
      const v4i a1 = *reinterpret_cast<const v4i*>(halo + hb0 + toff + 32);
      const v4i b0 = *reinterpret_cast<const v4i*>(wlane + (size_t)(tap * 2 + 0) * 2 * 64 * 16);
      const v4i b1 = *reinterpret_cast<const v4i*>(wlane + (size_t)(tap * 2 + 1) * 2 * 64 * 16);
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc1, 0, 0, 0);
    }
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (OUT) {
      float sw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      float* ob = p.out + yoff_cur;
      const float* ab = (p.att != nullptr) ? p.att + yoff_cur / p.C2 : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int arel = ((r >> 2) * p.OW) + (r & 3) + 4 * lh;
        const float o = (float)(acc0[r] + acc1[r]) * scale + bv;
        ob[yrel[r]] = o;
        const float d = o - ycur[r];
        // (both channel halves of a voxel add its weighted error: the sum runs over the values, as the plain one)
        const float wgt = ab ? ab[arel] : 1.0f;
        s[r & 3] = __builtin_fmaf(d, d, s[r & 3]);
        sw[r & 3] = __builtin_fmaf(wgt * d, d, sw[r & 3]);
      }
      l0w += (double)((sw[0] + sw[1]) + (sw[2] + sw[3]));
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = ((float)(acc0[r] + acc1[r]) * scale + bv) - ycur[r];
        s[r & 3] = __builtin_fmaf(d, d, s[r & 3]);
      }
    }
    l0 += (double)((s[0] + s[1]) + (s[2] + s[3]));
  };

  Regs A, B;
#pragma unroll
  for (int k = 0; k < NHL; ++k) A.h[k] = B.h[k] = v4i{0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 16; ++r) A.y[r] = B.y[r] = 0.0f;
  A.hmask = B.hmask = 0;
  A.yoff = B.yoff = 0;
  if (t_begin < t_end) fetch(t_begin, A);
  if (t_begin + 1 < t_end) fetch(t_begin + 1, B);
  for (int tile = t_begin; tile < t_end; tile += 2) {
    body(tile, A, tile + 2 < t_end);
    if (tile + 1 < t_end) body(tile + 1, B, tile + 3 < t_end);
  }
  double v[2] = {l0, OUT ? l0w : l0};
  grid_sum_finish<2>(v, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.x, gridDim.x);
}

struct I8Plan {
  ConvI8Params p;
  dim3 grid;
  size_t nblk, wq_bytes;
};

static bool i8_stream64() {
  static const int on = getenv("EFFQ_I8G64") ? atoi(getenv("EFFQ_I8G64")) : 0;
  return on != 0;
}
static bool i8_two_plane(const effq_geom* g) { return g->C1 == 32; }

static int i8_plan(const effq_geom* g, I8Plan* pl) {
  EFFQ_CHECK_ARG(g != nullptr);
  EFFQ_CHECK_ARG(g->KD == 3 && g->KH == 3 && g->KW == 3 && g->SD == 1 && g->SH == 1 && g->SW == 1);
  EFFQ_CHECK_ARG(g->C1 == 32 || g->C1 == 64 || g->C1 == 128 || g->C1 == 256 || g->C1 == 512);
  EFFQ_CHECK_ARG(g->C2 > 0 && (g->C2 % 32) == 0 && (g->C1 != 512 || (g->C2 % 64) == 0));
  EFFQ_CHECK_ARG(g->N > 0 && g->D > 0 && g->H > 0 && g->W > 0 && g->PD >= 0 && g->PH >= 0 && g->PW >= 0);
  ConvI8Params& p = pl->p;
  memset(&p, 0, sizeof(p));
  p.N = g->N; p.C1 = g->C1; p.C2 = g->C2; p.c2p = g->C2; p.D = g->D; p.H = g->H; p.W = g->W;
  p.PD = g->PD; p.PH = g->PH; p.PW = g->PW;
  p.OD = g->D + 2 * g->PD - 2; p.OH = g->H + 2 * g->PH - 2; p.OW = g->W + 2 * g->PW - 2;
  EFFQ_CHECK_ARG(p.OD > 0 && p.OH > 0 && p.OW > 0);
  const int td = i8_two_plane(g) ? L2_TD : (g->C1 == 512) ? G2_TD : ITD;
  p.tiles_d = (p.OD + td - 1) / td;
  p.tiles_h = (p.OH + ITH - 1) / ITH;
  p.tiles_w = (p.OW + ITW - 1) / ITW;
  const long long nt = (long long)p.N * p.tiles_d * p.tiles_h * p.tiles_w;
  EFFQ_CHECK_ARG(nt < (1ll << 30));
  p.ntiles = (int)nt;
  const int ny = (g->C1 == 512) ? p.C2 / 64 : p.C2 / 32;
  static const int wpc128 = getenv("EFFQ_I8_WPC128") ? atoi(getenv("EFFQ_I8_WPC128")) : 2;   // tuning aid
  static const int wpc64 = getenv("EFFQ_I8_WPC64") ? atoi(getenv("EFFQ_I8_WPC64")) : 2;      // tuning aid
  const int wg_per_cu = (g->C1 == 32) ? 2 : (g->C1 == 128) ? wpc128
                        : (g->C1 == 64 && i8_stream64()) ? wpc64 : 1;
  int gx = (256 * wg_per_cu + ny - 1) / ny;
  if (gx < 32) gx = 32;
  // The persistent grid leaves a few workgroup slots empty: the chain kernels of the NEXT iterate (prox GEMM, scale fixed
  // point, projection) run beside this kernel and otherwise find no room until the persistent workgroups retire - chain
  // and loss then serialise (32-channel layers: 0.104 ms per prox solve in situ against 0.046 with 15 slots free (grid 497; 482 workgroups measured worse again),
  // 1222 -> 1190 ms per calibration).  The grid is trimmed to equal runs of tiles.  EFFQ_I8_RESERVE overrides (tuning aid).
  static const int reserve = getenv("EFFQ_I8_RESERVE") ? atoi(getenv("EFFQ_I8_RESERVE")) : 15;
  static const int reserve_wide = getenv("EFFQ_I8_RESERVE_WIDE") ? atoi(getenv("EFFQ_I8_RESERVE_WIDE")) : -1;   // tuning aid
  const int rsv = (g->C1 >= 128 && reserve_wide >= 0) ? reserve_wide : reserve;
  if (rsv > 0 && gx * ny > 2 * rsv) {
    int g0 = gx - (rsv + ny - 1) / ny;
    const int per = (p.ntiles + g0 - 1) / g0;
    gx = (p.ntiles + per - 1) / per;
  }
  // EFFQ_I8_TPW = tiles per workgroup: > 0 launches ntiles / TPW short-lived workgroups instead of a persistent grid
  static const int tpw = getenv("EFFQ_I8_TPW") ? atoi(getenv("EFFQ_I8_TPW")) : 0;          // tuning aid
  if (tpw > 0) gx = (p.ntiles + tpw - 1) / tpw;
  if (gx > p.ntiles) gx = p.ntiles;
  pl->grid = dim3((unsigned)gx, (unsigned)ny, 1);
  pl->nblk = (size_t)gx * ny;
  if (pl->nblk < 256) pl->nblk = 256;      // k_conv3d_i8w runs one workgroup per CU whatever the plan above says
  pl->wq_bytes = (size_t)27 * p.c2p * p.C1;
  return EFFQ_OK;
}

}  // namespace effq

using namespace effq;

extern "C" {

int effq_conv_i8_supported(const effq_geom* g, int act_levels, int w_levels) {
  if (g == nullptr) return 0;
  if (!(g->KD == 3 && g->KH == 3 && g->KW == 3 && g->SD == 1 && g->SH == 1 && g->SW == 1)) return 0;
  if (!(g->C1 == 32 || g->C1 == 64 || g->C1 == 128 || g->C1 == 256 || g->C1 == 512) || (g->C2 % 32) != 0) return 0;
  if (g->C1 == 512 && (g->C2 % 64) != 0) return 0;
  if (act_levels < 2 || act_levels > 128 || w_levels < 2 || w_levels > 128) return 0;
  // int32 accumulator range: 27*C1 products of at most (La-1)*(Lw-1)
  if ((double)27 * g->C1 * (act_levels - 1) * (w_levels - 1) >= 2147483647.0) return 0;
  return 1;
}

size_t effq_conv_i8_ws_bytes(const effq_geom* g) {
  I8Plan pl;
  if (i8_plan(g, &pl) != EFFQ_OK) return 0;
  return 256 + pl.nblk * 2 * sizeof(double) + pl.wq_bytes + 256;
}

// which of the kernels that can also store their output serves the shape: 1 = k_conv3d_i8l2e (32 -> 32), 2 = k_conv3d_i8w
// (64 -> 64), 0 = none
static int i8_out_kernel(const ConvI8Params& p) {
  static const bool fast_off = getenv("EFFQ_I8L2E") != nullptr && atoi(getenv("EFFQ_I8L2E")) == 0;
  static const bool w64_off = getenv("EFFQ_I8W") != nullptr && atoi(getenv("EFFQ_I8W")) == 0;
  if (p.C1 == 32 && p.C2 == 32 && !fast_off && p.OD % L2_TD == 0 && p.OH % ITH == 0 && p.OW % ITW == 0) return 1;
  if (p.C1 == 64 && p.C2 == 64 && !w64_off && !i8_stream64() && p.OD % ITD == 0 && p.OH % ITH == 0 && p.OW % ITW == 0) return 2;
  return 0;
}

int effq_conv_i8_out_supported(const effq_geom* g, int act_levels, int w_levels) {
  if (g == nullptr || !effq_conv_i8_supported(g, act_levels, w_levels)) return 0;
  I8Plan pl;
  if (i8_plan(g, &pl) != EFFQ_OK) return 0;
  return i8_out_kernel(pl.p) != 0 ? 1 : 0;
}

static int conv_i8_impl(const uint8_t* xidx_ndhwc, const int8_t* Gq, const float* bias, const float* y_fp,
                        const effq_geom* g, const float* act_alpha_dev, int act_levels,
                        const effq_fp_state* w_state_dev, int w_levels, double* sqerr_out, void* ws, size_t ws_bytes,
                        void* stream, float* out, const float* att) {
  EFFQ_CHECK_ARG(xidx_ndhwc && Gq && y_fp && g && act_alpha_dev && w_state_dev && sqerr_out && ws);
  EFFQ_CHECK_ARG(effq_conv_i8_supported(g, act_levels, w_levels));
  I8Plan pl;
  int rc = i8_plan(g, &pl);
  if (rc != EFFQ_OK) return rc;
  const size_t need = 256 + pl.nblk * 2 * sizeof(double) + pl.wq_bytes + 256;
  if (ws_bytes < need) {
    set_error("conv_i8: workspace %zu < required %zu", ws_bytes, need);
    return EFFQ_ERR_WORKSPACE;
  }
  char* base = reinterpret_cast<char*>(ws);
  ConvI8Params& p = pl.p;
  p.ticket = reinterpret_cast<unsigned int*>(base);
  p.partials = reinterpret_cast<double*>(base + 256);
  int8_t* wq = reinterpret_cast<int8_t*>(base + 256 + pl.nblk * 2 * sizeof(double));
  wq = reinterpret_cast<int8_t*>((reinterpret_cast<uintptr_t>(wq) + 15) & ~(uintptr_t)15);
  p.x = reinterpret_cast<const int8_t*>(xidx_ndhwc);
  p.wq = wq;
  p.bias = bias;
  p.y = y_fp;
  p.act_alpha = act_alpha_dev;
  p.wstate = w_state_dev;
  p.inv_levels = 1.0 / ((double)(act_levels - 1) * (double)(w_levels - 1));
  p.sqerr = sqerr_out;
  p.debug = effq_ablate_env("EFFQ_I8_DEBUG");
  p.out = out;
  p.att = att;
  const int outk = (out != nullptr) ? i8_out_kernel(p) : 0;
  if (out != nullptr && outk == 0) {
    set_error("conv_i8: no output-storing kernel for this shape (effq_conv_i8_out_supported)");
    return EFFQ_ERR_ARG;
  }
  hipStream_t st = as_stream(stream);
  // (the ticket of the last-block reduction is left at zero by the kernel that used it: the caller zero-fills
  //  the workspace once, effq_hip.h)
  static const bool w64_off = getenv("EFFQ_I8W") != nullptr && atoi(getenv("EFFQ_I8W")) == 0;      // A/B switch
  if (p.C1 == 64 && p.C2 == 64 && !w64_off && !i8_stream64() && p.OD % ITD == 0 && p.OH % ITH == 0 && p.OW % ITW == 0) {
    size_t nb = (pl.wq_bytes + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_pack_weight_i8g<4>, dim3((unsigned)((p.c2p + 3) / 4)), dim3(256), (size_t)4 * p.C1 * 27, st, Gq, wq, p.C1, p.C2, 27, p.c2p);
    const size_t lds = (size_t)W64_WLB + W64_HALOB;
    static bool attr_set = false;
    if (!attr_set) {
      EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3d_i8w<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3d_i8w<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
    int gx = 256;                           // one workgroup per CU (LDS); pl.nblk = 256 partial slots
    if (gx > p.ntiles) gx = p.ntiles;
    if ((size_t)gx > pl.nblk) gx = (int)pl.nblk;
    if (out != nullptr)
      hipLaunchKernelGGL(k_conv3d_i8w<true>, dim3((unsigned)gx), dim3(512), lds, st, p);
    else
      hipLaunchKernelGGL(k_conv3d_i8w<false>, dim3((unsigned)gx), dim3(512), lds, st, p);
    EFFQ_LAUNCH_CHECK();
    return EFFQ_OK;
  }
  {
    size_t nb = (pl.wq_bytes + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (p.C1 < 64 || (p.C1 == 64 && !i8_stream64()))
      hipLaunchKernelGGL(k_pack_weight_i8, dim3((unsigned)nb), dim3(256), 0, st, Gq, wq, p.C1, p.C2, 27, p.c2p);
    else {
      // the pack sits on the loss stream beside the ADMM chain: one output channel per workgroup (128 - 512 workgroups
      // instead of 32 - 128) shortens it.  EFFQ_I8_PACK_RPW=4: the 4-channel form (A/B switch)
      static const int rpw = getenv("EFFQ_I8_PACK_RPW") ? atoi(getenv("EFFQ_I8_PACK_RPW")) : 1;
      if (rpw == 4)
        hipLaunchKernelGGL(k_pack_weight_i8g<4>, dim3((unsigned)((p.c2p + 3) / 4)), dim3(256), (size_t)4 * p.C1 * 27, st, Gq, wq, p.C1, p.C2, 27, p.c2p);
      else
        hipLaunchKernelGGL(k_pack_weight_i8g<1>, dim3((unsigned)p.c2p), dim3(256), (size_t)p.C1 * 27, st, Gq, wq, p.C1, p.C2, 27, p.c2p);
    }
    EFFQ_LAUNCH_CHECK();
  }
  if (p.C1 == 32) {
    static const bool fast_off = getenv("EFFQ_I8L2E") != nullptr && atoi(getenv("EFFQ_I8L2E")) == 0;   // A/B switch
    if (out != nullptr)
      hipLaunchKernelGGL(k_conv3d_i8l2e<true>, pl.grid, dim3(256), 0, st, p);
    else if (!fast_off && p.C2 == 32 && p.OD % L2_TD == 0 && p.OH % ITH == 0 && p.OW % ITW == 0)
      hipLaunchKernelGGL(k_conv3d_i8l2e<false>, pl.grid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL(k_conv3d_i8l2, pl.grid, dim3(256), 0, st, p);
  } else if (p.C1 == 64 && !i8_stream64()) {
    hipLaunchKernelGGL(k_conv3d_i8<2>, pl.grid, dim3(256), 0, st, p);
  } else {
    const int cg = p.C1 / 32;
    const size_t lds = (size_t)((I_NH * (32 * cg + 16) + I_HD * I_HH * halo_row_pad(cg) + 15) / 16) * 16 +
                       (size_t)128 * I_TS * sizeof(float);
    if (cg == 16) {
      const size_t lds2 = (size_t)((G2_NH * (32 * 16 + 16) + G2_HD * I_HH * halo_row_pad(16) + 15) / 16) * 16 +
                          (size_t)64 * G2_TS * sizeof(float);
      EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3d_i8g2<16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      hipLaunchKernelGGL(k_conv3d_i8g2<16>, pl.grid, dim3(256), lds2, st, p);
    } else if (cg == 2) {
      EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3d_i8g<2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_conv3d_i8g<2>, pl.grid, dim3(256), lds, st, p);
    } else if (cg == 4) {
      EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3d_i8g<4>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_conv3d_i8g<4>, pl.grid, dim3(256), lds, st, p);
    } else {
      EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3d_i8g<8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_conv3d_i8g<8>, pl.grid, dim3(256), lds, st, p);
    }
  }
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int conv3d_calib_step_i8(const uint8_t* xidx_ndhwc, const int8_t* Gq, const float* bias, const float* y_fp,
                         const effq_geom* g, const float* act_alpha_dev, int act_levels,
                         const effq_fp_state* w_state_dev, int w_levels, double* sqerr_out, void* ws, size_t ws_bytes,
                         void* stream) {
  return conv_i8_impl(xidx_ndhwc, Gq, bias, y_fp, g, act_alpha_dev, act_levels, w_state_dev, w_levels, sqerr_out, ws,
                      ws_bytes, stream, nullptr, nullptr);
}

int conv3d_quant_forward_i8(const uint8_t* xidx_ndhwc, const int8_t* Gq, const float* bias, const float* y_fp,
                            const float* att, const effq_geom* g, const float* act_alpha_dev, int act_levels,
                            const effq_fp_state* w_state_dev, int w_levels, double* sqerr_out, float* out, void* ws,
                            size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG(out != nullptr);
  return conv_i8_impl(xidx_ndhwc, Gq, bias, y_fp, g, act_alpha_dev, act_levels, w_state_dev, w_levels, sqerr_out, ws,
                      ws_bytes, stream, out, att);
}

}  // extern "C"
