// conv3d_calib_step_i8s: exact integer loss evaluation for the layers the tiled i8 kernels do not take -
// few input taps*channels (K = T*C1 <= 256: the 4-channel first conv, the 1x1x1 convs, the classifier) and/or
// 256 quantisation levels (q_first / q_last of the reference's recipes).
// Reference: EfficientQConv.py:118-122 (conv3d(Qactivation, G, b*) + mse_loss, 200x per layer).
//
// Integers.  Activation level u in [0, La-1]; for La > 128 the byte is re-centred u' = u - 128 (one XOR 0x80,
// zero padding becomes -128 by the same XOR).  Weight level k in [0, Lw-1], numerator m = 2k - (Lw-1); for
// Lw <= 128 the int8 operand is m itself, for Lw > 128 it is k' = k - 128 and m = 2k' + 1.  With
//   acc  = sum over ALL taps (padded ones included) of (weight operand) * (activation operand)   [i8 MFMA, int32]
//   Ksum = sum over all taps of the weight operand (per output channel)
//   Su   = sum over the receptive field of u (per output voxel; computed once per layer by the same kernel)
// the exact numerator of the conv is  wmul * (acc + aoff * Ksum) + (Lw > 128 ? Su : 0),  wmul = Lw > 128 ? 2 : 1,
// aoff = La > 128 ? 128 : 0, and out = alpha_a * alpha_w / ((La-1)(Lw-1)) * numerator + bias.
//
// Mapping: one wave = 32 consecutive output voxels x all output channels.  K is so short that the whole packed
// weight tensor sits in registers as MFMA B operands; the A operand of a lane (voxel l&31, K half l>>5) is
// gathered straight from global/L2 (16 consecutive K bytes = four 4-channel pixels of four taps, or 16 channels
// of one tap) - no LDS staging, latency is covered by occupancy.  The kernel streams C2*4 bytes of target
// and ~C1/stride^3 bytes of level ids per output voxel: HBM bound.
#include <stdint.h>
#include <stdlib.h>
#include "common.h"

namespace effq {

typedef int s_v4i __attribute__((ext_vector_type(4)));
typedef int s_v16i __attribute__((ext_vector_type(16)));

struct ConvI8sParams {
  const uint8_t* x;
  const int8_t* wq;       // packed B operands [step j][col tile ct][k half h][32 cols][16 bytes]
  const int* ksum;        // [c2p]
  const int* su_in;       // [V] (Lw > 128) or null
  int* su_out;            // sum mode: write Su instead of the loss
  const float* bias;
  const float* y;
  const float* act_alpha;
  const effq_fp_state* wstate;
  double inv_levels;
  int N, C1, C2, D, H, W, OD, OH, OW, KD, KH, KW, SD, SH, SW, PD, PH, PW;
  int T, K, NJ, CT;       // taps, T*C1, MFMA K steps, column tiles
  int wmul, aoff, xor80;
  long long V;
  int ntiles;             // 32-voxel wave tiles
  double* partials;
  unsigned int* ticket;
  double* sqerr;
};

// Gq [C2][C1][T] (reference weight layout, int8 operands) -> packed B operands + Ksum
__global__ __launch_bounds__(256) void k_pack_weight_i8s(const int8_t* __restrict__ Gq, int8_t* __restrict__ wq,
                                                         int* __restrict__ ksum, int C1, int C2, int T, int K, int NJ,
                                                         int CT) {
  const size_t total = (size_t)NJ * CT * 2 * 32 * 16;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int b = (int)(e & 15);
    size_t r = e >> 4;
    const int n = (int)(r & 31);
    r >>= 5;
    const int h = (int)(r & 1);
    r >>= 1;
    const int ct = (int)(r % CT);
    const int j = (int)(r / CT);
    const int k = 32 * j + 16 * h + b, col = 32 * ct + n;
    int8_t val = 0;
    if (k < K && col < C2) {
      const int tap = k / C1, c = k - tap * C1;
      val = Gq[((size_t)col * C1 + c) * T + tap];
    }
    wq[e] = val;
  }
  // Ksum: one thread per output channel (K <= 256)
  for (size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x; col < (size_t)CT * 32; col += stride) {
    int s = 0;
    if (col < (size_t)C2)
      for (int k = 0; k < K; ++k) s += (int)Gq[col * K + k];    // [C1][T] contiguous per output channel
    ksum[col] = s;
  }
}

// all-ones weights in column 0 (sum mode)
__global__ __launch_bounds__(256) void k_pack_ones_i8s(int8_t* __restrict__ wq, int K, int NJ) {
  const size_t total = (size_t)NJ * 2 * 32 * 16;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int b = (int)(e & 15);
    size_t r = e >> 4;
    const int n = (int)(r & 31);
    r >>= 5;
    const int h = (int)(r & 1);
    const int j = (int)(r >> 1);
    const int k = 32 * j + 16 * h + b;
    wq[e] = (k < K && n == 0) ? (int8_t)1 : (int8_t)0;
  }
}

// C4: C1 == 4 (a lane's 16 K bytes are four taps x 4 channels); otherwise C1 % 16 == 0 (16 channels of one tap).
// NJ = K steps (<= 8), CT = column tiles; NB = NJ*CT B operands per lane live in registers.
template <bool C4, int NJ, int CT>
__global__ __launch_bounds__(256) void k_conv3d_i8s(ConvI8sParams p) {
  __shared__ double red_smem[2 * 16];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int NL = C4 ? 4 : 1;          // loads per K step

  // B operands
  s_v4i breg[NJ][CT];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      breg[j][ct] = *reinterpret_cast<const s_v4i*>(p.wq + ((((size_t)j * CT + ct) * 2 + lh) * 32 + li) * 16);

  // per-lane tap table of its K chunks: linear input offset (elements of C1 bytes) and tap id (>= T: dummy)
  int toff[NJ][NL], tbit[NJ][NL];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int k0 = 32 * j + 16 * lh + (C4 ? 4 * q : 0);
      const int tap = k0 / p.C1, c0 = k0 - tap * p.C1;
      const int kw = tap % p.KW, t2 = tap / p.KW;
      const int kh = t2 % p.KH, kd = t2 / p.KH;
      toff[j][q] = ((kd * p.H + kh) * p.W + kw) * p.C1 + c0;
      tbit[j][q] = (tap < p.T) ? tap : 31;      // bit 31 of the mask is never set
    }

  const float scale = (float)((double)(*p.act_alpha) * (double)(float)p.wstate->alpha * p.inv_levels);
  float bv[CT];
  int ks[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = 32 * ct + li;
    bv[ct] = (p.bias != nullptr && col < p.C2) ? p.bias[col] : 0.0f;
    ks[ct] = p.ksum ? p.ksum[col] : 0;
  }
  const unsigned xorv = p.xor80 ? 0x80808080u : 0u;

  // Two tiles in flight per wave: the gathers and targets of tile i+1 are issued before the MFMA chain and the epilogue
  // of tile i (a wave walks 4-8 tiles; with the loads inside the iteration every tile paid a full memory round trip).
  constexpr bool PRE = CT <= 2;               // targets prefetched too (CT x 16 registers per buffer)
  struct TileRegs {
    s_v4i a[NJ];
    float yv[PRE ? CT : 1][16];
    unsigned okbits;                          // bit 4j + q: load (j, q) is a real tap
    unsigned okmask[CT];                      // bit r: (row r of this lane, column 32 ct + li) is a real output
    int su_lane;
    long long v0;
  };
  auto fetch = [&](int tile, TileRegs& R) {
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
    // 27-bit validity mask of the taps (bit tap = kd*KH*KW + kh*KW + kw); T <= 27
    unsigned m = 0;
    {
      unsigned wm = 0, hm = 0, dm = 0;
      for (int kw = 0; kw < p.KW; ++kw) wm |= (unsigned)(iw0 + kw >= 0 && iw0 + kw < p.W) << kw;
      for (int kh = 0; kh < p.KH; ++kh) hm |= (unsigned)(ih0 + kh >= 0 && ih0 + kh < p.H) << kh;
      for (int kd = 0; kd < p.KD; ++kd) dm |= (unsigned)(id0 + kd >= 0 && id0 + kd < p.D) << kd;
      unsigned m9 = 0;
      for (int kh = 0; kh < p.KH; ++kh) m9 |= ((hm >> kh) & 1u) ? (wm << (kh * p.KW)) : 0u;
      for (int kd = 0; kd < p.KD; ++kd) m |= ((dm >> kd) & 1u) ? (m9 << (kd * p.KH * p.KW)) : 0u;
      if (!vvalid) m = 0;
    }
    const int xbase = (((n * p.D + id0) * p.H + ih0) * p.W + iw0) * p.C1;
    unsigned okb = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (C4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = (m >> tbit[j][q]) & 1u;
          okb |= (ok ? 1u : 0u) << (4 * j + q);
          R.a[j][q] = (int)*reinterpret_cast<const unsigned*>(p.x + (ok ? (xbase + toff[j][q]) : 0));
        }
      } else {
        const bool ok = (m >> tbit[j][0]) & 1u;
        okb |= (ok ? 1u : 0u) << (4 * j);
        R.a[j] = *reinterpret_cast<const s_v4i*>(p.x + (ok ? (xbase + toff[j][0]) : 0));
      }
    }
    R.okbits = okb;
    R.v0 = (long long)tile * 32;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int col = 32 * ct + li;
      unsigned mk = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long vr = R.v0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = vr < p.V && col < p.C2;
        mk |= (ok ? 1u : 0u) << r;
        if (PRE && p.su_out == nullptr) R.yv[PRE ? ct : 0][r] = p.y[ok ? vr * p.C2 + col : 0];
      }
      R.okmask[ct] = mk;
    }
    R.su_lane = (p.su_in != nullptr) ? p.su_in[vvalid ? v : 0] : 0;
  };

  double l0 = 0.0;
  auto compute = [&](const TileRegs& R) {
    s_v16i acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      s_v4i a;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = (R.okbits >> (4 * j + (C4 ? q : 0))) & 1u;
        a[q] = (int)((ok ? (unsigned)R.a[j][q] : 0u) ^ xorv);
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, breg[j][ct], acc[ct], 0, 0, 0);
    }
    // epilogue: acc[ct][r] belongs to voxel row (r&3) + 8*(r>>2) + 4*lh of the tile, channel 32*ct + li
    if (p.su_out != nullptr) {       // sum mode: column 0 holds sum of the activation operands
      if (li == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long vr = R.v0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (vr < p.V) p.su_out[vr] = acc[0][r] + p.aoff * p.K;
        }
      }
      return;
    }
    // squared errors: fp32 within the tile (4 chains), one fp64 addition per tile
    float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int su = (p.su_in != nullptr) ? __shfl(R.su_lane, row) : 0;    // lane `row` (first half) holds voxel row's Su
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const bool ok = (R.okmask[ct] >> r) & 1u;
        const float yy = PRE ? R.yv[PRE ? ct : 0][r] : p.y[ok ? (R.v0 + row) * p.C2 + 32 * ct + li : 0];
        const int num = p.wmul * (acc[ct][r] + p.aoff * ks[ct]) + su;
        const float dd = (scale * (float)num + bv[ct]) - yy;
        s4[r & 3] = ok ? __builtin_fmaf(dd, dd, s4[r & 3]) : s4[r & 3];
      }
    }
    l0 += (double)((s4[0] + s4[1]) + (s4[2] + s4[3]));
  };

  const int wave_global = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;
  if (wave_global < p.ntiles) {
    TileRegs A, B;
    fetch(wave_global, A);
    for (int tile = wave_global; tile < p.ntiles; tile += 2 * nwaves) {
      const bool m1 = tile + nwaves < p.ntiles;
      if (m1) fetch(tile + nwaves, B);
      compute(A);
      if (!m1) break;
      const bool m2 = tile + 2 * nwaves < p.ntiles;
      if (m2) fetch(tile + 2 * nwaves, A);
      compute(B);
    }
  }
  if (p.su_out != nullptr) return;
  double vsum[2] = {l0, l0};
  grid_sum_finish<2>(vsum, p.partials, p.ticket, p.sqerr, red_smem, &s_last, blockIdx.x, gridDim.x);
}

struct I8sPlan {
  ConvI8sParams p;
  int c4, grid;
  size_t wq_bytes, su_bytes;
};

static int i8s_plan(const effq_geom* g, int act_levels, int w_levels, I8sPlan* pl) {
  EFFQ_CHECK_ARG(g != nullptr);
  EFFQ_CHECK_ARG(g->N > 0 && g->C1 > 0 && g->C2 > 0 && g->D > 0 && g->H > 0 && g->W > 0);
  EFFQ_CHECK_ARG(g->KD >= 1 && g->KH >= 1 && g->KW >= 1 && g->SD >= 1 && g->SH >= 1 && g->SW >= 1);
  EFFQ_CHECK_ARG(g->PD >= 0 && g->PH >= 0 && g->PW >= 0);
  EFFQ_CHECK_ARG(act_levels >= 2 && act_levels <= 256 && w_levels >= 2 && w_levels <= 256);
  ConvI8sParams& p = pl->p;
  memset(&p, 0, sizeof(p));
  p.N = g->N; p.C1 = g->C1; p.C2 = g->C2; p.D = g->D; p.H = g->H; p.W = g->W;
  p.KD = g->KD; p.KH = g->KH; p.KW = g->KW; p.SD = g->SD; p.SH = g->SH; p.SW = g->SW;
  p.PD = g->PD; p.PH = g->PH; p.PW = g->PW;
  p.OD = (g->D + 2 * g->PD - g->KD) / g->SD + 1;
  p.OH = (g->H + 2 * g->PH - g->KH) / g->SH + 1;
  p.OW = (g->W + 2 * g->PW - g->KW) / g->SW + 1;
  EFFQ_CHECK_ARG(p.OD > 0 && p.OH > 0 && p.OW > 0);
  p.T = p.KD * p.KH * p.KW;
  p.K = p.T * p.C1;
  EFFQ_CHECK_ARG(p.T <= 27 && p.K <= 256);
  EFFQ_CHECK_ARG(p.C1 == 4 || (p.C1 % 16) == 0);
  pl->c4 = (p.C1 == 4) ? 1 : 0;
  p.NJ = (p.K + 31) / 32;
  p.CT = (p.C2 + 31) / 32;
  EFFQ_CHECK_ARG(p.NJ * p.CT <= 16 && p.CT <= 8);
  p.V = (long long)p.N * p.OD * p.OH * p.OW;
  EFFQ_CHECK_ARG(p.V < (1ll << 31) && (long long)g->N * g->D * g->H * g->W * g->C1 < (1ll << 31));
  EFFQ_CHECK_ARG(p.V * p.C2 < (1ll << 40));
  p.ntiles = (int)((p.V + 31) / 32);
  p.wmul = (w_levels > 128) ? 2 : 1;
  p.aoff = (act_levels > 128) ? 128 : 0;
  p.xor80 = (act_levels > 128) ? 1 : 0;
  p.inv_levels = 1.0 / ((double)(act_levels - 1) * (double)(w_levels - 1));
  // int32 range of the numerator
  EFFQ_CHECK_ARG((double)p.K * 255.0 * 255.0 * 2.0 < 2147483647.0);
  // persistent: every workgroup pays a prologue (B operands, tap table) and one same-address ticket atomic (~12 ns
  // each, serialised): a few workgroups per CU with several tiles per wave, not one tile per wave
  static const int cap = getenv("EFFQ_I8S_GRID") ? atoi(getenv("EFFQ_I8S_GRID")) : 512;      // tuning aid
  int grid = (p.ntiles + 3) / 4;
  if (grid > cap) grid = cap;
  pl->grid = grid;
  pl->wq_bytes = (size_t)p.NJ * p.CT * 2 * 32 * 16;
  pl->su_bytes = (w_levels > 128) ? (size_t)p.V * sizeof(int) : 0;
  return EFFQ_OK;
}

template <bool C4>
static int i8s_launch(const I8sPlan& pl, hipStream_t st, bool sum_mode) {
  const ConvI8sParams& p = pl.p;
  const int nj = p.NJ, ct = sum_mode ? 1 : p.CT;
  const dim3 grid((unsigned)pl.grid), block(256);
#define EFFQ_I8S_CASE(J, C)                                                        \
  if (nj == J && ct == C) {                                                        \
    hipLaunchKernelGGL((k_conv3d_i8s<C4, J, C>), grid, block, 0, st, p);           \
    return EFFQ_OK;                                                                \
  }
  EFFQ_I8S_CASE(1, 1) EFFQ_I8S_CASE(1, 2) EFFQ_I8S_CASE(1, 4) EFFQ_I8S_CASE(1, 8)
  EFFQ_I8S_CASE(2, 1) EFFQ_I8S_CASE(2, 2) EFFQ_I8S_CASE(2, 4) EFFQ_I8S_CASE(2, 8)
  EFFQ_I8S_CASE(4, 1) EFFQ_I8S_CASE(4, 2) EFFQ_I8S_CASE(4, 4)
  EFFQ_I8S_CASE(8, 1) EFFQ_I8S_CASE(8, 2)
#undef EFFQ_I8S_CASE
  set_error("conv_i8s: no kernel for %d K steps x %d column tiles", nj, ct);
  return EFFQ_ERR_ARG;
}

static bool i8s_has_variant(int nj, int ct) {
  if (nj == 1 || nj == 2) return ct == 1 || ct == 2 || ct == 4 || ct == 8;
  if (nj == 4) return ct == 1 || ct == 2 || ct == 4;
  if (nj == 8) return ct == 1 || ct == 2;
  return false;
}

}  // namespace effq

using namespace effq;

extern "C" {

int effq_conv_i8s_supported(const effq_geom* g, int act_levels, int w_levels) {
  I8sPlan pl;
  if (g == nullptr) return 0;
  if (g->KD * g->KH * g->KW > 27 || g->KD * g->KH * g->KW * g->C1 > 256) return 0;
  if (!(g->C1 == 4 || (g->C1 % 16) == 0)) return 0;
  if (act_levels < 2 || act_levels > 256 || w_levels < 2 || w_levels > 256) return 0;
  const int nj = (g->KD * g->KH * g->KW * g->C1 + 31) / 32, ct = (g->C2 + 31) / 32;
  if (!i8s_has_variant(nj, ct)) return 0;
  if (i8s_plan(g, act_levels, w_levels, &pl) != EFFQ_OK) return 0;
  return 1;
}

size_t effq_conv_i8s_ws_bytes(const effq_geom* g, int act_levels, int w_levels) {
  I8sPlan pl;
  if (i8s_plan(g, act_levels, w_levels, &pl) != EFFQ_OK) return 0;
  return 256 + (size_t)pl.grid * 2 * sizeof(double) + pl.wq_bytes + 256 + (size_t)pl.p.CT * 32 * sizeof(int) + 256 +
         pl.su_bytes + 256;
}

int conv3d_calib_step_i8s(const uint8_t* xidx_ndhwc, const int8_t* Gq, const float* bias, const float* y_fp,
                          const effq_geom* g, const float* act_alpha_dev, int act_levels,
                          const effq_fp_state* w_state_dev, int w_levels, int prepare, double* sqerr_out, void* ws,
                          size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG(xidx_ndhwc && Gq && y_fp && g && act_alpha_dev && w_state_dev && sqerr_out && ws);
  EFFQ_CHECK_ARG(effq_conv_i8s_supported(g, act_levels, w_levels));
  I8sPlan pl;
  int rc = i8s_plan(g, act_levels, w_levels, &pl);
  if (rc != EFFQ_OK) return rc;
  const size_t need = effq_conv_i8s_ws_bytes(g, act_levels, w_levels);
  if (ws_bytes < need) {
    set_error("conv_i8s: workspace %zu < required %zu", ws_bytes, need);
    return EFFQ_ERR_WORKSPACE;
  }
  char* base = reinterpret_cast<char*>(ws);
  ConvI8sParams& p = pl.p;
  size_t off = 0;
  p.ticket = reinterpret_cast<unsigned int*>(base);
  off += 256;
  p.partials = reinterpret_cast<double*>(base + off);
  off += (size_t)pl.grid * 2 * sizeof(double);
  off = (off + 255) & ~(size_t)255;
  int8_t* wq = reinterpret_cast<int8_t*>(base + off);
  off += pl.wq_bytes;
  off = (off + 255) & ~(size_t)255;
  int* ksum = reinterpret_cast<int*>(base + off);
  off += (size_t)p.CT * 32 * sizeof(int);
  off = (off + 255) & ~(size_t)255;
  int* su = pl.su_bytes ? reinterpret_cast<int*>(base + off) : nullptr;
  p.x = xidx_ndhwc;
  p.wq = wq;
  p.bias = bias;
  p.y = y_fp;
  p.act_alpha = act_alpha_dev;
  p.wstate = w_state_dev;
  p.sqerr = sqerr_out;
  hipStream_t st = as_stream(stream);
  if (prepare && su != nullptr) {      // Su depends on the level ids only: once per layer
    hipLaunchKernelGGL(k_pack_ones_i8s, dim3(8), dim3(256), 0, st, wq, p.K, p.NJ);
    EFFQ_LAUNCH_CHECK();
    ConvI8sParams ps = p;
    ps.su_out = su;
    ps.ksum = nullptr;
    ps.su_in = nullptr;
    I8sPlan pls = pl;
    pls.p = ps;
    rc = pl.c4 ? i8s_launch<true>(pls, st, true) : i8s_launch<false>(pls, st, true);
    if (rc != EFFQ_OK) return rc;
    EFFQ_LAUNCH_CHECK();
  }
  // (the ticket of the last-block reduction is left at zero by the kernel that used it: the caller zero-fills
  //  the workspace once, effq_hip.h)
  {
    size_t nb = (pl.wq_bytes + 255) / 256;
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(k_pack_weight_i8s, dim3((unsigned)nb), dim3(256), 0, st, Gq, wq, ksum, p.C1, p.C2, p.T, p.K,
                       p.NJ, p.CT);
    EFFQ_LAUNCH_CHECK();
  }
  p.ksum = ksum;
  p.su_in = su;
  p.su_out = nullptr;
  rc = pl.c4 ? i8s_launch<true>(pl, st, false) : i8s_launch<false>(pl, st, false);
  if (rc != EFFQ_OK) return rc;
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
