// Proximal step of the ADMM: system matrix, fp64 SPD inverse (cached per rho by the caller) and the
// per-iteration  What = B * A^-1  product on the f32 matrix cores.
// Reference: solver.py:316-325 (getAB), :327-345 (solve; torch.linalg.solve = LU, every iteration).
//
// A = A0 + rho*I' + eta*I is symmetric positive definite (PSD Gram + eta*I, eta>0), and it changes
// only when rho does (5 values per layer), so the inverse is formed once per rho in fp64 by a
// blocked in-place Gauss-Jordan sweep (no pivoting needed for SPD) whose bulk is a rank-64 fp64
// MFMA update (v_mfma_f64_16x16x4_f64), then rounded once to fp32.
#include <cstdlib>
#include <vector>
#include "common.h"

namespace effq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

constexpr int NBK = 64;        // Gauss-Jordan block size
constexpr int LDA_S = 66;      // LDS leading dims (doubles): conflict-free ds_read_b64 operand fetches
constexpr int LDB_S = 80;

// A64 (npad x npad, row-major) = A0 + diag, identity on the padding
__global__ __launch_bounds__(256) void k_build_a64(const float* __restrict__ A0, int n, int npad, int has_bias,
                                                   double rho, double eta, double* __restrict__ A64) {
  const size_t tot = (size_t)npad * npad;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) {
    const int i = (int)(e / npad), j = (int)(e % npad);
    double v = 0.0;
    if (i < n && j < n) {
      v = (double)A0[(size_t)i * n + j];
      if (i == j) v += eta + ((has_bias && i == n - 1) ? 0.0 : rho);
    } else if (i == j) {
      v = 1.0;
    }
    A64[e] = v;
  }
}

// Step 1: Dinv = inv(A[kb:kb+64, kb:kb+64]) by unblocked in-place Gauss-Jordan (SPD: no pivoting).
// Lane = row, wave w = columns 16w..16w+15, the 16 values of a row live in registers.  Per pivot p the
// multiplier column and 1/pivot go through a parity-double-buffered LDS vector (ONE barrier per
// pivot), the pivot row is broadcast inside each wave with v_readlane.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// In-register Gauss-Jordan inverse of a 64 x 64 block held as lane = row, wave w = columns CPW w .. CPW w + CPW - 1
// (64 / CPW waves take part).  fcol: [2][NBK + 1] doubles of LDS.
template <int CPW>
__device__ __forceinline__ void gj_invert_regs(double (&reg)[CPW], double (*fcol)[NBK + 1], int lane, int wid) {
#pragma unroll
  for (int p = 0; p < NBK; ++p) {      // fully unrolled: pivot column / lane selectors become immediates
    const int par = p & 1, wp = p / CPW, jp = p % CPW;
    if (wid == wp) {
      double cp = reg[0];
#pragma unroll
      for (int j = 1; j < CPW; ++j) cp = (jp == j) ? reg[j] : cp;
      fcol[par][lane] = cp;
      if (lane == p) fcol[par][NBK] = 1.0 / cp;
    }
    lds_barrier();
    const double piv = fcol[par][NBK];
    const double f = fcol[par][lane];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const double rp = readlane_f64(reg[j], p) * piv;      // scaled pivot-row entry of this column
      const bool colp = (wid == wp) && (jp == j);
      double v;
      if (lane == p)
        v = colp ? piv : rp;
      else
        v = colp ? (-f * piv) : (reg[j] - f * rp);
      reg[j] = v;
    }
  }
}

// 16 waves, 4 columns of the 64 x 64 block per wave (lane = row): a pivot step costs every wave 4 column updates
// (the 4-wave version, 16 columns per wave, took 32 us per block - 64 serial steps of ~1500 cycles - on the critical
// path of every inverse).  Only the FIRST pivot block of an inverse is inverted by this kernel: the later ones are
// inverted by a workgroup of the preceding trailing update (k_gj_trail_sym), under that update.
constexpr int GJD_T = 1024, GJD_CPW = NBK / (GJD_T / 64);     // columns per wave = 4
__global__ __launch_bounds__(GJD_T) void k_gj_diag(const double* __restrict__ A, int npad, int kb,
                                                   double* __restrict__ Dinv) {
  __shared__ double fcol[2][NBK + 1];   // [parity][row] multipliers, [NBK] = 1/pivot
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double reg[GJD_CPW];
#pragma unroll
  for (int j = 0; j < GJD_CPW; ++j) reg[j] = A[(size_t)(kb + lane) * npad + kb + wid * GJD_CPW + j];
  gj_invert_regs<GJD_CPW>(reg, fcol, lane, wid);
#pragma unroll
  for (int j = 0; j < GJD_CPW; ++j) Dinv[lane * NBK + wid * GJD_CPW + j] = reg[j];
}

// 64x64x64 fp64 tile product on the matrix cores: acc += As(64x64) * Bs(64x64), one 32x32 quadrant per wave.
__device__ __forceinline__ void tile_mma64(const double* As, const double* Bs, f64x4 (&acc)[2][2]) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll 4
  for (int ks = 0; ks < NBK / 4; ++ks) {
    double a[2], b[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a[mi] = As[(wr * 32 + mi * 16 + lr) * LDA_S + ks * 4 + lk];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) b[ni] = Bs[(ks * 4 + lk) * LDB_S + wc * 32 + ni * 16 + lr];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  }
}

__device__ __forceinline__ void load_tile(double* dst, int ld, const double* src, size_t src_ld) {
  // 64x64 doubles, 256 threads, 16 per thread, coalesced along rows
  const int tid = threadIdx.x;
  for (int e = tid; e < NBK * NBK; e += 256) {
    const int r = e >> 6, c = e & 63;
    dst[r * ld + c] = src[(size_t)r * src_ld + c];
  }
}

constexpr int GJ_CT = 4;     // column blocks per workgroup of the trailing update

// ---- symmetric block Gauss-Jordan ---------------------------------------------------------------------
// A is symmetric, and block Gauss-Jordan keeps a signed symmetry: with S = the pivot blocks already processed
// and U the rest, M_ji = M_ij^T when i, j are both in S or both in U, and M_ji = -M_ij^T otherwise.  Only the
// UPPER block triangle is therefore stored and updated: half the flops and - what matters, the rank-64 update
// is HBM-bound - half the bytes per elimination step.  Per pivot block k:
//   X_t = M_tk for every block row t != k   (t < k: block (t,k);  t > k: block (k,t)^T)
//   Z_t = X_t * Dinv                        (Dinv = inv(M_kk), k_gj_diag)
//   M_ij -= Z_i * (s_j X_j^T)  for i <= j, i, j != k,  s_j = -1 for j < k (j in S, k in U), +1 for j > k
//   new M_tk = -Z_t (t < k),  new M_kt = Z_t^T (t > k),  new M_kk = Dinv.
// k_gj_panel forms NZ = -Z (npad x 64, the A operand of the update) and XT = s * X^T (64 x npad, the B operand
// in MFMA layout, coalesced) and writes the new pivot row/column; k_gj_trail_sym is the old trailing update on
// the upper triangle with those operands.
__global__ __launch_bounds__(256) void k_gj_panel(double* __restrict__ A, int npad, int kb,
                                                  const double* __restrict__ Dinv, double* __restrict__ NZ,
                                                  double* __restrict__ XT) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* As = sm;                  // X_t  [64][LDA_S]
  double* Bs = sm + NBK * LDA_S;    // Dinv [64][LDB_S]
  const int kblk = kb / NBK;
  const int t = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  if (t == kblk) {                  // A[kb,kb] = Dinv
    for (int e = tid; e < NBK * NBK; e += 256) A[(size_t)(kb + (e >> 6)) * npad + kb + (e & 63)] = Dinv[e];
    return;
  }
  const bool upper = t < kblk;      // X_t is the stored block (t,k); otherwise the transpose of block (k,t)
  const double sgn = upper ? -1.0 : 1.0;
  for (int e = tid; e < NBK * NBK; e += 256) {
    const int r = e >> 6, c = e & 63;
    // stored element (r,c) of the source block; for the transposed case it is X[c][r]
    const double v = upper ? A[(size_t)(t * NBK + r) * npad + kb + c] : A[(size_t)(kb + r) * npad + (size_t)t * NBK + c];
    if (upper) {
      As[r * LDA_S + c] = v;                                        // X[r][c]
    } else {
      As[c * LDA_S + r] = v;                                        // X[c][r] = block(k,t)[r][c]
    }
  }
  load_tile(Bs, LDB_S, Dinv, NBK);
  lds_barrier();
  // XT[kk][t*64 + r] = s * X[r][kk]   (coalesced along r)
  for (int e = tid; e < NBK * NBK; e += 256) {
    const int kk = e >> 6, r = e & 63;
    XT[(size_t)kk * npad + (size_t)t * NBK + r] = sgn * As[r * LDA_S + kk];
  }
  f64x4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
  tile_mma64(As, Bs, acc);          // Z = X * Dinv
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * 32 + mi * 16 + lk + 4 * r;
        const int col = wc * 32 + ni * 16 + lr;
        const double z = acc[mi][ni][r];
        NZ[(size_t)(t * NBK + row) * NBK + col] = -z;
        if (upper)
          A[(size_t)(t * NBK + row) * npad + kb + col] = -z;          // new M_tk
        else
          A[(size_t)(kb + col) * npad + (size_t)t * NBK + row] = z;   // new M_kt = Z^T
      }
}

// blockIdx.y = 0 is the LOOK-AHEAD row: its first workgroup forms the pivot block of the NEXT step - tile (k+1, k+1)
// after this step's update, which it computes itself from the not yet updated tile (the regular workgroups leave that
// tile alone: k_gj_panel of the next step overwrites it with its inverse anyway) - and inverts it in registers, while the
// other workgroups update the rest of the triangle.  The 64 serial pivots (~30 us with 4 waves) that used to sit
// between two trailing updates as a kernel of their own (k_gj_diag: 24 us + a launch gap per 64 columns) are hidden
// under the update.  blockIdx.y = ib + 1 for the regular rows.
__global__ __launch_bounds__(256) void k_gj_trail_sym(double* __restrict__ A, int npad, int kb,
                                                      const double* __restrict__ NZ, const double* __restrict__ XT,
                                                      double* __restrict__ Dinv_next) {
  __shared__ __attribute__((aligned(16))) double As[NBK * LDA_S];
  __shared__ double fcol[2][NBK + 1];
  const int kblk = kb / NBK, nblk = npad / NBK;
  const int kn = kblk + 1;                                         // the next pivot block
  const bool ahead = Dinv_next != nullptr && kn < nblk;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int lr = lane & 15, lk = lane >> 4;
  if (blockIdx.y == 0) {
    if (blockIdx.x != 0 || !ahead) return;
    for (int e = tid; e < NBK * NBK; e += 256) {
      const int r = e >> 6, c = e & 63;
      As[r * LDA_S + c] = NZ[(size_t)(kn * NBK + r) * NBK + c];      // -Z_kn
    }
    lds_barrier();
    const double* Cb = A + (size_t)(kn * NBK + wr * 32) * npad + (size_t)kn * NBK + wc * 32;
    const double* Rb = XT + (size_t)kn * NBK + wc * 32;
    f64x4 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = Cb[(size_t)(mi * 16 + lk + 4 * r) * npad + ni * 16 + lr];
#pragma unroll 4
    for (int ks = 0; ks < NBK / 4; ++ks) {
      double a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = As[(wr * 32 + mi * 16 + lr) * LDA_S + ks * 4 + lk];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[ni] = Rb[(size_t)(ks * 4 + lk) * npad + ni * 16 + lr];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    lds_barrier();                                                 // As (the A operand) has been consumed
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          As[(wr * 32 + mi * 16 + lk + 4 * r) * LDA_S + wc * 32 + ni * 16 + lr] = acc[mi][ni][r];
    lds_barrier();
    constexpr int CPW = NBK / 4;                                   // 4 waves, 16 columns each, lane = row
    double reg[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) reg[j] = As[lane * LDA_S + wid * CPW + j];
    gj_invert_regs<CPW>(reg, fcol, lane, wid);
#pragma unroll
    for (int j = 0; j < CPW; ++j) Dinv_next[lane * NBK + wid * CPW + j] = reg[j];
    return;
  }
  const int ib = (int)blockIdx.y - 1;
  if (ib == kblk) return;
  if ((int)(blockIdx.x * GJ_CT + GJ_CT - 1) < ib) return;        // the whole strip lies below the diagonal
  for (int e = tid; e < NBK * NBK; e += 256) {
    const int r = e >> 6, c = e & 63;
    As[r * LDA_S + c] = NZ[(size_t)(ib * NBK + r) * NBK + c];      // -Z_i
  }
  lds_barrier();
  // The strip's tiles in turn, the C tile of the NEXT one fetched into a second register set before the MFMAs of the
  // current one (the update moves 64 KB per 0.5 MFLOP tile: it is bound by memory latency, not by the matrix cores)
  auto live = [&](int t) -> bool {
    const int jb = blockIdx.x * GJ_CT + t;
    return t < GJ_CT && jb < nblk && jb != kblk && jb >= ib && !(ahead && ib == kn && jb == kn);
  };
  auto fetch = [&](int t, f64x4 (&c)[2][2]) {
    const int jb = blockIdx.x * GJ_CT + t;
    const double* Cb = A + (size_t)(ib * NBK + wr * 32) * npad + (size_t)jb * NBK + wc * 32;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[mi][ni][r] = Cb[(size_t)(mi * 16 + lk + 4 * r) * npad + ni * 16 + lr];
  };
  f64x4 nxt[2][2];
  int t = 0;
  while (t < GJ_CT && !live(t)) ++t;
  if (t < GJ_CT) fetch(t, nxt);
  while (t < GJ_CT) {
    const int jb = blockIdx.x * GJ_CT + t;
    f64x4 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = nxt[mi][ni];
    int tn = t + 1;
    while (tn < GJ_CT && !live(tn)) ++tn;
    if (tn < GJ_CT) fetch(tn, nxt);
    double* Cb = A + (size_t)(ib * NBK + wr * 32) * npad + (size_t)jb * NBK + wc * 32;
    const double* Rb = XT + (size_t)jb * NBK + wc * 32;
#pragma unroll 4
    for (int ks = 0; ks < NBK / 4; ++ks) {
      double a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = As[(wr * 32 + mi * 16 + lr) * LDA_S + ks * 4 + lk];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[ni] = Rb[(size_t)(ks * 4 + lk) * npad + ni * 16 + lr];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cb[(size_t)(mi * 16 + lk + 4 * r) * npad + ni * 16 + lr] = acc[mi][ni][r];
    t = tn;
  }
}

__global__ __launch_bounds__(256) void k_a64_to_f32(const double* __restrict__ A64, int n, int npad,
                                                    float* __restrict__ Ainv, int lda) {
  const size_t tot = (size_t)n * lda;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) {
    const int i = (int)(e / lda), j = (int)(e % lda);
    // only the upper block triangle is maintained (symmetric Gauss-Jordan): mirror it.  Columns >= n are padding.
    const int a = (i < j) ? i : j, b = (i < j) ? j : i;
    Ainv[e] = (j < n) ? (float)A64[(size_t)a * npad + b] : 0.0f;
  }
}

// ======================================================================================================================
// The same symmetric block Gauss-Jordan with pivot blocks of 256 rows (WB = 4 of the 64-blocks above).
//
// The rank-64 sweep reads and writes the stored triangle once per 64 columns: 8 flop per byte moved, an HBM / Infinity-
// Cache stream that left the fp64 matrix cores at a third of their rate (n = 6913: 12.9 ms alone, 27 - 33 ms beside the
// ADMM chain).  With a 256-row pivot block P the elimination step is the same algebra on bigger blocks,
//   X_t = M_tP (t outside P),  Z_t = X_t inv(M_PP),  M_ij -= Z_i (s_j X_j^T),  new M_tP = -+Z_t,  new M_PP = inv(M_PP),
// and the update of the triangle is a rank-256 product: 32 flop per byte of C, four times fewer passes over the matrix.
// Per pivot block:
//   k_gj_panel_w   Z = X inv(M_PP) for every 64-row block outside P -> the operands of the update, both K-major
//                  (NZT = -Z^T, XTW = s X^T: [256][ldx]), nothing written to the matrix yet (the workgroups of one block
//                  row read each other's blocks);
//   k_gj_wback_w   the new pivot rows / columns and inv(M_PP) into the matrix;
//   k_gj_big       the update, 128 x 128 tiles of the upper triangle, K = 256 staged through LDS in chunks of 16;
//   the NEXT pivot block: k_gj_big on its own 256 x 256 tile alone (out of place, into the scratch matrix D), the rank-64
//                  sweep above on D (a 256 x 256 matrix of its own), k_mirror_blocks.  None of that depends on the big
//                  update, which leaves that tile alone: it runs on a helper stream BESIDE the big update, so the serial
//                  pivot work (4 x 64 dependent pivots per pivot block) is off the critical path for n >= ~5000.
constexpr int WB = 4, WK = WB * NBK;
constexpr int BG_T = 128, BG_KC = 16, BG_LD = 144;      // macro tile, K chunk, LDS row pitch (doubles): see the bank notes below
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct GjBig {
  const double* Cin;      // the matrix, addressed in place: block (i, j) at Cin[(i*64)*ldin + j*64]
  size_t ldin;
  double* Cout;           // output: Cout[(i*64 - orow)*ldout + (j*64 - ocol)]  (Cin itself for the in-place update)
  size_t ldout;
  int orow, ocol;
  const double* NZT;      // [kdim][ldx]: -Z^T  (A operand, K-major)
  const double* XTW;      // [kdim][ldx]: s X^T (B operand)
  size_t ldx;
  int kdim;               // 64 * (64-blocks of the pivot); 0 = plain copy
  int nblk;               // 64-blocks per side
  int i0m, j0m;           // macro-tile offset of the grid
  int skip_lo, skip_hi;   // 64-blocks of the pivot: rows / columns left alone
  int la_lo, la_hi;       // 64-blocks of the NEXT pivot: blocks with BOTH indices inside are left alone
};

// One 128 x 128 macro tile per workgroup, one 64 x 64 block per wave (16 accumulator tiles of 16 x 16), K in chunks of 16
// through a double-buffered LDS stage: global -> registers (issued before the MFMAs of the current chunk) -> LDS, one
// barrier per chunk.  Both operands are K-major ([k][row]): an operand fetch is one ds_read_b64 per lane, lanes 0-15 /
// 16-31 of a half-wave read rows k and k+1, 144 doubles = 32 banks (mod 64) apart: conflict-free; the staging stores
// are ds_write_b128 with the 8 lanes of a group 16 B apart.  64 MFMAs (64 cycles each) per wave and chunk against 32
// operand reads and 4 + 4 staging stores: the kernel is paced by the matrix pipe, and by its C traffic (16 B per 512 flop).
__global__ __launch_bounds__(256, 2) void k_gj_big(const GjBig p) {
  __shared__ __attribute__((aligned(16))) double As[2][BG_KC * BG_LD];
  __shared__ __attribute__((aligned(16))) double Bs[2][BG_KC * BG_LD];
  const int I = p.i0m + (int)blockIdx.y, J = p.j0m + (int)blockIdx.x;
  if (J < I) return;
  auto blk_live = [&](int bi, int bj) -> bool {
    if (bi > bj || bj >= p.nblk) return false;
    if ((bi >= p.skip_lo && bi < p.skip_hi) || (bj >= p.skip_lo && bj < p.skip_hi)) return false;
    if (bi >= p.la_lo && bi < p.la_hi && bj >= p.la_lo && bj < p.la_hi) return false;
    return true;
  };
  if (!(blk_live(2 * I, 2 * J) || blk_live(2 * I, 2 * J + 1) || blk_live(2 * I + 1, 2 * J) || blk_live(2 * I + 1, 2 * J + 1)))
    return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int bi = 2 * I + wr, bj = 2 * J + wc;
  const bool live = blk_live(bi, bj);
  f64x4 acc[4][4];
  if (live) {
    const double* Cb = p.Cin + (size_t)bi * NBK * p.ldin + (size_t)bj * NBK;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = Cb[(size_t)(mi * 16 + lk + 4 * r) * p.ldin + ni * 16 + lr];
  } else {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
  }
  const int nch = p.kdim / BG_KC;
  if (nch > 0) {
    const int sk = tid >> 4, sc = (tid & 15) * 2;       // staging: K row, first column (doubles) of the 4 pieces 32 apart
    const double* Ag = p.NZT + (size_t)sk * p.ldx + (size_t)I * BG_T + sc;
    const double* Bg = p.XTW + (size_t)sk * p.ldx + (size_t)J * BG_T + sc;
    f64x2 ra[4], rb[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      ra[v] = *reinterpret_cast<const f64x2*>(Ag + v * 32);
      rb[v] = *reinterpret_cast<const f64x2*>(Bg + v * 32);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      *reinterpret_cast<f64x2*>(&As[0][sk * BG_LD + sc + v * 32]) = ra[v];
      *reinterpret_cast<f64x2*>(&Bs[0][sk * BG_LD + sc + v * 32]) = rb[v];
    }
    lds_barrier();
    for (int c = 0; c < nch; ++c) {
      const int buf = c & 1;
      if (c + 1 < nch) {
        const size_t off = (size_t)(c + 1) * BG_KC * p.ldx;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          ra[v] = *reinterpret_cast<const f64x2*>(Ag + off + v * 32);
          rb[v] = *reinterpret_cast<const f64x2*>(Bg + off + v * 32);
        }
      }
      if (live) {
#pragma unroll
        for (int ks = 0; ks < BG_KC / 4; ++ks) {
          double a[4], b[4];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) a[mi] = As[buf][(ks * 4 + lk) * BG_LD + wr * 64 + mi * 16 + lr];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) b[ni] = Bs[buf][(ks * 4 + lk) * BG_LD + wc * 64 + ni * 16 + lr];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
      }
      if (c + 1 < nch) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          *reinterpret_cast<f64x2*>(&As[buf ^ 1][sk * BG_LD + sc + v * 32]) = ra[v];
          *reinterpret_cast<f64x2*>(&Bs[buf ^ 1][sk * BG_LD + sc + v * 32]) = rb[v];
        }
      }
      lds_barrier();
    }
  }
  if (live) {
    double* Ob = p.Cout + ((size_t)bi * NBK - p.orow) * p.ldout + ((size_t)bj * NBK - p.ocol);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ob[(size_t)(mi * 16 + lk + 4 * r) * p.ldout + ni * 16 + lr] = acc[mi][ni][r];
  }
}

// Z_t[:, c] = sum_a X_t[:, a] * Dw[a, c] for the 64-row block t outside the pivot (blocks kb0 .. kb0 + m - 1) and the
// pivot's column block c; Dw = the full symmetric inverse of the pivot block ([64 m][64 m]).  Writes only the operand
// panels: XTW[c*64 + kk][t*64 + r] = s X[r][c*64 + kk], NZT[c*64 + col][t*64 + row] = -Z[row][col].
__global__ __launch_bounds__(256) void k_gj_panel_w(const double* __restrict__ A, int npad, int kb0, int m,
                                                    const double* __restrict__ Dw, double* __restrict__ NZT,
                                                    double* __restrict__ XTW, size_t ldx) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* As = sm;                  // X_t[:, a]  [64][LDA_S]
  double* Bs = sm + NBK * LDA_S;    // Dw[a, c]   [64][LDB_S]
  const int t = blockIdx.x, c = blockIdx.y;
  if (t >= kb0 && t < kb0 + m) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const bool upper = t < kb0;
  const double sgn = upper ? -1.0 : 1.0;
  const int ldd = m * NBK;
  f64x4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
  for (int a = 0; a < m; ++a) {
    for (int e = tid; e < NBK * NBK; e += 256) {
      const int r = e >> 6, cc = e & 63;
      if (upper)
        As[r * LDA_S + cc] = A[(size_t)(t * NBK + r) * npad + (size_t)(kb0 + a) * NBK + cc];       // X[r][cc]
      else
        As[cc * LDA_S + r] = A[(size_t)((kb0 + a) * NBK + r) * npad + (size_t)t * NBK + cc];       // X[cc][r]
    }
    load_tile(Bs, LDB_S, Dw + (size_t)a * NBK * ldd + (size_t)c * NBK, (size_t)ldd);
    lds_barrier();
    if (a == c)
      for (int e = tid; e < NBK * NBK; e += 256) {
        const int kk = e >> 6, r = e & 63;
        XTW[(size_t)(c * NBK + kk) * ldx + (size_t)t * NBK + r] = sgn * As[r * LDA_S + kk];
      }
    tile_mma64(As, Bs, acc);
    lds_barrier();
  }
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        As[(wr * 32 + mi * 16 + lk + 4 * r) * LDA_S + wc * 32 + ni * 16 + lr] = acc[mi][ni][r];     // Z[row][col]
  lds_barrier();
  for (int e = tid; e < NBK * NBK; e += 256) {
    const int kk = e >> 6, r = e & 63;
    NZT[(size_t)(c * NBK + kk) * ldx + (size_t)t * NBK + r] = -As[r * LDA_S + kk];
  }
}

// The pivot rows / columns after the step: block (t, kb0 + c) = -Z_t[:, c] for t above the pivot, block (kb0 + c, t) =
// Z_t[:, c]^T for t below it, and the pivot block itself = Dw (upper 64-blocks).
__global__ __launch_bounds__(256) void k_gj_wback_w(double* __restrict__ A, int npad, int kb0, int m,
                                                    const double* __restrict__ Dw, const double* __restrict__ NZT,
                                                    size_t ldx) {
  __shared__ double Ts[NBK * (NBK + 1)];
  const int t = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
  const int ldd = m * NBK;
  if (t >= kb0 && t < kb0 + m) {
    const int a = t - kb0;
    if (a > c) return;
    for (int e = tid; e < NBK * NBK; e += 256) {
      const int r = e >> 6, cc = e & 63;
      A[(size_t)(t * NBK + r) * npad + (size_t)(kb0 + c) * NBK + cc] = Dw[(size_t)(a * NBK + r) * ldd + (size_t)c * NBK + cc];
    }
    return;
  }
  if (t < kb0) {
    for (int e = tid; e < NBK * NBK; e += 256) {
      const int kk = e >> 6, r = e & 63;
      Ts[kk * (NBK + 1) + r] = NZT[(size_t)(c * NBK + kk) * ldx + (size_t)t * NBK + r];      // -Z[r][kk]
    }
    __syncthreads();
    for (int e = tid; e < NBK * NBK; e += 256) {
      const int r = e >> 6, kk = e & 63;
      A[(size_t)(t * NBK + r) * npad + (size_t)(kb0 + c) * NBK + kk] = Ts[kk * (NBK + 1) + r];
    }
  } else {
    for (int e = tid; e < NBK * NBK; e += 256) {
      const int kk = e >> 6, r = e & 63;
      A[(size_t)((kb0 + c) * NBK + kk) * npad + (size_t)t * NBK + r] = -NZT[(size_t)(c * NBK + kk) * ldx + (size_t)t * NBK + r];
    }
  }
}

// The next pivot block of the 128-row sweep in ONE launch of ONE workgroup (16 waves): the 128 x 128 block is held in
// LDS (133 KB), the pending rank-kdim update of its own tile applied while it is loaded (operands straight from the
// K-major panels), then inverted by blocks: D0 = inv(M00), Z = M10 D0, S = M11 - Z M01, inv(S), W = inv(S) Z,
//   inverse = [[D0 + Z^T W, -W^T], [-W, inv(S)]],
// the two 64 x 64 inversions in registers (gj_invert_regs, 16 waves), the four 64^3 products on the matrix cores (one
// 16 x 16 tile per wave).  Replaces a chain of ~8 launches per pivot block (tile update, diag, panel, trailing, mirror).
constexpr int PF_LD = 130;      // LDS pitch (doubles): operand fetches indexed [row = lane & 15][k = lane >> 4] are conflict-free
__device__ __forceinline__ f64x4 mm16(f64x4 acc, const double* Aop, int a_rs, int a_cs, const double* Bop, int b_rs,
                                      int b_cs, int K, int lr, int lk) {
#pragma unroll 4
  for (int ks = 0; ks < K / 4; ++ks) {
    const double a = Aop[lr * a_rs + (ks * 4 + lk) * a_cs];
    const double b = Bop[(ks * 4 + lk) * b_rs + lr * b_cs];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

__global__ __launch_bounds__(1024) void k_gj_pivot_fused(const double* __restrict__ A, int npad, int k0, int mk,
                                                         const double* __restrict__ NZT, const double* __restrict__ XTW,
                                                         size_t ldx, int kdim, double* __restrict__ Dw) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* M = sm;                                                        // [128][PF_LD]
  double(*fcol)[NBK + 1] = reinterpret_cast<double(*)[NBK + 1]>(sm + 2 * NBK * PF_LD);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tr = wid >> 2, tc = wid & 3;
  const int lr = lane & 15, lk = lane >> 4;
  // ---- the block with the pending update applied: upper 64-blocks computed, the lower one mirrored
  for (int bi = 0; bi < mk; ++bi)
    for (int bj = bi; bj < mk; ++bj) {
      const size_t row0 = (size_t)(k0 + bi) * NBK + tr * 16, col0 = (size_t)(k0 + bj) * NBK + tc * 16;
      f64x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = A[(row0 + lk + 4 * r) * npad + col0 + lr];
      const double* Ag = NZT + row0 + lr;
      const double* Bg = XTW + col0 + lr;
#pragma unroll 8
      for (int ks = 0; ks < kdim / 4; ++ks) {
        const double a = Ag[(size_t)(ks * 4 + lk) * ldx];
        const double b = Bg[(size_t)(ks * 4 + lk) * ldx];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = bi * NBK + tr * 16 + lk + 4 * r, cc = bj * NBK + tc * 16 + lr;
        M[rr * PF_LD + cc] = acc[r];
        if (bi < bj) M[cc * PF_LD + rr] = acc[r];
      }
    }
  lds_barrier();
  constexpr int CPW = NBK / 16;                                          // 16 waves, 4 columns each, lane = row
  auto invert64 = [&](int off) {
    double reg[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) reg[j] = M[(off + lane) * PF_LD + off + wid * CPW + j];
    gj_invert_regs<CPW>(reg, fcol, lane, wid);
#pragma unroll
    for (int j = 0; j < CPW; ++j) M[(off + lane) * PF_LD + off + wid * CPW + j] = reg[j];
    lds_barrier();
  };
  auto zero = []() {
    f64x4 z;
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = 0.0;
    return z;
  };
  auto tile_store = [&](int rb, int cb, f64x4 v, double sgn) {
#pragma unroll
    for (int r = 0; r < 4; ++r) M[(rb + tr * 16 + lk + 4 * r) * PF_LD + cb + tc * 16 + lr] = sgn * v[r];
  };
  auto tile_load = [&](int rb, int cb) {
    f64x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = M[(rb + tr * 16 + lk + 4 * r) * PF_LD + cb + tc * 16 + lr];
    return v;
  };
  invert64(0);                                                            // M00 <- D0
  if (mk == 2) {
    // Z = M10 D0 (into the M10 region, once every wave has read its operands)
    f64x4 z = mm16(zero(), M + (NBK + tr * 16) * PF_LD, PF_LD, 1, M + tc * 16, PF_LD, 1, NBK, lr, lk);
    lds_barrier();
    tile_store(NBK, 0, z, 1.0);
    lds_barrier();
    // S = M11 - Z M01, in place
    f64x4 sacc = tile_load(NBK, NBK);
#pragma unroll
    for (int r = 0; r < 4; ++r) sacc[r] = -sacc[r];
    sacc = mm16(sacc, M + (NBK + tr * 16) * PF_LD, PF_LD, 1, M + NBK + tc * 16, PF_LD, 1, NBK, lr, lk);
    tile_store(NBK, NBK, sacc, -1.0);
    lds_barrier();
    invert64(NBK);                                                        // M11 <- inv(S)
    // W = inv(S) Z -> the M01 region (M01 itself is no longer needed)
    f64x4 w = mm16(zero(), M + (NBK + tr * 16) * PF_LD + NBK, PF_LD, 1, M + NBK * PF_LD + tc * 16, PF_LD, 1, NBK, lr, lk);
    tile_store(0, NBK, w, 1.0);
    lds_barrier();
    // M00 <- D0 + Z^T W   (A operand (m, k) = Z[k][m])
    f64x4 d = tile_load(0, 0);
    d = mm16(d, M + NBK * PF_LD + tr * 16, 1, PF_LD, M + NBK + tc * 16, PF_LD, 1, NBK, lr, lk);
    tile_store(0, 0, d, 1.0);
    lds_barrier();
  }
  // ---- out: [[M00, -W^T], [-W, M11]] with W in the M01 region
  const int nd = mk * NBK;
  for (int e = tid; e < nd * nd; e += 1024) {
    const int i = e / nd, j = e % nd;
    double v;
    if (i < NBK && j >= NBK)
      v = -M[(j - NBK) * PF_LD + NBK + i];
    else if (i >= NBK && j < NBK)
      v = -M[(i - NBK) * PF_LD + NBK + j];
    else
      v = M[i * PF_LD + j];
    Dw[e] = v;
  }
}

// the rank-64 sweep keeps the upper 64-block triangle: fill in the blocks below the diagonal (the finished inverse is
// symmetric)
__global__ __launch_bounds__(256) void k_mirror_blocks(double* __restrict__ D, int nd) {
  const size_t tot = (size_t)nd * nd;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / nd), j = (int)(e % nd);
    if ((i / NBK) > (j / NBK)) D[e] = D[(size_t)j * nd + i];
  }
}

// Bm[c2][ldb] = B0 + eta*[W0|b0] + rho*[G-dual|0]     (solver.py:316-322, fp32, reference op order)
__global__ __launch_bounds__(256) void k_build_b(const float* __restrict__ B0, const float* __restrict__ W0,
                                                 const float* __restrict__ b0, const float* __restrict__ G,
                                                 const float* __restrict__ dual, int c2, int n, int has_bias,
                                                 float rho, float eta, float* __restrict__ Bm, int ldb, int c2p,
                                                 const float* __restrict__ wprev, float shift) {
  const size_t tot = (size_t)c2p * ldb;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int nw = n - has_bias;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) {
    const int r = (int)(e / ldb), k = (int)(e % ldb);
    float v = 0.0f;
    if (r < c2 && k < n) {
      if (k < nw) {
        const size_t wi = (size_t)r * nw + k;
        v = B0[(size_t)r * n + k] + eta * W0[wi];
        v = v + rho * (G[wi] - dual[wi]);
        if (wprev != nullptr) v = v + shift * wprev[wi];     // effq_prox_solve_shifted: + d * W_prev * I'
      } else {
        v = B0[(size_t)r * n + k] + eta * b0[r];
      }
    }
    Bm[e] = v;
  }
}

// The same, four columns per thread (row per blockIdx.y): for weight rows that are a multiple of 4 long (every layer of the
// shipped nets but a 1-channel first conv) - 16-byte loads, no integer division per element.
__global__ __launch_bounds__(256) void k_build_b4(const float* __restrict__ B0, const float* __restrict__ W0,
                                                  const float* __restrict__ b0, const float* __restrict__ G,
                                                  const float* __restrict__ dual, int c2, int n, int has_bias,
                                                  float rho, float eta, float* __restrict__ Bm, int ldb,
                                                  const float* __restrict__ wprev, float shift) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams

  const int r = blockIdx.y;
  const int nw = n - has_bias;
  float* out = Bm + (size_t)r * ldb;
  for (int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4; k < ldb; k += gridDim.x * blockDim.x * 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < c2 && k < nw) {                       // nw % 4 == 0: a group of four never straddles the bias column
      const size_t wi = (size_t)r * nw + k;
      const float* bp = B0 + (size_t)r * n + k;   // (rows of B0 are n long: not 16-byte aligned in general)
      const float4 w0 = *reinterpret_cast<const float4*>(W0 + wi), g = *reinterpret_cast<const float4*>(G + wi),
                   du = *reinterpret_cast<const float4*>(dual + wi);
      v.x = bp[0] + eta * w0.x;
      v.y = bp[1] + eta * w0.y;
      v.z = bp[2] + eta * w0.z;
      v.w = bp[3] + eta * w0.w;
      v.x = v.x + rho * (g.x - du.x);
      v.y = v.y + rho * (g.y - du.y);
      v.z = v.z + rho * (g.z - du.z);
      v.w = v.w + rho * (g.w - du.w);
      if (wprev != nullptr) {
        const float4 wp = *reinterpret_cast<const float4*>(wprev + wi);
        v.x = v.x + shift * wp.x;
        v.y = v.y + shift * wp.y;
        v.z = v.z + shift * wp.z;
        v.w = v.w + shift * wp.w;
      }
    } else if (r < c2 && k == nw && has_bias) {
      v.x = B0[(size_t)r * n + k] + eta * b0[r];
    }
    *reinterpret_cast<float4*>(out + k) = v;
  }
}

__global__ __launch_bounds__(256) void k_prox_reduce4(const float* __restrict__ part, int ldp, int nsplit, int c2, int n,
                                                      int has_bias, float* __restrict__ wstar, float* __restrict__ bstar) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams

  const int r = blockIdx.y;
  const int nw = n - has_bias;
  const size_t tot = (size_t)c2 * ldp;
  const float* row = part + (size_t)r * ldp;
  for (int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4; k < ldp; k += gridDim.x * blockDim.x * 4) {
    if (k >= n) continue;
    float4 v = *reinterpret_cast<const float4*>(row + k);
    for (int z = 1; z < nsplit; ++z) {            // slice order: deterministic
      const float4 t = *reinterpret_cast<const float4*>(row + (size_t)z * tot + k);
      v.x += t.x;
      v.y += t.y;
      v.z += t.z;
      v.w += t.w;
    }
    if (k < nw)
      *reinterpret_cast<float4*>(wstar + (size_t)r * nw + k) = v;
    else
      bstar[r] = v.x;
  }
}

// What = Bm * Ainv on the f32 matrix cores.  Ainv is exactly symmetric, so What[r][c] = sum_k Bm[r][k] *
// Ainv[c][k]: BOTH operands are read along K (contiguous, 16-byte loads), staged as [row][32+4] tiles in
// LDS (conflict-free ds_read_b128, 4 MFMAs per pair of reads) with the next K tile prefetched into
// registers under the MFMAs of the current one.  Wave tile = (32*MT) x (32*NTN); workgroup tile =
// (32*MT*WM) x (32*NTN*WN), 4 waves.
constexpr int PBK = 32, PLD = PBK + 4;

template <int MT, int WM, int WN, int NTN>
__global__ __launch_bounds__(256) void k_prox_gemm(const float* __restrict__ Bm, int ldb, const float* __restrict__ Ainv,
                                                   int lda, int n, int c2, int has_bias, float* __restrict__ wstar,
                                                   float* __restrict__ bstar, float* __restrict__ part, int ldp) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams

  constexpr int BM = 32 * MT * WM, BN = 32 * WN * NTN;
  constexpr int NA = BM * 8 / 256, NB = BN * 8 / 256;   // 16-byte loads per thread per K tile
  static_assert(WM * WN == 4 && NA >= 1 && NB >= 1, "4 waves");
  __shared__ __attribute__((aligned(16))) float As[BM * PLD];
  __shared__ __attribute__((aligned(16))) float Bs[BN * PLD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;

  f32x16 acc[MT][NTN];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

  // The prefetch is written out twice (prologue and loop) rather than as a lambda over ra/rb: captured by
  // reference the arrays stay in scratch memory and every prefetch is waited for and spilled on the spot.
  // Columns >= n are clamped (their outputs are dropped below): an unconditional load keeps the prefetch
  // asynchronous, a branch around it makes hipcc wait for it inside the branch.
  f32x4 ra[NA], rb[NB];
  const float* pa[NA];
  const float* pb[NB];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int u = tid + q * 256, r = u >> 3, c4 = u & 7;
    pa[q] = Bm + (size_t)(row0 + r) * ldb + c4 * 4;   // Bm is zero padded
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int u = tid + q * 256, r = u >> 3, c4 = u & 7;
    pb[q] = Ainv + (size_t)min(col0 + r, n - 1) * lda + c4 * 4;   // lda padded
  }
#define EFFQ_PROX_FETCH(k0)                                                         \
  {                                                                                 \
    _Pragma("unroll") for (int q = 0; q < NA; ++q)                                  \
        ra[q] = *reinterpret_cast<const f32x4*>(pa[q] + (k0));                      \
    _Pragma("unroll") for (int q = 0; q < NB; ++q)                                  \
        rb[q] = *reinterpret_cast<const f32x4*>(pb[q] + (k0));                      \
  }
  // split K: slice z of gridDim.z takes K tiles [kt0, kt1); with more than one slice the partial tile goes
  // to part[z] and k_prox_reduce adds the slices in a fixed order
  const int nkt = ldb / PBK;   // ldb is a multiple of 32
  const int kt0 = (int)(((long long)nkt * blockIdx.z) / gridDim.z);
  const int nk = (int)(((long long)nkt * (blockIdx.z + 1)) / gridDim.z);
  EFFQ_PROX_FETCH(kt0 * PBK)
  for (int kt = kt0; kt < nk; ++kt) {
    lds_barrier();
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int u = tid + q * 256;
      *reinterpret_cast<f32x4*>(&As[(u >> 3) * PLD + (u & 7) * 4]) = ra[q];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int u = tid + q * 256;
      *reinterpret_cast<f32x4*>(&Bs[(u >> 3) * PLD + (u & 7) * 4]) = rb[q];
    }
    lds_barrier();
    EFFQ_PROX_FETCH(min(kt + 1, nk - 1) * PBK)   // unconditional: the last one is a harmless re-read
    __builtin_amdgcn_sched_barrier(0);           // keep the loads ahead of the MFMAs they hide under
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a[MT], b[NTN];
#pragma unroll
      for (int m = 0; m < MT; ++m)
        a[m] = *reinterpret_cast<const f32x4*>(&As[((wm * MT + m) * 32 + li) * PLD + q * 8 + 4 * lh]);
#pragma unroll
      for (int t = 0; t < NTN; ++t)
        b[t] = *reinterpret_cast<const f32x4*>(&Bs[((wn * NTN + t) * 32 + li) * PLD + q * 8 + 4 * lh]);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTN; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][e], b[t][e], acc[m][t], 0, 0, 0);
    }
  }
#undef EFFQ_PROX_FETCH
  const int nw = n - has_bias;
  if (gridDim.z > 1) {
    float* P = part + (size_t)blockIdx.z * c2 * ldp;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTN; ++t) {
        const int col = col0 + (wn * NTN + t) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < c2 && col < n) P[(size_t)row * ldp + col] = acc[m][t][r];
        }
      }
    return;
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
      const int col = col0 + (wn * NTN + t) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < c2 && col < n) {
          if (col < nw)
            wstar[(size_t)row * nw + col] = acc[m][t][r];
          else
            bstar[row] = acc[m][t][r];
        }
      }
    }
}


// ---- the same product on the bf16 matrix cores with both operands split in three --------------------------------
// An fp32 value x is the exact sum of three bf16 values x1 + x2 + x3 (8 + 8 + 8 mantissa bits; bf16 has fp32's
// exponent range, so no scaling is needed).  x y = sum_{i,j} x_i y_j; the six products with i + j <= 4 carry everything
// down to 2^-24 of x y - the size of the rounding of an fp32 product itself - and each is EXACT in the fp32 accumulator
// of v_mfma_f32_32x32x16_bf16 (8 x 8 bits).  Six bf16 MFMAs (32 x 32 x 16 each, 32 cycles) replace eight fp32 MFMAs
// (32 x 32 x 2, 64 cycles) per K = 16: 0.375 of the matrix-core time for the same fp32-grade result.  The operands are
// split ON THE FLY while a K tile is staged into LDS (3 planes of [rows][32 + 8] bf16, conflict-free b128 fragment
// reads), so HBM / L2 still carry 4 bytes per element and nothing upstream changes.
// Workgroup tile (128 MT_) x 256, 8 waves as WM_ x WN_; K tiles of 32; split K like k_prox_gemm.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int B3_K = 16;                       // K tile = one MFMA step
constexpr int B3_LD = B3_K + 8;                // bf16 elements per LDS row (48 bytes: conflict-free b128 fragment reads)

__device__ __forceinline__ void split3(const f32x4 v, bf16x4& p0, bf16x4& p1, bf16x4& p2) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 b0 = (__bf16)v[e];
    const float r1 = v[e] - (float)b0;          // exact
    const __bf16 b1 = (__bf16)r1;
    const float r2 = r1 - (float)b1;            // exact
    p0[e] = b0;
    p1[e] = b1;
    p2[e] = (__bf16)r2;
  }
}

// Two LDS stages: while the MFMAs of K tile kt read stage kt & 1, the tile kt + 1 (fetched into registers one
// iteration earlier) is split and stored into the other stage and the global loads of tile kt + 2 are issued - one
// barrier per K tile, the conversion and the loads hide under the matrix-core work of the co-resident waves.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(512) void k_prox_gemm_b3(const float* __restrict__ Bm, int ldb, const float* __restrict__ Ainv,
                                                      int lda, int n, int c2, int has_bias, float* __restrict__ wstar,
                                                      float* __restrict__ bstar, float* __restrict__ part, int ldp) {
  __builtin_amdgcn_s_setprio(2);   // ADMM chain (critical path) over the loss / inverse streams
  constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
  constexpr int NA = BM * 4 / 512, NB = BN * 4 / 512;          // 16-byte global loads per thread per K tile
  constexpr int STAGE = 3 * (BM + BN) * B3_LD;                  // bf16 elements per stage
  static_assert(WM * WN == 8 && MT >= 1 && NT >= 1 && NA >= 1 && NB >= 1, "8 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char b3_lds[];
  __bf16* lds = reinterpret_cast<__bf16*>(b3_lds);              // [2][ As [3][BM][B3_LD] | Bs [3][BN][B3_LD] ]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

  f32x4 ra[NA], rb[NB];
  const float* pa[NA];
  const float* pb[NB];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int u = tid + q * 512, r = u >> 2, c4 = u & 3;
    pa[q] = Bm + (size_t)(row0 + r) * ldb + c4 * 4;             // Bm is zero padded to the row tile
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int u = tid + q * 512, r = u >> 2, c4 = u & 3;
    pb[q] = Ainv + (size_t)min(col0 + r, n - 1) * lda + c4 * 4; // lda padded; columns >= n are dropped below
  }
#define EFFQ_B3_FETCH(k0)                                                           \
  {                                                                                 \
    _Pragma("unroll") for (int q = 0; q < NA; ++q)                                  \
        ra[q] = *reinterpret_cast<const f32x4*>(pa[q] + (k0));                      \
    _Pragma("unroll") for (int q = 0; q < NB; ++q)                                  \
        rb[q] = *reinterpret_cast<const f32x4*>(pb[q] + (k0));                      \
  }
#define EFFQ_B3_STORE(stage)                                                                        \
  {                                                                                                 \
    __bf16* As_ = lds + (stage) * STAGE;                                                            \
    __bf16* Bs_ = As_ + 3 * BM * B3_LD;                                                             \
    _Pragma("unroll") for (int q = 0; q < NA; ++q) {                                                \
      const int u = tid + q * 512, off = (u >> 2) * B3_LD + (u & 3) * 4;                            \
      bf16x4 p0, p1, p2;                                                                            \
      split3(ra[q], p0, p1, p2);                                                                    \
      *reinterpret_cast<bf16x4*>(&As_[off]) = p0;                                                   \
      *reinterpret_cast<bf16x4*>(&As_[BM * B3_LD + off]) = p1;                                      \
      *reinterpret_cast<bf16x4*>(&As_[2 * BM * B3_LD + off]) = p2;                                  \
    }                                                                                               \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                                \
      const int u = tid + q * 512, off = (u >> 2) * B3_LD + (u & 3) * 4;                            \
      bf16x4 p0, p1, p2;                                                                            \
      split3(rb[q], p0, p1, p2);                                                                    \
      *reinterpret_cast<bf16x4*>(&Bs_[off]) = p0;                                                   \
      *reinterpret_cast<bf16x4*>(&Bs_[BN * B3_LD + off]) = p1;                                      \
      *reinterpret_cast<bf16x4*>(&Bs_[2 * BN * B3_LD + off]) = p2;                                  \
    }                                                                                               \
  }
  const int nkt = ldb / B3_K;                   // ldb is a multiple of 32
  const int kt0 = (int)(((long long)nkt * blockIdx.z) / gridDim.z);
  const int nk = (int)(((long long)nkt * (blockIdx.z + 1)) / gridDim.z);
  EFFQ_B3_FETCH(kt0 * B3_K)
  EFFQ_B3_STORE(0)
  EFFQ_B3_FETCH(min(kt0 + 1, nk - 1) * B3_K)
  lds_barrier();
  for (int kt = kt0; kt < nk; ++kt) {
    const int st = (kt - kt0) & 1;
    if (kt + 1 < nk) EFFQ_B3_STORE(st ^ 1)       // tile kt + 1 (in registers since the previous iteration)
    EFFQ_B3_FETCH(min(kt + 2, nk - 1) * B3_K)    // unconditional: past the end a harmless re-read
    const __bf16* As = lds + st * STAGE;
    const __bf16* Bs = As + 3 * BM * B3_LD;
    bf16x8 a[MT][3], b[NT][3];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[m][p] = *reinterpret_cast<const bf16x8*>(&As[p * BM * B3_LD + ((wm * MT + m) * 32 + li) * B3_LD + 8 * lh]);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        b[t][p] = *reinterpret_cast<const bf16x8*>(&Bs[p * BN * B3_LD + ((wn * NT + t) * 32 + li) * B3_LD + 8 * lh]);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        // smallest terms first (x1 y3, x2 y2, x3 y1), then x1 y2, x2 y1, then x1 y1
        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[t][2], acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[t][1], acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][2], b[t][0], acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[t][1], acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[t][0], acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[t][0], acc[m][t], 0, 0, 0);
      }
    lds_barrier();                               // stage st has been read by every wave, stage st ^ 1 is complete
  }
#undef EFFQ_B3_FETCH
#undef EFFQ_B3_STORE
  const int nw = n - has_bias;
  float* P = (gridDim.z > 1) ? part + (size_t)blockIdx.z * c2 * ldp : nullptr;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + (wn * NT + t) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < c2 && col < n) {
          if (P != nullptr)
            P[(size_t)row * ldp + col] = acc[m][t][r];
          else if (col < nw)
            wstar[(size_t)row * nw + col] = acc[m][t][r];
          else
            bstar[row] = acc[m][t][r];
        }
      }
    }
}

// What = sum_z part[z] in slice order (deterministic), scattered to [wstar | bstar]
__global__ __launch_bounds__(256) void k_prox_reduce(const float* __restrict__ part, int ldp, int nsplit, int c2, int n,
                                                     int has_bias, float* __restrict__ wstar, float* __restrict__ bstar) {
  const size_t tot = (size_t)c2 * ldp, stride = (size_t)gridDim.x * blockDim.x;
  const int nw = n - has_bias;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) {
    const int row = (int)(e / ldp), col = (int)(e % ldp);
    if (col >= n) continue;
    float v = part[e];
    for (int z = 1; z < nsplit; ++z) v += part[(size_t)z * tot + e];
    if (col < nw)
      wstar[(size_t)row * nw + col] = v;
    else
      bstar[row] = v;
  }
}

// workgroup tile variant and K split of the prox GEMM: ~440 workgroups, >= 6 K tiles per slice (measured
// best on MI355X for the BraTS sizes: scripts/prof_prox.py with EFFQ_PROX_SPLIT).  The 256-row variant holds 3
// workgroups per CU (104 VGPR + 64 AGPR, 45 KB LDS): ~760 workgroups fill the 768 slots evenly - with 545 some CUs
// carried 3 and others 2 (c2 = 256, n = 6913: 325 -> 266 us per solve).
struct ProxPlan { int variant, gx, gy, nsplit, c2p, ldb; };
static ProxPlan prox_plan(int c2, int n) {
  ProxPlan p;
  p.c2p = (c2 > 128) ? round_up(c2, 256) : (c2 > 64) ? 128 : round_up(c2, 32);
  p.ldb = round_up(n, 32);
  static const int wide = getenv("EFFQ_PROX_WIDE") ? atoi(getenv("EFFQ_PROX_WIDE")) : 0;   // tuning aid
  // A/B switch of the bf16 x 3 kernel: bit 0 = rows > 128, bit 1 = the 128-row layers too (0 = f32 matrix cores)
  static const int b3 = getenv("EFFQ_PROX_B3") ? atoi(getenv("EFFQ_PROX_B3")) : 3;
  if (b3 && (p.c2p >= 256 || (p.c2p == 128 && (b3 & 2))) && n >= 1024) {
    // 256 (128) x 256 tiles on the bf16 matrix cores (k_prox_gemm_b3), one workgroup of 8 waves per CU: K split so that
    // the grid is about one round of the 256 CUs, at least 16 K tiles per slice
    p.variant = (p.c2p >= 256) ? 5 : 6; p.gx = (n + 255) / 256; p.gy = (p.c2p >= 256) ? p.c2p / 256 : 1;
    // 128 rows: 128 x 128 tiles (28 column tiles x 9 K slices at n = 3457: half the partial slabs of the 128 x 256 tiling,
    // two workgroups per CU): 37.7 against 43.1 us alone, 35.8 against 44.3 in situ.  EFFQ_PROX_B3_128=0: A/B switch
    static const int b3_128 = getenv("EFFQ_PROX_B3_128") ? atoi(getenv("EFFQ_PROX_B3_128")) : 1;
    if (p.variant == 6 && b3_128) { p.variant = 7; p.gx = (n + 127) / 128; }
    const int tiles5 = p.gx * p.gy, nkt5 = p.ldb / B3_K;
    int s5 = (256 + tiles5 / 2) / tiles5;
    if (s5 > nkt5 / 16) s5 = nkt5 / 16;
    if (s5 < 1) s5 = 1;
    static const int force5 = getenv("EFFQ_PROX_SPLIT") ? atoi(getenv("EFFQ_PROX_SPLIT")) : 0;   // tuning aid
    if (force5 > 0 && force5 <= nkt5) s5 = force5;
    p.nsplit = s5;
    return p;
  }
  if (p.c2p >= 256 && wide) { p.variant = 4; p.gx = (n + 127) / 128; p.gy = p.c2p / 256; }   // 256x128, waves 64x128
  else if (p.c2p >= 256) { p.variant = 0; p.gx = (n + 63) / 64; p.gy = p.c2p / 256; }        // 256x64, waves 64x64
  else if (p.c2p == 128) { p.variant = 1; p.gx = (n + 63) / 64; p.gy = 1; }             // 128x64, waves 32x64
  else if (p.c2p == 64) { p.variant = 2; p.gx = (n + 63) / 64; p.gy = 1; }              // 64x64, waves 32x32
  else { p.variant = 3; p.gx = (n + 127) / 128; p.gy = 1; }                             // 32x128, waves 32x32
  const int tiles = p.gx * p.gy, nkt = p.ldb / PBK;
  static const int target_env = getenv("EFFQ_PROX_WGS") ? atoi(getenv("EFFQ_PROX_WGS")) : 0;   // tuning aid
  int s = ((target_env > 0 ? target_env : 440) + tiles - 1) / tiles;
  if (s > nkt / 6) s = nkt / 6;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  if (p.variant == 0 && target_env == 0) {
    // fill the 768 slots (3 workgroups per CU) in whole rounds: the split with the best fill, smaller splits preferred
    const int smax = s > 1 ? 16 : 1;
    double best = -1.0;
    for (int c = 1; c <= smax && c <= nkt / 6; ++c) {
      const int wgs = tiles * c, rounds = (wgs + 767) / 768;
      const double score = (double)wgs / (768.0 * rounds) - 0.01 * c;
      if (score > best) { best = score; s = c; }
    }
  }
  static const int force = getenv("EFFQ_PROX_SPLIT") ? atoi(getenv("EFFQ_PROX_SPLIT")) : 0;   // tuning aid
  if (force > 0 && force <= nkt) s = force;
  p.nsplit = s;
  return p;
}

}  // namespace effq

using namespace effq;

extern "C" {

// workspace of the rank-64 sweep on an npad x npad matrix that is already in place: XT [64][npad], NZ [npad][64], 2 Dinv
static size_t gj64_aux_doubles(size_t npad) { return 2 * (size_t)NBK * npad + 2 * NBK * NBK; }

// the wide sweep (pivot blocks of 256) from this many 64-blocks on; EFFQ_GJ_WIDE=0: the rank-64 sweep everywhere (A/B)
static int gj_wide_min_blocks() {
  static const int off = getenv("EFFQ_GJ_WIDE") != nullptr && atoi(getenv("EFFQ_GJ_WIDE")) == 0;
  // measured inside the calibration (ms per calibration, one box): rank-64 sweep everywhere 689.6; 256-row pivot blocks
  // from n = 6400 on 671.0; 256-row (or 128-row, fused pivot launch) blocks for every n >= 512: 683.5 / 681.8 - below
  // n ~ 5000 an inverse is a chain of dependent launches either way, and the rank-64 sweep has the shortest one
  static const int mn = getenv("EFFQ_GJ_WIDE_MIN") ? atoi(getenv("EFFQ_GJ_WIDE_MIN")) : 100;
  return off ? (1 << 30) : mn;
}

size_t effq_spd_inverse_ws_bytes(int n) {
  if (n <= 0) return 0;
  const size_t npad = (size_t)round_up(n, NBK);
  size_t d = npad * npad + gj64_aux_doubles(npad);
  if ((int)(npad / NBK) >= gj_wide_min_blocks()) {
    const size_t ldx = (size_t)round_up((int)npad, BG_T);
    d = npad * npad + 2 * (size_t)WK * ldx + (size_t)WK * WK + gj64_aux_doubles(WK);
  }
  return d * sizeof(double) + 256;
}

int effq_ainv_ld(int n) { return n > 0 ? round_up(n, 32) : 0; }

// in-place rank-64 symmetric Gauss-Jordan sweep of the npad x npad matrix A64 (upper 64-block triangle kept)
static int gj64_sweep(double* A64, int npad, double* aux, hipStream_t st) {
  const int nblk = npad / NBK;
  double* XT = aux;                              // [64][npad]
  double* NZ = XT + (size_t)NBK * npad;          // [npad][64]
  double* Dinv = NZ + (size_t)NBK * npad;
  const size_t lds = (size_t)(NBK * LDA_S + NBK * LDB_S) * sizeof(double);
  // The trailing update of step k also forms the inverse of the NEXT pivot block (look-ahead workgroup, see
  // k_gj_trail_sym): only step 0 needs the stand-alone k_gj_diag.  Dinv is double-buffered by step parity.
  static const bool ahead_off = getenv("EFFQ_GJ_AHEAD") != nullptr && atoi(getenv("EFFQ_GJ_AHEAD")) == 0;   // A/B switch
  for (int k = 0; k < nblk; ++k) {
    const int kb = k * NBK;
    double* Dk = Dinv + (size_t)(k & 1) * NBK * NBK;
    double* Dn = Dinv + (size_t)((k + 1) & 1) * NBK * NBK;
    if (k == 0 || ahead_off) hipLaunchKernelGGL(k_gj_diag, dim3(1), dim3(GJD_T), 0, st, A64, npad, kb, Dk);
    hipLaunchKernelGGL(k_gj_panel, dim3(nblk), dim3(256), lds, st, A64, npad, kb, Dk, NZ, XT);
    if (nblk > 1)
      hipLaunchKernelGGL(k_gj_trail_sym, dim3((nblk + GJ_CT - 1) / GJ_CT, nblk + 1), dim3(256), 0, st, A64, npad, kb, NZ,
                         XT, ahead_off ? (double*)nullptr : Dn);
    EFFQ_LAUNCH_CHECK();
  }
  return EFFQ_OK;
}

// helper stream + two events per (thread, caller stream): the next pivot block is inverted beside the big update
struct GjWideCtx {
  int dev;
  hipStream_t caller, helper;
  hipEvent_t e1, e2;
};
static thread_local std::vector<GjWideCtx> g_gj_ctx;

static int gj_wide_ctx(hipStream_t caller, GjWideCtx** out) {
  int dev = 0;
  EFFQ_HIP(hipGetDevice(&dev));
  for (GjWideCtx& c : g_gj_ctx)
    if (c.dev == dev && c.caller == caller) {
      *out = &c;
      return EFFQ_OK;
    }
  GjWideCtx c;
  c.dev = dev;
  c.caller = caller;
  int prio = 0;
  if (hipStreamGetPriority(caller, &prio) != hipSuccess) prio = 0;
  EFFQ_HIP(hipStreamCreateWithPriority(&c.helper, hipStreamNonBlocking, prio));
  EFFQ_HIP(hipEventCreateWithFlags(&c.e1, hipEventDisableTiming));
  EFFQ_HIP(hipEventCreateWithFlags(&c.e2, hipEventDisableTiming));
  g_gj_ctx.push_back(c);
  *out = &g_gj_ctx.back();
  return EFFQ_OK;
}

static int gj_wide_sweep(double* A64, int npad, double* aux, hipStream_t st) {
  const int nblk = npad / NBK;
  // pivot blocks of 128 rows (next pivot block by ONE fused launch) below EFFQ_GJ_WB4_MIN 64-blocks, of 256 rows from there
  // on (the triangle no longer fits the Infinity Cache: the rank-256 update halves the bytes per flop once more)
  static const int wb4_min = getenv("EFFQ_GJ_WB4_MIN") ? atoi(getenv("EFFQ_GJ_WB4_MIN")) : 100;
  const int WB = (nblk >= wb4_min) ? effq::WB : 2;
  const size_t ldx = (size_t)round_up(npad, BG_T);
  double* NZT = aux;                               // [256][ldx]
  double* XTW = NZT + (size_t)WK * ldx;            // [256][ldx]
  double* Dw = XTW + (size_t)WK * ldx;             // [64 m][64 m]: the pivot block, then its inverse
  double* aux64 = Dw + (size_t)WK * WK;
  const int nK = (nblk + WB - 1) / WB, nm = (nblk + 1) / 2;
  static const bool overlap_on = !(getenv("EFFQ_GJ_OVERLAP") != nullptr && atoi(getenv("EFFQ_GJ_OVERLAP")) == 0);   // A/B
  GjWideCtx* ctx = nullptr;
  if (overlap_on && nK > 1) {
    const int rc = gj_wide_ctx(st, &ctx);
    if (rc != EFFQ_OK) return rc;
  }
  hipStream_t s2 = ctx ? ctx->helper : st;
  const size_t lds = (size_t)(NBK * LDA_S + NBK * LDB_S) * sizeof(double);
  const size_t pf_lds = (size_t)(2 * NBK * PF_LD + 2 * (NBK + 1)) * sizeof(double);
  auto pivot_block = [&](int k0, int mk, int kdim, hipStream_t s) -> int {
    // Dw = the pivot block (blocks k0 .. k0 + mk - 1) with the pending rank-kdim update applied, then inverted
    GjBig g;
    memset(&g, 0, sizeof(g));
    g.Cin = A64; g.ldin = (size_t)npad;
    g.Cout = Dw; g.ldout = (size_t)mk * NBK; g.orow = k0 * NBK; g.ocol = k0 * NBK;
    g.NZT = NZT; g.XTW = XTW; g.ldx = ldx; g.kdim = kdim; g.nblk = nblk;
    g.i0m = k0 / 2; g.j0m = k0 / 2;
    g.skip_lo = g.skip_hi = -1; g.la_lo = g.la_hi = -1;
    if (WB == 2) {
      hipLaunchKernelGGL(k_gj_pivot_fused, dim3(1), dim3(1024), pf_lds, s, A64, npad, k0, mk, NZT, XTW, ldx, kdim, Dw);
      EFFQ_LAUNCH_CHECK();
      return EFFQ_OK;
    }
    const int span = (mk + 1) / 2;
    hipLaunchKernelGGL(k_gj_big, dim3(span, span), dim3(256), 0, s, g);
    EFFQ_LAUNCH_CHECK();
    const int rc = gj64_sweep(Dw, mk * NBK, aux64, s);
    if (rc != EFFQ_OK) return rc;
    hipLaunchKernelGGL(k_mirror_blocks, dim3(64), dim3(256), 0, s, Dw, mk * NBK);
    EFFQ_LAUNCH_CHECK();
    return EFFQ_OK;
  };
  {
    const int rc = pivot_block(0, nblk < WB ? nblk : WB, 0, st);
    if (rc != EFFQ_OK) return rc;
  }
  for (int K = 0; K < nK; ++K) {
    const int kb0 = K * WB, m = (nblk - kb0 < WB) ? nblk - kb0 : WB;
    const int kn0 = kb0 + WB, mn = (K + 1 < nK) ? ((nblk - kn0 < WB) ? nblk - kn0 : WB) : 0;
    hipLaunchKernelGGL(k_gj_panel_w, dim3(nblk, m), dim3(256), lds, st, A64, npad, kb0, m, Dw, NZT, XTW, ldx);
    EFFQ_LAUNCH_CHECK();
    if (ctx) {
      EFFQ_HIP(hipEventRecord(ctx->e1, st));
      EFFQ_HIP(hipStreamWaitEvent(s2, ctx->e1, 0));
    }
    // (the write-back reads the inverse of THIS pivot block from Dw: before the next pivot block overwrites it.  It is
    // a copy of n x 64 m doubles; the pivot work behind it on the helper stream is the part that must not wait)
    hipLaunchKernelGGL(k_gj_wback_w, dim3(nblk, m), dim3(256), 0, s2, A64, npad, kb0, m, Dw, NZT, ldx);
    EFFQ_LAUNCH_CHECK();
    if (mn > 0) {
      const int rc = pivot_block(kn0, mn, m * NBK, s2);
      if (rc != EFFQ_OK) return rc;
    }
    if (nblk > m) {
      GjBig g;
      memset(&g, 0, sizeof(g));
      g.Cin = A64; g.ldin = (size_t)npad;
      g.Cout = A64; g.ldout = (size_t)npad; g.orow = 0; g.ocol = 0;
      g.NZT = NZT; g.XTW = XTW; g.ldx = ldx; g.kdim = m * NBK; g.nblk = nblk;
      g.i0m = 0; g.j0m = 0;
      g.skip_lo = kb0; g.skip_hi = kb0 + m;
      g.la_lo = mn > 0 ? kn0 : -1; g.la_hi = mn > 0 ? kn0 + mn : -1;
      hipLaunchKernelGGL(k_gj_big, dim3(nm, nm), dim3(256), 0, st, g);
      EFFQ_LAUNCH_CHECK();
    }
    if (ctx) {
      EFFQ_HIP(hipEventRecord(ctx->e2, s2));
      EFFQ_HIP(hipStreamWaitEvent(st, ctx->e2, 0));
    }
  }
  return EFFQ_OK;
}

// The helper stream of the 256-row sweep for `stream`, created NOW instead of at the first large inverse: streams take their
// hardware queue in the order they are created, and a communicator created in between (RCCL brings streams of its own)
// moved the calibration's streams onto shared queues (hip_ops.HipOps.warm_streams)
int effq_spd_inverse_prepare(void* stream) {
  GjWideCtx* ctx = nullptr;
  return gj_wide_ctx(as_stream(stream), &ctx);
}

int effq_spd_inverse(const float* A0, int n, int has_bias, double rho, double eta, float* Ainv, void* ws,
                     size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG(A0 && Ainv && ws && n > 0);
  EFFQ_CHECK_ARG(eta > 0.0 && rho >= 0.0);
  if (ws_bytes < effq_spd_inverse_ws_bytes(n)) {
    set_error("spd_inverse: workspace %zu < required %zu", ws_bytes, effq_spd_inverse_ws_bytes(n));
    return EFFQ_ERR_WORKSPACE;
  }
  const int npad = round_up(n, NBK);
  double* A64 = reinterpret_cast<double*>(ws);
  double* aux = A64 + (size_t)npad * npad;
  hipStream_t st = as_stream(stream);
  {
    size_t nb = ((size_t)npad * npad + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_build_a64, dim3((unsigned)nb), dim3(256), 0, st, A0, n, npad, has_bias, rho, eta, A64);
    EFFQ_LAUNCH_CHECK();
  }
  const size_t lds = (size_t)(NBK * LDA_S + NBK * LDB_S) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gj_panel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gj_panel_w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gj_pivot_fused), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)((2 * NBK * PF_LD + 2 * (NBK + 1)) * sizeof(double))));
    attr_set = true;
  }
  const int rc = (npad / NBK >= gj_wide_min_blocks()) ? gj_wide_sweep(A64, npad, aux, st) : gj64_sweep(A64, npad, aux, st);
  if (rc != EFFQ_OK) return rc;
  {
    const int lda = effq_ainv_ld(n);
    size_t nb = ((size_t)n * lda + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_a64_to_f32, dim3((unsigned)nb), dim3(256), 0, st, A64, n, npad, Ainv, lda);
    EFFQ_LAUNCH_CHECK();
  }
  return EFFQ_OK;
}

size_t effq_prox_ws_bytes(int c2, int n) {
  if (c2 <= 0 || n <= 0) return 0;
  const ProxPlan p = prox_plan(c2, n);
  return ((size_t)p.c2p + (p.nsplit > 1 ? (size_t)p.nsplit * c2 : 0)) * p.ldb * sizeof(float) + 256;
}

static int prox_solve_impl(const float* B0, const float* Ainv, const float* W0, const float* b0, const float* G,
                           const float* dual, int c2, int n, int has_bias, double rho, double eta, double shift,
                           int nterms, float* wstar, float* bstar, void* ws, size_t ws_bytes, void* stream,
                           bool prebuilt = false, const float** part_out = nullptr, int* nsplit_out = nullptr,
                           int* ldp_out = nullptr) {
  EFFQ_CHECK_ARG(B0 && Ainv && W0 && G && dual && wstar && ws && c2 > 0 && n > 0 && nterms >= 1);
  EFFQ_CHECK_ARG(!has_bias || (b0 != nullptr && bstar != nullptr));
  if (ws_bytes < effq_prox_ws_bytes(c2, n)) {
    set_error("prox_solve: workspace %zu < required %zu", ws_bytes, effq_prox_ws_bytes(c2, n));
    return EFFQ_ERR_WORKSPACE;
  }
  // rows padded to the workgroup tile, K to the 32-wide K tile (zero filled by k_build_b)
  const ProxPlan pl = prox_plan(c2, n);
  const int c2p = pl.c2p, ldb = pl.ldb;
  const int lda = effq_ainv_ld(n);
  float* Bm = reinterpret_cast<float*>(ws);
  float* part = Bm + (size_t)c2p * ldb;
  hipStream_t st = as_stream(stream);
  for (int term = 0; term < nterms; ++term) {
    if (!(prebuilt && term == 0)) {     // prebuilt: the previous projection already left Bm (effq_project_dual_next)
      const int nw = n - (has_bias ? 1 : 0);
      if ((nw % 4) == 0) {
        const unsigned bx = (unsigned)((ldb / 4 + 255) / 256);
        hipLaunchKernelGGL(k_build_b4, dim3(bx, (unsigned)c2p), dim3(256), 0, st, B0, W0, b0, G, dual, c2, n,
                           has_bias ? 1 : 0, (float)rho, (float)eta, Bm, ldb, (term > 0) ? wstar : nullptr, (float)shift);
      } else {
        size_t nb = ((size_t)c2p * ldb + 255) / 256;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(k_build_b, dim3((unsigned)nb), dim3(256), 0, st, B0, W0, b0, G, dual, c2, n, has_bias ? 1 : 0,
                           (float)rho, (float)eta, Bm, ldb, c2p, (term > 0) ? wstar : nullptr, (float)shift);
      }
      EFFQ_LAUNCH_CHECK();
    }
    const dim3 grid(pl.gx, pl.gy, pl.nsplit);
#define EFFQ_PROX_LAUNCH(MT, WM, WN, NTN)                                                                              \
  hipLaunchKernelGGL((k_prox_gemm<MT, WM, WN, NTN>), grid, dim3(256), 0, st, Bm, ldb, Ainv, lda, n, c2, has_bias ? 1 : 0, \
                     wstar, bstar, part, ldb)
    if (pl.variant == 5 || pl.variant == 6 || pl.variant == 7) {
      const size_t lds5 = (size_t)2 * 3 * (256 + 256) * B3_LD * sizeof(__bf16);      // two stages
      const size_t lds6 = (size_t)2 * 3 * (128 + 256) * B3_LD * sizeof(__bf16);
      const size_t lds7 = (size_t)2 * 3 * (128 + 128) * B3_LD * sizeof(__bf16);
      static bool attr5[64] = {};
      int dev5 = 0;
      EFFQ_HIP(hipGetDevice(&dev5));
      EFFQ_CHECK_ARG(dev5 >= 0 && dev5 < 64);
      if (!attr5[dev5]) {
        EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_prox_gemm_b3<256, 256, 4, 2>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5));
        EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_prox_gemm_b3<128, 256, 2, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6));
        EFFQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_prox_gemm_b3<128, 128, 2, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds7));
        attr5[dev5] = true;
      }
      if (pl.variant == 5)
        hipLaunchKernelGGL((k_prox_gemm_b3<256, 256, 4, 2>), grid, dim3(512), lds5, st, Bm, ldb, Ainv, lda, n, c2,
                           has_bias ? 1 : 0, wstar, bstar, part, ldb);
      else if (pl.variant == 6)
        hipLaunchKernelGGL((k_prox_gemm_b3<128, 256, 2, 4>), grid, dim3(512), lds6, st, Bm, ldb, Ainv, lda, n, c2,
                           has_bias ? 1 : 0, wstar, bstar, part, ldb);
      else
        hipLaunchKernelGGL((k_prox_gemm_b3<128, 128, 2, 4>), grid, dim3(512), lds7, st, Bm, ldb, Ainv, lda, n, c2,
                           has_bias ? 1 : 0, wstar, bstar, part, ldb);
    } else
    switch (pl.variant) {
      case 0: EFFQ_PROX_LAUNCH(2, 4, 1, 2); break;
      case 4: EFFQ_PROX_LAUNCH(2, 4, 1, 4); break;
      case 1: EFFQ_PROX_LAUNCH(1, 4, 1, 2); break;
      case 2: EFFQ_PROX_LAUNCH(1, 2, 2, 1); break;
      default: EFFQ_PROX_LAUNCH(1, 1, 4, 1); break;
    }
#undef EFFQ_PROX_LAUNCH
    EFFQ_LAUNCH_CHECK();
    if (part_out != nullptr) {                  // the caller's next kernel adds the K slices up itself
      *part_out = (pl.nsplit > 1) ? part : nullptr;
      *nsplit_out = pl.nsplit;
      *ldp_out = ldb;
    } else if (pl.nsplit > 1) {
      const int nw = n - (has_bias ? 1 : 0);
      if ((nw % 4) == 0) {
        const unsigned bx = (unsigned)((ldb / 4 + 255) / 256);
        hipLaunchKernelGGL(k_prox_reduce4, dim3(bx, (unsigned)c2), dim3(256), 0, st, part, ldb, pl.nsplit, c2, n,
                           has_bias ? 1 : 0, wstar, bstar);
      } else {
        size_t nb = ((size_t)c2 * ldb + 255) / 256;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(k_prox_reduce, dim3((unsigned)nb), dim3(256), 0, st, part, ldb, pl.nsplit, c2, n,
                           has_bias ? 1 : 0, wstar, bstar);
      }
      EFFQ_LAUNCH_CHECK();
    }
  }
  return EFFQ_OK;
}

int effq_prox_solve(const float* B0, const float* Ainv, const float* W0, const float* b0, const float* G,
                    const float* dual, int c2, int n, int has_bias, double rho, double eta, float* wstar, float* bstar,
                    void* ws, size_t ws_bytes, void* stream) {
  return prox_solve_impl(B0, Ainv, W0, b0, G, dual, c2, n, has_bias, rho, eta, 0.0, 1, wstar, bstar, ws, ws_bytes,
                         stream);
}

// internal (admm_run.hip): Bm = the start of the prox workspace, its row length; the solve on a Bm that
// effq_project_dual_next has already written
float* effq_prox_bm(void* ws, int c2, int n, int* ldb) {
  *ldb = prox_plan(c2, n).ldb;
  return reinterpret_cast<float*>(ws);
}
int effq_prox_solve_prebuilt(const float* B0, const float* Ainv, const float* W0, const float* b0, const float* G,
                             const float* dual, int c2, int n, int has_bias, double rho, double eta, float* wstar,
                             float* bstar, void* ws, size_t ws_bytes, void* stream) {
  return prox_solve_impl(B0, Ainv, W0, b0, G, dual, c2, n, has_bias, rho, eta, 0.0, 1, wstar, bstar, ws, ws_bytes,
                         stream, true);
}

// internal (admm_run.hip): the product only.  *part_out != NULL: nsplit K slices of [c2][ldp] floats that the NEXT kernel
// adds up (effq_fixed_point_traj_parts); NULL: the product had one slice and wstar / bstar are complete
int effq_prox_solve_prebuilt_parts(const float* B0, const float* Ainv, const float* W0, const float* b0, const float* G,
                                   const float* dual, int c2, int n, int has_bias, double rho, double eta, float* wstar,
                                   float* bstar, void* ws, size_t ws_bytes, void* stream, const float** part_out,
                                   int* nsplit_out, int* ldp_out) {
  EFFQ_CHECK_ARG(part_out && nsplit_out && ldp_out);
  return prox_solve_impl(B0, Ainv, W0, b0, G, dual, c2, n, has_bias, rho, eta, 0.0, 1, wstar, bstar, ws, ws_bytes,
                         stream, true, part_out, nsplit_out, ldp_out);
}

int effq_prox_solve_shifted(const float* B0, const float* Ainv, const float* W0, const float* b0, const float* G,
                            const float* dual, int c2, int n, int has_bias, double rho, double eta, double rho_inv,
                            int nterms, float* wstar, float* bstar, void* ws, size_t ws_bytes, void* stream) {
  EFFQ_CHECK_ARG(rho_inv >= rho && rho > 0.0 && eta >= 0.0);
  return prox_solve_impl(B0, Ainv, W0, b0, G, dual, c2, n, has_bias, rho, eta, rho_inv - rho, nterms, wstar, bstar, ws,
                         ws_bytes, stream);
}

int effq_project_dual_checked(const float* v, const float* wstar, const effq_fp_state* state_dev, int levels, float* G,
                              float* dual, float dual_div, int8_t* Gq_out, size_t n, int32_t* err_flag_dev,
                              void* stream);   // quant_reduce.hip (internal to the library)

}  // extern "C"
