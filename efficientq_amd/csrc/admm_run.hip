// The whole ADMM loop of one layer (EfficientQConv.py:99-144) enqueued by ONE C call.
//
// The loop body is launch-bound for most layers of a network (a few tens of microseconds of GPU work per iteration),
// so issuing it from the host language costs more than running it.  effq_admm_run() enqueues all iterations on up to
// three caller-owned HIP streams:
//   main : prox solve -> scale fixed point -> projection + dual update            (the serial chain)
//   loss : conv + squared error of iteration i, while main computes i+1            (best-iterate selection only)
//   side : the inverses of A(rho) for the later rho values, under the iterations that precede their first use
// Every per-iteration result the loss stream reads lives in a RING indexed by the iteration (G, its int8 operands,
// b*, the scale state): slots are written once, so main never waits for loss.  The squared errors land in hist[i];
// the best iterate is picked AFTER the loop (effq_admm_select_best), which is what lets a data-parallel caller
// all-reduce the whole history with one collective per layer instead of one per iteration.
#include <stdlib.h>
#include <vector>
#include "common.h"
#include "project_dual.h"

extern "C" {
int effq_fixed_point_small_fused(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                                 double tol, int max_iter, effq_fp_state* state_dev, const effq::ProjFused* pf_in,
                                 void* stream);   // quant_reduce.hip (internal)
int effq_fixed_point_bucket_fused(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                                  double tol, int max_iter, effq_fp_state* state_dev, void* ws, size_t ws_bytes,
                                  void* pred_dev, const effq::ProjFused* pf_in, int* fused_out, void* stream);   // fixed_point_bucket.hip (internal)
int effq_project_dual_next(const float* v, const float* wstar, const effq_fp_state* state_dev, int levels, float* G,
                           float* dual, float dual_div, int8_t* Gq_out, size_t n, int32_t* err_flag_dev, float* Bm,
                           const float* B0, const float* W0, int nwrow, int nb0, int ldb, double rho_next, double eta,
                           void* stream);
float* effq_prox_bm(void* ws, int c2, int n, int* ldb);
int effq_prox_solve_prebuilt(const float* B0, const float* Ainv, const float* W0, const float* b0, const float* G,
                             const float* dual, int c2, int n, int has_bias, double rho, double eta, float* wstar,
                             float* bstar, void* ws, size_t ws_bytes, void* stream);
int effq_prox_solve_prebuilt_parts(const float* B0, const float* Ainv, const float* W0, const float* b0, const float* G,
                                   const float* dual, int c2, int n, int has_bias, double rho, double eta, float* wstar,
                                   float* bstar, void* ws, size_t ws_bytes, void* stream, const float** part_out,
                                   int* nsplit_out, int* ldp_out);                      // solve.hip
int effq_fixed_point_traj_parts(const float* part, int nsplit, int ldp, int c2, int nwrow, int has_bias, const float* dual,
                                float* wstar_out, float* bstar_out, float* v_out, int levels, double lo, double hi, double tol,
                                int max_iter, effq_fp_state* state_dev, void* pred_dev, void* ws, size_t ws_bytes,
                                void* stream);                                          // fixed_point_traj.hip
int effq_project_dual_checked(const float* v, const float* wstar, const effq_fp_state* state_dev, int levels, float* G,
                              float* dual, float dual_div, int8_t* Gq_out, size_t n, int32_t* err_flag_dev,
                              void* stream);   // quant_reduce.hip (internal)
}

namespace effq {

// ---- sampling profiler of effq_admm_run (off by default): HIP-event pairs, on the stream the work is launched on,
// around the ops of every `every`-th iteration.  bench.py reads them back after the timed region.
struct ProfRec {
  int kind, iter, loss_kind, c2, n;
  effq_geom geom;
  hipEvent_t e0, e1;
};
static thread_local std::vector<ProfRec> g_prof;
static thread_local int g_prof_every = 0;
// PROF_WAIT brackets the points where the MAIN stream waits for another stream (the inverse of the next rho, the joins
// at the end of the layer): two records on the main stream around the wait = the time the chain stood still there
enum { PROF_PROX = 1, PROF_FIXED_POINT = 2, PROF_PROJECT = 3, PROF_LOSS = 4, PROF_INVERSE = 5, PROF_WAIT = 6 };

struct ProfScope {          // records e0 now, e1 at close()
  bool on;
  hipStream_t stream;
  ProfRec rec;
  ProfScope(bool enabled, int kind, int iter, const effq_admm_run_args* a, hipStream_t st) : on(enabled), stream(st) {
    if (!on) return;
    rec.kind = kind;
    rec.iter = iter;
    rec.loss_kind = a->loss_kind;
    rec.c2 = a->c2;
    rec.n = a->n;
    rec.geom = a->geom;
    if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(rec.e0, stream);
  }
  void close() {
    if (!on) return;
    (void)hipEventRecord(rec.e1, stream);
    g_prof.push_back(rec);
    on = false;
  }
};

constexpr int ADMM_MAX_RHOS = 16;

struct RhoPlan {
  int count;                       // distinct rho values, in order of first use
  double rho[ADMM_MAX_RHOS];
  int first_iter[ADMM_MAX_RHOS];   // iteration of first use
  bool shifted_first;              // rho[0] serves iteration 0 only: solved through the inverse of A(rho[1])
  bool overflow;
};

// EfficientQConv.py:129-137: at i % period == 0 (after the iteration), rho doubles while 2*rho <= rho_max, else -> rho_max
static RhoPlan plan_rhos(double rho, double rho_max, int iters, int period) {
  RhoPlan p;
  p.count = 0;
  p.overflow = false;
  double r = rho;
  for (int i = 0; i < iters; ++i) {
    if (p.count == 0 || p.rho[p.count - 1] != r) {
      if (p.count == ADMM_MAX_RHOS) {
        p.overflow = true;
        break;
      }
      p.rho[p.count] = r;
      p.first_iter[p.count] = i;
      ++p.count;
    }
    if (i % period == 0) r = (r * 2 <= rho_max) ? r * 2 : rho_max;
  }
  p.shifted_first = p.count > 1 && p.first_iter[1] == 1 && p.rho[1] > p.rho[0];
  return p;
}

static int shift_terms(double rho, double eta, double rho_inv) {
  // sweeps of the contraction (factor d / (rho_inv + eta)) to reach 2^-26
  const double d = rho_inv - rho;
  if (d <= 0) return 1;
  int t = (int)ceil(-26.0 * log(2.0) / log(d / (rho_inv + eta)));
  return t < 2 ? 2 : (t > 64 ? 64 : t);
}

// ---- Gram system packed for the data-parallel exchange: upper triangle of A0 (row-major, n(n+1)/2) then B0 (c2 x n) ----
__global__ __launch_bounds__(256) void k_gram_pack(const float* __restrict__ A0, const float* __restrict__ B0, int n, int c2,
                                                   float* __restrict__ buf) {
  const size_t tri = (size_t)n * (n + 1) / 2, nb = (size_t)c2 * n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n * n + nb; e += stride) {
    if (e < (size_t)n * n) {
      const int i = (int)(e / n), j = (int)(e % n);
      if (j >= i) buf[(size_t)i * n - (size_t)i * (i - 1) / 2 + (j - i)] = A0[e];
    } else {
      buf[tri + (e - (size_t)n * n)] = B0[e - (size_t)n * n];
    }
  }
}
__global__ __launch_bounds__(256) void k_gram_unpack(const float* __restrict__ buf, int n, int c2, float* __restrict__ A0,
                                                     float* __restrict__ B0) {
  const size_t tri = (size_t)n * (n + 1) / 2, nb = (size_t)c2 * n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n * n + nb; e += stride) {
    if (e < (size_t)n * n) {
      int i = (int)(e / n), j = (int)(e % n);
      if (j < i) {              // mirror: A0 is exactly symmetric
        const int t = i;
        i = j;
        j = t;
      }
      A0[e] = buf[(size_t)i * n - (size_t)i * (i - 1) / 2 + (j - i)];
    } else {
      B0[e - (size_t)n * n] = buf[tri + (e - (size_t)n * n)];
    }
  }
}

__global__ __launch_bounds__(256) void k_select_best(const double* __restrict__ hist, int iters,
                                                     const float* __restrict__ G_ring, const float* __restrict__ b_ring,
                                                     size_t nw, size_t nb, float* __restrict__ best_G,
                                                     float* __restrict__ best_b, double* __restrict__ best_out) {
  // "if i == 0 or lossf < best" (EfficientQConv.py:139-142): the EARLIEST minimum; every thread scans the same doubles
  int bi = 0;
  double bl = hist[0];
  for (int i = 1; i < iters; ++i) {
    const double l = hist[2 * (size_t)i];
    if (l < bl) {
      bl = l;
      bi = i;
    }
  }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float* g = G_ring + (size_t)bi * nw;
  for (size_t i = t0; i < nw; i += stride) best_G[i] = g[i];
  if (b_ring != nullptr)
    for (size_t i = t0; i < nb; i += stride) best_b[i] = b_ring[(size_t)bi * nb + i];
  if (t0 == 0) {
    best_out[0] = bl;
    best_out[1] = (double)bi;
  }
}

// lwq_verbose (EfficientQConv.py:114-116): sum (w* - G)^2 and sum (G - G0)^2 of an iteration -> res[0], res[1] (the host
// takes the roots and multiplies the second by rho).  A diagnostic: fp64 atomics, printed to four decimals.
__global__ __launch_bounds__(256) void k_admm_residuals(const float* __restrict__ wstar, const float* __restrict__ G,
                                                        const float* __restrict__ G0, size_t n, double* __restrict__ res) {
  __shared__ double sh[2][4];
  double a = 0.0, b = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double p = (double)wstar[i] - (double)G[i], q = (double)G[i] - (double)G0[i];
    a += p * p;
    b += q * q;
  }
  a = wave_sum_f64_dpp(a);
  b = wave_sum_f64_dpp(b);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    sh[0][wid] = a;
    sh[1][wid] = b;
  }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(res + threadIdx.x, (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]));
}

}  // namespace effq
using namespace effq;

extern "C" {

// whether effq_admm_run would use effq_fixed_point_traj for a layer of nw weights at w_levels levels (the caller then
// provides fp_pred / fp_traj_ws; otherwise both may be NULL)
int effq_admm_uses_traj(size_t nw, int w_levels) {
  static const bool traj_on = !(getenv("EFFQ_FP_TRAJ") && atoi(getenv("EFFQ_FP_TRAJ")) == 0);
  static const int traj_levels = getenv("EFFQ_FP_TRAJ_LEVELS") ? atoi(getenv("EFFQ_FP_TRAJ_LEVELS")) : 4;
  static const size_t traj_min = getenv("EFFQ_FP_TRAJ_MIN") ? (size_t)atoll(getenv("EFFQ_FP_TRAJ_MIN")) : 65536;
  return (traj_on && w_levels <= traj_levels && nw >= traj_min && nw <= effq_fp_traj_max()) ? 1 : 0;
}

int effq_admm_num_inverses(double rho, double rho_max, int iters, int period) {
  if (!(rho > 0.0) || iters <= 0 || period <= 0) return -1;
  const RhoPlan p = plan_rhos(rho, rho_max, iters, period);
  if (p.overflow) return -1;
  return p.count - (p.shifted_first ? 1 : 0);
}

int effq_admm_run(const effq_admm_run_args* a) {
  EFFQ_CHECK_ARG(a != nullptr);
  EFFQ_CHECK_ARG(a->A0 && a->B0 && a->W0 && a->dual && a->wstar && a->v && a->G_ring && a->state_ring && a->hist &&
                 a->err_flag && a->ainv_pool && a->prox_ws && a->red_ws && a->inv_ws && a->conv_ws && a->y_fp);
  EFFQ_CHECK_ARG(a->c2 > 0 && a->n > 1 && a->iters > 0 && a->rho_period > 0 && a->w_levels >= 2 && a->w_levels <= 256);
  EFFQ_CHECK_ARG((a->has_bias != 0) == (a->b0 != nullptr) && (a->has_bias != 0) == (a->b_ring != nullptr));
  EFFQ_CHECK_ARG((a->loss_kind >= 0 && a->loss_kind <= 2) || a->loss_kind == 4 || a->loss_kind == 5);
  if (a->loss_kind == 4)
    EFFQ_CHECK_ARG(a->loss_Au != nullptr && a->loss_Bu != nullptr && a->loss_syy != nullptr);
  else if (a->loss_kind == 5)
    EFFQ_CHECK_ARG(a->loss_Au != nullptr && a->loss_Bu != nullptr && a->loss_syy != nullptr && a->loss_planes != nullptr &&
                   a->loss_nplanes > 0 && a->Gq_ring != nullptr && a->act_alpha_dev != nullptr &&
                   effq_gram_loss_i8_supported(a->c2, a->n, a->has_bias, a->w_levels));
  else
    EFFQ_CHECK_ARG(a->loss_kind == 0 ? (a->xq != nullptr) : (a->xidx != nullptr && a->Gq_ring != nullptr &&
                                                            a->act_alpha_dev != nullptr));
  const int c2 = a->c2, n = a->n, has_b = a->has_bias ? 1 : 0;
  const size_t nw = (size_t)c2 * (size_t)(n - has_b);
  EFFQ_CHECK_ARG(nw == (size_t)a->geom.C2 * a->geom.C1 * a->geom.KD * a->geom.KH * a->geom.KW);
  if (nw > effq_fp_coop_max()) {
    set_error("admm_run: %zu weights exceed the single-launch fixed points", nw);
    return EFFQ_ERR_ARG;
  }
  // weight-scale fixed point, by measured speed on MI355X (scripts/prof_fp.py, microseconds per call at 4 levels,
  // all-values kernel / bucketed: 2048 values 16 / 22; 8192 36 / 28; 27648 84 / 37; 110592 125 / 51; 442368 143 / 65;
  // 1.77 M 171 / 163 alone but slower inside the loop (1261 vs 1224 ms per calibration); at 256 levels the all-values
  // kernels win at every size)
  static const size_t bucket_max = getenv("EFFQ_FP_BUCKET_MAX") ? (size_t)atoll(getenv("EFFQ_FP_BUCKET_MAX"))
                                                                : ((size_t)1 << 19);          // tuning aid
  const bool bucket = a->fp_ws != nullptr && a->w_levels <= 16 && nw > 4096 && nw <= bucket_max;
  if (bucket && a->fp_ws_bytes < effq_fp_bucket_ws_bytes(nw)) {
    set_error("admm_run: fixed-point workspace %zu < %zu", a->fp_ws_bytes, effq_fp_bucket_ws_bytes(nw));
    return EFFQ_ERR_WORKSPACE;
  }
  // ... and, where it is the faster one, the projection that starts from the previous iteration's iterates
  // (effq_fixed_point_traj: one launch, no grid barrier).  Its last workgroup scans lists whose length follows the drift
  // of the iterates from call to call, which is largest right after rho has changed: measured inside the calibration
  // (ms per calibration, all on one box: older kernels only 715; trajectory kernel from the third iteration of a layer on
  // 743; from 5 / 10 / 15 iterations after a change of rho, layers of 65 536 ... 2^20 weights only: 702 / 703 / 699; and
  // for larger layers from 30 iterations after: 698).  The iterations in between run the kernels above, which leave
  // their iterates behind.  EFFQ_FP_TRAJ=0, EFFQ_FP_TRAJ_AFTER[_BIG], EFFQ_FP_TRAJ_MIN, EFFQ_FP_TRAJ_LEVELS: A/B switches
  static const bool traj_rho_old = !(getenv("EFFQ_FP_TRAJ_RHO") && atoi(getenv("EFFQ_FP_TRAJ_RHO")) == 0);
  // (at 16 levels the fixed point takes ~50 iterations: the one hull slot for everything past the seventh keeps a third
  // of the values on the list, and the older kernels are faster - measured, scripts/prof_fp_traj.py)
  static const int traj_after_env = getenv("EFFQ_FP_TRAJ_AFTER") ? atoi(getenv("EFFQ_FP_TRAJ_AFTER")) : 12;
  static const int traj_after_big = getenv("EFFQ_FP_TRAJ_AFTER_BIG") ? atoi(getenv("EFFQ_FP_TRAJ_AFTER_BIG")) : 30;
  const int traj_after = (nw > ((size_t)1 << 20)) ? traj_after_big : traj_after_env;
  const bool traj = a->fp_pred != nullptr && a->fp_traj_ws != nullptr && effq_admm_uses_traj(nw, a->w_levels) != 0;
  if (traj && a->fp_traj_ws_bytes < effq_fp_traj_ws_bytes(nw)) {
    set_error("admm_run: trajectory fixed-point workspace %zu < %zu", a->fp_traj_ws_bytes, effq_fp_traj_ws_bytes(nw));
    return EFFQ_ERR_WORKSPACE;
  }
  const RhoPlan plan = plan_rhos(a->rho, a->rho_max, a->iters, a->rho_period);
  EFFQ_CHECK_ARG(!plan.overflow);
  const int first = plan.shifted_first ? 1 : 0;
  const int n_inv = plan.count - first;
  EFFQ_CHECK_ARG(a->n_ainv >= n_inv);
  const size_t ainv_elems = (size_t)n * (size_t)effq_ainv_ld(n);

  hipStream_t s_main = as_stream(a->stream_main);
  hipStream_t s_loss = a->stream_loss ? as_stream(a->stream_loss) : s_main;
  hipStream_t s_side = (a->stream_side && a->inv_ws_side) ? as_stream(a->stream_side) : s_main;
  const bool fork_loss = s_loss != s_main, fork_side = s_side != s_main;
  // a second side stream: the later inverses alternate between the two (a Gauss-Jordan sweep is a chain of ~100 dependent
  // launches with serial pivot phases: two sweeps side by side fill each other's bubbles, and the last inverse of a wide
  // layer is ready before the chain reaches the iteration that needs it)
  hipStream_t s_side2 = (fork_side && a->stream_side2 && a->inv_ws_side2) ? as_stream(a->stream_side2) : s_side;
  const bool two_sides = s_side2 != s_side;

  // events: one per inverse formed on the side stream, a small pool for main -> loss, one each for the joins.  They
  // come from a per-thread, per-device pool that is never destroyed (a wait captures the record that precedes it, so
  // an event may be re-recorded by the next call while an older wait on it is still queued).
  constexpr int EV_POOL = 4;
  int dev_id = 0;
  EFFQ_HIP(hipGetDevice(&dev_id));
  static thread_local std::vector<std::vector<hipEvent_t>> g_events;
  if ((int)g_events.size() <= dev_id) g_events.resize(dev_id + 1);
  std::vector<hipEvent_t>& pool = g_events[dev_id];
  size_t pool_used = 0;
  auto new_event = [&](hipEvent_t* e) -> hipError_t {
    if (pool_used == pool.size()) {
      hipEvent_t fresh;
      hipError_t rc = hipEventCreateWithFlags(&fresh, hipEventDisableTiming);
      if (rc != hipSuccess) return rc;
      pool.push_back(fresh);
    }
    *e = pool[pool_used++];
    return hipSuccess;
  };
  auto destroy_events = [&]() {};
#define ADMM_HIP(call)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      effq::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));     \
      destroy_events();                                                                       \
      return EFFQ_ERR_HIP;                                                                    \
    }                                                                                         \
  } while (0)
#define ADMM_RC(call)          \
  do {                         \
    int rc_ = (call);          \
    if (rc_ != EFFQ_OK) {      \
      destroy_events();        \
      return rc_;              \
    }                          \
  } while (0)

  hipEvent_t ev_inv[ADMM_MAX_RHOS] = {};
  hipEvent_t ev_main[EV_POOL] = {};
  hipEvent_t ev_fork = nullptr, ev_join_loss = nullptr, ev_join_side = nullptr;

  ADMM_HIP(hipMemsetAsync(a->dual, 0, nw * sizeof(float), s_main));                 // dual <- 0 (EfficientQConv.py:40)
  if (a->res_ring != nullptr) ADMM_HIP(hipMemsetAsync(a->res_ring, 0, 2 * (size_t)a->iters * sizeof(double), s_main));
  if (traj) ADMM_HIP(hipMemsetAsync(a->fp_pred, 0, effq_fp_traj_pred_bytes(), s_main));   // nothing known about this layer
  // the inverse the first iterations need, on the main stream; the later ones on the side stream, which starts at once,
  // beside the first inverse (all of them only read A0): with n = 13825 an inverse takes longer than the 50 iterations it has
  // to be ready after, and the chain waited for each of the three later ones in turn (LiTS: 3.08 -> 3.03 s per calibration;
  // BraTS 900 -> 893 ms).  EFFQ_SIDE_EARLY_N = smallest n that does so (tuning aid).
  static const int early_n = getenv("EFFQ_SIDE_EARLY_N") ? atoi(getenv("EFFQ_SIDE_EARLY_N")) : 0;
  // ... unless the later inverses are quick enough to be ready in time when they start AFTER the first one (n below
  // EFFQ_SIDE_SERIAL_BELOW): beside two other sweeps the first inverse - which the whole chain waits for - took 2 x as
  // long as alone (n = 3457: 7.3 against 3.6 ms, 1729: 3.1 against 1.5), and the later ones then run one after the other
  // on ONE side stream, the one that is needed next always first.
  static const int serial_below = getenv("EFFQ_SIDE_SERIAL_BELOW") ? atoi(getenv("EFFQ_SIDE_SERIAL_BELOW")) : 0;
  const bool side_serial = fork_side && n_inv > 1 && n < serial_below;
  const bool side_early = fork_side && n_inv > 1 && n >= early_n;
  if (side_early && !side_serial) {
    ADMM_HIP(new_event(&ev_fork));
    ADMM_HIP(hipEventRecord(ev_fork, s_main));         // A0 (and everything before the call) is ready
    ADMM_HIP(hipStreamWaitEvent(s_side, ev_fork, 0));
    if (two_sides) ADMM_HIP(hipStreamWaitEvent(s_side2, ev_fork, 0));
  }
  {
    ProfScope ps(g_prof_every > 0, PROF_INVERSE, -1, a, s_main);
    ADMM_RC(effq_spd_inverse(a->A0, n, has_b, plan.rho[first], a->eta, a->ainv_pool, a->inv_ws, a->inv_ws_bytes, s_main));
    ps.close();
  }
  if (side_serial) {
    ADMM_HIP(new_event(&ev_fork));
    ADMM_HIP(hipEventRecord(ev_fork, s_main));         // the first inverse has finished
    ADMM_HIP(hipStreamWaitEvent(s_side, ev_fork, 0));
  }
  // The later inverses (side stream) are ENQUEUED a few iterations into the loop, not here: their ~30 - 650 launches take
  // the host 0.2 - 2.6 ms, during which the main stream - done with its own inverse on the small layers - had nothing queued
  // (under a profiler, at 3 x the launch cost, 6 ms per layer).  The side stream still starts from the fork event recorded
  // above, i.e. as early as before.
  bool side_enqueued = (n_inv <= 1);
  auto enqueue_side_inverses = [&]() -> int {
    side_enqueued = true;
    if (fork_side && !side_early && !side_serial) {
      hipError_t e1 = new_event(&ev_fork);
      if (e1 == hipSuccess) e1 = hipEventRecord(ev_fork, s_main);       // A0 (and everything before the call) is ready
      if (e1 == hipSuccess) e1 = hipStreamWaitEvent(s_side, ev_fork, 0);
      if (e1 == hipSuccess && two_sides) e1 = hipStreamWaitEvent(s_side2, ev_fork, 0);
      if (e1 != hipSuccess) {
        effq::set_error("admm_run: side-stream fork -> %s", hipGetErrorString(e1));
        return EFFQ_ERR_HIP;
      }
    }
    for (int r = first + 1; r < plan.count; ++r) {
      float* dst = a->ainv_pool + (size_t)(r - first) * ainv_elems;
      const bool on2 = two_sides && !side_serial && ((r - first) % 2 == 0);   // first later inverse on side 1, the next on side 2 ...
      hipStream_t sr = on2 ? s_side2 : s_side;
      void* ws = !fork_side ? a->inv_ws : (on2 ? a->inv_ws_side2 : a->inv_ws_side);
      const size_t wsb = !fork_side ? a->inv_ws_bytes : (on2 ? a->inv_ws_side2_bytes : a->inv_ws_side_bytes);
      ProfScope ps(g_prof_every > 0, PROF_INVERSE, -1 - r, a, sr);
      const int rc = effq_spd_inverse(a->A0, n, has_b, plan.rho[r], a->eta, dst, ws, wsb, sr);
      if (rc != EFFQ_OK) return rc;
      ps.close();
      if (fork_side) {
        hipError_t e2 = new_event(&ev_inv[r]);
        if (e2 == hipSuccess) e2 = hipEventRecord(ev_inv[r], sr);
        if (e2 != hipSuccess) {
          effq::set_error("admm_run: side-stream event -> %s", hipGetErrorString(e2));
          return EFFQ_ERR_HIP;
        }
      }
    }
    return EFFQ_OK;
  };
  // without a side stream the later inverses run on the main stream: they must be queued before the iterations that use them
  if (!fork_side && !side_enqueued) ADMM_RC(enqueue_side_inverses());
  constexpr int SIDE_AFTER_ITERS = 8;
  if (fork_loss)
    for (int e = 0; e < EV_POOL; ++e) ADMM_HIP(new_event(&ev_main[e]));

  double rho = a->rho;
  int cur = -1;     // index into plan.rho of the inverse in use
  static const bool fuse_off = getenv("EFFQ_FUSE_BUILD") != nullptr && atoi(getenv("EFFQ_FUSE_BUILD")) == 0;   // A/B switch
  const bool fuse_build = !fuse_off;
  int bm_ld = 0;
  float* bm = effq_prox_bm(a->prox_ws, c2, n, &bm_ld);
  bool bm_ready = false;
  const float* Ainv = nullptr;
  // group size: the last group is evaluated after the chain has finished (it delays the join by one group of losses), so
  // the cheap losses from the Gram system travel in larger groups than the conv passes.  EFFQ_LOSS_GROUP[_CONV]: A/B switches
  static const int group_gram = getenv("EFFQ_LOSS_GROUP") ? atoi(getenv("EFFQ_LOSS_GROUP")) : 8;
  static const int group_conv = getenv("EFFQ_LOSS_GROUP_CONV") ? atoi(getenv("EFFQ_LOSS_GROUP_CONV")) : 4;
  const int group_env = (a->loss_kind == 4 || a->loss_kind == 5) ? group_gram : group_conv;
  const int loss_group = (fork_loss && group_env > 1) ? group_env : 1;
  int loss_next = 0;
  bool rho_changed_last = false;
  int rho_changed_at = 0;          // first iteration that ran with the current rho
  for (int i = 0; i < a->iters; ++i) {
    const bool use_shift = plan.shifted_first && i == 0;
    if (!side_enqueued && (i == SIDE_AFTER_ITERS || (first + 1 < plan.count && i + 1 >= plan.first_iter[first + 1])))
      ADMM_RC(enqueue_side_inverses());
    if (!use_shift && (cur < 0 || plan.rho[cur] != rho)) {
      int r = first;
      while (r < plan.count && plan.rho[r] != rho) ++r;
      EFFQ_CHECK_ARG(r < plan.count);
      if (fork_side && ev_inv[r] != nullptr) {
        ProfScope p_wait(g_prof_every > 0, PROF_WAIT, i, a, s_main);
        ADMM_HIP(hipStreamWaitEvent(s_main, ev_inv[r], 0));
        p_wait.close();
      }
      Ainv = a->ainv_pool + (size_t)(r - first) * ainv_elems;
      cur = r;
    }
    float dual_div = 1.0f;
    if (i % a->rho_period == 0) dual_div = (rho * 2 <= a->rho_max) ? 2.0f : (float)(a->rho_max / rho);
    const float* G_prev = (i == 0) ? a->W0 : a->G_ring + (size_t)(i - 1) * nw;
    float* G = a->G_ring + (size_t)i * nw;
    int8_t* Gq = a->Gq_ring ? a->Gq_ring + (size_t)i * nw : nullptr;
    float* bstar = has_b ? a->b_ring + (size_t)i * c2 : nullptr;
    effq_fp_state* st = a->state_ring + i;
    // (rho changes at the end of iterations 0, period, 2 period ...: the iteration after sees a rescaled dual)
    const bool after_rho = i > 0 && (i - 1) % a->rho_period == 0 && rho_changed_last;
    // (iterations since rho last changed: the drift from call to call - and with it the length of the lists the
    // trajectory kernel's single last workgroup has to scan - is largest right after a change)
    const int since_rho = i - rho_changed_at;
    const bool use_traj = traj && i > 1 && !(traj_rho_old && after_rho) && since_rho >= traj_after;
    // ... and then the trajectory kernel adds the K slices of the product up in its prologue: one launch less
    static const bool fuse_reduce = !(getenv("EFFQ_FUSE_REDUCE") && atoi(getenv("EFFQ_FUSE_REDUCE")) == 0);   // A/B switch
    const float* parts = nullptr;
    int parts_n = 1, parts_ld = 0;
    // ---- the chain (main stream) ----
    const bool prof = g_prof_every > 0 && !use_shift && (i % g_prof_every) == g_prof_every / 2;
    ProfScope p_prox(prof, PROF_PROX, i, a, s_main);
    if (!use_shift && bm_ready && use_traj && fuse_reduce && c2 <= 512)
      ADMM_RC(effq_prox_solve_prebuilt_parts(a->B0, Ainv, a->W0, a->b0, G_prev, a->dual, c2, n, has_b, rho, a->eta, a->wstar,
                                             bstar, a->prox_ws, a->prox_ws_bytes, s_main, &parts, &parts_n, &parts_ld));
    else if (use_shift)
      ADMM_RC(effq_prox_solve_shifted(a->B0, a->ainv_pool, a->W0, a->b0, G_prev, a->dual, c2, n, has_b, rho, a->eta,
                                      plan.rho[1], shift_terms(rho, a->eta, plan.rho[1]), a->wstar, bstar, a->prox_ws,
                                      a->prox_ws_bytes, s_main));
    else if (bm_ready)
      ADMM_RC(effq_prox_solve_prebuilt(a->B0, Ainv, a->W0, a->b0, G_prev, a->dual, c2, n, has_b, rho, a->eta, a->wstar,
                                       bstar, a->prox_ws, a->prox_ws_bytes, s_main));
    else
      ADMM_RC(effq_prox_solve(a->B0, Ainv, a->W0, a->b0, G_prev, a->dual, c2, n, has_b, rho, a->eta, a->wstar, bstar,
                              a->prox_ws, a->prox_ws_bytes, s_main));
    p_prox.close();
    ProfScope p_fp(prof, PROF_FIXED_POINT, i, a, s_main);
    // The projection + dual update as the EPILOGUE of the fixed point where that is a single workgroup (weights <= 32768):
    // one launch per iteration less.  The projection also leaves the right-hand side of the NEXT prox solve in the prox
    // workspace; the first solve of the layer builds Bm itself (bias column, padding).
    double rho_next = rho;
    if (i % a->rho_period == 0) rho_next = (rho * 2 <= a->rho_max) ? rho * 2 : a->rho_max;
    const bool next_rhs = fuse_build && i + 1 < a->iters;
    ProjFused pf;
    memset(&pf, 0, sizeof(pf));
    bool fuse_proj = false;
    {
      const int nwrow = n - has_b;
      const uintptr_t al16 = reinterpret_cast<uintptr_t>(a->v) | reinterpret_cast<uintptr_t>(a->wstar) |
                             reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(a->dual) |
                             (next_rhs ? (reinterpret_cast<uintptr_t>(a->W0) | reinterpret_cast<uintptr_t>(bm)) : 0);
      const bool vec_ok = (al16 & 15) == 0 && (reinterpret_cast<uintptr_t>(Gq) & 3) == 0 && (nw % 4) == 0 &&
                          (!next_rhs || ((nwrow % 4) == 0 && (bm_ld % 4) == 0));
      // measured (us per iteration, fused against fixed point + projection): 2048 weights 13.8 against 12.2 + 7.1, 3456 at
      // 256 levels 231.8 against 229.3 + 8.3 - but 27648 weights on the bucketed kernel 56.3 against 33.6 + 7.1: only the
      // 256 threads of its iteration phase are left for the epilogue.  So: the all-values kernel's layers only
      // (EFFQ_FUSE_PROJ=2 also fuses the bucketed single-workgroup kernel: A/B switch)
      static const int fuse_proj_mode = getenv("EFFQ_FUSE_PROJ") ? atoi(getenv("EFFQ_FUSE_PROJ")) : 1;
      const bool one_wg = (bucket && nw <= 32768 && fuse_proj_mode >= 2) || (!bucket && nw <= effq_fp_small_max());
      if (vec_ok && one_wg && fuse_proj_mode != 0) {
        fuse_proj = true;
        pf.wstar = a->wstar; pf.G = G; pf.dual = a->dual; pf.Gq = Gq; pf.err_flag = a->err_flag;
        pf.d = 2.0 / (double)(a->w_levels - 1); pf.dual_div = dual_div; pf.lm1 = a->w_levels - 1;
        pf.n4 = (unsigned)(nw / 4);
        if (next_rhs) {
          pf.nx.Bm = bm; pf.nx.B0 = a->B0; pf.nx.W0 = a->W0; pf.nx.nwrow = nwrow; pf.nx.n = n; pf.nx.ldb = bm_ld;
          pf.nx.rho = (float)rho_next; pf.nx.eta = (float)a->eta;
        }
      }
    }
    void* rec = traj ? a->fp_pred : nullptr;
    if (use_traj && parts != nullptr) {
      fuse_proj = false;
      ADMM_RC(effq_fixed_point_traj_parts(parts, parts_n, parts_ld, c2, n - has_b, has_b ? 1 : 0, a->dual, a->wstar, bstar,
                                          a->v, a->w_levels, -1.0, 1.0, a->tol, 100 * a->w_levels, st, a->fp_pred,
                                          a->fp_traj_ws, a->fp_traj_ws_bytes, s_main));
    } else if (use_traj) {
      fuse_proj = false;
      ADMM_RC(effq_fixed_point_traj(a->wstar, a->dual, a->v, nw, a->w_levels, -1.0, 1.0, a->tol, 100 * a->w_levels, st,
                                    a->fp_pred, a->fp_traj_ws, a->fp_traj_ws_bytes, s_main));
    } else if (bucket) {
      int fused = 0;
      ADMM_RC(effq_fixed_point_bucket_fused(a->wstar, a->dual, a->v, nw, a->w_levels, -1.0, 1.0, a->tol,
                                            100 * a->w_levels, st, a->fp_ws, a->fp_ws_bytes, rec,
                                            fuse_proj ? &pf : nullptr, &fused, s_main));
      fuse_proj = fused != 0;
    } else if (nw <= effq_fp_small_max()) {
      ADMM_RC(effq_fixed_point_small_fused(a->wstar, a->dual, a->v, nw, a->w_levels, -1.0, 1.0, a->tol,
                                           100 * a->w_levels, st, fuse_proj ? &pf : nullptr, s_main));
    } else {
      fuse_proj = false;
      ADMM_RC(effq_fixed_point_coop_rec(a->wstar, a->dual, a->v, nw, a->w_levels, -1.0, 1.0, a->tol, 100 * a->w_levels,
                                        st, a->red_ws, rec, s_main));
    }
    p_fp.close();
    if (fuse_proj) {
      bm_ready = next_rhs;
    } else {
      ProfScope p_pr(prof, PROF_PROJECT, i, a, s_main);
      if (next_rhs) {
        ADMM_RC(effq_project_dual_next(a->v, a->wstar, st, a->w_levels, G, a->dual, dual_div, Gq, nw, a->err_flag, bm,
                                       a->B0, a->W0, n - has_b, n, bm_ld, rho_next, a->eta, s_main));
        bm_ready = true;
      } else {
        ADMM_RC(effq_project_dual_checked(a->v, a->wstar, st, a->w_levels, G, a->dual, dual_div, Gq, nw, a->err_flag,
                                          s_main));
        bm_ready = false;
      }
      p_pr.close();
    }
    if (a->res_ring != nullptr) {                 // lwq_verbose: residuals of this iteration (w* is overwritten by the next)
      size_t nb = (nw + 1023) / 1024;
      if (nb > 256) nb = 256;
      hipLaunchKernelGGL(k_admm_residuals, dim3((unsigned)nb), dim3(256), 0, s_main, a->wstar, G, G_prev, nw,
                         a->res_ring + 2 * (size_t)i);
    }
    // ---- the losses (loss stream), in groups of LOSS_GROUP iterates ----
    // An event record is a barrier packet in the main queue: the next chain kernel starts ~7 us later than it would
    // behind a kernel (kernel trace: the only gap of an iteration sat between the projection and the next prox GEMM).
    // The iterates are kept in rings, so the loss stream may as well pick them up a few at a time.
    if (fork_loss && ((i + 1) % loss_group == 0 || i + 1 == a->iters)) {
      hipEvent_t e = ev_main[(i / loss_group) % EV_POOL];
      ADMM_HIP(hipEventRecord(e, s_main));
      ADMM_HIP(hipStreamWaitEvent(s_loss, e, 0));
    }
    if (a->loss_kind == 5 && (!fork_loss || (i + 1) % loss_group == 0 || i + 1 == a->iters)) {
      // the whole group in one launch pair (the digit planes of the Gram system are read once per group)
      for (int j = loss_next; j <= i; j += 16) {
        const int cnt = (i - j + 1 < 16) ? (i - j + 1) : 16;
        ProfScope p_loss(g_prof_every > 0 && (j / loss_group) % 2 == 1, PROF_LOSS, j, a, s_loss);
        ADMM_RC(effq_gram_loss_i8(a->loss_planes, a->loss_nplanes, a->loss_Au, a->loss_Bu, a->loss_syy,
                                  a->Gq_ring + (size_t)j * nw, has_b ? a->b_ring + (size_t)j * c2 : nullptr, a->state_ring + j,
                                  a->act_alpha_dev, a->act_levels, a->w_levels, c2, n, has_b, cnt, a->hist + 2 * (size_t)j,
                                  a->conv_ws, a->conv_ws_bytes, s_loss));
        p_loss.close();
      }
      loss_next = i + 1;
    } else if (!fork_loss || (i + 1) % loss_group == 0 || i + 1 == a->iters) {
      for (int j = loss_next; j <= i; ++j) {
        const float* Gj = a->G_ring + (size_t)j * nw;
        const int8_t* Gqj = a->Gq_ring ? a->Gq_ring + (size_t)j * nw : nullptr;
        const float* bj = has_b ? a->b_ring + (size_t)j * c2 : nullptr;
        const effq_fp_state* stj = a->state_ring + j;
        double* sq = a->hist + 2 * (size_t)j;
        const bool profj = g_prof_every > 0 && !(plan.shifted_first && j == 0) && (j % g_prof_every) == g_prof_every / 2;
        ProfScope p_loss(profj, PROF_LOSS, j, a, s_loss);
        if (a->loss_kind == 1)
          ADMM_RC(conv3d_calib_step_i8(a->xidx, Gqj, bj, a->y_fp, &a->geom, a->act_alpha_dev, a->act_levels, stj,
                                       a->w_levels, sq, a->conv_ws, a->conv_ws_bytes, s_loss));
        else if (a->loss_kind == 2)
          ADMM_RC(conv3d_calib_step_i8s(a->xidx, Gqj, bj, a->y_fp, &a->geom, a->act_alpha_dev, a->act_levels, stj,
                                        a->w_levels, j == 0 ? 1 : 0, sq, a->conv_ws, a->conv_ws_bytes, s_loss));
        else if (a->loss_kind == 4)
          ADMM_RC(effq_gram_loss(a->loss_Au, a->loss_Bu, a->loss_syy, Gj, bj, c2, n, has_b, sq, a->conv_ws,
                                 a->conv_ws_bytes, s_loss));
        else
          ADMM_RC(conv3d_quant_calib_step(a->xq, Gj, bj, a->y_fp, nullptr, &a->geom, nullptr, 0, sq, nullptr, a->conv_ws,
                                          a->conv_ws_bytes, s_loss));   // unweighted MSE (quirk Q5)
        p_loss.close();
      }
      loss_next = i + 1;
    }
    if (i % a->rho_period == 0) {
      const double rho_new = (rho * 2 <= a->rho_max) ? rho * 2 : a->rho_max;
      rho_changed_last = rho_new != rho;
      if (rho_changed_last) rho_changed_at = i + 1;
      rho = rho_new;
    }
  }
  if (!side_enqueued) ADMM_RC(enqueue_side_inverses());      // (fewer iterations than SIDE_AFTER_ITERS)
  // join: everything the caller reads next (hist, rings) is ordered on the main stream
  ProfScope p_join(g_prof_every > 0 && (fork_loss || fork_side), PROF_WAIT, a->iters, a, s_main);
  if (fork_loss) {
    ADMM_HIP(new_event(&ev_join_loss));
    ADMM_HIP(hipEventRecord(ev_join_loss, s_loss));
    ADMM_HIP(hipStreamWaitEvent(s_main, ev_join_loss, 0));
  }
  if (fork_side && n_inv > 1) {
    ADMM_HIP(new_event(&ev_join_side));
    ADMM_HIP(hipEventRecord(ev_join_side, s_side));
    ADMM_HIP(hipStreamWaitEvent(s_main, ev_join_side, 0));
    if (two_sides) {
      hipEvent_t ev_join_side2 = nullptr;
      ADMM_HIP(new_event(&ev_join_side2));
      ADMM_HIP(hipEventRecord(ev_join_side2, s_side2));
      ADMM_HIP(hipStreamWaitEvent(s_main, ev_join_side2, 0));
    }
  }
  p_join.close();
  destroy_events();
#undef ADMM_HIP
#undef ADMM_RC
  return EFFQ_OK;
}

size_t effq_gram_packed_elems(int n, int c2) {
  return (n > 0 && c2 > 0) ? (size_t)n * (n + 1) / 2 + (size_t)c2 * n : 0;
}

int effq_gram_pack(const float* A0, const float* B0, int n, int c2, float* buf, void* stream) {
  EFFQ_CHECK_ARG(A0 && B0 && buf && n > 0 && c2 > 0);
  size_t blocks = ((size_t)n * n + (size_t)c2 * n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_gram_pack, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), A0, B0, n, c2, buf);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_gram_unpack(const float* buf, int n, int c2, float* A0, float* B0, void* stream) {
  EFFQ_CHECK_ARG(A0 && B0 && buf && n > 0 && c2 > 0);
  size_t blocks = ((size_t)n * n + (size_t)c2 * n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_gram_unpack, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), buf, n, c2, A0, B0);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

int effq_prof_enable(int every) {
  for (ProfRec& r : g_prof) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  g_prof_every = every > 0 ? every : 0;
  return EFFQ_OK;
}

int effq_prof_count(void) { return (int)g_prof.size(); }

int effq_prof_read(int i, effq_prof_record* out) {
  EFFQ_CHECK_ARG(out != nullptr && i >= 0 && i < (int)g_prof.size());
  const ProfRec& r = g_prof[(size_t)i];
  float ms = 0.0f;
  EFFQ_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));      // the caller has synchronised the device
  out->kind = r.kind;
  out->iter = r.iter;
  out->loss_kind = r.loss_kind;
  out->c2 = r.c2;
  out->n = r.n;
  out->geom = r.geom;
  out->ms = ms;
  return EFFQ_OK;
}

int effq_admm_select_best(const double* hist, int iters, const float* G_ring, const float* b_ring, size_t nw, size_t nb,
                          float* best_G, float* best_b, double* best_out, void* stream) {
  EFFQ_CHECK_ARG(hist && G_ring && best_G && best_out && iters > 0 && nw > 0);
  EFFQ_CHECK_ARG((b_ring == nullptr) == (best_b == nullptr));
  size_t blocks = (nw + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_select_best, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), hist, iters, G_ring, b_ring,
                     nw, nb, best_G, best_b, best_out);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
