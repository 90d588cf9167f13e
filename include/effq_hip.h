/*
 * effq_hip.h -- C ABI of the MI355X (gfx950) hot path of EfficientQ's layer-wise
 * PTQ calibration.  This is the drop-in boundary: plain pointers and sizes, no
 * torch types, every call returns an int status (EFFQ_OK == 0; no exceptions
 * cross the ABI), every buffer is a caller-owned DEVICE pointer unless the
 * parameter says "host", every call takes the HIP stream to enqueue on
 * (a hipStream_t passed as void*; NULL = default stream) and never
 * synchronises unless documented.  The reference has no FFI of its own (it is
 * pure Python); each entry point cites the reference code it
 * replaces (paths relative to the reference checkout).
 *
 * Layouts
 *   activations / targets : NDHWC fp32 (torch channels_last_3d), x[n][d][h][w][c]
 *   attention mask        : [n][D'][H'][W'] fp32 (one weight per output voxel)
 *   weights               : reference layout [c2][c1][kd][kh][kw] fp32
 *   Gram system           : A0 [n x n], B0 [c2 x n] row-major fp32, n = c1*k^3 (+1 bias),
 *                           row order (c1,kd,kh,kw)+bias exactly as solver.py:104-108,256
 */
#ifndef EFFQ_HIP_H
#define EFFQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  EFFQ_OK = 0,
  EFFQ_ERR_ARG = 1,        /* bad argument (null pointer, bad shape, unsupported geometry) */
  EFFQ_ERR_HIP = 2,        /* a HIP runtime call failed; see effq_last_error() */
  EFFQ_ERR_WORKSPACE = 3,  /* workspace too small; query the *_ws_bytes function */
  EFFQ_ERR_NO_DEVICE = 4,
  EFFQ_ERR_NOT_CONVERGED = 5
};

/* Conv geometry.  Dilation and groups are 1 (the reference's solver ignores
 * them, solver.py:86-111, and both shipped configs use 1). */
typedef struct effq_geom {
  int32_t N, C1, C2;
  int32_t D, H, W;          /* input spatial size  */
  int32_t KD, KH, KW;       /* 1 or 3 per axis      */
  int32_t SD, SH, SW;       /* stride               */
  int32_t PD, PH, PW;       /* symmetric zero pad   */
} effq_geom;

const char* effq_last_error(void);
int effq_version(void);
/* Number of HIP devices visible; does not initialise a context. */
int effq_device_count(int* count);

/* ---- a1/a3: discretize + PTQConv._quantize_act -------------------------------
 * layer_helper.py:25-37, PTQConv.py:114-116.
 * fp32 path: y = (rint((clamp(x/alpha,lo,hi)-lo)/d)*d+lo)*alpha with d=f32((hi-lo)/(L-1)),
 * IEEE divisions, round-half-even, no FMA contraction.  alpha is read from DEVICE
 * memory (one float).  idx_out (uint8 level ids) and y_out may each be NULL. */
int effq_quant_dequant_f32(const float* x, const float* alpha_dev, float lo, float hi, int levels,
                           float* y_out, uint8_t* idx_out, size_t n, void* stream);

/* fp64 path used during calibration (project_by_iter's final discretize,
 * layer_helper.py:50-66, then "a * b", EfficientQConv.py:68-70):
 * b = f32(discretize(f64(x)/alpha)), y = f32(alpha)*b.  alpha is a DEVICE double. */
int effq_quant_dequant_f64path(const float* x, const double* alpha_dev, double lo, double hi, int levels,
                               float* y_out, float* b_out, uint8_t* idx_out, size_t n, void* stream);

/* ---- a2: project_by_iter pieces (layer_helper.py:40-70) -----------------------
 * Workspace for all reductions below: effq_reduce_ws_bytes() bytes, zero-initialised
 * once by the caller (hipMemset) and then owned by this library between calls. */
size_t effq_reduce_ws_bytes(void);

/* sums_out[0] = sum |x_i| (fp64), sums_out[1] = n (2 doubles).  (a = var.abs().mean(), layer_helper.py:51) */
int effq_abs_sum_f64(const float* x, size_t n, double* sums_out, void* ws, void* stream);

/* sums_out[0] = sum x, [1] = sum x^2 (fp64), [2] = n.  (Tensor.std(), EfficientQConv.py:46,48) */
int effq_moments_f64(const float* x, size_t n, double* sums_out, void* ws, void* stream);

/* One fixed-point statistics pass: b = discretize(f64(x)/alpha); sums_out[0] = sum b*x,
 * sums_out[1] = sum b*b (layer_helper.py:57-59).  alpha is a DEVICE double.  If
 * done_flag_dev is non-NULL and *done_flag_dev != 0 the pass is skipped. */
int effq_alpha_stats_f64(const float* x, const double* alpha_dev, double lo, double hi, int levels,
                         size_t n, double* sums_out, const int32_t* done_flag_dev, void* ws, void* stream);

/* Fixed-point state on the device: {alpha, alpha_prev, sums[2], iters, done}. */
typedef struct effq_fp_state {
  double alpha;
  double alpha_prev;
  double sums[2];
  int32_t iters;
  int32_t done;      /* 1 converged, 2 hit max_iter (the reference raises, layer_helper.py:62-64) */
} effq_fp_state;

/* state.alpha = sums[0]/sums[1] (abs-mean start), alpha_prev=-999, iters=0, done=0. */
int effq_fp_init(effq_fp_state* state_dev, const double* abs_sums_dev, void* stream);
/* alpha_prev=alpha; alpha=sums[0]/sums[1]; ++iters; done when |alpha-alpha_prev|<=tol or iters==max_iter.
 * Reads state->sums (so an all-reduce of state->sums may run between stats and update). */
int effq_fp_update(effq_fp_state* state_dev, double tol, int max_iter, void* stream);

/* Fixed point on the stream without host round trips: runs `n_iters` fused iterations (statistics pass
 * whose last block also applies the update; each a no-op once done).  The caller checks state.done. */
int effq_alpha_fixed_point(const float* x, size_t n, int levels, double lo, double hi, double tol,
                           int max_iter, int n_iters, effq_fp_state* state_dev, void* ws, void* stream);

/* Whole project_by_iter of a SMALL tensor (n <= effq_fp_small_max()) in one launch, no host round trip:
 * v = a + b (b may be NULL; v is written to v_out when given, required if b != NULL), alpha0 = mean|v|,
 * then the fixed point runs on chip until |d alpha| <= tol or max_iter.  The ADMM weight projection of
 * most layers (EfficientQConv.py:108 -> layer_helper.py:40-70). */
size_t effq_fp_small_max(void);
int effq_fixed_point_small(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                           double tol, int max_iter, effq_fp_state* state_dev, void* stream);
/* Same contract for larger tensors (n <= effq_fp_coop_max()): one COOPERATIVE launch of ceil(n/27648) <= 256
 * workgroups (one per CU, slice of v resident in LDS) that meet at a bounded-spin grid barrier once per
 * iteration; partial sums are combined in workgroup order by every workgroup (deterministic).  state.done = 3
 * reports a barrier time-out.  ws: the reduction workspace (effq_reduce_ws_bytes()), zero-filled once by the caller:
 * the kernel keeps its barrier words in the tail of it and leaves them at zero.  A time-out POISONS the workspace: every
 * later launch on it (e.g. the following ADMM iterations, already enqueued) returns at once with state.done = 3 and
 * touches nothing, until the caller zero-fills the workspace again.
 * effq_fp_coop_set_spin_limit: polls a workgroup waits at the barrier before it gives up (0 = the default, 2^24); a
 * process-wide test hook - tests force the time-out path with a small value. */
size_t effq_fp_coop_max(void);
int effq_fixed_point_coop(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                          double tol, int max_iter, effq_fp_state* state_dev, void* ws, void* stream);
int effq_fp_coop_set_spin_limit(unsigned int polls);
/* Same contract again (n <= effq_fp_bucket_max(), levels <= 256) without a per-iteration pass over the tensor: the
 * values are counted into equal-width buckets (exact integer sum per bucket) and regrouped by bucket once; each
 * iteration then reads prefix tables and looks only at the values of the bucket a level boundary falls into (those in
 * a 1e-6 relative guard band around the boundary are classified with the reference's own fp64 arithmetic).  Level
 * counts are exactly the reference's, alpha agrees to ~1e-14 relative (fp64 sums in another order), same iteration
 * count; deterministic (integer partial sums).  n <= 32768: one workgroup, everything in LDS, ws unused.  Larger:
 * four launches (sum|v| and range / count with global atomics / scan / regroup + iterate in the last workgroup to
 * finish).  ws: effq_fp_bucket_ws_bytes(n) bytes, ZERO-FILLED once by the caller and owned by the library between
 * calls (its counters are left at zero). */
size_t effq_fp_bucket_max(void);
size_t effq_fp_bucket_ws_bytes(size_t n);
int effq_fixed_point_bucket(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                            double tol, int max_iter, effq_fp_state* state_dev, void* ws, size_t ws_bytes,
                            void* stream);

/* project_by_iter for the weight projection INSIDE the ADMM loop (EfficientQConv.py:108 -> layer_helper.py:40-70) in ONE
 * launch, from the previous ADMM iteration's iterates: v_k = w*_k + dual_{k-1} differs little from v_{k-1}, so the i-th
 * iterate of this call lies within 1e-3 ... 1e-9 of the i-th iterate of the last one.  `pred_dev`
 * (effq_fp_traj_pred_bytes() bytes of device memory, zero-filled = nothing known) carries them from call to call with a
 * margin each.  One pass over the values by all workgroups forms v, sums |v| and tallies - in integers - every value
 * whose level is the same at both ends of a predicted bracket (the level is monotone in the scale); the few per cent
 * that are not go to a list.  The last workgroup to finish then iterates: an iterate inside its predicted bracket costs
 * one scan of the list; one outside costs a pass of that single workgroup over v (slow, rare).  Levels are exactly the
 * reference's, alpha within ~1e-15 of the fp64 kernels', same iteration count, deterministic.  levels <= 16,
 * n <= effq_fp_traj_max().  ws: effq_fp_traj_ws_bytes(n), zero-filled once (its counters are left at zero).
 * The *_rec variants of the older fixed points do the same work as their namesakes and, when pred_dev != NULL, leave
 * their iterates in it (the first call of a layer, calls after rho has changed). */
size_t effq_fp_traj_max(void);
size_t effq_fp_traj_ws_bytes(size_t n);
size_t effq_fp_traj_pred_bytes(void);
int effq_fixed_point_traj(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                          double tol, int max_iter, effq_fp_state* state_dev, void* pred_dev, void* ws, size_t ws_bytes,
                          void* stream);
int effq_fixed_point_bucket_rec(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                                double tol, int max_iter, effq_fp_state* state_dev, void* ws, size_t ws_bytes,
                                void* pred_dev, void* stream);
int effq_fixed_point_coop_rec(const float* a, const float* b, float* v_out, size_t n, int levels, double lo, double hi,
                              double tol, int max_iter, effq_fp_state* state_dev, void* ws, void* pred_dev, void* stream);

/* project_by_iter (layer_helper.py:40-70) on tensors far too large for the chip - the activations of a layer
 * (PTQConv.py:74-78, EfficientQConv.py:64-72) - without a pass over the whole tensor per iteration.  The level of a value
 * is monotone in the scale, so a value whose level is the same at both ends of a bracket that confines the remaining
 * iterates is settled: a narrowing pass moves its contribution into integer tallies and leaves the unsettled values in
 * a compact list, which is all the following iterations read (and narrow further).  The bracket is a prediction from
 * the last iterates (up to the limit of the sequence, or a horizon of a few iterations while it converges slowly); an
 * iterate that leaves it restarts from the tensor or the list of its non-zeros (correctness never depends on it).
 * Levels are exactly the reference's in every iteration; sums are integers in units of 2^-e (order-independent:
 * deterministic, identical for any split of the tensor); alpha agrees with the fp64 kernels to ~1e-13, same iteration
 * count.  levels <= 256.
 *   init  : state.alpha = abs_sums[0] / abs_sums[1] (the all-reduced sum|x| and count: abs-mean start), unit and plan;
 *           list_first != 0: the first pass already lists the values that are not settled for every scale (the
 *           non-zeros - worth it for post-ReLU tensors, a wasted copy for dense ones).
 *   run   : n_iters x {iteration pass + finish (sums, scalar update, plan of the next pass)}; no-ops once state.done.
 *   stats : one iteration pass, state.sums = this rank's [sum b*x, sum b*b]   } with data-parallel ranks the caller
 *   update: scalar update from state.sums + plan of the next pass              } all-reduces state.sums in between
 * ws: effq_fp_bracket_ws_bytes(n) bytes (three lists of n floats + tallies), owned by the fit between init and its end.
 * The first 128 bytes of ws are sixteen 8-byte words of diagnostics (bracket, plan, escapes, narrowings, values read). */
size_t effq_fp_bracket_ws_bytes(size_t n);
int effq_fp_bracket_init(effq_fp_state* state_dev, const double* abs_sums_dev, size_t n, int levels, int list_first,
                         void* ws, size_t ws_bytes, void* stream);
int effq_fp_bracket_run(const float* x, size_t n, int levels, double lo, double hi, double tol, int max_iter, int n_iters,
                        effq_fp_state* state_dev, void* ws, void* stream);
int effq_fp_bracket_stats(const float* x, size_t n, int levels, double lo, double hi, effq_fp_state* state_dev, void* ws,
                          void* stream);
int effq_fp_bracket_update(size_t n, int levels, double lo, double hi, double tol, int max_iter,
                           effq_fp_state* state_dev, void* ws, void* stream);
/* Data-parallel ranks, "gather once": after a few all-reduced iterations (stats / update above) a rank's shard is, under
 * the current bracket, four integer tallies of the decided values + the list of the undecided ones.
 *   export: out (effq_fp_bracket_export_words() int64): [0..3] = the tallies, [4] = the list's length (-1: no list yet, -2:
 *           the current bracket is a horizon the iterates are meant to leave - many levels - so not worth exchanging),
 *           [5], [6] = the bit patterns of the bracket [blo, bhi] the tallies and the list are valid under (ranks plan on
 *           their own shard: brackets may differ), the rest scratch; list_out[0 .. list_cap) = the list, zero-filled behind
 *           it (all zeros if there is none or it does not fit).  The caller all-reduces a pack {tallies[4], then per rank
 *           (length, blo bits, bhi bits)} and all-gathers the list_cap floats of every rank (zero padding is harmless for
 *           the unsigned quantiser lo = 0: an exact zero has level 0 at every scale);
 *   import: sets up ws_dst (effq_fp_bracket_ws_bytes(world * list_cap)) as a fit over the gathered lists with the summed
 *           tallies as its constant part and the iterates / unit of ws_src, valid under the INTERSECTION of the ranks'
 *           brackets; effq_fp_bracket_run(gathered, world * list_cap, ..., ws_dst) then finishes WITHOUT collectives,
 *           bit-identically on every rank.  Whether the exchange is usable (every list fitted, the iterate lies in the
 *           intersection) is decided on the device - the host never needs the lengths: if not, or once an iterate leaves
 *           that bracket, state.done = 4 (the launches that follow are no-ops) and the caller goes on with stats / update
 *           on the rank's own workspace after effq_fp_bracket_rebase. */
size_t effq_fp_bracket_export_words(void);
int effq_fp_bracket_export(const void* ws, size_t n, long long* out, float* list_out, size_t list_cap, void* stream);
int effq_fp_bracket_import(const void* ws_src, size_t n_src, const long long* pack_dev, int world, size_t list_cap,
                           effq_fp_state* state_dev, void* ws_dst, size_t ws_dst_bytes, void* stream);
/* after state.done = 4: clears it and makes the rank's OWN workspace start its next pass from the base (its list and
 * tallies belong to a bracket the iterates have moved on from while the imported fit ran) */
int effq_fp_bracket_rebase(effq_fp_state* state_dev, void* ws, size_t n, void* stream);
/* Sticky device-side check used by stream-resident loops: *err_flag_dev = 2 (cap hit; the reference
 * raises, layer_helper.py:62-64) or 3 (not finished) unless state.done == 1. */
int effq_fp_check(const effq_fp_state* state_dev, int32_t* err_flag_dev, void* stream);

/* ---- a5/a6: im2col + getA0B0 (solver.py:86-111, 282-314), never materialising x_col ----
 * A0 = 2*sum_v att_v xhat_v xhat_v^T, B0 = 2*sum_v att_v y_v xhat_v^T; xhat has a trailing 1
 * when has_bias.  att may be NULL (all ones).  accumulate!=0 adds into A0/B0 (sharded volumes).
 * ws: effq_gram_ws_bytes(geom) bytes. */
size_t effq_gram_ws_bytes(const effq_geom* g, int has_bias);
int effq_gram_accum(const float* x_ndhwc, const float* att, const float* y_ndhwc, const effq_geom* g,
                    int has_bias, float* A0, float* B0, int accumulate, void* ws, size_t ws_bytes,
                    void* stream);

/* The same A0/B0 for a layer whose input is already quantised (EfficientQConv.py:64-72 ran first), evaluated
 * exactly on the i8 matrix cores: xidx = level ids of the quantised input (uint8, NDHWC, value k means
 * xhat = alpha_act*k/(act_levels-1)), act_alpha_dev = device float.  The attention weights enter as a voxel
 * list sorted by weight value: vox_list[n_list] (output-voxel indices, -1 = padding; every run of 128 entries
 * has one weight), chunk_cls[n_list/128] = class of each run, cls_w_dev[ncls] = the class weights (device
 * floats, ncls <= 16).  vox_list == NULL: all weights 1 (then chunk_cls = NULL, ncls = 1, n_list = 0).
 * Requires effq_gram_i8_supported (C1 % 16 == 0, at most 31 taps, act_levels <= 128); results equal
 * effq_gram_accum on xhat up to the fp32 rounding of that path (integer sums here are exact).
 * ws: effq_gram_i8_ws_bytes(geom, ncls). */
int effq_gram_i8_supported(const effq_geom* g, int act_levels);
size_t effq_gram_i8_ws_bytes(const effq_geom* g, int ncls);
int effq_gram_accum_i8(const uint8_t* xidx_ndhwc, const float* y_ndhwc, const effq_geom* g, int has_bias,
                       const float* act_alpha_dev, int act_levels, const int32_t* vox_list,
                       const int32_t* chunk_cls, const float* cls_w_dev, int ncls, long long n_list,
                       float* A0, float* B0, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* The same pass with the UNWEIGHTED system as a by-product, in fp64: Au [n][n] = sum_v xhat xhat^T (reference row order,
 * ones row included, no factor 2), Bu [c2][n] = sum_v y xhat^T - the integer class slabs summed without the attention
 * weights.  Au / Bu may both be NULL (= effq_gram_accum_i8). */
int effq_gram_accum_i8_unw(const uint8_t* xidx_ndhwc, const float* y_ndhwc, const effq_geom* g, int has_bias,
                           const float* act_alpha_dev, int act_levels, const int32_t* vox_list, const int32_t* chunk_cls,
                           const float* cls_w_dev, int ncls, long long n_list, float* A0, float* B0, int accumulate,
                           double* Au, double* Bu, void* ws, size_t ws_bytes, void* stream);

/* The unweighted system of a layer whose input is NOT quantised (first conv / classifier, q_first = q_last = "256,-1":
 * definer.py:296-299, model_blk.py:98-107), in fp64 on the matrix cores: Au [n][n] = sum_v xhat xhat^T with
 * xhat = [im2col patch of x (solver.py:86-111 row order); 1 if has_bias], Bu [c2][n] = sum_v y xhat^T.  fp32 inputs, exact
 * products, fp64 accumulation, deterministic (partial slabs added in workgroup order).  Operands of effq_gram_loss for
 * those layers.  Requires effq_gram_f64_supported (n = C1*KD*KH*KW + has_bias <= 128, C2 <= 64).
 * ws: effq_gram_f64_ws_bytes(geom, has_bias). */
int effq_gram_f64_supported(const effq_geom* g, int has_bias);
size_t effq_gram_f64_ws_bytes(const effq_geom* g, int has_bias);
int effq_gram_f64(const float* x_ndhwc, const float* y_ndhwc, const effq_geom* g, int has_bias, double* Au, double* Bu,
                  void* ws, size_t ws_bytes, void* stream);

/* ---- the loss of one iterate from the unweighted Gram system (EfficientQConv.py:118-122 without the pass over the voxels)
 * sum_v,c (conv(Qx, G, b)_v,c - y_v,c)^2 = sum_c g_c^T Au g_c - 2 sum_c g_c . Bu_c + syy with g_c = [G[c,:], b_c], in fp64:
 * sqerr_out[0] = sqerr_out[1] = that sum (the unweighted squared error, as the conv entry points report it).  c2 n^2
 * multiply-adds on an n x n matrix instead of a pass over all voxels: for layers whose voxel count is far above n.
 * ws: effq_gram_loss_ws_bytes(n), zero-filled once by the caller. */
size_t effq_gram_loss_ws_bytes(int n);
int effq_gram_loss(const double* Au, const double* Bu, const double* syy_dev, const float* G, const float* b, int c2, int n,
                   int has_bias, double* sqerr_out, void* ws, size_t ws_bytes, void* stream);

/* ---- the same losses for a GROUP of iterates of a wide layer, the quadratic form on the i8 matrix cores (exact integers)
 * sum_c w_c^T Aww w_c = s_w^2 s_a^2 <K, J^T J> with K = Aww / s_a^2 (integer: sums of products of level ids) and
 * J = the int8 level numerators of the iterate (Gq ring of effq_admm_run, as for conv3d_calib_step_i8).
 *   effq_gram_loss_i8_supported: c2 % 32 == 0, (n - has_bias) % 64 == 0, w_levels <= 64;
 *   effq_gram_loss_i8_num_planes(kmax): balanced base-256 digit planes for entries up to kmax (<= (La-1)^2 * voxels), -1 if > 6;
 *   effq_gram_loss_i8_prepare: planes [P][round_up(n - has_bias, 256)][n - has_bias] int8 from Au (effq_gram_accum_i8_unw),
 *     once per layer; *err_flag_dev is set non-zero if Au is not the integer system it should be;
 *   effq_gram_loss_i8: hist_out[j][0] = hist_out[j][1] = sum (out - y)^2 of iterate j = 0 .. count-1 (count <= 16):
 *     Gq [count][c2][n - has_bias], b [count][c2] (NULL without bias), states[j].alpha = the iterate's weight scale;
 *     out = f32(alpha_a) f32(alpha_w) / ((La-1)(Lw-1)) * (J . k) + b in exact arithmetic (the contract of
 *     conv3d_calib_step_i8), evaluated in integers and fp64.  ws: effq_gram_loss_i8_ws_bytes(), zero-filled once. */
int effq_gram_loss_i8_supported(int c2, int n, int has_bias, int w_levels);
int effq_gram_loss_i8_num_planes(long long kmax);
size_t effq_gram_loss_i8_planes_bytes(int n, int has_bias, int nplanes);
int effq_gram_loss_i8_prepare(const double* Au, int n, int has_bias, const float* act_alpha_dev, int act_levels,
                              int nplanes, int8_t* planes, int32_t* err_flag_dev, void* stream);
size_t effq_gram_loss_i8_ws_bytes(void);
int effq_gram_loss_i8(const int8_t* planes, int nplanes, const double* Au, const double* Bu, const double* syy_dev,
                      const int8_t* Gq, const float* b, const effq_fp_state* states, const float* act_alpha_dev,
                      int act_levels, int w_levels, int c2, int n, int has_bias, int count, double* hist_out, void* ws,
                      size_t ws_bytes, void* stream);


/* The voxel list of an attention mask, by three small kernels (distinct weights + counts, segment layout, scatter): the
 * class weights are the few integers quirk Q1 leaves (ptqer.py:161-165).  vox_list: V + 2048 int32, chunk_cls: V/128 + 16
 * int32, cls_w_dev: 16 floats, ws: effq_att_classes_ws_bytes().  info_host_out[3] = {ncls, n_list, overflow}; the call
 * synchronises the stream to return them (once per mask; the layers of a pyramid level share it).  overflow != 0: more
 * than 16 distinct weights (use effq_gram_accum).  The order of the voxels inside a class is not fixed; the sums of
 * effq_gram_accum_i8 are exact integers, so its results do not depend on it. */
size_t effq_att_classes_ws_bytes(void);
int effq_att_classes(const float* att, long long V, int32_t* vox_list, int32_t* chunk_cls, float* cls_w_dev,
                     int32_t* info_host_out, void* ws, void* stream);

/* Creates the helper stream the 256-row sweep keeps per caller stream (otherwise created by the first large inverse): call it
 * for every stream inverses will run on BEFORE anything else creates streams (a communicator, a framework pool), so that the
 * calibration's streams keep hardware queues of their own. */
int effq_spd_inverse_prepare(void* stream);
/* ---- a7: getAB + solve (solver.py:316-345) --------------------------------------
 * Ainv = (A0 + rho*I' + eta*I)^-1 in fp64 (I' has 0 on the bias diagonal), stored fp32 as n rows of
 * effq_ainv_ld(n) floats (row padding is zero; exactly symmetric).
 * The reference refactorises per iteration; A only changes with rho (5 values per layer). */
int effq_ainv_ld(int n);
size_t effq_spd_inverse_ws_bytes(int n);
int effq_spd_inverse(const float* A0, int n, int has_bias, double rho, double eta, float* Ainv,
                     void* ws, size_t ws_bytes, void* stream);

/* What = (B0 + eta*[W0|b0] + rho*[G-dual|0]) * Ainv ; splits into wstar [c2 x (n-1|n)] and bstar [c2].
 * W0, G, dual, wstar in reference weight layout (contiguous c2 x c1k).  b0/bstar NULL when !has_bias.
 * rho/eta are host doubles.  ws: effq_prox_ws_bytes(c2,n). */
size_t effq_prox_ws_bytes(int c2, int n);
int effq_prox_solve(const float* B0, const float* Ainv, const float* W0, const float* b0, const float* G,
                    const float* dual, int c2, int n, int has_bias, double rho, double eta, float* wstar,
                    float* bstar, void* ws, size_t ws_bytes, void* stream);

/* The same solve for A(rho) when only Ainv = A(rho_inv)^-1 is at hand (rho_inv >= rho): A(rho) = A(rho_inv) -
 * d*I' with d = rho_inv - rho, so What = (B + d*[What_w|0]) * Ainv is a contraction with factor
 * < d/(rho_inv + eta) (1/2 for the reference's doubling schedule, EfficientQConv.py:129-137).  nterms sweeps of
 * the GEMM; 26 reach fp32 resolution.  Used for iteration 0, whose rho serves that one iteration only. */
int effq_prox_solve_shifted(const float* B0, const float* Ainv, const float* W0, const float* b0, const float* G,
                            const float* dual, int c2, int n, int has_bias, double rho, double eta,
                            double rho_inv, int nterms, float* wstar, float* bstar, void* ws, size_t ws_bytes,
                            void* stream);

/* ---- a4: ADMM elementwise steps (EfficientQConv.py:108-111,129-137,139-142) ----
 * v = wstar + dual                                  (input of the weight projection) */
int effq_admm_presum(const float* wstar, const float* dual, float* v, size_t n, void* stream);
/* G = f32(alpha)*b with b=f32(discretize(f64(v)/alpha,-1,1)); dual = (wstar - G + dual) / dual_div.
 * dual_div is 1, or 2 / (rho_max/rho) on the rho-schedule iterations (i % 50 == 0).  alpha from state_dev. */
int effq_admm_project_dual(const float* v, const float* wstar, const effq_fp_state* state_dev, int levels,
                           float* G, float* dual, float dual_div, int8_t* Gq_out, size_t n, void* stream);
/* Gq_out (optional, levels <= 128): signed level numerators j' = 2*level-(L-1), so that G = alpha_w*j'/(L-1);
 * the operand of the exact-integer conv below. */
/* ---- the entry point north_star names ------------------------------------------------
 * One ADMM iteration's device work (EfficientQConv.py:118-122,161-165; PTQConv.py:154-167):
 * out = conv3d(xq, G, bias) in fp32 on the matrix cores (f32 MFMA, exact fp32 fma chains),
 * fused with sqerr_out[0] = sum (out-y)^2 and sqerr_out[1] = sum att*(out-y)^2 (fp64 scalars).
 * y_fp may be NULL (plain forward), out may be NULL (loss only), att may be NULL (sqerr[1]=sqerr[0]).
 * If act_alpha_dev != NULL the fp32 quant-dequant of PTQConv._quantize_act (levels act_levels,
 * range [0,1]) is applied to x while staging it (quantised forward, PTQConv.py:163-167).
 * ws: effq_conv_ws_bytes(geom) bytes, ZERO-FILLED once by the caller (it holds the ticket of the last-block
 * reduction, which every launch leaves at zero again; the same holds for the i8 conv workspaces below). */
size_t effq_conv_ws_bytes(const effq_geom* g);
int conv3d_quant_calib_step(const float* xq_ndhwc, const float* G, const float* bias, const float* y_fp,
                            const float* att, const effq_geom* g, const float* act_alpha_dev, int act_levels,
                            double* sqerr_out, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- exact-integer ("int-simulated") form of the per-iteration loss evaluation -------------------
 * Same quantity as conv3d_quant_calib_step(xq, G, bias, y_fp, NULL, ...)'s sqerr_out[0], for quantised
 * activations and projected weights: x = alpha_a*k/(La-1) with level ids k (uint8, NDHWC) and
 * G = alpha_w*j'/(Lw-1) with Gq = j' (int8, reference weight layout).  The contraction runs on the i8
 * matrix cores with exact int32 accumulation; out = f32(alpha_a)*f32(alpha_w)/((La-1)(Lw-1)) * acc + bias.
 * Supported: 3x3x3, stride 1, C1 in {32,64,128,256,512}, C2 % 32 == 0, levels <= 128 (query effq_conv_i8_supported).
 * alpha_a: device float; alpha_w: w_state_dev->alpha.  sqerr_out[0] = sqerr_out[1] = sum (out-y)^2. */
int effq_conv_i8_supported(const effq_geom* g, int act_levels, int w_levels);
size_t effq_conv_i8_ws_bytes(const effq_geom* g);
int conv3d_calib_step_i8(const uint8_t* xidx_ndhwc, const int8_t* Gq, const float* bias, const float* y_fp,
                         const effq_geom* g, const float* act_alpha_dev, int act_levels,
                         const effq_fp_state* w_state_dev, int w_levels, double* sqerr_out, void* ws,
                         size_t ws_bytes, void* stream);
/* The quantised FORWARD of a calibrated layer on the same kernels (PTQConv.py:160-167 with quantised input and weights,
 * and the final loss of EfficientQConv.py:161-166 from the same pass): out (fp32, NDHWC) = the conv output, sqerr_out[0] =
 * sum (out - y)^2, sqerr_out[1] = sum att * (out - y)^2 (att: one weight per output voxel, [N][OD][OH][OW], or NULL: the
 * plain sum).  An exact integer contraction and ONE fp32 multiply-add per output instead of c1 k^3 fp32 products: what
 * the f32 conv of conv3d_quant_calib_step computes, without its rounding.  32 -> 32 and 64 -> 64 channels on volumes the
 * kernels' tiles divide (effq_conv_i8_out_supported); ws as for conv3d_calib_step_i8. */
int effq_conv_i8_out_supported(const effq_geom* g, int act_levels, int w_levels);
int conv3d_quant_forward_i8(const uint8_t* xidx_ndhwc, const int8_t* Gq, const float* bias, const float* y_fp,
                            const float* att, const effq_geom* g, const float* act_alpha_dev, int act_levels,
                            const effq_fp_state* w_state_dev, int w_levels, double* sqerr_out, float* out, void* ws,
                            size_t ws_bytes, void* stream);

/* The same exact-integer loss for the layers the tiled kernels above do not take: few taps*channels
 * (KD*KH*KW*C1 <= 256 with C1 == 4 or C1 % 16 == 0: the first conv, the 1x1x1 convs, the classifier), any
 * stride/padding, and up to 256 levels on either side (q_first/q_last = 256 in the reference's recipes).
 * Gq holds the int8 operands effq_admm_project_dual emits (2*level-(Lw-1), or level-128 when Lw > 128).
 * prepare != 0 (first call of a layer) also rebuilds the per-voxel level sums the Lw > 128 form needs; they
 * live in ws between calls.  ws: effq_conv_i8s_ws_bytes(geom, act_levels, w_levels). */
int effq_conv_i8s_supported(const effq_geom* g, int act_levels, int w_levels);
size_t effq_conv_i8s_ws_bytes(const effq_geom* g, int act_levels, int w_levels);
int conv3d_calib_step_i8s(const uint8_t* xidx_ndhwc, const int8_t* Gq, const float* bias, const float* y_fp,
                          const effq_geom* g, const float* act_alpha_dev, int act_levels,
                          const effq_fp_state* w_state_dev, int w_levels, int prepare, double* sqerr_out,
                          void* ws, size_t ws_bytes, void* stream);

/* ---- f3: tune_activation_range (ptqer.py:238-272) - Adam on every alpha_act, end-to-end MSE, STE through discretize ----
 * Backward of q = discretize(x / alpha, L, 0, 1) * alpha (PTQConv.py:114-116; round with identity gradient,
 * layer_helper.py:13-22; clamp with torch's inclusive mask) given gq = dLoss/dq:
 *   gx_out = gq * mask (may be NULL),  *galpha_out (device double) = sum gq * (r - mask * x / alpha).
 * ws: the reduction workspace (effq_reduce_ws_bytes()).  The input gradient of the conv itself is the conv entry point
 * applied to the output gradient with flipped, transposed weights (stride 1). */
int effq_act_quant_backward(const float* x, const float* alpha_dev, int levels, const float* gq, float* gx_out,
                            double* galpha_out, size_t n, void* ws, void* stream);
/* torch.optim.Adam step (no weight decay, no amsgrad) on n parameters; t = step number starting at 1. */
int effq_adam_step(float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                   int t, size_t n, void* stream);

/* ---- the whole ADMM loop of a layer in ONE call (EfficientQConv.py:99-144) -----------------------------------
 * Enqueues `iters` iterations of { prox solve, weight-scale fixed point, projection + dual update } on stream_main,
 * the loss of each iterate (conv + squared error against y_fp, the reference's per-iteration F.conv3d + F.mse_loss)
 * on stream_loss one iteration behind, and the inverses of A(rho) for the later rho values on stream_side.  stream_loss
 * / stream_side may be NULL (that work then runs on stream_main).  No host synchronisation; on return stream_main is
 * ordered after everything the call enqueued on the other two.
 * Results are kept PER ITERATION (slot i of each ring is written once and never reused, so the three streams need no
 * back-pressure): G_ring [iters][nw] projected weights, Gq_ring [iters][nw] their int8 operands (required for
 * loss_kind 1/2, else may be NULL), b_ring [iters][c2] (NULL without bias), state_ring [iters] scale states
 * (state_ring[iters-1].alpha is the reference's final alpha_w, quirk Q6), hist [iters][2] = {sum (out-y)^2, same}.
 * The best iterate is chosen afterwards by effq_admm_select_best(); a data-parallel caller all-reduces hist first
 * (ONE collective per layer for the 200 per-iteration losses).
 * rho schedule: after iteration i with i % rho_period == 0, rho doubles while 2*rho <= rho_max (else rho = rho_max)
 * and dual is divided by the same factor (EfficientQConv.py:129-137).  One inverse per distinct rho that serves more
 * than one iteration: ainv_pool holds n_ainv >= effq_admm_num_inverses(...) matrices of n*effq_ainv_ld(n) floats.
 * loss_kind: 0 = conv3d_quant_calib_step on xq (fp32), 1 = conv3d_calib_step_i8, 2 = conv3d_calib_step_i8s (both on
 * xidx, act_alpha_dev, act_levels), 4 = effq_gram_loss (no pass over the voxels; loss_Au / loss_Bu / loss_syy below).  Workspaces as the respective entry points document them (conv_ws zero-filled
 * once; red_ws = the reduction workspace; fp_ws = effq_fp_bucket_ws_bytes(nw), may be NULL -> cooperative fixed point;
 * inv_ws / inv_ws_side = effq_spd_inverse_ws_bytes(n) each, the second only with stream_side).
 * *err_flag (device int32, zeroed by the caller) is set when a weight fixed point hits its cap (layer_helper.py:62-64). */
typedef struct effq_admm_run_args {
  const float* A0; const float* B0; const float* W0; const float* b0;
  int32_t c2, n, has_bias, w_levels;
  int32_t iters, rho_period;
  double rho, rho_max, eta, tol;
  effq_geom geom;
  int32_t loss_kind, act_levels;
  const float* xq; const uint8_t* xidx; const float* y_fp; const float* act_alpha_dev;
  float* dual; float* wstar; float* v;
  float* G_ring; int8_t* Gq_ring; float* b_ring; effq_fp_state* state_ring; double* hist;
  int32_t* err_flag;
  float* ainv_pool; int32_t n_ainv;
  void* prox_ws; size_t prox_ws_bytes;
  void* red_ws;
  void* fp_ws; size_t fp_ws_bytes;
  /* effq_fixed_point_traj for the weight projection (where effq_admm_uses_traj(weights, w_levels) says so): fp_pred =
   * effq_fp_traj_pred_bytes() of device memory (the run zero-fills it), fp_traj_ws = effq_fp_traj_ws_bytes(weights),
   * zero-filled once.  NULL: the older fixed points only. */
  void* fp_pred; void* fp_traj_ws; size_t fp_traj_ws_bytes;
  void* inv_ws; size_t inv_ws_bytes;
  void* inv_ws_side; size_t inv_ws_side_bytes;
  void* conv_ws; size_t conv_ws_bytes;
  void* stream_main; void* stream_loss; void* stream_side;
  /* optional second side stream with its own inverse workspace: the later inverses alternate between the two side
   * streams (their serial pivot phases overlap); NULL: one side stream */
  void* stream_side2; void* inv_ws_side2; size_t inv_ws_side2_bytes;
  /* loss_kind 4: the loss of an iterate from the layer's unweighted Gram system (effq_gram_loss): Au [n][n], Bu [c2][n]
   * (effq_gram_accum_i8_unw), syy = one device double, sum y^2 over this rank's voxels; conv_ws = effq_gram_loss_ws_bytes(n)
   * zero-filled once.  NULL for the other kinds. */
  const double* loss_Au; const double* loss_Bu; const double* loss_syy;
  /* loss_kind 5: the same from effq_gram_loss_i8, in the groups the loss stream picks the iterates up in: loss_Au / Bu / syy
   * as above, loss_planes / loss_nplanes from effq_gram_loss_i8_prepare, Gq_ring and act_alpha_dev as for loss_kind 1;
   * conv_ws = effq_gram_loss_i8_ws_bytes() zero-filled once. */
  const int8_t* loss_planes; int32_t loss_nplanes;
  /* lwq_verbose (EfficientQConv.py:114-127): iters x 2 device doubles, sum (w* - G)^2 and sum (G - G_prev)^2 of every
   * iteration (the primal residual is the root of the first, the dual residual rho times the root of the second); NULL: not
   * computed (one small launch per iteration) */
  double* res_ring;
} effq_admm_run_args;
/* 1 if effq_admm_run takes the trajectory weight projection (effq_fixed_point_traj) for a layer of nw weights at
 * w_levels levels - the caller then passes fp_pred (effq_fp_traj_pred_bytes(), zero-filled) and fp_traj_ws
 * (effq_fp_traj_ws_bytes(nw)); otherwise both may be NULL and nothing needs to be allocated. */
int effq_admm_uses_traj(size_t nw, int w_levels);
int effq_admm_num_inverses(double rho, double rho_max, int iters, int rho_period);
int effq_admm_run(const effq_admm_run_args* a);
/* best = the EARLIEST iterate with the smallest hist[i][0] ("if i == 0 or lossf < best", EfficientQConv.py:139-142):
 * copies its G / b* out of the rings; best_out[0] = its loss sum, best_out[1] = its index (as a double). */
int effq_admm_select_best(const double* hist, int iters, const float* G_ring, const float* b_ring, size_t nw, size_t nb,
                          float* best_G, float* best_b, double* best_out, void* stream);

/* ---- the Gram system as ONE data-parallel message ------------------------------------------------------------------
 * A0 is symmetric: a rank's partial sums travel as [upper triangle of A0, row-major: n(n+1)/2 floats | B0: c2*n floats]
 * (effq_gram_packed_elems), one all-reduce per layer instead of two and half the bytes of the full matrix
 * (solver.py:302-312 sums the per-sample contributions the same way).  unpack mirrors the triangle back. */
size_t effq_gram_packed_elems(int n, int c2);
int effq_gram_pack(const float* A0, const float* B0, int n, int c2, float* buf, void* stream);
int effq_gram_unpack(const float* buf, int n, int c2, float* A0, float* B0, void* stream);

/* ---- measurement aid: sampling profiler of effq_admm_run (off by default, per host thread) ---------------------
 * effq_prof_enable(every > 0): from now on every `every`-th iteration of effq_admm_run brackets its ops with HIP-event
 * pairs recorded on the stream the op is launched on; effq_prof_enable(0) stops and drops the records.  After the
 * caller has synchronised the device, effq_prof_read(i) returns record i: kind 1 prox solve, 2 weight-scale fixed point,
 * 3 projection + dual update, 4 loss evaluation (conv + squared error, with its weight pack), 5 inverse of A(rho)
 * (iter < 0); ms = elapsed time between the two events. */
typedef struct effq_prof_record {
  int32_t kind, iter, loss_kind, c2, n;
  effq_geom geom;
  float ms;
} effq_prof_record;
int effq_prof_enable(int every);
int effq_prof_count(void);
int effq_prof_read(int i, effq_prof_record* out);

/* ---- glue between the quantised convs (row a11): x2 trilinear up-sampling of the decoder (factory_blk.py:70-93,
 * nn.Upsample(scale_factor, mode='trilinear'), align_corners = False) on NDHWC tensors; per-axis scale 1 or 2. */
int effq_upsample_trilinear(const float* x_ndhwc, int N, int D, int H, int W, int C, int sd, int sh, int sw,
                            float* y_ndhwc, void* stream);

/* ---- f2: bit-packed storage of level ids ---------------------------------------------------
 * The reference stores one uint8 per weight (store_int_weight, PTQConv.py:125-152); these pack the level ids
 * at 1/2/4/8 bits each (little-endian bit stream: element i in bits [i*bits, (i+1)*bits)) and back. */
size_t effq_packed_bytes(size_t n, int bits);
int effq_pack_levels(const uint8_t* idx, size_t n, int bits, uint8_t* packed, void* stream);
int effq_unpack_levels(const uint8_t* packed, size_t n, int bits, uint8_t* idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EFFQ_HIP_H */
