/*
 * lcrec.h -- C-ABI of the MI355X (gfx950) implementation of LC-Rec's
 * item-indexing hot path.
 *
 * The reference (jiaozihao18/LC-Rec) is pure Python/PyTorch and has no FFI
 * layer; the seam this library sits under is the reference's nn.Module API
 * (SURVEY.md section 8b).  Each entry point below names the reference code it
 * replaces (paths relative to the reference root).  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every pointer marked "device" is a HIP device
 *     pointer to a contiguous row-major buffer owned by the caller;
 *     pointers marked "host" are ordinary host arrays (shape descriptors);
 *   - the library allocates nothing that outlives a call: scratch space is a
 *     caller-provided workspace whose size is returned by *_workspace().  The one
 *     exception is explicit: an lcrec_context (create/destroy below) owns the helper
 *     streams, events and the pinned upload ring that the two calls taking a context
 *     may use; with a NULL context those calls keep every launch on `stream`;
 *   - all work is enqueued on `stream` (a hipStream_t, NULL = default stream), or on
 *     a context's helper streams forked from it and joined back into it before the
 *     call returns (on every exit path, errors included): a call is ordered after
 *     prior and before later work on `stream`; calls do not synchronise the device
 *     (the one documented host wait: lcrec_sinkhorn_assign with many groups and a
 *     NULL context);
 *   - no global mutable state apart from the diagnostic trace log; a context is used
 *     by one host thread at a time (one process per GPU needs no locking).  The library
 *     reads a handful of environment variables ONCE per process, on the first call that
 *     consults them (function-local statics): they select between kernels that compute
 *     the same bits and exist for A/B measurements, not for configuration --
 *       LCREC_GEMM_PP, LCREC_GEMM_PP3, LCREC_GEMM_FAST, LCREC_GEMM_SMALL, LCREC_GEMM_TUNE,
 *       LCREC_GEMM_S16, LCREC_GEMM_S16_TILES
 *                                  which tiling / kernel form lcrec_linear_forward (and the
 *                                  dX product of lcrec_linear_backward) takes
 *       LCREC_GEMM_SPLITK           cap on the K-runs of the weight gradient (changes S of
 *                                  lcrec_linear_backward_splits, hence its documented sum)
 *       LCREC_RQ_SPLIT              the batch-sized form of lcrec_rq_assign
 *       LCREC_SINKHORN_SCALING, LCREC_SINKHORN_PERSISTENT, LCREC_SK_RW, LCREC_SK_BLOCKS,
 *       LCREC_SK_LOCAL              which batch-sized Sinkhorn solver runs, and whether its
 *                                  exchange stays inside one XCD (assignments may differ only
 *                                  inside the 1e-9 margin stated at the call)
 *       LCREC_STRIP_COLS, LCREC_BN_V4, LCREC_BN_CACHED_MIN
 *                                  strip width / kernel form of the BatchNorm column reductions
 *                                  (the forms sum a column's rows in different orders: last-bit
 *                                  differences in the statistics, within the stated tolerances)
 *     A deployment sets none of them; being process-global they are outside the
 *     per-call / per-context contract above;
 *   - return 0 on success, a negative LCREC_E* code otherwise; the message for
 *     the calling thread's last error is lcrec_last_error(); nothing throws;
 *   - arithmetic contract: every contraction is one fp32 fused-multiply-add
 *     chain over k ascending starting at 0 (v_mfma_f32_32x32x2_f32 semantics);
 *     results are bit-identical to oracle/lcrec_oracle.c.  One documented
 *     exception: the weight gradient of lcrec_linear_backward is an ordered sum
 *     of a few such chains over runs of the batch (see there);
 *   - `ticket` arguments (ABI 3): a device pointer to ONE 4-byte word owned by the caller, or NULL.
 *     The word must be zero before its first use and every call that takes it leaves it zero.
 *     A call that sums per-workgroup partials finishes that sum in a second, one-workgroup launch;
 *     given a ticket, the workgroups count their arrivals in it and the last one to arrive adds
 *     the partials instead -- in the same order, so the result has the same bits -- and the second
 *     launch disappears (a training step makes ~80 launches and a fifth of them are such tails,
 *     4-5 us each).  Nothing waits on the word, so the dispatch order cannot deadlock a call.
 *     Calls that share a ticket must be ordered on one stream (or one captured graph branch).
 */
#ifndef LCREC_H
#define LCREC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCREC_ABI_VERSION 3

#define LCREC_OK 0
#define LCREC_EINVAL (-1)      /* bad argument (shape, NULL pointer, unsupported size) */
#define LCREC_EUNSUPPORTED (-2) /* valid request this build has no kernel for */
#define LCREC_EWORKSPACE (-3)  /* workspace too small */
#define LCREC_EHIP (-4)        /* HIP runtime error (launch failed, no gfx950 device) */

#define LCREC_MAX_LEVELS 16
#define LCREC_MAX_LAYERS 16

/* LCREC_ABI_VERSION of the loaded library. */
int lcrec_version(void);

/* Message for the last error returned on the calling thread ("" if none). */
const char *lcrec_last_error(void);

/* Opaque per-device resources for the calls that can overlap independent launches (SURVEY.md section 8b:
 * "nothing persistent except an opaque handle with explicit create/destroy").  The reference has no
 * counterpart: its index/ stage issues every op on one stream (index/main.py:38, one device).
 * A context belongs to the HIP device that is current when it is created and holds
 *   - two helper streams with their fork/join events: ONE pool shared by the chunk pipelines of
 *     lcrec_encode_assign and the size-class launches of lcrec_sinkhorn_assign (so a process never owns
 *     more than two library streams per device, whatever mix of calls it makes);
 *   - a ring of pinned host buffers through which lcrec_sinkhorn_assign uploads its group table without
 *     waiting on the host;
 *   - settings: the number of chunk pipelines of lcrec_encode_assign (1 or 2; default 1).
 * Streams, events and pinned buffers are created on first use and released by lcrec_context_destroy,
 * which first waits for the helper streams to drain. */
typedef struct lcrec_context lcrec_context;
int lcrec_context_create(lcrec_context **out);
int lcrec_context_destroy(lcrec_context *ctx);
/* chunk pipelines of lcrec_encode_assign: 1 (default) = every launch on the caller's stream; 2 = odd chunks on a helper
 * stream: +1.5-1.9 % on C3 when the two streams' persistent GEMM launches share the CUs evenly, but the hardware dispatcher
 * sometimes starves one of them (measured: a 61 ms pass taking 101 ms, round 1: 185 ms), so it is opt-in. */
int lcrec_context_set_pipelines(lcrec_context *ctx, int pipelines);

/* One MLP layer: y = [relu]( [bn]( x @ W^T + b ) ).
 * Replaces one Dropout(p=0)/Linear/[BatchNorm1d eval]/[ReLU] group of
 * MLPLayers.forward, index/models/layers.py:18-30,42 (nn.Linear at :23,
 * BatchNorm1d at :26, activation at :28-30).
 *   x [n][in_dim], W [out_dim][in_dim] (nn.Linear.weight layout), b [out_dim] or NULL,
 *   bn_scale/bn_shift [out_dim] or both NULL: eval-mode BatchNorm folded by the
 *   caller to y = t*scale + shift (scale = gamma/sqrt(var+eps), shift = beta - mean*scale),
 *   y [n][out_dim].  All device pointers. */
int lcrec_linear_forward(const float *x, int64_t n, int in_dim, const float *W, const float *b,
                         const float *bn_scale, const float *bn_shift, int relu, int out_dim,
                         float *y, void *stream);

/* One Linear of a TRAINING step with the BatchNorms on either side of it folded in (index/models/layers.py:23-30 in train();
 * the reference runs Linear, BatchNorm1d and ReLU as three modules with a full activation tensor between each pair):
 *   u = x * in_scale + in_shift, clamped at 0 from below when in_relu != 0   (in_scale / in_shift [in_dim], both NULL: u = x)
 *         -- the PREVIOUS layer's BatchNorm + ReLU, applied while the operand tile is staged: that layer's activation
 *            is never written, its pre-BatchNorm output is all that exists;
 *   t_out [n][out_dim] = u @ W^T + b   (the same fp32 fma chains as lcrec_linear_forward, bit for bit, on u);
 *   want_stats != 0: the batch statistics of t_out over its n rows -- THIS layer's BatchNorm (layers.py:26) -- taken in the
 *         product's epilogue: mean_out, rstd_out = 1/sqrt(biased var + eps), scale_out = gamma * rstd, shift_out = beta -
 *         mean * scale_out (what the next lcrec_linear_bn_forward takes as in_scale / in_shift), and running_mean /
 *         running_var (may be NULL) updated with `momentum` as nn.BatchNorm1d does (unbiased variance).  gamma / beta may
 *         be NULL (1 / 0).  Every tile publishes per-column partial sums; the last tile of a column strip to arrive merges
 *         them in tile order (deterministic; see "ticket arguments"): tickets = ceil(out_dim / 64) zeroed words, left zero.
 *   workspace: lcrec_linear_bn_forward_workspace(n, out_dim) bytes (want_stats only).
 * Batch-sized launches only (n <= a few thousand rows: the tile rule of a training step), in_dim % 32 == 0, out_dim % 4 == 0,
 * out_dim <= 4096; anything else returns LCREC_EUNSUPPORTED and the caller uses lcrec_linear_forward + lcrec_bn_relu_forward.
 * Statistics agree with torch's to ~1e-7 relative (another summation order); pinned by fixtures F4 and F11 through the engine. */
size_t lcrec_linear_bn_forward_workspace(int64_t n, int out_dim);
int lcrec_linear_bn_forward(const float *x, int64_t n, int in_dim, const float *in_scale, const float *in_shift, int in_relu,
                            const float *W, const float *b, int out_dim, float *t_out, int want_stats, const float *gamma,
                            const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                            float *mean_out, float *rstd_out, float *scale_out, float *shift_out, void *workspace,
                            size_t workspace_bytes, unsigned int *tickets, void *stream);

/* The two backward products of one Linear layer, as autograd derives them for nn.Linear in
 * MLPLayers (index/models/layers.py:23; driven by loss.backward() at index/trainer.py:117,
 * SURVEY.md row a9):
 *   gx [n][in_dim]       = gy [n][out_dim] * W [out_dim][in_dim]
 *   gw [out_dim][in_dim] = gy^T * x [n][in_dim]
 * gy is the gradient w.r.t. the layer's pre-activation output (the caller has already applied the
 * ReLU mask); the bias gradient is the caller's column sum of gy.  Every operand is read in the
 * layout it is stored in -- no transposed copies -- and every output element is one fp32 fma chain
 * over the contracted index ascending (out_dim for gx), like the forward kernel.  For gw the
 * contracted index is the batch: it is cut into S = lcrec_linear_backward_splits(n, in_dim, out_dim)
 * runs of 32*ceil(ceil(n/32)/S) consecutive items (S = 1 for the wide layers, up to 16 for the narrow
 * ones, so that a small [out_dim][in_dim] output still fills the chip); each run is one fma chain
 * from 0 and the runs are added in order, ((p0 + p1) + p2) + ...  S depends only on the three sizes.
 * gx_out / gw_out may be NULL to skip a product.  out_dim % 32 == 0 is required for gx, in_dim and
 * out_dim % 4 == 0 for both; sized for training batches (n * max(in_dim, out_dim) < 2^29).
 * workspace: device scratch of lcrec_linear_backward_workspace() bytes (S partial products; 0 if S = 1). */
int lcrec_linear_backward_splits(int64_t n, int in_dim, int out_dim);
size_t lcrec_linear_backward_workspace(int64_t n, int in_dim, int out_dim);
int lcrec_linear_backward(const float *gy, const float *x, const float *W, int64_t n, int in_dim, int out_dim,
                          float *gx_out, float *gw_out, void *workspace, size_t workspace_bytes, void *stream);

/* The weight gradients of SEVERAL Linear layers in one launch: gw_p = gy_p^T * x_p for p < count (count <= 16), each
 * exactly what lcrec_linear_backward computes for that layer -- the same S = lcrec_linear_backward_splits runs, added in
 * order -- so the results are bit-identical to per-layer calls.  A training step's narrow layers are a handful of tiles
 * each; launched together their workgroups fill the chip (index/trainer.py:117: autograd produces these one at a time).
 * A problem's `splits` field (> 0) sets its S itself -- the same definition of the runs and of their sum, another S.  A
 * launch of a whole step's problems has thousands of tiles, and cutting the narrow layers' batches there only adds partial
 * products to write, read and add (the training engine passes 1: one fma chain over the batch per element).
 * `problems` is a HOST array; all pointers inside are device pointers.  Widths must be multiples of 4. */
typedef struct {
    const float *gy;   /* [n][out_dim] gradient w.r.t. the layer's pre-activation output */
    const float *x;    /* [n][in_dim]  the layer's input */
    float *gw;         /* [out_dim][in_dim] */
    int64_t n;
    int in_dim, out_dim;
    /* ABI 3: both NULL, or [in_dim] device vectors: the layer's input is then u = x * x_scale + x_shift, clamped at 0 from
     * below when x_relu != 0 -- x being the PRE-BatchNorm output of the previous layer as lcrec_linear_bn_forward leaves it
     * and (x_scale, x_shift) that BatchNorm's folded affine; u is formed on the operand's way into LDS. */
    const float *x_scale, *x_shift;
    int x_relu;
    int splits;        /* ABI 3: 0 = S of lcrec_linear_backward_splits(n, in_dim, out_dim); > 0 = this S (at most ceil(n/32)) */
} lcrec_dw_problem;
size_t lcrec_linear_backward_weights_workspace(const lcrec_dw_problem *problems, int count);
int lcrec_linear_backward_weights(const lcrec_dw_problem *problems, int count, void *workspace, size_t workspace_bytes,
                                  void *stream);

/* L-level residual quantisation with hard (argmin) assignment.
 * Replaces ResidualVectorQuantizer.forward, index/models/rq.py:39-55, over
 * VectorQuantizer.forward with use_sk=False, index/models/vq.py:63-99
 * (distance :71-73, argmin :75, gather :87, losses :90-92, STE :95).
 *   z          device [n][e]         latents
 *   codebooks  device, level l is [K[l]][e] at float offset sum_{m<l} K[m]*e
 *                                    (= torch.cat of rq.get_codebook() rows, rq.py:32-37)
 *   K          host   [L]            codes per level (any positive count; a level must fit
 *                                    in LDS: roundup32(K)*(e+5)*4 bytes <= ~156 KB)
 *   idx_out    device int64, item i's level-l code at idx_out[i*idx_stride + l] (rq.py:54);
 *   idx_stride elements between rows, >= L; 0 = L (a dense [n][L] matrix).  A run of levels can so be
 *                                    written straight into the columns of a wider index matrix
 *   xq_out     device [n][e] or NULL sum over levels of the straight-through x_res (rq.py:48);
 *                                    if xq_accumulate != 0 its current contents are the initial
 *                                    value of the sum (chaining a run of levels after another)
 *   sse_out    device [L] double or NULL  per-level sum of (c - r)^2; the level loss of
 *                                    vq.py:90-92 is (1+beta)*sse/(n*e)
 *   resid_out  device [L+1][n][e] or NULL  entry l = residual fed into level l (input of
 *                                    lcrec_code_stats), entry L = residual after the last level
 *   margin_out device [n][L] float or NULL  near-tie audit: per item and level, the second smallest
 *                                    distance minus the smallest, both as computed above (0 on an exact
 *                                    tie, +inf when K[l] == 1)
 *   neartie_out device [n] uint32 or NULL   bit l set when margin_l <= tie_tau * (xx_l + cc_l[idx_l]),
 *                                    the magnitude at which the winning distance of level l was rounded
 *   tie_tau    threshold for neartie_out (>= 0; ignored when neartie_out is NULL)
 *   workspace  device scratch of lcrec_rq_assign_workspace() bytes
 *   ticket     NULL, or see "ticket arguments" above (finishes sse_out inside the launch)
 * e must be 16, 32 or 64.
 * Why the audit outputs exist: the reference decides vq.py:75 on fp32 distances whose summation order (MKL,
 * vectorised reductions) is unspecified and batch-size dependent (hard part 1), so an item
 * whose two best codes are closer than that noise can get a different code from the reference's CPU run than from
 * this library's canonical order -- and, its residual then differing, different codes on the levels after it.
 * Items whose neartie_out is 0 at the tau recorded in tests/golden/f9_neartie_*.npz carry the reference's tuple;
 * DESIGN.md section 2 states the measured rates. */
size_t lcrec_rq_assign_workspace(int64_t n, int e, const int *K, int L);
int lcrec_rq_assign(const float *z, int64_t n, int e, const float *codebooks, const int *K, int L,
                    int64_t *idx_out, int64_t idx_stride, float *xq_out, int xq_accumulate, double *sse_out,
                    float *resid_out, float *margin_out, uint32_t *neartie_out, float tie_tau,
                    void *workspace, size_t workspace_bytes, unsigned int *ticket, void *stream);

/* Encoder MLP + residual quantisation: item embeddings -> index tuples.
 * Replaces RQVAE.get_indices(xs, use_sk=False), index/models/rqvae.py:68-72
 * (encoder = MLPLayers, layers.py:42; rq = rq.py:39-55).  This is the path
 * BASELINE.json's metric measures.
 *   x          device [n][dims[0]]
 *   dims       host   [n_layers+1]   in_dim, hidden sizes..., e_dim (rqvae.py:46)
 *   W, b, bn_scale, bn_shift  host arrays of n_layers device pointers (bn_* arrays
 *              may be NULL, or hold NULL entries for layers without BatchNorm);
 *              ReLU follows every layer but the last (layers.py:27-30)
 *   latent_out device [n][e] or NULL  encoder output
 *   others     as lcrec_rq_assign (margin_out / neartie_out / tie_tau: the near-tie audit of the quantiser)
 *   ctx        NULL, or a context of the current device
 * Items are processed in chunks of lcrec_encode_assign_chunk_rows() = 524 288.  With a context whose pipelines setting is 2 (opt-in), odd chunks run on
 * the context's first helper stream, forked from `stream` by an event at entry and joined back into it before
 * the quantiser pass (and on every error exit), so everything is ordered after prior work on `stream` and
 * before later work on it, without any host synchronisation; with ctx == NULL or pipelines == 1 every launch
 * is on `stream`. */
size_t lcrec_encode_assign_workspace(int64_t n, const int *dims, int n_layers, const int *K, int L);
/* rows per chunk of the walk described above (a build constant; results do not depend on it) */
int64_t lcrec_encode_assign_chunk_rows(void);
int lcrec_encode_assign(const float *x, int64_t n, const int *dims, int n_layers,
                        const float *const *W, const float *const *b,
                        const float *const *bn_scale, const float *const *bn_shift,
                        const float *codebooks, const int *K, int L, int64_t *idx_out,
                        float *latent_out, float *xq_out, double *sse_out,
                        float *margin_out, uint32_t *neartie_out, float tie_tau,
                        void *workspace, size_t workspace_bytes, lcrec_context *ctx, void *stream);

/* Sinkhorn ("uniform semantic") assignment of one level, for one or many independent
 * groups of rows.  Replaces the use_sk branch of VectorQuantizer.forward,
 * index/models/vq.py:76-83: distances (:71-73, same fp32 fma chains as lcrec_rq_assign),
 * centring on the group's global max/min (:51-61), fp64 sinkhorn_algorithm
 * (index/models/layers.py:85-108: Q=exp(-d/eps), /total, iters x {/rowsum, /B, /colsum, /K},
 * *B, every division a true fp64 division in that order) and argmax (:83, first maximum).
 * Training calls it with one group (the batch, index/trainer.py:114); index generation
 * with one group per set of colliding items (index/generate_indices.py:113-119), which the
 * reference runs one forward at a time and this runs as one batched launch.
 *   resid          device [n][e]   residual entering the level
 *   codebook       device [K][e]
 *   group_offsets  HOST  [n_groups+1] ascending row offsets; group g = rows [off[g], off[g+1])
 *   idx_out        device int64, element i written at idx_out[i*idx_stride]
 *   ctx            NULL, or a context of the current device
 * e must be 16, 32 or 64; K <= 1024 for groups too large for LDS (rows*K > 16384).
 * With several groups the call buckets them by size, one launch per bucket.  With a context the buckets go to
 * `stream` and to the context's two helper streams, forked from `stream` by an event and joined back into it
 * before the call returns (error exits included), and the group table reaches the device through the context's
 * pinned ring: no host wait.  With ctx == NULL every launch is on `stream` and the call waits once on the host
 * for the upload of the group table (many-group calls only; a single training batch never synchronises).
 * A lone group of more than 16384/K rows is solved by one multi-workgroup launch whose workgroups must all be
 * resident: it is chosen only when the occupancy the runtime reports for its LDS size times the CU count covers
 * the grid (<= 128 workgroups), otherwise the multi-launch solver runs; every spin in it is bounded (~0.1 s in
 * total) and a timeout -- other work holding the CUs -- fills idx_out with -1 instead of hanging.
 * ticket (NULL, or see "ticket arguments"): the batch-sized solver's control words and exchange slots are then set up by the
 * distance launch itself (its last workgroup reduces the global max/min of vq.py:52-53 from the workgroups' partials) instead
 * of by a launch of their own in front of it. */
size_t lcrec_sinkhorn_assign_workspace(int64_t n, int K, const int64_t *group_offsets, int n_groups);
int lcrec_sinkhorn_assign(const float *resid, int64_t n, int e, const float *codebook, int K,
                          const int64_t *group_offsets, int n_groups, double epsilon, int iters,
                          int64_t *idx_out, int64_t idx_stride, void *workspace, size_t workspace_bytes,
                          lcrec_context *ctx, unsigned int *ticket, void *stream);

/* Apply a given assignment to one level: gather, squared error, straight-through estimator and
 * residual update (index/models/vq.py:87-95, index/models/rq.py:47-48) -- what follows the
 * argmin/argmax inside VectorQuantizer.forward, for levels whose indices came from
 * lcrec_sinkhorn_assign.
 *   xq        device [n][e] or NULL: x_q sum, += x_res (starts from 0 unless xq_accumulate)
 *   resid_out device [n][e] or NULL: residual after the level (may alias resid_in)
 *   sse_out   device double[1] or NULL; workspace must then hold 8 KB
 *   ticket    NULL, or see "ticket arguments" (finishes sse_out inside the launch) */
int lcrec_rq_apply_level(const float *resid_in, int64_t n, int e, const float *codebook, int K,
                         const int64_t *idx, int64_t idx_stride, float *xq, int xq_accumulate,
                         float *resid_out, double *sse_out, void *workspace, size_t workspace_bytes,
                         unsigned int *ticket, void *stream);

/* Per-code count and sum of the residuals assigned to it:
 *   count[k] = #{i : idx[i*idx_stride] == k},  sum[k][:] = sum_i resid[i][:]  (item order, fp32).
 * Replaces the scatter_add_/index_add_ block of index_improve/models/vq.py:151-167 and is the
 * segmented reduce behind the codebook gradient autograd derives from vq.py:90-92
 * (dL/dC[k] = w * (count[k]*C[k] - sum[k])).  Bit-identical to the CPU order. */
int lcrec_code_stats(const int64_t *idx, int64_t idx_stride, const float *resid, int64_t n, int e, int K,
                     float *count, float *sum, void *stream);

/* lcrec_code_stats for ALL levels of a quantiser in one launch, optionally fused with lcrec_codebook_grad:
 *   idx      device [n][L] int64 (ResidualVectorQuantizer's index matrix, rq.py:54)
 *   resid    HOST array of L device pointers, resid[l] [n][e] = the residual that entered level l
 *   K        HOST [L];  count / sum: HOST arrays of L device pointers ([K[l]], [K[l]][e])
 *   codebooks / grad_out: both NULL, or HOST arrays of L device pointers: grad_out[l] =
 *            (scale * (count*C_l - sum)) * weight, see lcrec_codebook_grad
 * Same bits as the per-level calls. */
int lcrec_code_stats_levels(const int64_t *idx, const float *const *resid, int64_t n, int e, const int *K, int L,
                            float *const *count, float *const *sum, const float *const *codebooks, float *const *grad_out,
                            float scale, float weight, void *stream);

/* EMA codebook update, index_improve/models/vq.py:155-184, in place:
 *   ema_count = ema_count*decay + alpha*count;  ema_sum = ema_sum*decay + alpha*sum;
 *   where ema_count > eps:  codebook = codebook*keep + (ema_sum/(ema_count+eps))*alpha.
 * decay, alpha = 1-decay and keep = 1-(1-decay) are the reference's python doubles rounded to fp32
 * by the caller.  skip_flag: NULL, or a device byte; when it is non-zero the call changes nothing (the sticky
 * "loss was NaN" flag of lcrec_step_losses: the reference raises before the offending step updates anything,
 * index/trainer.py:116, so a caller that learns of the NaN later still finds the last good state). */
int lcrec_ema_update(float *ema_count, float *ema_sum, float *codebook, const float *count,
                     const float *sum, int K, int e, float decay, float alpha, float keep, float eps,
                     const unsigned char *skip_flag, void *stream);

/* ---- the element-wise / column-reduction half of a training step -------------------
 * Device pointers throughout; batch-sized inputs (n <= 2^20 rows); deterministic (no atomics): a column is summed by
 * 8 row groups (rows r = g, g+8, ... in order) whose partials are added in group order.  Floating-point results are
 * within 1e-5 of the torch ops they replace (the reference's own order is unspecified); pinned by the reference's
 * F4 three-step trajectory. */

/* Training-mode BatchNorm1d (+ ReLU) on the output t [n][features] of a Linear: index/models/layers.py:25-30
 * (nn.BatchNorm1d at :26 in train(), activation at :28-30).  Batch mean and biased variance (two passes), y =
 * [relu]((t - mean) * rstd * gamma + beta), rstd = 1/sqrt(var + eps); running_mean / running_var (may be NULL) are
 * updated in place with `momentum`, running_var from the unbiased variance; mean_out / rstd_out [features] are what
 * the backward needs.  n >= 2. */
int lcrec_bn_relu_forward(const float *t, int64_t n, int features, const float *gamma, const float *beta, float eps,
                          float momentum, float *running_mean, float *running_var, float *y, float *mean_out,
                          float *rstd_out, int relu, void *stream);

/* Backward of the above for gy = dL/dy (what autograd derives for layers.py:25-30 under loss.backward(),
 * index/trainer.py:117):  g = gy * [y > 0];  dbeta = sum g;  dgamma = sum g*xhat;
 * dt = gamma*rstd*(g - dbeta/n - xhat*dgamma/n);  dbias = sum dt -- the gradient of the Linear bias feeding the
 * BatchNorm (zero up to rounding, as in the reference).  dgamma/dbeta/dbias may be NULL; dt may alias gy.
 * y may be NULL when relu != 0 and fold_scale / fold_shift [features] are given: the mask is then [t * fold_scale +
 * fold_shift > 0] (one fma), the expression the consumer of lcrec_linear_bn_forward's output evaluated.  With y NULL,
 * fold_scale NULL and fold_shift = the layer's beta [features], the mask is [(t - mean) * rstd * gamma + beta > 0]
 * evaluated exactly as lcrec_bn_relu_forward evaluates y: the same bits as passing that y, one array less to read. */
int lcrec_bn_relu_backward(const float *gy, const float *t, const float *y, int64_t n, int features, const float *gamma,
                           const float *mean, const float *rstd, int relu, float *dt_out, float *dgamma_out,
                           float *dbeta_out, float *dbias_out, const float *fold_scale, const float *fold_shift, void *stream);

/* The same two operations split where an item-sharded (data-parallel) run exchanges statistics, so that the batch is the
 * union of all ranks' rows (torch.nn.SyncBatchNorm semantics; the reference's index/ stage is single-process, SURVEY.md
 * section 8e item 4):
 *   lcrec_bn_stats            local mean and M2 = sum (t - mean)^2 of this rank's n rows      -> exchanged by the caller
 *   lcrec_bn_merge_stats      rows[world][2*features+1] = every rank's (n_r, mean_r[], M2_r[]) -> mean, rstd of the union of
 *                             the rows (Chan et al., merged in rank order: the same bits on every rank) and the running
 *                             statistics as nn.BatchNorm1d keeps them (running_* may be NULL)
 *   lcrec_bn_relu_apply       y = [relu]((t - mean) * rstd * gamma + beta) with the merged statistics
 *   lcrec_bn_backward_reduce  local sum g and sum g*xhat (g = gy * [y > 0])                    -> all-reduced by the caller;
 *                             dbeta_out / dgamma_out (may be NULL) receive a copy: this rank's share of the parameter gradients
 *   lcrec_bn_backward_apply   dt = gamma*rstd*(g - sum_g/n_total - xhat*sum_gx/n_total), dbias = local column sums of dt */
int lcrec_bn_stats(const float *t, int64_t n, int features, float *mean_out, float *m2_out, void *stream);
int lcrec_bn_merge_stats(const float *rows, int world, int features, float eps, float momentum, float *mean_out, float *rstd_out,
                         float *running_mean, float *running_var, void *stream);
int lcrec_bn_relu_apply(const float *t, int64_t n, int features, const float *gamma, const float *beta, const float *mean,
                        const float *rstd, int relu, float *y, void *stream);
int lcrec_bn_backward_reduce(const float *gy, const float *t, const float *y, int64_t n, int features, const float *mean,
                             const float *rstd, int relu, float *sum_g_out, float *sum_gx_out, float *dbeta_out,
                             float *dgamma_out, void *stream);
int lcrec_bn_backward_apply(const float *gy, const float *t, const float *y, int64_t n, int features, const float *gamma,
                            const float *mean, const float *rstd, int relu, const float *sum_g, const float *sum_gx,
                            float n_total, float *dt_out, float *dbias_out, void *stream);

/* ReLU mask and bias gradient of a Linear without BatchNorm (layers.py:23,28-30): g = gy * [y > 0] (relu != 0;
 * g_out may alias gy or be NULL), dbias = column sums of g. */
int lcrec_relu_bias_backward(const float *gy, const float *y, int64_t n, int features, int relu, float *g_out,
                             float *dbias_out, void *stream);

/* Scratch (bytes) of the two whole-tensor reductions below. */
size_t lcrec_train_reduce_workspace(void);

/* Reconstruction loss and its gradient, index/models/rqvae.py:74-85: l1 == 0: loss = mean (out-x)^2,
 * grad = 2 (out-x)/count; l1 != 0: loss = mean |out-x|, grad = sign(out-x)/count.  count = n * in_dim elements;
 * loss_out is a device float (fp64 accumulation); grad_out [count] or NULL.
 * count_total (0 = count): the element count of the GLOBAL batch when these `count` elements are one rank's shard of
 * it (item-sharded data parallel): both divisions use count_total, so loss_out is this rank's share of the global
 * mean and the ranks' gradients sum to the global-batch gradient. */
int lcrec_recon_loss_grad(const float *out, const float *x, int64_t count, int64_t count_total, int l1, float *grad_out,
                          float *loss_out, void *workspace, size_t workspace_bytes, unsigned int *ticket, void *stream);

/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) (index/trainer.py:118) on a flat gradient buffer:
 * norm_out[0] = ||grads||_2 (fp64 accumulation), norm_out[1] = min(1, max_norm / (norm + 1e-6)), the coefficient
 * lcrec_adamw_step applies. */
int lcrec_grad_norm_clip(const float *grads, int64_t count, float max_norm, float *norm_out, void *workspace,
                         size_t workspace_bytes, unsigned int *ticket, void *stream);

/* Codebook gradient from the per-code statistics of lcrec_code_stats -- what autograd derives from the two MSE terms
 * of index/models/vq.py:90-92: grad[k][:] = (scale * (count[k]*C[k][:] - sum[k][:])) * weight, with
 * scale = 2/(L*n*e) and weight = d loss / d rq_loss (quant_loss_weight, index/models/rqvae.py:83).  [K][e] each. */
int lcrec_codebook_grad(const float *count, const float *sum, const float *codebook, int K, int e, float scale,
                        float weight, float *grad_out, void *stream);

/* The scalar tail of a training step in one launch: level losses mse_l + beta*mse_l with mse_l = sse[l]/(n*e)
 * (index/models/vq.py:90-92), their mean (rq.py:53), loss = recon + quant_loss_weight * rq_loss (rqvae.py:83);
 * losses_out[3] = {loss, recon, rq_loss}; sums_inout[2] += {loss, recon} (trainer.py:122-123; may be NULL);
 * *nan_flag = 1 if the loss is NaN (trainer.py:116's check as a sticky device flag; may be NULL).  sse as written by
 * lcrec_rq_assign / lcrec_rq_apply_level (device double[L]); recon a device float (lcrec_recon_loss_grad).
 * poison_probe / poison_flag (both NULL, or both given): *poison_flag = 1 when *poison_probe < 0 -- the first assignment of a
 * batch-sized lcrec_sinkhorn_assign, which fills its output with -1 when its solver gives up; also sticky. */
int lcrec_step_losses(const double *sse, int L, int64_t n, int e, float beta, float quant_loss_weight, const float *recon,
                      float *losses_out, double *sums_inout, unsigned char *nan_flag, const int64_t *poison_probe,
                      unsigned char *poison_flag, void *stream);

/* Gradient reaching the encoder output z through the quantiser (autograd of vq.py:87-95 / rq.py:45-48):
 * out = (coef * (z - C0[idx0])) * weight + g_xq, coef = beta * 2/(L*n*e), weight = d loss / d rq_loss; idx0 = the
 * level-0 index column (element i at idx[i*idx_stride]); all [n][e]. */
int lcrec_quantizer_input_grad(const float *z, const float *codebook0, const int64_t *idx, int64_t idx_stride, int64_t n,
                               int e, float coef, float weight, const float *g_xq, float *out, void *stream);
/* The same, and dbias_out[e] = the column sums of `out`: the gradient of the bias of the encoder's last Linear, which has no
 * BatchNorm and no activation behind it (index/models/layers.py:19-30 skips both on the last layer), so what reaches z reaches
 * its pre-activation unchanged -- one launch for what lcrec_quantizer_input_grad + lcrec_relu_bias_backward(relu = 0) do in two.
 * dbias_out NULL: lcrec_quantizer_input_grad. */
int lcrec_quantizer_input_grad_bias(const float *z, const float *codebook0, const int64_t *idx, int64_t idx_stride, int64_t n,
                                    int e, float coef, float weight, const float *g_xq, float *out, float *dbias_out, void *stream);

/* One optimiser step on flat fp32 buffers: torch.optim.AdamW (decoupled != 0) or Adam (index/trainer.py:49-81,119),
 * preceded by the clipping of index/trainer.py:118 (grads *= clip[1], stored back; clip may be NULL) and with the
 * learning rate of index/trainer.py:83-92,120 evaluated on the device from the step counter:
 *   schedule -1: lr = base_lr;  0: constant after a linear warm-up;  1: linear warm-up, then linear decay to 0 at
 *   total_steps (transformers' get_{constant,linear}_schedule_with_warmup multipliers, in double).
 * *step (device int64) = optimiser steps taken so far; the call uses lr(*step) and bias corrections for step *step+1,
 * then increments it -- so a captured hipGraph of a training step replays without any host-side scalar.
 * lr_out: device float receiving the learning rate used, or NULL.
 * ticket: NULL, or see "ticket arguments": the increment is then made by the last workgroup of the update launch.
 * skip_flag: NULL, or a device byte; non-zero = update nothing and leave *step (see lcrec_ema_update). */
int lcrec_adamw_step(float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t count, const float *clip,
                     int64_t *step, double base_lr, double beta1, double beta2, double eps, double weight_decay,
                     int decoupled, int schedule, int64_t warmup_steps, int64_t total_steps, float *lr_out,
                     unsigned int *ticket, const unsigned char *skip_flag, void *stream);

/* Which items share an identical index tuple.  Replaces the Python string-set / dict passes of
 * index/trainer.py:139-150 (collision rate) and index/generate_indices.py:18-42
 * (check_collision, get_indices_count, get_collision_item), keeping get_collision_item's order:
 * groups (tuples held by >= 2 items) by first occurrence of the tuple in item order, item ids
 * ascending inside a group.
 *   idx               device [n][L] int64, K host [L] (codes per level; sum of ceil(log2 K) <= 128)
 *   members_out       device [n] int64 or NULL: ids of the items of all groups, group after group
 *   group_offsets_out device [n/2+2] int64 or NULL (given together with members_out):
 *                     group g = members_out[off[g] .. off[g+1])
 *   counters_out      device int64[4]: {distinct tuples, groups, items in groups, largest tuple count}
 *                     (entries 1 and 2 only when members_out is given);
 *                     collision_rate = (n - counters[0]) / n */
size_t lcrec_collision_groups_workspace(int64_t n, int L);
int lcrec_collision_groups(const int64_t *idx, int64_t n, int L, const int *K, int64_t *members_out,
                           int64_t *group_offsets_out, int64_t *counters_out, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Text of the `.index.json` entries for a run of items (host-side; no device work).  Replaces the
 * per-item Python of index/generate_indices.py:83-92 (token strings "<a_{i}>", "<b_{j}>", ...) and the
 * json.dump of :138-145, whose default separators (", " and ": ") every consumer relies on
 * (data.py:38-89 json.load's the file).  For items first_item .. first_item+n-1 the output is
 *     "<id>": ["<a_i>", "<b_j>", ...], "<id+1>": [...]
 * -- entries joined by ", ", no enclosing braces, no trailing separator, no NUL -- so that
 * "{" + chunk_0 + ", " + chunk_1 + ... + "}" is byte-for-byte json.dump's output for the whole dict.
 *   idx   HOST [n][L] int64 (row-major), 1 <= L <= 26 (prefix letters a..z)
 *   out   HOST buffer of capacity cap bytes; lcrec_index_json_bound(n, L) is always enough
 * Returns the number of bytes written, or a negative LCREC_E* code (LCREC_EWORKSPACE: cap too small). */
int64_t lcrec_index_json_bound(int64_t n, int L);
int64_t lcrec_index_json_format(const int64_t *idx, int64_t n, int L, int64_t first_item, char *out, int64_t cap);

/* Kernel tracing (diagnostic; the reference has no tracing on this path -- its only
 * timing is wall-clock per epoch, index/trainer.py:193-195).  While enabled, every
 * kernel this library launches is bracketed by a pair of hipEvents recorded on the
 * launch stream; lcrec_trace_collect() waits for the recorded events, sums elapsed
 * time per kernel and clears the log.  bench.py uses it to price the dominant
 * kernel against its roofline inside the timed region. */
typedef struct {
    const char *kernel;   /* static string, e.g. "linear_fwd_128x128" */
    int64_t launches;
    double total_ms;
} lcrec_trace_entry;
int lcrec_trace_enable(int on);
/* Returns the number of entries written (<= capacity), or a negative LCREC_E* code. */
int lcrec_trace_collect(lcrec_trace_entry *out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* LCREC_H */
