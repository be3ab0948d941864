// Internal helpers shared by the HIP translation units of liblcrec_hip.so.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, fp32 MFMA, 160 KB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lcrec.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LCREC_API extern "C" __attribute__((visibility("default")))

namespace lcrec {

// thread-local last-error text (lcrec_last_error)
char *err_buf();
int fail(int code, const char *fmt, ...);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LCREC_EHIP, "%s: %s", what, hipGetErrorString(e));
    return LCREC_OK;
}

// Kernel ids for lcrec_trace_*.
enum KernelId { K_LINEAR_128x128 = 0, K_LINEAR_128x64, K_LINEAR_128x32, K_RQ_ASSIGN, K_RQ_SSE_FINALIZE,
                K_VQ_DISTANCE, K_SINKHORN, K_SINKHORN_SMALL, K_APPLY_LEVEL, K_CODE_STATS, K_EMA_UPDATE, K_COLLISION,
                K_LINEAR_PP, K_LINEAR_64x64, K_SINKHORN_SLAB, K_SINKHORN_TINY, K_BN_FWD, K_BN_BWD, K_RELU_BIAS_BWD, K_LOSS,
                K_GRAD_NORM, K_ADAMW, K_LINEAR_32x64, K_COUNT };
extern const char *const kKernelNames[K_COUNT];
bool trace_on();
void trace_begin(int kernel, hipStream_t stream);
void trace_end(hipStream_t stream);

// Brackets the launches made in its scope with hipEvents when tracing is enabled.
struct TraceScope {
    hipStream_t s;
    bool on;
    TraceScope(int kernel, hipStream_t stream) : s(stream), on(trace_on()) { if (on) trace_begin(kernel, s); }
    ~TraceScope() { if (on) trace_end(s); }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace lcrec

// lcrec_context (include/lcrec.h): the only library-owned resources that outlive a call.
struct lcrec_context {
    static constexpr int HELPERS = 2, RING = 4;
    int device = 0;
    int pipelines = 1;
    bool streams_ready = false;
    hipStream_t helper[HELPERS] = {};
    hipEvent_t fork = nullptr, join[HELPERS] = {};
    // pinned upload ring: slot i may be rewritten once pin_done[i] (recorded after the copy that read it) has completed
    void *pin[RING] = {};
    size_t pin_bytes[RING] = {};
    hipEvent_t pin_done[RING] = {};
    bool pin_busy[RING] = {};
    int pin_next = 0;

    int ensure_streams();                                   // LCREC_OK or a failed code (last error set)
    void *ring_acquire(size_t bytes, int *slot);            // pinned host buffer of >= bytes, or NULL (last error set)
    void ring_release(int slot, hipStream_t after);         // call after enqueueing the copy that reads the slot
};

namespace lcrec {

// Fork a context's helper streams from `s` (those in `mask`) and join them back on scope exit -- every exit, so an
// error return never leaves helper work un-ordered against the caller's stream.
struct ForkJoin {
    lcrec_context *c;
    hipStream_t s;
    unsigned mask;
    ForkJoin(lcrec_context *ctx, hipStream_t stream, unsigned helpers) : c(ctx), s(stream), mask(ctx ? helpers : 0u)
    {
        if (!mask) return;
        (void)hipEventRecord(c->fork, s);
        for (int i = 0; i < lcrec_context::HELPERS; ++i)
            if (mask & (1u << i)) (void)hipStreamWaitEvent(c->helper[i], c->fork, 0);
    }
    ~ForkJoin()
    {
        for (int i = 0; i < lcrec_context::HELPERS; ++i)
            if (mask & (1u << i)) {
                (void)hipEventRecord(c->join[i], c->helper[i]);
                (void)hipStreamWaitEvent(s, c->join[i], 0);
            }
    }
    hipStream_t on(int i) const { return (mask & (1u << i)) ? c->helper[i] : s; }
    ForkJoin(const ForkJoin &) = delete;
    ForkJoin &operator=(const ForkJoin &) = delete;
};

// ---- "the last workgroup to arrive finishes the reduction" (include/lcrec.h, `ticket` arguments) -----------------------------
// Every workgroup hands its partial over with agent-scope (write-through) atomic stores made by ONE thread, that thread waits
// for them (vmcnt) and takes a ticket with a relaxed agent-scope add; the workgroup whose add returns count-1 is the last: it
// resets the ticket, and reads all partials back with agent-scope atomic loads behind an acquire.  No agent-scope RELEASE
// anywhere -- on gfx950 that writes the XCD's dirty L2 lines back (MI355X_MICROARCH.md, Valid forms; measured here at 38 us
// with 8 MB dirty) -- and nothing spins: the launch cannot deadlock whatever the dispatch order.
#ifdef __HIPCC__
__device__ __forceinline__ void handoff_put(double *slot, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double handoff_get(const double *slot)
{
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(slot), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void handoff_put(float *slot, float v)
{
    __hip_atomic_store(reinterpret_cast<unsigned *>(slot), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float handoff_get(const float *slot)
{
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned *>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// Call from the ONE thread that made this workgroup's handoff_put stores (if other threads stored too: after their own
// `s_waitcnt vmcnt(0)` and a workgroup barrier).  True in exactly one workgroup of the launch: the one that arrived last.
__device__ __forceinline__ bool ticket_is_last(unsigned *ticket, unsigned workgroups)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t + 1u != workgroups) return false;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // left zero for the next call
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return true;
}
// The same in two halves, for a caller with useful work to do while the add is in flight (the GEMM epilogue: the add's
// round trip through the fabric -- a microsecond or two -- passes under the tile's stores): ticket_take() after the
// hand-over stores have been waited for, ticket_finish() on its result when the answer is needed.
__device__ __forceinline__ unsigned ticket_take(unsigned *ticket)
{
    return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool ticket_finish(unsigned *ticket, unsigned taken, unsigned workgroups)
{
    if (taken + 1u != workgroups) return false;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return true;
}

// Sum of `count` handed-over partials (element b at base[b * stride]) in index order, by ONE wave: the lanes fetch them side by
// side (a chain of dependent agent-scope loads would cost a memory round trip each), the additions then run in order over
// shuffled-in values -- ((p0 + p1) + p2) + ..., the bits of a sequential loop.  count <= 64 * HANDOFF_MAX_PER_LANE; every lane
// of the wave must call it; the sum is returned in every lane.
constexpr int HANDOFF_MAX_PER_LANE = 4;
constexpr int TICKET_MAX_WORKGROUPS = 64 * HANDOFF_MAX_PER_LANE;   // also keeps same-address ticket adds cheap (~5 ns each, serialised)
__device__ __forceinline__ double handoff_sum_ordered(const double *base, int count, int stride)
{
    const int lane = threadIdx.x & 63;
    double x[HANDOFF_MAX_PER_LANE];
#pragma unroll
    for (int j = 0; j < HANDOFF_MAX_PER_LANE; ++j) {
        const int b = lane + 64 * j;
        x[j] = b < count ? handoff_get(base + (size_t)b * stride) : 0.0;
    }
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < HANDOFF_MAX_PER_LANE; ++j) {
        if (64 * j >= count) break;
        const int m = count - 64 * j < 64 ? count - 64 * j : 64;
        for (int b = 0; b < m; ++b) v += __shfl(x[j], b, 64);
    }
    return v;
}
#endif

// LCREC_OK when ctx is NULL or belongs to the current device
int check_context(const lcrec_context *ctx, const char *who);

// kernels' launchers (host side, enqueue only)
int linear_forward(const float *x, int64_t n, int in_dim, const float *W, const float *b,
                   const float *bn_scale, const float *bn_shift, int relu, int out_dim, float *y,
                   hipStream_t stream);

size_t linear_bn_forward_workspace(int64_t n, int out_dim);
int linear_bn_forward(const float *x, int64_t n, int in_dim, const float *in_scale, const float *in_shift, int in_relu,
                      const float *W, const float *b, int out_dim, float *t_out, int want_stats, const float *gamma,
                      const float *beta, float eps, float momentum, float *running_mean, float *running_var, float *mean_out,
                      float *rstd_out, float *scale_out, float *shift_out, void *workspace, size_t workspace_bytes,
                      unsigned *tickets, hipStream_t stream);

int linear_backward_splits(int64_t n, int in_dim, int out_dim);
size_t linear_backward_workspace(int64_t n, int in_dim, int out_dim);
int linear_backward(const float *gy, const float *x, const float *W, int64_t n, int in_dim, int out_dim, float *gx, float *gw,
                    void *workspace, size_t workspace_bytes, hipStream_t stream);

size_t linear_backward_weights_workspace(const lcrec_dw_problem *problems, int count);
int linear_backward_weights(const lcrec_dw_problem *problems, int count, void *workspace, size_t workspace_bytes, hipStream_t stream);

size_t rq_assign_workspace(int64_t n, int e, const int *K, int L);
int rq_assign(const float *z, int64_t n, int e, const float *codebooks, const int *K, int L,
              int64_t *idx_out, int64_t idx_stride, float *xq_out, int xq_accumulate, double *sse_out, float *resid_out,
              float *margin_out, uint32_t *neartie_out, float tie_tau,
              void *workspace, size_t workspace_bytes, unsigned *ticket, hipStream_t stream);

size_t sinkhorn_workspace(int64_t n, int K, const int64_t *offs, int G);
int sinkhorn_assign(const float *r, int64_t n, int e, const float *cb, int K, const int64_t *offs, int G, double eps,
                    int iters, int64_t *idx_out, int64_t idx_stride, void *workspace, size_t workspace_bytes,
                    lcrec_context *ctx, unsigned *ticket, hipStream_t stream);
int apply_level(const float *r_in, int64_t n, int e, const float *cb, int K, const int64_t *idx, int64_t idx_stride,
                float *xq, int xq_accumulate, float *r_out, double *sse_out, void *workspace, size_t workspace_bytes,
                unsigned *ticket, hipStream_t stream);
int code_stats(const int64_t *idx, int64_t idx_stride, const float *resid, int64_t n, int e, int K, float *count,
               float *sum, hipStream_t stream);
int code_stats_levels(const int64_t *idx, const float *const *resid, int64_t n, int e, const int *K, int L, float *const *count,
                      float *const *sum, const float *const *cb, float *const *grad, float scale, float weight, hipStream_t stream);
size_t collision_workspace(int64_t n, int L);
int collision_groups(const int64_t *idx, int64_t n, int L, const int *K, int64_t *members_out, int64_t *offsets_out,
                     int64_t *counters_out, void *workspace, size_t workspace_bytes, hipStream_t stream);
int ema_update(float *ema_count, float *ema_sum, float *codebook, const float *count, const float *sum, int K, int e,
               float decay, float alpha, float keep, float eps, const unsigned char *skip, hipStream_t stream);

// training-step element-wise / reduction kernels (train_ops.hip)
int bn_relu_forward(const float *t, int64_t n, int F, const float *gamma, const float *beta, float eps, float momentum,
                    float *running_mean, float *running_var, float *y, float *mean_out, float *rstd_out, int relu,
                    hipStream_t stream);
int bn_relu_backward(const float *gy, const float *t, const float *y, int64_t n, int F, const float *gamma, const float *mean,
                     const float *rstd, int relu, float *dt, float *dgamma, float *dbeta, float *dbias, const float *fold_scale,
                     const float *fold_shift, hipStream_t stream);
int bn_stats(const float *t, int64_t n, int F, float *mean_out, float *m2_out, hipStream_t stream);
int bn_merge_stats(const float *rows, int world, int F, float eps, float momentum, float *mean_out, float *rstd_out,
                   float *running_mean, float *running_var, hipStream_t stream);
int bn_relu_apply(const float *t, int64_t n, int F, const float *gamma, const float *beta, const float *mean, const float *rstd,
                  int relu, float *y, hipStream_t stream);
int bn_backward_reduce(const float *gy, const float *t, const float *y, int64_t n, int F, const float *mean, const float *rstd,
                       int relu, float *sum_g, float *sum_gx, float *dbeta, float *dgamma, hipStream_t stream);
int bn_backward_apply(const float *gy, const float *t, const float *y, int64_t n, int F, const float *gamma, const float *mean,
                      const float *rstd, int relu, const float *sum_g, const float *sum_gx, float n_total, float *dt, float *dbias,
                      hipStream_t stream);
int relu_bias_backward(const float *gy, const float *y, int64_t n, int F, int relu, float *g_out, float *dbias, hipStream_t stream);
size_t train_reduce_workspace();
int recon_loss_grad(const float *out, const float *x, int64_t count, int64_t count_total, int l1, float *g, float *loss,
                    void *workspace, size_t workspace_bytes, unsigned *ticket, hipStream_t stream);
int grad_norm_clip(const float *g, int64_t count, float max_norm, float *norm_out, void *workspace, size_t workspace_bytes,
                   unsigned *ticket, hipStream_t stream);
int step_losses(const double *sse, int L, int64_t n, int e, float beta, float qlw, const float *recon, float *out3, double *sums2,
                unsigned char *nan_flag, const int64_t *probe, unsigned char *probe_flag, hipStream_t stream);
int quantizer_input_grad(const float *z, const float *cb0, const int64_t *idx, int64_t idx_stride, int64_t n, int e, float coef,
                         float weight, const float *g_xq, float *out, hipStream_t stream);
int quantizer_input_grad_bias(const float *z, const float *cb0, const int64_t *idx, int64_t idx_stride, int64_t n, int e, float coef,
                              float weight, const float *g_xq, float *out, float *dbias, hipStream_t stream);
int codebook_grad(const float *count, const float *sum, const float *cb, int K, int e, float scale, float weight, float *grad,
                  hipStream_t stream);
int adamw_step(float *p, float *g, float *m, float *v, int64_t count, const float *clip, int64_t *step, double base_lr,
               double beta1, double beta2, double eps, double weight_decay, int decoupled, int schedule, int64_t warmup_steps,
               int64_t total_steps, float *lr_out, unsigned *ticket, const unsigned char *skip, hipStream_t stream);

// host-side text (index_json.hip)
int64_t index_json_bound(int64_t n, int L);
int64_t index_json_format(const int64_t *idx, int64_t n, int L, int64_t first_item, char *out, int64_t cap);

}  // namespace lcrec
