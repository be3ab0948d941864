// extern "C" surface of liblcrec_hip.so (declared in include/lcrec.h).
#include "common.h"

#include <string.h>

#include <mutex>
#include <new>
#include <vector>

namespace lcrec {

static thread_local char g_err[512] = "";

char *err_buf() { return g_err; }

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const char *const kKernelNames[K_COUNT] = {"linear_fwd_128x128", "linear_fwd_128x64", "linear_fwd_128x32",
                                           "rq_assign", "rq_sse_finalize", "vq_distance", "sinkhorn",
                                           "sinkhorn_small", "rq_apply_level", "code_stats", "ema_update",
                                           "collision_groups", "linear_fwd_pp_256x128", "linear_fwd_64x64", "sinkhorn_slab",
                                           "sinkhorn_tiny", "bn_relu_forward", "bn_relu_backward", "relu_bias_backward",
                                           "recon_loss_grad", "grad_norm_clip", "adamw_step", "linear_fwd_32x64"};

struct TraceRec { int kernel; hipEvent_t start, stop; };
static std::mutex g_trace_mu;
static std::vector<TraceRec> g_trace;
static bool g_trace_on = false;

bool trace_on() { return g_trace_on; }

void trace_begin(int kernel, hipStream_t stream)
{
    TraceRec r;
    r.kernel = kernel;
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
    (void)hipEventRecord(r.start, stream);
    std::lock_guard<std::mutex> g(g_trace_mu);
    g_trace.push_back(r);
}

void trace_end(hipStream_t stream)
{
    std::lock_guard<std::mutex> g(g_trace_mu);
    if (!g_trace.empty()) (void)hipEventRecord(g_trace.back().stop, stream);
}

// Items per pass of the encoder chain: bounds the activation scratch (2 x chunk x widest hidden layer per pipeline: 8.6 GB at
// 524 288 x 2048 floats, of a 288 GB part) while keeping every GEMM launch >> 256 tiles.  Measured on C3 (round 3, one box, M
// items/s): 65 536: 16.26, 131 072 (rounds 1-2): 16.40, 262 144: 16.46, 524 288: 16.51 -- the dominant kernel's own rate does not
// move (0.930 of the roof throughout); fewer, longer launches mean fewer exposed first / last tiles of the narrow layers and
// fewer dispatch gaps.  Results do not depend on it (every output is one chain over k).
#ifndef LCREC_ENC_CHUNK
#define LCREC_ENC_CHUNK 524288
#endif
constexpr int64_t ENC_CHUNK = LCREC_ENC_CHUNK;
constexpr int ENC_PIPES_MAX = 2;      // measured: a third and fourth pipeline add nothing (and cost 2 GB of scratch each)

struct EncLayout {
    int64_t chunk;
    size_t act_bytes;     // one activation buffer
    size_t latent_bytes;  // [n][e]
    size_t rq_bytes;
};

static EncLayout enc_layout(int64_t n, const int *dims, int n_layers, const int *K, int L)
{
    EncLayout o;
    o.chunk = n < ENC_CHUNK ? (n > 0 ? n : 1) : ENC_CHUNK;
    int widest = 1;
    for (int l = 1; l < n_layers; ++l) widest = dims[l] > widest ? dims[l] : widest;
    o.act_bytes = align_up((size_t)o.chunk * widest * sizeof(float), 256);
    o.latent_bytes = align_up((size_t)(n > 0 ? n : 1) * dims[n_layers] * sizeof(float), 256);
    o.rq_bytes = rq_assign_workspace(n, dims[n_layers], K, L);
    return o;
}

int check_context(const lcrec_context *ctx, const char *who)
{
    if (!ctx) return LCREC_OK;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return fail(LCREC_EHIP, "%s: hipGetDevice failed", who);
    if (dev != ctx->device) return fail(LCREC_EINVAL, "%s: context belongs to device %d, current device is %d", who, ctx->device, dev);
    return LCREC_OK;
}

}  // namespace lcrec

using namespace lcrec;

int lcrec_context::ensure_streams()
{
    if (streams_ready) return LCREC_OK;
    bool ok = fork || hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < HELPERS && ok; ++i) {
        ok = helper[i] || hipStreamCreateWithFlags(&helper[i], hipStreamNonBlocking) == hipSuccess;
        ok = ok && (join[i] || hipEventCreateWithFlags(&join[i], hipEventDisableTiming) == hipSuccess);
    }
    if (!ok) return fail(LCREC_EHIP, "context: cannot create the helper streams: %s", hipGetErrorString(hipGetLastError()));
    streams_ready = true;
    return LCREC_OK;
}

void *lcrec_context::ring_acquire(size_t bytes, int *slot)
{
    const int i = pin_next;
    pin_next = (pin_next + 1) % RING;
    if (pin_busy[i]) {                      // RING uploads ago: complete unless the device is that far behind
        (void)hipEventSynchronize(pin_done[i]);
        pin_busy[i] = false;
    }
    if (pin_bytes[i] < bytes) {
        if (pin[i]) (void)hipHostFree(pin[i]);
        pin[i] = nullptr;
        pin_bytes[i] = 0;
        const size_t want = align_up(bytes + bytes / 2, 4096);
        if (hipHostMalloc(&pin[i], want, hipHostMallocDefault) != hipSuccess) {
            fail(LCREC_EHIP, "context: hipHostMalloc(%zu) failed", want);
            return nullptr;
        }
        pin_bytes[i] = want;
    }
    if (!pin_done[i] && hipEventCreateWithFlags(&pin_done[i], hipEventDisableTiming) != hipSuccess) {
        fail(LCREC_EHIP, "context: cannot create an event");
        return nullptr;
    }
    *slot = i;
    return pin[i];
}

void lcrec_context::ring_release(int slot, hipStream_t after)
{
    if (hipEventRecord(pin_done[slot], after) == hipSuccess) pin_busy[slot] = true;
}

LCREC_API int lcrec_context_create(lcrec_context **out)
{
    if (!out) return fail(LCREC_EINVAL, "context_create: NULL output");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return fail(LCREC_EHIP, "context_create: no HIP device");
    lcrec_context *c = new (std::nothrow) lcrec_context();
    if (!c) return fail(LCREC_EHIP, "context_create: out of host memory");
    c->device = dev;
    *out = c;
    return LCREC_OK;
}

LCREC_API int lcrec_context_destroy(lcrec_context *c)
{
    if (!c) return LCREC_OK;
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != c->device && hipSetDevice(c->device) == hipSuccess;
    for (int i = 0; i < lcrec_context::HELPERS; ++i) {
        if (c->helper[i]) { (void)hipStreamSynchronize(c->helper[i]); (void)hipStreamDestroy(c->helper[i]); }
        if (c->join[i]) (void)hipEventDestroy(c->join[i]);
    }
    if (c->fork) (void)hipEventDestroy(c->fork);
    for (int i = 0; i < lcrec_context::RING; ++i) {
        if (c->pin_busy[i]) (void)hipEventSynchronize(c->pin_done[i]);
        if (c->pin_done[i]) (void)hipEventDestroy(c->pin_done[i]);
        if (c->pin[i]) (void)hipHostFree(c->pin[i]);
    }
    if (switched) (void)hipSetDevice(cur);
    delete c;
    return LCREC_OK;
}

LCREC_API int lcrec_context_set_pipelines(lcrec_context *c, int pipelines)
{
    if (!c) return fail(LCREC_EINVAL, "context_set_pipelines: NULL context");
    if (pipelines < 1 || pipelines > 2) return fail(LCREC_EINVAL, "context_set_pipelines: %d (supported: 1, 2)", pipelines);
    c->pipelines = pipelines;
    return LCREC_OK;
}

LCREC_API int lcrec_version(void) { return LCREC_ABI_VERSION; }

LCREC_API const char *lcrec_last_error(void) { return err_buf(); }

LCREC_API int lcrec_linear_forward(const float *x, int64_t n, int in_dim, const float *W, const float *b,
                                   const float *bn_scale, const float *bn_shift, int relu, int out_dim,
                                   float *y, void *stream)
{
    return linear_forward(x, n, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, (hipStream_t)stream);
}

LCREC_API int lcrec_linear_backward_splits(int64_t n, int in_dim, int out_dim) { return linear_backward_splits(n, in_dim, out_dim); }

LCREC_API size_t lcrec_linear_backward_workspace(int64_t n, int in_dim, int out_dim)
{
    return linear_backward_workspace(n, in_dim, out_dim);
}

LCREC_API int lcrec_linear_backward(const float *gy, const float *x, const float *W, int64_t n, int in_dim, int out_dim,
                                    float *gx_out, float *gw_out, void *workspace, size_t workspace_bytes, void *stream)
{
    return linear_backward(gy, x, W, n, in_dim, out_dim, gx_out, gw_out, workspace, workspace_bytes, (hipStream_t)stream);
}

LCREC_API size_t lcrec_rq_assign_workspace(int64_t n, int e, const int *K, int L)
{
    return rq_assign_workspace(n, e, K, L);
}

LCREC_API int lcrec_rq_assign(const float *z, int64_t n, int e, const float *codebooks, const int *K, int L,
                              int64_t *idx_out, int64_t idx_stride, float *xq_out, int xq_accumulate, double *sse_out,
                              float *resid_out, float *margin_out, uint32_t *neartie_out, float tie_tau,
                              void *workspace, size_t workspace_bytes, unsigned int *ticket, void *stream)
{
    return rq_assign(z, n, e, codebooks, K, L, idx_out, idx_stride, xq_out, xq_accumulate, sse_out, resid_out, margin_out,
                     neartie_out, tie_tau, workspace, workspace_bytes, ticket, (hipStream_t)stream);
}

LCREC_API size_t lcrec_encode_assign_workspace(int64_t n, const int *dims, int n_layers, const int *K, int L)
{
    if (!dims || n_layers < 1 || n_layers > LCREC_MAX_LAYERS || !K || L < 1) return 0;
    EncLayout o = enc_layout(n, dims, n_layers, K, L);
    return 2 * ENC_PIPES_MAX * o.act_bytes + o.latent_bytes + o.rq_bytes;      // an activation ping-pong pair per chunk pipeline
}

LCREC_API int64_t lcrec_encode_assign_chunk_rows(void) { return ENC_CHUNK; }

LCREC_API int lcrec_encode_assign(const float *x, int64_t n, const int *dims, int n_layers,
                                  const float *const *W, const float *const *b,
                                  const float *const *bn_scale, const float *const *bn_shift,
                                  const float *codebooks, const int *K, int L, int64_t *idx_out,
                                  float *latent_out, float *xq_out, double *sse_out,
                                  float *margin_out, uint32_t *neartie_out, float tie_tau,
                                  void *workspace, size_t workspace_bytes, lcrec_context *ctx, void *stream)
{
    if (n == 0 && dims && K) return LCREC_OK;              // empty batch
    if (!x || !dims || !W || !b || !codebooks || !K || !idx_out)
        return fail(LCREC_EINVAL, "encode_assign: NULL pointer");
    if (n_layers < 1 || n_layers > LCREC_MAX_LAYERS)
        return fail(LCREC_EINVAL, "encode_assign: n_layers=%d out of range", n_layers);
    if (n < 0) return fail(LCREC_EINVAL, "encode_assign: n < 0");
    if (n == 0) return LCREC_OK;
    const size_t need = lcrec_encode_assign_workspace(n, dims, n_layers, K, L);
    if (!workspace || workspace_bytes < need)
        return fail(LCREC_EWORKSPACE, "encode_assign: workspace %zu B < required %zu B", workspace_bytes, need);
    if (int rc = check_context(ctx, "encode_assign")) return rc;
    const EncLayout o = enc_layout(n, dims, n_layers, K, L);
    char *ws = reinterpret_cast<char *>(workspace);
    float *latent = latent_out ? latent_out : reinterpret_cast<float *>(ws + 2 * ENC_PIPES_MAX * o.act_bytes);
    void *rq_ws = ws + 2 * ENC_PIPES_MAX * o.act_bytes + o.latent_bytes;
    const int e = dims[n_layers];
    hipStream_t s = (hipStream_t)stream;

    // Chunk pipelines: chunk c runs on pipeline c % P, pipeline 0 being the caller's stream and pipeline 1 the
    // context's first helper stream, forked from the caller's and joined back before the quantiser pass -- so two
    // layers' kernels are in flight at once and their workgroups interleave on the CUs: one kernel's store bursts,
    // prologues, narrow tail layers and last partial round run under the other's K loops.  Measured on C3: P = 2 is
    // +1.5 .. +1.9 % over P = 1 in the common case, P = 3 and 4 give nothing -- but in a few percent of passes one stream's
    // persistent launches are starved of CUs by the other's and the pass takes 1.6-3x (tools/generate_probe.py: wall 101 ms
    // with 125 ms of kernel brackets against 61 / 120), so P = 1 is the default.  The call does not synchronise with the host.
    const int64_t n_chunks = (n + o.chunk - 1) / o.chunk;
    int P = ctx ? ctx->pipelines : 1;
    if (P > ENC_PIPES_MAX) P = ENC_PIPES_MAX;
    if (n_chunks < P) P = (int)n_chunks;
    if (P > 1) {
        if (int rc = ctx->ensure_streams()) return rc;
    }
    {
        ForkJoin fj(ctx, s, P > 1 ? 1u : 0u);              // joins on every exit from this block
        int64_t c = 0;
        for (int64_t i0 = 0; i0 < n; i0 += o.chunk, ++c) {
            const int64_t m = (n - i0 < o.chunk) ? n - i0 : o.chunk;
            const float *src = x + i0 * dims[0];
            const int pipe = (int)(c % P);
            hipStream_t cs = pipe ? fj.on(pipe - 1) : s;
            float *act[2] = {reinterpret_cast<float *>(ws + (size_t)(2 * pipe) * o.act_bytes),
                             reinterpret_cast<float *>(ws + (size_t)(2 * pipe + 1) * o.act_bytes)};
            for (int l = 0; l < n_layers; ++l) {
                const bool last = l == n_layers - 1;
                float *dst = last ? latent + i0 * e : act[l & 1];
                int rc = linear_forward(src, m, dims[l], W[l], b[l], bn_scale ? bn_scale[l] : nullptr,
                                        bn_shift ? bn_shift[l] : nullptr, last ? 0 : 1, dims[l + 1], dst, cs);
                if (rc) return rc;
                src = dst;
            }
        }
    }
    return rq_assign(latent, n, e, codebooks, K, L, idx_out, L, xq_out, 0, sse_out, nullptr, margin_out, neartie_out, tie_tau,
                     rq_ws, o.rq_bytes, nullptr, s);
}

LCREC_API int lcrec_trace_enable(int on)
{
    std::lock_guard<std::mutex> g(g_trace_mu);
    for (auto &r : g_trace) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    g_trace.clear();
    g_trace_on = on != 0;
    return LCREC_OK;
}

LCREC_API int lcrec_trace_collect(lcrec_trace_entry *out, int capacity)
{
    if (!out || capacity < 1) return fail(LCREC_EINVAL, "trace_collect: no output buffer");
    std::lock_guard<std::mutex> g(g_trace_mu);
    int64_t launches[K_COUNT] = {0};
    double ms[K_COUNT] = {0.0};
    for (auto &r : g_trace) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(r.stop);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, r.start, r.stop);
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
        if (e != hipSuccess) { g_trace.clear(); return fail(LCREC_EHIP, "trace_collect: %s", hipGetErrorString(e)); }
        launches[r.kernel] += 1;
        ms[r.kernel] += t;
    }
    g_trace.clear();
    int n = 0;
    for (int k = 0; k < K_COUNT && n < capacity; ++k)
        if (launches[k]) { out[n].kernel = kKernelNames[k]; out[n].launches = launches[k]; out[n].total_ms = ms[k]; ++n; }
    return n;
}

LCREC_API size_t lcrec_sinkhorn_assign_workspace(int64_t n, int K, const int64_t *group_offsets, int n_groups)
{
    if (!group_offsets || n_groups < 1) return 256;
    return sinkhorn_workspace(n, K, group_offsets, n_groups);
}

LCREC_API int lcrec_sinkhorn_assign(const float *resid, int64_t n, int e, const float *codebook, int K,
                                    const int64_t *group_offsets, int n_groups, double epsilon, int iters,
                                    int64_t *idx_out, int64_t idx_stride, void *workspace, size_t workspace_bytes,
                                    lcrec_context *ctx, unsigned int *ticket, void *stream)
{
    return sinkhorn_assign(resid, n, e, codebook, K, group_offsets, n_groups, epsilon, iters, idx_out, idx_stride,
                           workspace, workspace_bytes, ctx, ticket, (hipStream_t)stream);
}

LCREC_API int lcrec_rq_apply_level(const float *resid_in, int64_t n, int e, const float *codebook, int K,
                                   const int64_t *idx, int64_t idx_stride, float *xq, int xq_accumulate,
                                   float *resid_out, double *sse_out, void *workspace, size_t workspace_bytes,
                                   unsigned int *ticket, void *stream)
{
    return apply_level(resid_in, n, e, codebook, K, idx, idx_stride, xq, xq_accumulate, resid_out, sse_out, workspace,
                       workspace_bytes, ticket, (hipStream_t)stream);
}

LCREC_API int lcrec_code_stats(const int64_t *idx, int64_t idx_stride, const float *resid, int64_t n, int e, int K,
                               float *count, float *sum, void *stream)
{
    return code_stats(idx, idx_stride, resid, n, e, K, count, sum, (hipStream_t)stream);
}

LCREC_API int lcrec_ema_update(float *ema_count, float *ema_sum, float *codebook, const float *count,
                               const float *sum, int K, int e, float decay, float alpha, float keep, float eps,
                               const unsigned char *skip_flag, void *stream)
{
    return ema_update(ema_count, ema_sum, codebook, count, sum, K, e, decay, alpha, keep, eps, skip_flag, (hipStream_t)stream);
}

LCREC_API int64_t lcrec_index_json_bound(int64_t n, int L) { return index_json_bound(n, L); }

LCREC_API int64_t lcrec_index_json_format(const int64_t *idx, int64_t n, int L, int64_t first_item, char *out, int64_t cap)
{
    return index_json_format(idx, n, L, first_item, out, cap);
}

LCREC_API size_t lcrec_collision_groups_workspace(int64_t n, int L) { return collision_workspace(n, L); }

LCREC_API int lcrec_collision_groups(const int64_t *idx, int64_t n, int L, const int *K, int64_t *members_out,
                                     int64_t *group_offsets_out, int64_t *counters_out, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    return collision_groups(idx, n, L, K, members_out, group_offsets_out, counters_out, workspace, workspace_bytes,
                            (hipStream_t)stream);
}

LCREC_API int lcrec_bn_relu_forward(const float *t, int64_t n, int features, const float *gamma, const float *beta, float eps,
                                    float momentum, float *running_mean, float *running_var, float *y, float *mean_out,
                                    float *rstd_out, int relu, void *stream)
{
    return bn_relu_forward(t, n, features, gamma, beta, eps, momentum, running_mean, running_var, y, mean_out, rstd_out, relu,
                           (hipStream_t)stream);
}

LCREC_API int lcrec_bn_relu_backward(const float *gy, const float *t, const float *y, int64_t n, int features, const float *gamma,
                                     const float *mean, const float *rstd, int relu, float *dt_out, float *dgamma_out,
                                     float *dbeta_out, float *dbias_out, const float *fold_scale, const float *fold_shift,
                                     void *stream)
{
    return bn_relu_backward(gy, t, y, n, features, gamma, mean, rstd, relu, dt_out, dgamma_out, dbeta_out, dbias_out, fold_scale,
                            fold_shift, (hipStream_t)stream);
}

LCREC_API size_t lcrec_linear_bn_forward_workspace(int64_t n, int out_dim) { return linear_bn_forward_workspace(n, out_dim); }

LCREC_API int lcrec_linear_bn_forward(const float *x, int64_t n, int in_dim, const float *in_scale, const float *in_shift, int in_relu,
                                      const float *W, const float *b, int out_dim, float *t_out, int want_stats, const float *gamma,
                                      const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                      float *mean_out, float *rstd_out, float *scale_out, float *shift_out, void *workspace,
                                      size_t workspace_bytes, unsigned int *tickets, void *stream)
{
    return linear_bn_forward(x, n, in_dim, in_scale, in_shift, in_relu, W, b, out_dim, t_out, want_stats, gamma, beta, eps, momentum,
                             running_mean, running_var, mean_out, rstd_out, scale_out, shift_out, workspace, workspace_bytes, tickets,
                             (hipStream_t)stream);
}

LCREC_API int lcrec_relu_bias_backward(const float *gy, const float *y, int64_t n, int features, int relu, float *g_out,
                                       float *dbias_out, void *stream)
{
    return relu_bias_backward(gy, y, n, features, relu, g_out, dbias_out, (hipStream_t)stream);
}

LCREC_API size_t lcrec_train_reduce_workspace(void) { return train_reduce_workspace(); }

LCREC_API int lcrec_recon_loss_grad(const float *out, const float *x, int64_t count, int64_t count_total, int l1, float *grad_out,
                                    float *loss_out, void *workspace, size_t workspace_bytes, unsigned int *ticket, void *stream)
{
    return recon_loss_grad(out, x, count, count_total, l1, grad_out, loss_out, workspace, workspace_bytes, ticket, (hipStream_t)stream);
}

LCREC_API int lcrec_grad_norm_clip(const float *grads, int64_t count, float max_norm, float *norm_out, void *workspace,
                                   size_t workspace_bytes, unsigned int *ticket, void *stream)
{
    return grad_norm_clip(grads, count, max_norm, norm_out, workspace, workspace_bytes, ticket, (hipStream_t)stream);
}

LCREC_API int lcrec_adamw_step(float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t count, const float *clip,
                               int64_t *step, double base_lr, double beta1, double beta2, double eps, double weight_decay,
                               int decoupled, int schedule, int64_t warmup_steps, int64_t total_steps, float *lr_out,
                               unsigned int *ticket, const unsigned char *skip_flag, void *stream)
{
    return adamw_step(params, grads, exp_avg, exp_avg_sq, count, clip, step, base_lr, beta1, beta2, eps, weight_decay, decoupled,
                      schedule, warmup_steps, total_steps, lr_out, ticket, skip_flag, (hipStream_t)stream);
}

LCREC_API int lcrec_codebook_grad(const float *count, const float *sum, const float *codebook, int K, int e, float scale,
                                  float weight, float *grad_out, void *stream)
{
    return codebook_grad(count, sum, codebook, K, e, scale, weight, grad_out, (hipStream_t)stream);
}

LCREC_API int lcrec_bn_stats(const float *t, int64_t n, int features, float *mean_out, float *m2_out, void *stream)
{
    return bn_stats(t, n, features, mean_out, m2_out, (hipStream_t)stream);
}

LCREC_API int lcrec_bn_relu_apply(const float *t, int64_t n, int features, const float *gamma, const float *beta, const float *mean,
                                  const float *rstd, int relu, float *y, void *stream)
{
    return bn_relu_apply(t, n, features, gamma, beta, mean, rstd, relu, y, (hipStream_t)stream);
}

LCREC_API int lcrec_bn_backward_reduce(const float *gy, const float *t, const float *y, int64_t n, int features, const float *mean,
                                       const float *rstd, int relu, float *sum_g_out, float *sum_gx_out, float *dbeta_out,
                                       float *dgamma_out, void *stream)
{
    return bn_backward_reduce(gy, t, y, n, features, mean, rstd, relu, sum_g_out, sum_gx_out, dbeta_out, dgamma_out, (hipStream_t)stream);
}

LCREC_API int lcrec_bn_merge_stats(const float *rows, int world, int features, float eps, float momentum, float *mean_out,
                                   float *rstd_out, float *running_mean, float *running_var, void *stream)
{
    return bn_merge_stats(rows, world, features, eps, momentum, mean_out, rstd_out, running_mean, running_var, (hipStream_t)stream);
}

LCREC_API int lcrec_bn_backward_apply(const float *gy, const float *t, const float *y, int64_t n, int features, const float *gamma,
                                      const float *mean, const float *rstd, int relu, const float *sum_g, const float *sum_gx,
                                      float n_total, float *dt_out, float *dbias_out, void *stream)
{
    return bn_backward_apply(gy, t, y, n, features, gamma, mean, rstd, relu, sum_g, sum_gx, n_total, dt_out, dbias_out,
                             (hipStream_t)stream);
}

LCREC_API size_t lcrec_linear_backward_weights_workspace(const lcrec_dw_problem *problems, int count)
{
    if (!problems || count < 1) return 0;
    return linear_backward_weights_workspace(problems, count);
}

LCREC_API int lcrec_linear_backward_weights(const lcrec_dw_problem *problems, int count, void *workspace, size_t workspace_bytes,
                                            void *stream)
{
    return linear_backward_weights(problems, count, workspace, workspace_bytes, (hipStream_t)stream);
}

LCREC_API int lcrec_step_losses(const double *sse, int L, int64_t n, int e, float beta, float quant_loss_weight, const float *recon,
                                float *losses_out, double *sums_inout, unsigned char *nan_flag, const int64_t *poison_probe,
                                unsigned char *poison_flag, void *stream)
{
    return step_losses(sse, L, n, e, beta, quant_loss_weight, recon, losses_out, sums_inout, nan_flag, poison_probe, poison_flag,
                       (hipStream_t)stream);
}

LCREC_API int lcrec_quantizer_input_grad(const float *z, const float *codebook0, const int64_t *idx, int64_t idx_stride, int64_t n,
                                         int e, float coef, float weight, const float *g_xq, float *out, void *stream)
{
    return quantizer_input_grad(z, codebook0, idx, idx_stride, n, e, coef, weight, g_xq, out, (hipStream_t)stream);
}

LCREC_API int lcrec_quantizer_input_grad_bias(const float *z, const float *codebook0, const int64_t *idx, int64_t idx_stride, int64_t n,
                                              int e, float coef, float weight, const float *g_xq, float *out, float *dbias_out, void *stream)
{
    return quantizer_input_grad_bias(z, codebook0, idx, idx_stride, n, e, coef, weight, g_xq, out, dbias_out, (hipStream_t)stream);
}

LCREC_API int lcrec_code_stats_levels(const int64_t *idx, const float *const *resid, int64_t n, int e, const int *K, int L,
                                      float *const *count, float *const *sum, const float *const *codebooks, float *const *grad_out,
                                      float scale, float weight, void *stream)
{
    return code_stats_levels(idx, resid, n, e, K, L, count, sum, codebooks, grad_out, scale, weight, (hipStream_t)stream);
}
