// The element-wise / column-reduction half of a training step, for gfx950 -- what the reference leaves to a dozen
// aten kernels per layer ("full training step on device"):
//
//   bn_relu_forward    training-mode BatchNorm1d (+ ReLU) after a Linear: index/models/layers.py:25-30
//   bn_relu_backward   its backward (autograd of the above under loss.backward(), index/trainer.py:117), incl. the
//                      gradient of the Linear's bias
//   relu_bias_backward ReLU mask + bias gradient of a Linear without BatchNorm (layers.py:23,28-30)
//   recon_loss_grad    mse / l1 reconstruction loss and its gradient (index/models/rqvae.py:74-85)
//   grad_norm_clip     global L2 norm of all gradients + clip coefficient (clip_grad_norm_(.., 1.0), trainer.py:118)
//   adamw_step         clipped AdamW / Adam update with the warm-up schedule evaluated on the device
//                      (trainer.py:49-92,119-120)
//
// All of it is HBM/L2-bound streaming over [batch][features] or over the flat parameter buffer; no MFMA.  Column
// reductions (BatchNorm statistics, bias gradients) are done by one workgroup per strip of 32 columns: 8 row groups of
// 32 lanes read 128-byte row segments (coalesced), each lane sums its rows in order, the 8 partials are added in a fixed
// order through LDS -- deterministic, no atomics.  Batch-sized inputs only (a training batch is 1-8 k rows; the whole
// strip is re-read from L2 for the second pass).
#include "common.h"

namespace lcrec {

constexpr int COLS = 32;          // columns per workgroup
constexpr int RGS = 8;            // row groups per workgroup
constexpr int CR_THREADS = COLS * RGS;

// sum over the RGS row groups of one value per (row group, column); result valid in every thread of the column
__device__ __forceinline__ float strip_sum(float v, float (*sm)[COLS], int rg, int c)
{
    __syncthreads();                       // previous use of sm is over
    sm[rg][c] = v;
    __syncthreads();
    float s = sm[0][c];
#pragma unroll
    for (int g = 1; g < RGS; ++g) s += sm[g][c];
    return s;
}

// Training-mode BatchNorm1d (+ReLU).  torch semantics: batch mean, biased variance for the normalisation,
// running_mean/var updated with `momentum` (running_var from the unbiased variance), eps inside the square root.
__global__ __launch_bounds__(CR_THREADS) void bn_relu_forward_kernel(const float *__restrict__ t, int64_t n, int F,
                                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                      float eps, float momentum, float *running_mean,
                                                                      float *running_var, float *__restrict__ y, float *mean_out,
                                                                      float *rstd_out, int relu)
{
    __shared__ float sm[RGS][COLS];
    const int c = threadIdx.x % COLS, rg = threadIdx.x / COLS;
    const int col = blockIdx.x * COLS + c;
    const bool live = col < F;
    const float inv_n = 1.0f / (float)n;
    float s = 0.f;
    if (live)
        for (int64_t r = rg; r < n; r += RGS) s += t[r * F + col];
    const float mean = strip_sum(s, sm, rg, c) * inv_n;
    float q = 0.f;
    if (live)
        for (int64_t r = rg; r < n; r += RGS) { const float d = t[r * F + col] - mean; q = __builtin_fmaf(d, d, q); }
    const float m2 = strip_sum(q, sm, rg, c);
    const float var = m2 * inv_n;
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    if (!live) return;
    const float g = gamma ? gamma[col] : 1.0f, b = beta ? beta[col] : 0.0f;
    for (int64_t r = rg; r < n; r += RGS) {
        float v = (t[r * F + col] - mean) * rstd * g + b;
        if (relu) v = v > 0.f ? v : 0.f;
        y[r * F + col] = v;
    }
    if (rg == 0) {
        mean_out[col] = mean;
        rstd_out[col] = rstd;
        if (running_mean) running_mean[col] = (1.0f - momentum) * running_mean[col] + momentum * mean;
        if (running_var) {
            const float unbiased = n > 1 ? m2 / (float)(n - 1) : var;
            running_var[col] = (1.0f - momentum) * running_var[col] + momentum * unbiased;
        }
    }
}

// Backward of y = [relu](bn(t)) for gy = dL/dy:
//   g = gy * [y > 0];  dbeta = sum g;  dgamma = sum g * xhat;  dt = gamma * rstd * (g - dbeta/n - xhat * dgamma/n)
// and the gradient of the Linear bias that produced t: dbias = sum dt (zero up to rounding, as in autograd).
__global__ __launch_bounds__(CR_THREADS) void bn_relu_backward_kernel(const float *__restrict__ gy, const float *__restrict__ t,
                                                                       const float *__restrict__ y, int64_t n, int F,
                                                                       const float *__restrict__ gamma, const float *__restrict__ mean,
                                                                       const float *__restrict__ rstd, int relu, float *__restrict__ dt,
                                                                       float *dgamma, float *dbeta, float *dbias)
{
    __shared__ float sm[RGS][COLS];
    const int c = threadIdx.x % COLS, rg = threadIdx.x / COLS;
    const int col = blockIdx.x * COLS + c;
    const bool live = col < F;
    const float mu = live ? mean[col] : 0.f, rs = live ? rstd[col] : 0.f, gm = live ? (gamma ? gamma[col] : 1.0f) : 0.f;
    float sg = 0.f, sgx = 0.f;
    if (live)
        for (int64_t r = rg; r < n; r += RGS) {
            float g = gy[r * F + col];
            if (relu && !(y[r * F + col] > 0.f)) g = 0.f;
            const float xh = (t[r * F + col] - mu) * rs;
            sg += g;
            sgx = __builtin_fmaf(g, xh, sgx);
        }
    const float db = strip_sum(sg, sm, rg, c);
    const float dg = strip_sum(sgx, sm, rg, c);
    const float inv_n = 1.0f / (float)n;
    const float k = gm * rs, mdb = db * inv_n, mdg = dg * inv_n;
    float sdt = 0.f;
    if (live)
        for (int64_t r = rg; r < n; r += RGS) {
            float g = gy[r * F + col];
            if (relu && !(y[r * F + col] > 0.f)) g = 0.f;
            const float xh = (t[r * F + col] - mu) * rs;
            const float v = k * (g - mdb - xh * mdg);
            dt[r * F + col] = v;
            sdt += v;
        }
    const float dbs = strip_sum(sdt, sm, rg, c);
    if (live && rg == 0) {
        if (dgamma) dgamma[col] = dg;
        if (dbeta) dbeta[col] = db;
        if (dbias) dbias[col] = dbs;
    }
}

// g = gy * [y > 0] (in place allowed), dbias = column sums of g
__global__ __launch_bounds__(CR_THREADS) void relu_bias_backward_kernel(const float *gy, const float *__restrict__ y, int64_t n, int F,
                                                                         int relu, float *g_out, float *dbias)
{
    __shared__ float sm[RGS][COLS];
    const int c = threadIdx.x % COLS, rg = threadIdx.x / COLS;
    const int col = blockIdx.x * COLS + c;
    const bool live = col < F;
    float s = 0.f;
    if (live)
        for (int64_t r = rg; r < n; r += RGS) {
            float g = gy[r * F + col];
            if (relu && !(y[r * F + col] > 0.f)) g = 0.f;
            if (g_out) g_out[r * F + col] = g;
            s += g;
        }
    const float db = strip_sum(s, sm, rg, c);
    if (live && rg == 0 && dbias) dbias[col] = db;
}

// ---- whole-tensor reductions: per-workgroup fp64 partials, then one finishing workgroup (fixed order)
constexpr int RED_THREADS = 256;
constexpr int RED_MAX_BLOCKS = 1024;

__device__ __forceinline__ double block_sum(double v, double *sm)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < RED_THREADS / 64; ++w) s += sm[w];
    return s;                              // valid in thread 0
}

// mse: loss = mean (out - x)^2, g = 2 (out - x) / count;  l1: loss = mean |out - x|, g = sign(out - x) / count
__global__ __launch_bounds__(RED_THREADS) void recon_loss_grad_kernel(const float *__restrict__ out, const float *__restrict__ x,
                                                                       int64_t count, int l1, float *__restrict__ g, double *partial)
{
    __shared__ double sm[RED_THREADS / 64];
    const float scale = (l1 ? 1.0f : 2.0f) / (float)count;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < count; i += (int64_t)gridDim.x * RED_THREADS) {
        const float d = out[i] - x[i];
        if (l1) {
            acc += (double)__builtin_fabsf(d);
            if (g) g[i] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
        } else {
            acc += (double)d * (double)d;
            if (g) g[i] = d * scale;
        }
    }
    const double s = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(RED_THREADS) void recon_loss_finish_kernel(const double *partial, int blocks, int64_t count, float *loss)
{
    __shared__ double sm[RED_THREADS / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < blocks; i += RED_THREADS) acc += partial[i];
    const double s = block_sum(acc, sm);
    if (threadIdx.x == 0) *loss = (float)(s / (double)count);
}

__global__ __launch_bounds__(RED_THREADS) void sumsq_kernel(const float *__restrict__ g, int64_t count, double *partial)
{
    __shared__ double sm[RED_THREADS / 64];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < count; i += (int64_t)gridDim.x * RED_THREADS) {
        const double v = (double)g[i];
        acc += v * v;
    }
    const double s = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// norm_out[0] = ||g||_2, norm_out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(RED_THREADS) void grad_norm_finish_kernel(const double *partial, int blocks, float max_norm, float *norm_out)
{
    __shared__ double sm[RED_THREADS / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < blocks; i += RED_THREADS) acc += partial[i];
    const double s = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        const float norm = (float)__builtin_sqrt(s);
        float coef = max_norm / (norm + 1e-6f);
        coef = coef > 1.0f ? 1.0f : coef;
        norm_out[0] = norm;
        norm_out[1] = coef;
    }
}

struct AdamParams {
    float *p, *g, *m, *v;
    int64_t count;
    const float *clip;       // [2]: norm, coefficient; or NULL (no clipping)
    int64_t *step;           // device step counter: optimiser steps taken so far (incremented by the kernel's last act)
    double base_lr, beta1, beta2, eps, weight_decay;
    int decoupled;           // 1 = AdamW, 0 = Adam (L2 added to the gradient)
    int schedule;            // 0 = constant after warm-up, 1 = linear decay to 0 at total_steps, -1 = no schedule (lr = base_lr)
    int64_t warmup_steps, total_steps;
    float *lr_out;           // device float: the learning rate this step used (logging / tests), or NULL
};

// learning rate of optimiser step `s` (0-based): base_lr * lambda(s), the python-double arithmetic of
// transformers' get_{linear,constant}_schedule_with_warmup (reference index/trainer.py:83-92)
__device__ __forceinline__ double lr_at(const AdamParams &a, int64_t s)
{
    if (a.schedule < 0) return a.base_lr;
    double f;
    if (s < a.warmup_steps) {
        f = (double)s / (double)(a.warmup_steps > 1 ? a.warmup_steps : 1);
    } else if (a.schedule == 1) {
        const int64_t den = a.total_steps - a.warmup_steps;
        f = (double)(a.total_steps - s) / (double)(den > 1 ? den : 1);
        f = f > 0.0 ? f : 0.0;
    } else {
        f = 1.0;
    }
    return a.base_lr * f;
}

// torch.optim.AdamW / Adam, single-tensor formulation (the fused kernel's arithmetic: state in fp32, hyper-parameters in
// double), on the flat parameter buffer, with the clip coefficient applied to the gradient on the way in (and stored back,
// as clip_grad_norm_ leaves it).
__global__ __launch_bounds__(256) void adamw_step_kernel(AdamParams a)
{
    const int64_t s = *a.step;                       // steps taken before this one
    const double lr = lr_at(a, s);
    const double t = (double)(s + 1);
    const double bc1 = 1.0 - pow(a.beta1, t);
    const double bc2_sqrt = sqrt(1.0 - pow(a.beta2, t));
    const float coef = a.clip ? a.clip[1] : 1.0f;
    const float step_size = (float)(lr / bc1);
    const float b1 = (float)a.beta1, b2 = (float)a.beta2, one_m_b1 = (float)(1.0 - a.beta1), one_m_b2 = (float)(1.0 - a.beta2);
    const float epsf = (float)a.eps, bc2s = (float)bc2_sqrt;
    const float decay = (float)(lr * a.weight_decay), wd = (float)a.weight_decay;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.count; i += (int64_t)gridDim.x * 256) {
        float p = a.p[i], g = a.g[i] * coef, m = a.m[i], v = a.v[i];
        a.g[i] = g;
        if (a.weight_decay != 0.0) {
            if (a.decoupled) p -= decay * p;
            else g += p * wd;
        }
        m = m + one_m_b1 * (g - m);                  // lerp(m, g, 1 - beta1)
        v = b2 * v + one_m_b2 * g * g;
        const float denom = __builtin_sqrtf(v) / bc2s + epsf;
        p -= step_size * m / denom;
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
    (void)b1;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.lr_out) *a.lr_out = (float)lr;
}

__global__ void step_advance_kernel(int64_t *step) { *step += 1; }

// ---------------------------------------------------------------- host side

static int strips(int F) { return (F + COLS - 1) / COLS; }

int bn_relu_forward(const float *t, int64_t n, int F, const float *gamma, const float *beta, float eps, float momentum,
                    float *running_mean, float *running_var, float *y, float *mean_out, float *rstd_out, int relu,
                    hipStream_t stream)
{
    if (n == 0 || F == 0) return LCREC_OK;
    if (!t || !y || !mean_out || !rstd_out) return fail(LCREC_EINVAL, "bn_relu_forward: NULL pointer");
    if (n < 2) return fail(LCREC_EINVAL, "bn_relu_forward: training-mode BatchNorm needs more than 1 row (n=%lld)", (long long)n);
    if (n > (1 << 20) || F < 1) return fail(LCREC_EUNSUPPORTED, "bn_relu_forward: sized for training batches (n=%lld)", (long long)n);
    TraceScope trace(K_BN_FWD, stream);
    hipLaunchKernelGGL(bn_relu_forward_kernel, dim3(strips(F)), dim3(CR_THREADS), 0, stream, t, n, F, gamma, beta, eps, momentum,
                       running_mean, running_var, y, mean_out, rstd_out, relu);
    return check_launch("bn_relu_forward_kernel");
}

int bn_relu_backward(const float *gy, const float *t, const float *y, int64_t n, int F, const float *gamma, const float *mean,
                     const float *rstd, int relu, float *dt, float *dgamma, float *dbeta, float *dbias, hipStream_t stream)
{
    if (n == 0 || F == 0) return LCREC_OK;
    if (!gy || !t || !mean || !rstd || !dt || (relu && !y)) return fail(LCREC_EINVAL, "bn_relu_backward: NULL pointer");
    if (n > (1 << 20) || F < 1) return fail(LCREC_EUNSUPPORTED, "bn_relu_backward: sized for training batches (n=%lld)", (long long)n);
    TraceScope trace(K_BN_BWD, stream);
    hipLaunchKernelGGL(bn_relu_backward_kernel, dim3(strips(F)), dim3(CR_THREADS), 0, stream, gy, t, y, n, F, gamma, mean, rstd, relu,
                       dt, dgamma, dbeta, dbias);
    return check_launch("bn_relu_backward_kernel");
}

int relu_bias_backward(const float *gy, const float *y, int64_t n, int F, int relu, float *g_out, float *dbias, hipStream_t stream)
{
    if (n == 0 || F == 0) return LCREC_OK;
    if (!gy || (relu && !y)) return fail(LCREC_EINVAL, "relu_bias_backward: NULL pointer");
    if (n > (1 << 20) || F < 1) return fail(LCREC_EUNSUPPORTED, "relu_bias_backward: sized for training batches (n=%lld)", (long long)n);
    TraceScope trace(K_RELU_BIAS_BWD, stream);
    hipLaunchKernelGGL(relu_bias_backward_kernel, dim3(strips(F)), dim3(CR_THREADS), 0, stream, gy, y, n, F, relu, g_out, dbias);
    return check_launch("relu_bias_backward_kernel");
}

static int red_blocks(int64_t count)
{
    int64_t b = (count + (int64_t)RED_THREADS * 16 - 1) / ((int64_t)RED_THREADS * 16);
    return (int)(b < 1 ? 1 : (b > RED_MAX_BLOCKS ? RED_MAX_BLOCKS : b));
}

size_t train_reduce_workspace() { return RED_MAX_BLOCKS * sizeof(double); }

int recon_loss_grad(const float *out, const float *x, int64_t count, int l1, float *g, float *loss, void *workspace,
                    size_t workspace_bytes, hipStream_t stream)
{
    if (!out || !x || !loss) return fail(LCREC_EINVAL, "recon_loss_grad: NULL pointer");
    if (count < 1) return fail(LCREC_EINVAL, "recon_loss_grad: empty input");
    if (!workspace || workspace_bytes < train_reduce_workspace()) return fail(LCREC_EWORKSPACE, "recon_loss_grad: workspace too small");
    const int blocks = red_blocks(count);
    TraceScope trace(K_LOSS, stream);
    hipLaunchKernelGGL(recon_loss_grad_kernel, dim3(blocks), dim3(RED_THREADS), 0, stream, out, x, count, l1, g, (double *)workspace);
    hipLaunchKernelGGL(recon_loss_finish_kernel, dim3(1), dim3(RED_THREADS), 0, stream, (const double *)workspace, blocks, count, loss);
    return check_launch("recon_loss_grad_kernel");
}

int grad_norm_clip(const float *g, int64_t count, float max_norm, float *norm_out, void *workspace, size_t workspace_bytes,
                   hipStream_t stream)
{
    if (!g || !norm_out) return fail(LCREC_EINVAL, "grad_norm_clip: NULL pointer");
    if (count < 1) return fail(LCREC_EINVAL, "grad_norm_clip: empty input");
    if (!workspace || workspace_bytes < train_reduce_workspace()) return fail(LCREC_EWORKSPACE, "grad_norm_clip: workspace too small");
    const int blocks = red_blocks(count);
    TraceScope trace(K_GRAD_NORM, stream);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(RED_THREADS), 0, stream, g, count, (double *)workspace);
    hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(RED_THREADS), 0, stream, (const double *)workspace, blocks, max_norm, norm_out);
    return check_launch("grad_norm kernels");
}

int adamw_step(float *p, float *g, float *m, float *v, int64_t count, const float *clip, int64_t *step, double base_lr,
               double beta1, double beta2, double eps, double weight_decay, int decoupled, int schedule, int64_t warmup_steps,
               int64_t total_steps, float *lr_out, hipStream_t stream)
{
    if (!p || !g || !m || !v || !step) return fail(LCREC_EINVAL, "adamw_step: NULL pointer");
    if (count < 1) return fail(LCREC_EINVAL, "adamw_step: empty parameter buffer");
    if (schedule < -1 || schedule > 1) return fail(LCREC_EINVAL, "adamw_step: schedule %d (supported: -1 none, 0 constant, 1 linear)", schedule);
    AdamParams a = {p, g, m, v, count, clip, step, base_lr, beta1, beta2, eps, weight_decay, decoupled, schedule, warmup_steps,
                    total_steps, lr_out};
    int64_t blocks = (count + 256 * 8 - 1) / (256 * 8);
    if (blocks > 2048) blocks = 2048;
    TraceScope trace(K_ADAMW, stream);
    hipLaunchKernelGGL(adamw_step_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, stream, step);
    return check_launch("adamw_step_kernel");
}

}  // namespace lcrec
