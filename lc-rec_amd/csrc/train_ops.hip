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
// reductions (BatchNorm statistics, bias gradients) are done by one 1024-thread workgroup per strip of 8-32 columns
// (see Strip below): deterministic, no atomics.  Batch-sized inputs only (a training batch is 1-8 k rows; the strip is
// re-read from L2 for the later passes).  First form (256 threads, one load in flight per lane): 64 us per call at
// 1024 x 2048 -- a third of the training step; 16 waves with 8 loads in flight each: see profiles/.
#include "common.h"

#include <stdlib.h>
#include <stdint.h>
#include <initializer_list>

namespace lcrec {

constexpr int CR_THREADS = 1024;  // 16 waves per strip: enough loads in flight to cover HBM/L2 latency from one CU
constexpr int CR_WAVES = CR_THREADS / 64;
#ifndef LCREC_CR_UNROLL
#define LCREC_CR_UNROLL 8
#endif
constexpr int CR_UNROLL = LCREC_CR_UNROLL;

// Strip geometry: COLS columns x RGS = 1024 / COLS row groups; lane l of a wave holds column l % COLS, so a wave spans
// 64 / COLS consecutive row groups.  Row group g owns rows g, g + RGS, ... (each lane adds its rows in ascending order,
// CR_UNROLL loads in flight); the groups of a wave are added by xor-shuffles (a fixed tree), the 16 waves through LDS in
// wave order -- the same bits on every run.
template <int COLS>
struct Strip {
    static constexpr int RGS = CR_THREADS / COLS;
    int c, rg, col;
    bool live;
    // Strips are narrower than a 128-byte line (32 floats), so neighbours share lines -- and consecutive workgroups go to different
    // XCDs (round-robin over 8), each with an L2 of its own that would fetch the shared line again.  When the grid is a multiple
    // of 8, workgroup b takes strip (b % 8) * (grid / 8) + b / 8: the strips of one XCD are neighbours (measured on the float4
    // form below: backward of 1024 x 2048 in 16-column strips 13.3 -> 8.0 us).
    static __device__ __forceinline__ int strip_of_block()
    {
        const int b = blockIdx.x, g = gridDim.x;
        return (COLS < 32 && (g & 7) == 0) ? (b & 7) * (g >> 3) + (b >> 3) : b;
    }
    __device__ Strip(int F) : c(threadIdx.x % COLS), rg(threadIdx.x / COLS), col(strip_of_block() * COLS + threadIdx.x % COLS), live(col < F) {}

    // f(row) -> value; returns the sum over this lane's rows in ascending row order
    template <typename Fn>
    __device__ __forceinline__ float rows(int64_t n, Fn f) const
    {
        float s = 0.f;
        int64_t r = rg;
        for (; r + (int64_t)(CR_UNROLL - 1) * RGS < n; r += (int64_t)CR_UNROLL * RGS) {
            float v[CR_UNROLL];
#pragma unroll
            for (int u = 0; u < CR_UNROLL; ++u) v[u] = f(r + (int64_t)u * RGS);
#pragma unroll
            for (int u = 0; u < CR_UNROLL; ++u) s += v[u];
        }
        for (; r < n; r += RGS) s += f(r);
        return s;
    }

    // the same walk with two running sums: f(row, a, b) adds the row's contribution to both
    template <typename Fn>
    __device__ __forceinline__ void rows2(int64_t n, float &s0, float &s1, Fn f) const
    {
        s0 = 0.f;
        s1 = 0.f;
        int64_t r = rg;
        for (; r + (int64_t)(CR_UNROLL - 1) * RGS < n; r += (int64_t)CR_UNROLL * RGS) {
            float v0[CR_UNROLL], v1[CR_UNROLL];
#pragma unroll
            for (int u = 0; u < CR_UNROLL; ++u) f(r + (int64_t)u * RGS, v0[u], v1[u]);
#pragma unroll
            for (int u = 0; u < CR_UNROLL; ++u) { s0 += v0[u]; s1 += v1[u]; }
        }
        for (; r < n; r += RGS) {
            float a, b;
            f(r, a, b);
            s0 += a;
            s1 += b;
        }
    }

    // sum of one value per lane over all row groups of the lane's column; valid in every lane
    __device__ __forceinline__ float sum(float v, float (*sm)[COLS]) const
    {
#pragma unroll
        for (int o = COLS; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();                       // previous use of sm is over
        if ((threadIdx.x & 63) < COLS) sm[threadIdx.x >> 6][c] = v;
        __syncthreads();
        float s = sm[0][c];
#pragma unroll
        for (int w = 1; w < CR_WAVES; ++w) s += sm[w][c];
        return s;
    }
};

// Training-mode BatchNorm1d (+ReLU).  torch semantics: batch mean, biased variance for the normalisation,
// running_mean/var updated with `momentum` (running_var from the unbiased variance), eps inside the square root.
// CACHED (n <= CR_MAXR rows per lane): the lane's rows stay in registers between the statistics pass and the apply pass --
// t is read from memory once.  Same sums in the same order as the two-pass form, hence the same bits.
constexpr int CR_MAXR = 32;

template <int COLS, bool CACHED>
__global__ __launch_bounds__(CR_THREADS) void bn_relu_forward_kernel(const float *__restrict__ t, int64_t n, int F,
                                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                      float eps, float momentum, float *running_mean,
                                                                      float *running_var, float *__restrict__ y, float *mean_out,
                                                                      float *rstd_out, int relu)
{
    __shared__ float sm[CR_WAVES][COLS];
    constexpr int RGS = Strip<COLS>::RGS;
    const Strip<COLS> st(F);
    const int col = st.live ? st.col : 0;          // dead lanes read column 0 and write nothing
    const float inv_n = 1.0f / (float)n;
    const float *tc = t + col;
    // one statistics pass: sums of (t - pivot) and (t - pivot)^2 with pivot = the column's first row, a sample of the
    // column -- so |mean - pivot| is of the order of the standard deviation and m2 = s2 - s1^2/n loses a bit or two, not
    // the digits the textbook E[t^2] - mean^2 loses when |mean| >> std
    const float pivot = tc[0];
    float tv[CACHED ? CR_MAXR : 1];
    float s1 = 0.f, s2 = 0.f;
    if (CACHED) {
#pragma unroll
        for (int u = 0; u < CR_MAXR; ++u) {
            const int64_t r = st.rg + (int64_t)u * RGS;
            tv[u] = r < n ? tc[r * F] : pivot;
        }
#pragma unroll
        for (int u = 0; u < CR_MAXR; ++u) {
            const float d = tv[u] - pivot;          // rows past n contribute exact zeros, which change no sum
            if (st.rg + (int64_t)u * RGS < n) { s1 += d; s2 += d * d; }
        }
    } else {
        st.rows2(n, s1, s2, [&](int64_t r, float &a, float &b) { const float d = tc[r * F] - pivot; a = d; b = d * d; });
    }
    s1 = st.sum(s1, sm);
    s2 = st.sum(s2, sm);
    const float dmean = s1 * inv_n;
    const float mean = pivot + dmean;
    float m2 = s2 - s1 * dmean;
    m2 = m2 > 0.f ? m2 : 0.f;
    const float var = m2 * inv_n;
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    if (!st.live) return;
    const float g = gamma ? gamma[col] : 1.0f, b = beta ? beta[col] : 0.0f;
    float *yc = y + col;
    if (CACHED) {
#pragma unroll
        for (int u = 0; u < CR_MAXR; ++u) {
            const int64_t r = st.rg + (int64_t)u * RGS;
            if (r < n) {
                float v = (tv[u] - mean) * rstd * g + b;
                if (relu) v = v > 0.f ? v : 0.f;
                yc[r * F] = v;
            }
        }
    } else {
#pragma unroll 4
        for (int64_t r = st.rg; r < n; r += RGS) {
            float v = (tc[r * F] - mean) * rstd * g + b;
            if (relu) v = v > 0.f ? v : 0.f;
            yc[r * F] = v;
        }
    }
    if (st.rg == 0) {
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
template <int COLS, bool CACHED>
__global__ __launch_bounds__(CR_THREADS) void bn_relu_backward_kernel(const float *gy, const float *__restrict__ t,
                                                                       const float *__restrict__ y, int64_t n, int F,
                                                                       const float *__restrict__ gamma, const float *__restrict__ mean,
                                                                       const float *__restrict__ rstd, int relu, float *dt,
                                                                       float *dgamma, float *dbeta, float *dbias,
                                                                       const float *__restrict__ fold_scale,
                                                                       const float *__restrict__ fold_shift)
{
    __shared__ float sm[CR_WAVES][COLS];
    constexpr int RGS = Strip<COLS>::RGS;
    const Strip<COLS> st(F);
    const int col = st.live ? st.col : 0;
    const float mu = mean[col], rs = rstd[col], gm = gamma ? gamma[col] : 1.0f;
    const float *gc = gy + col, *tc = t + col, *yc = (relu && y) ? y + col : nullptr;
    // the ReLU mask from the stored activation y, or -- when the forward never wrote one (lcrec_linear_bn_forward hands t to the
    // next layer, which applies max(t * scale + shift, 0) itself) -- from the same fused expression the consumer evaluated; or,
    // with fold_shift alone (= beta), from lcrec_bn_relu_forward's own expression (t - mean) * rstd * gamma + beta: the bits of
    // the y it wrote, without reading it
    const bool from_beta = relu && !y && !fold_scale;
    const float fs = (relu && !y && fold_scale) ? fold_scale[col] : 0.f, fh = (relu && !y) ? fold_shift[col] : 0.f;
    auto gval = [&](int64_t r) {
        float g = gc[r * F];
        if (relu) {
            const bool on = yc ? yc[r * F] > 0.f
                               : (from_beta ? (tc[r * F] - mu) * rs * gm + fh > 0.f : __builtin_fmaf(tc[r * F], fs, fh) > 0.f);
            if (!on) g = 0.f;
        }
        return g;
    };
    float gv[CACHED ? CR_MAXR : 1], xv[CACHED ? CR_MAXR : 1];     // CACHED: the lane's masked gradients and xhat, read once
    float db = 0.f, dg = 0.f;
    if (CACHED) {
#pragma unroll
        for (int u = 0; u < CR_MAXR; ++u) {
            const int64_t r = st.rg + (int64_t)u * RGS;
            const bool in = r < n;
            gv[u] = in ? gval(r) : 0.f;
            xv[u] = in ? (tc[r * F] - mu) * rs : 0.f;
        }
#pragma unroll
        for (int u = 0; u < CR_MAXR; ++u)
            if (st.rg + (int64_t)u * RGS < n) { db += gv[u]; dg += gv[u] * xv[u]; }
    } else {
        st.rows2(n, db, dg, [&](int64_t r, float &a, float &b) { const float g = gval(r); a = g; b = g * ((tc[r * F] - mu) * rs); });
    }
    db = st.sum(db, sm);
    dg = st.sum(dg, sm);
    const float inv_n = 1.0f / (float)n;
    const float k = gm * rs, mdb = db * inv_n, mdg = dg * inv_n;
    float *dc = dt + col;
    float sdt = 0.f;
    if (st.live) {
        if (CACHED) {
#pragma unroll
            for (int u = 0; u < CR_MAXR; ++u) {
                const int64_t r = st.rg + (int64_t)u * RGS;
                if (r < n) {
                    const float v = k * (gv[u] - mdb - xv[u] * mdg);
                    dc[r * F] = v;
                    sdt += v;
                }
            }
        } else {
#pragma unroll 4
            for (int64_t r = st.rg; r < n; r += RGS) {
                const float xh = (tc[r * F] - mu) * rs;
                const float v = k * (gval(r) - mdb - xh * mdg);
                dc[r * F] = v;
                sdt += v;
            }
        }
    }
    const float dbs = st.sum(sdt, sm);
    if (st.live && st.rg == 0) {
        if (dgamma) dgamma[col] = dg;
        if (dbeta) dbeta[col] = db;
        if (dbias) dbias[col] = dbs;
    }
}

// ---- the two kernels again for the shapes a training step has: F a multiple of 4, rows 16-byte aligned, and few enough rows
// that a lane's share of the strip stays in registers (n <= RMAX * RGS).  Lanes read float4s: a strip of COLS columns is COLS / 4
// lanes wide and RGS = 1024 / (COLS / 4) row groups deep, so a 1024 x 16 strip is FOUR 16-byte loads per lane and array, all in
// flight at once, where the dword form above issues 16 per lane -- on one CU the dword form is bound by the texture-address
// rate (13 us for a 1024 x 16 backward strip however few strips the launch has; r03 kernel trace by grid size).
// Row order of the sums (fixed, so the same bits on every run): a lane's rows ascending; the row groups of a 16-lane row by DPP
// rotations; the workgroup's 64 such rows as four chains of 16 in row order, ((c0 + c1) + (c2 + c3)) -- see sum() below.
template <int COLS>
struct Strip4 {
    static constexpr int L4 = COLS / 4, RGS = CR_THREADS / L4;
    int c4, rg, col;
    bool live;
    // the strip of a workgroup: neighbours on one XCD, as in Strip::strip_of_block
    static __device__ __forceinline__ int strip_of_block()
    {
        const int b = blockIdx.x, g = gridDim.x;
        return (COLS < 32 && (g & 7) == 0) ? (b & 7) * (g >> 3) + (b >> 3) : b;
    }
    __device__ Strip4(int F) : c4(threadIdx.x % L4), rg(threadIdx.x / L4), col(strip_of_block() * COLS + 4 * (threadIdx.x % L4)), live(col < F) {}

    // v + (v rotated right by N lanes within its row of 16 lanes): one v_add_f32 with a DPP operand, no LDS crossbar
    template <int N>
    static __device__ __forceinline__ float add_ror(float v)
    {
        return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
    }

    // sums over all row groups of NV four-column values per lane; valid in every lane afterwards.
    //  1. inside a row of 16 lanes (16 / L4 row groups): DPP rotations, so lane c4 of the row ends with the row's sum of its
    //     columns (each lane adds in its own rotated order; only lanes c4 < L4 of each row are used, a fixed order);
    //  2. the 64 row partials of the workgroup (16 waves x 4 rows) through LDS, added by ONE lane per (value, column) as four
    //     chains of 16 in row order -- every lane reading all partials of its four columns would be 8x the LDS traffic;
    //  3. the totals back through LDS.
    template <int NV>
    __device__ __forceinline__ void sum(float (&v)[NV][4], float *sm) const
    {
        constexpr int PR = CR_WAVES * 4;                                          // partial rows
        float(*part)[NV * COLS] = reinterpret_cast<float(*)[NV * COLS]>(sm);     // [PR][NV * COLS]
        float *tot = sm + PR * NV * COLS;                                        // [NV * COLS]
#pragma unroll
        for (int q = 0; q < NV; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (L4 <= 1) v[q][e] = add_ror<1>(v[q][e]);
                if (L4 <= 2) v[q][e] = add_ror<2>(v[q][e]);
                if (L4 <= 4) v[q][e] = add_ror<4>(v[q][e]);
                if (L4 <= 8) v[q][e] = add_ror<8>(v[q][e]);
            }
        __syncthreads();                       // previous use of sm is over
        if ((threadIdx.x & 15) < L4)
#pragma unroll
            for (int q = 0; q < NV; ++q)
                *reinterpret_cast<float4 *>(&part[threadIdx.x >> 4][q * COLS + 4 * c4]) = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
        __syncthreads();
        if (threadIdx.x < NV * COLS) {
            float a[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) a[h] = part[h * (PR / 4)][threadIdx.x];
#pragma unroll 3                              // 12 reads in flight: fully unrolled, the 64 of them crowd out the cached rows
            for (int w = 1; w < PR / 4; ++w)
#pragma unroll
                for (int h = 0; h < 4; ++h) a[h] += part[h * (PR / 4) + w][threadIdx.x];
            tot[threadIdx.x] = (a[0] + a[1]) + (a[2] + a[3]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const float4 a = *reinterpret_cast<const float4 *>(&tot[q * COLS + 4 * c4]);
            v[q][0] = a.x; v[q][1] = a.y; v[q][2] = a.z; v[q][3] = a.w;
        }
    }
    static constexpr int SM_FLOATS = (CR_WAVES * 4 + 1) * 2 * COLS;      // for NV <= 2
};

__device__ __forceinline__ void ld4(float (&d)[4], const float *p)
{
    const float4 q = *reinterpret_cast<const float4 *>(p);
    d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
}
__device__ __forceinline__ void st4(float *p, const float (&d)[4]) { *reinterpret_cast<float4 *>(p) = make_float4(d[0], d[1], d[2], d[3]); }

// STATS: only the local statistics of a rank's rows (lcrec_bn_stats: mean -> mean_out, M2 -> rstd_out), nothing applied
template <int COLS, int RMAX, bool STATS = false>
__global__ __launch_bounds__(CR_THREADS) void bn_relu_forward_v4_kernel(const float *__restrict__ t, int64_t n, int F,
                                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                         float eps, float momentum, float *running_mean,
                                                                         float *running_var, float *__restrict__ y, float *mean_out,
                                                                         float *rstd_out, int relu)
{
    using S = Strip4<COLS>;
    __shared__ __attribute__((aligned(16))) float sm[S::SM_FLOATS];
    const S st(F);
    const int col = st.live ? st.col : 0;          // dead lanes read columns 0..3 and write nothing
    const float inv_n = 1.0f / (float)n;
    const float *tc = t + col;
    float pivot[4];                                // the columns' first row, as in the dword form
    ld4(pivot, tc);
    float tv[RMAX][4];
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int r = st.rg + u * S::RGS;         // 32-bit offsets from the uniform base (n * F < 2^29)
        ld4(tv[u], t + ((unsigned)(r < (int)n ? r : 0) * (unsigned)F + (unsigned)col));      // no branch between the loads; rows past n read the pivot row: exact zeros below
    }
    float s[2][4] = {};
#pragma unroll
    for (int u = 0; u < RMAX; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = tv[u][e] - pivot[e];
            s[0][e] += d;
            s[1][e] += d * d;
        }
    float g[4] = {1.f, 1.f, 1.f, 1.f}, b[4] = {};          // in flight with the rows
    if (gamma) ld4(g, gamma + col);
    if (beta) ld4(b, beta + col);
    st.template sum<2>(s, sm);
    if (!st.live) return;
    float mean[4], rstd[4], m2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float dmean = STATS ? s[0][e] / (float)n : s[0][e] * inv_n;        // (bn_stats_kernel divides; keep its expression)
        mean[e] = pivot[e] + dmean;
        const float m = s[1][e] - s[0][e] * dmean;
        m2[e] = m > 0.f ? m : 0.f;
        rstd[e] = 1.0f / __builtin_sqrtf(m2[e] * inv_n + eps);
    }
    if (STATS) {
        if (st.rg == 0) { st4(mean_out + col, mean); st4(rstd_out + col, m2); }
        return;
    }
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int r = st.rg + u * S::RGS;
        if (r < (int)n) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (tv[u][e] - mean[e]) * rstd[e] * g[e] + b[e];
                if (relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            st4(y + ((unsigned)r * (unsigned)F + (unsigned)col), v);
        }
    }
    if (st.rg == 0) {
        st4(mean_out + col, mean);
        st4(rstd_out + col, rstd);
        if (running_mean) {
            float rm[4];
            ld4(rm, running_mean + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) rm[e] = (1.0f - momentum) * rm[e] + momentum * mean[e];
            st4(running_mean + col, rm);
        }
        if (running_var) {
            float rv[4];
            ld4(rv, running_var + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float unbiased = n > 1 ? m2[e] / (float)(n - 1) : m2[e] * inv_n;
                rv[e] = (1.0f - momentum) * rv[e] + momentum * unbiased;
            }
            st4(running_var + col, rv);
        }
    }
}

struct Bn4Bwd {
    const float *gy, *t, *y;
    int64_t n;
    int F;
    const float *gamma, *mean, *rstd;
    int relu;
    float *dt, *dgamma, *dbeta, *dbias;
    const float *fold_scale, *fold_shift;
    float *sum_g, *sum_gx;      // MODE 1: out, this rank's column sums of g and g * xhat; MODE 2: in, the sums over all ranks
    float n_total;              // MODE 2: rows of the global batch
};
// MODE 0: the whole backward of one process.  The data-parallel split (SyncBatchNorm semantics, lcrec_amd/layers.py all-reduces
// between the halves): MODE 1 = lcrec_bn_backward_reduce (the two column sums of this rank's rows, nothing written to dt),
// MODE 2 = lcrec_bn_backward_apply (dt and dbias from the global sums).
template <int COLS, int RMAX, int MODE = 0>
__global__ __launch_bounds__(CR_THREADS) void bn_relu_backward_v4_kernel(Bn4Bwd p)
{
    using S = Strip4<COLS>;
    __shared__ __attribute__((aligned(16))) float sm[S::SM_FLOATS];
    const float *gy = p.gy, *__restrict__ t = p.t, *__restrict__ y = p.y;
    const int64_t n = p.n;
    const int F = p.F, relu = p.relu;
    const S st(F);
    const int col = st.live ? st.col : 0;
    float mu[4], rs[4], gm[4] = {1.f, 1.f, 1.f, 1.f}, fs[4] = {}, fh[4] = {};
    ld4(mu, p.mean + col);
    ld4(rs, p.rstd + col);
    if (MODE != 1 && p.gamma) ld4(gm, p.gamma + col);
    // the ReLU mask: the stored activation; or the consumer's fused expression fma(t, fold_scale, fold_shift); or, with fold_shift
    // alone (= beta), the forward's own expression -- the bits of the y it wrote, one array less to read (in the step: 2 us a call)
    const bool from_y = relu && y, from_fold = relu && !y && p.fold_scale, from_beta = relu && !y && !p.fold_scale;
    if (from_fold) ld4(fs, p.fold_scale + col);
    if (from_fold || from_beta) ld4(fh, p.fold_shift + col);
    float gv[RMAX][4], xv[RMAX][4];               // the lane's masked gradients and xhat, read once
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        // Four rows per lane: every load issued unconditionally (rows past n read row 0 and are zeroed) -- no branch, nothing
        // waits between the 12 loads.  Eight rows per lane: a branch per row keeps the compiler from hoisting all 24 loads, which
        // do not fit the 128 registers a 1024-thread workgroup leaves each lane (measured with the spills: 8.8 -> 13.8 us at
        // 1024 x 2048 in 32-column strips).
        const int r = st.rg + u * S::RGS;
        const bool in = r < (int)n;
        const unsigned off = (unsigned)(in ? r : 0) * (unsigned)F + (unsigned)col;      // one 32-bit offset for the three arrays (n * F < 2^29)
        if (RMAX <= 4 || in) {
            float tt[4], yy[4] = {1.f, 1.f, 1.f, 1.f};
            ld4(gv[u], gy + off);
            ld4(tt, t + off);
            if (from_y) ld4(yy, y + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (tt[e] - mu[e]) * rs[e];
                const bool on = from_fold ? __builtin_fmaf(tt[e], fs[e], fh[e]) > 0.f : (from_beta ? xh * gm[e] + fh[e] > 0.f : yy[e] > 0.f);
                gv[u][e] = (in && on) ? gv[u][e] : 0.f;
                xv[u][e] = in ? xh : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { gv[u][e] = 0.f; xv[u][e] = 0.f; }
        }
    }
    float s[2][4] = {};
#pragma unroll
    for (int u = 0; u < RMAX; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[0][e] += gv[u][e];
            s[1][e] += gv[u][e] * xv[u][e];
        }
    if (MODE == 2) {
        ld4(s[0], p.sum_g + col);
        ld4(s[1], p.sum_gx + col);
    } else {
        st.template sum<2>(s, sm);
    }
    if (MODE == 1) {
        if (st.live && st.rg == 0) {
            st4(p.sum_g + col, s[0]);
            st4(p.sum_gx + col, s[1]);
            if (p.dbeta) st4(p.dbeta + col, s[0]);           // this rank's share of the parameter gradients
            if (p.dgamma) st4(p.dgamma + col, s[1]);
        }
        return;
    }
    float *dt = p.dt, *dgamma = MODE == 0 ? p.dgamma : nullptr, *dbeta = MODE == 0 ? p.dbeta : nullptr, *dbias = p.dbias;
    const float inv_n = 1.0f / (MODE == 2 ? p.n_total : (float)n);
    float k[4], mdb[4], mdg[4], sdt[1][4] = {};
#pragma unroll
    for (int e = 0; e < 4; ++e) { k[e] = gm[e] * rs[e]; mdb[e] = s[0][e] * inv_n; mdg[e] = s[1][e] * inv_n; }
    if (st.live) {
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int r = st.rg + u * S::RGS;
            if (r < (int)n) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = k[e] * (gv[u][e] - mdb[e] - xv[u][e] * mdg[e]);
                    sdt[0][e] += v[e];
                }
                st4(dt + ((unsigned)r * (unsigned)F + (unsigned)col), v);
            }
        }
    }
    st.template sum<1>(sdt, sm);
    if (st.live && st.rg == 0) {
        if (dgamma) st4(dgamma + col, s[1]);
        if (dbeta) st4(dbeta + col, s[0]);
        if (dbias) st4(dbias + col, sdt[0]);
    }
}

// ---- the same, split at the points where a data-parallel run exchanges statistics (SyncBatchNorm semantics: the
// batch is the union of the ranks' rows; lcrec_amd/layers.py all-reduces between the halves)
//
// local statistics of a rank's rows: mean and M2 = sum (t - mean)^2, to be merged over ranks (Chan et al.)
template <int COLS>
__global__ __launch_bounds__(CR_THREADS) void bn_stats_kernel(const float *__restrict__ t, int64_t n, int F, float *mean_out, float *m2_out)
{
    __shared__ float sm[CR_WAVES][COLS];
    const Strip<COLS> st(F);
    const int col = st.live ? st.col : 0;
    const float *tc = t + col;
    const float pivot = tc[0];
    float s1, s2;
    st.rows2(n, s1, s2, [&](int64_t r, float &a, float &b) { const float d = tc[r * F] - pivot; a = d; b = d * d; });
    s1 = st.sum(s1, sm);
    s2 = st.sum(s2, sm);
    const float dmean = s1 / (float)n;
    float m2 = s2 - s1 * dmean;
    m2 = m2 > 0.f ? m2 : 0.f;
    if (st.live && st.rg == 0) { mean_out[col] = pivot + dmean; m2_out[col] = m2; }
}

// Statistics of the union of the ranks' rows from their (n_r, mean_r[F], M2_r[F]) rows (Chan et al., merged in rank order so
// every rank computes the same bits): mean = sum n_r mean_r / N, M2 = sum (M2_r + n_r (mean_r - mean)^2); rstd and the
// running statistics as nn.BatchNorm1d keeps them (biased variance normalises, unbiased goes into running_var).
__global__ __launch_bounds__(256) void bn_merge_stats_kernel(const float *__restrict__ rows, int world, int F, float eps, float momentum,
                                                             float *mean_out, float *rstd_out, float *running_mean, float *running_var)
{
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= F) return;
    const int64_t stride = 2 * (int64_t)F + 1;
    float total = 0.f, acc = 0.f;
    for (int r = 0; r < world; ++r) {
        const float nr = rows[r * stride];
        total += nr;
        acc += nr * rows[r * stride + 1 + col];
    }
    const float mean = acc / total;
    float m2 = 0.f;
    for (int r = 0; r < world; ++r) {
        const float nr = rows[r * stride];
        const float d = rows[r * stride + 1 + col] - mean;
        m2 += rows[r * stride + 1 + F + col] + nr * (d * d);
    }
    mean_out[col] = mean;
    rstd_out[col] = 1.0f / __builtin_sqrtf(m2 / total + eps);
    if (running_mean) running_mean[col] = (1.0f - momentum) * running_mean[col] + momentum * mean;
    if (running_var) running_var[col] = (1.0f - momentum) * running_var[col] + momentum * (m2 / (total > 1.f ? total - 1.f : 1.f));
}

// y = [relu]((t - mean) * rstd * gamma + beta) with given (global) statistics
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float *__restrict__ t, int64_t n, int F, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, int relu, float *__restrict__ y)
{
    const int64_t total = n * F;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % F);
        float v = (t[i] - mean[col]) * rstd[col] * (gamma ? gamma[col] : 1.0f) + (beta ? beta[col] : 0.0f);
        if (relu) v = v > 0.f ? v : 0.f;
        y[i] = v;
    }
}

// local sums of the backward: sum g and sum g * xhat over this rank's rows (g = gy * [y > 0], xhat from the GLOBAL statistics)
template <int COLS>
__global__ __launch_bounds__(CR_THREADS) void bn_backward_reduce_kernel(const float *__restrict__ gy, const float *__restrict__ t,
                                                                        const float *__restrict__ y, int64_t n, int F,
                                                                        const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                        int relu, float *sum_g, float *sum_gx, float *dbeta, float *dgamma)
{
    __shared__ float sm[CR_WAVES][COLS];
    const Strip<COLS> st(F);
    const int col = st.live ? st.col : 0;
    const float mu = mean[col], rs = rstd[col];
    const float *gc = gy + col, *tc = t + col, *yc = relu ? y + col : nullptr;
    float db, dg;
    st.rows2(n, db, dg, [&](int64_t r, float &a, float &b) {
        float g = gc[r * F];
        if (relu && !(yc[r * F] > 0.f)) g = 0.f;
        a = g;
        b = g * ((tc[r * F] - mu) * rs);
    });
    db = st.sum(db, sm);
    dg = st.sum(dg, sm);
    if (st.live && st.rg == 0) {
        sum_g[col] = db; sum_gx[col] = dg;
        if (dbeta) dbeta[col] = db;          // this rank's share of the parameter gradients (the gradient all-reduce sums them)
        if (dgamma) dgamma[col] = dg;
    }
}

// dt = gamma * rstd * (g - sum_g / n_total - xhat * sum_gx / n_total) with the GLOBAL sums; dbias = local column sums of dt
template <int COLS>
__global__ __launch_bounds__(CR_THREADS) void bn_backward_apply_kernel(const float *gy, const float *__restrict__ t,
                                                                       const float *__restrict__ y, int64_t n, int F,
                                                                       const float *__restrict__ gamma, const float *__restrict__ mean,
                                                                       const float *__restrict__ rstd, int relu,
                                                                       const float *__restrict__ sum_g, const float *__restrict__ sum_gx,
                                                                       float n_total, float *dt, float *dbias)
{
    __shared__ float sm[CR_WAVES][COLS];
    const Strip<COLS> st(F);
    const int col = st.live ? st.col : 0;
    const float mu = mean[col], rs = rstd[col], gm = gamma ? gamma[col] : 1.0f;
    const float *gc = gy + col, *tc = t + col, *yc = relu ? y + col : nullptr;
    const float inv_n = 1.0f / n_total;
    const float k = gm * rs, mdb = sum_g[col] * inv_n, mdg = sum_gx[col] * inv_n;
    float *dc = dt + col;
    float sdt = 0.f;
    if (st.live) {
#pragma unroll 4
        for (int64_t r = st.rg; r < n; r += Strip<COLS>::RGS) {
            float g = gc[r * F];
            if (relu && !(yc[r * F] > 0.f)) g = 0.f;
            const float xh = (tc[r * F] - mu) * rs;
            const float v = k * (g - mdb - xh * mdg);
            dc[r * F] = v;
            sdt += v;
        }
    }
    const float dbs = st.sum(sdt, sm);
    if (st.live && st.rg == 0 && dbias) dbias[col] = dbs;
}

// g = gy * [y > 0] (in place allowed), dbias = column sums of g
template <int COLS>
__global__ __launch_bounds__(CR_THREADS) void relu_bias_backward_kernel(const float *gy, const float *__restrict__ y, int64_t n, int F,
                                                                         int relu, float *g_out, float *dbias)
{
    __shared__ float sm[CR_WAVES][COLS];
    const Strip<COLS> st(F);
    const int col = st.live ? st.col : 0;
    const float *gc = gy + col, *yc = relu ? y + col : nullptr;
    float *oc = (g_out && st.live) ? g_out + col : nullptr;
    const float s = st.rows(n, [&](int64_t r) {
        float g = gc[r * F];
        if (relu && !(yc[r * F] > 0.f)) g = 0.f;
        if (oc) oc[r * F] = g;
        return g;
    });
    const float db = st.sum(s, sm);
    if (st.live && st.rg == 0 && dbias) dbias[col] = db;
}

// ---- whole-tensor reductions: per-workgroup fp64 partials, then one finishing workgroup (fixed order)
// 1024-thread workgroups, at most 256 of them: as many loads in flight as 1024 x 256 threads, a quarter of the partials -- and
// of the ticket adds, which serialise on their one address (measured: 1024 arrivals cost the gradient-norm launch 8 us)
constexpr int RED_THREADS = 1024;
constexpr int RED_MAX_BLOCKS = TICKET_MAX_WORKGROUPS;

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

// Sum of `blocks` per-workgroup partials in the finishing kernels' order (thread i adds partials i, i + 256, ...; then the
// workgroup tree) -- shared by the second-launch form and by the last-workgroup form, so both give the same bits.
template <bool HANDOFF>
__device__ __forceinline__ double sum_partials_ordered(const double *partial, int blocks, double *sm)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < blocks; i += RED_THREADS) acc += HANDOFF ? handoff_get(partial + i) : partial[i];
    return block_sum(acc, sm);            // valid in thread 0
}

// `ticket` form of a partial-sum kernel's tail: thread 0 publishes the workgroup's partial and takes a ticket; returns true
// (in every thread) in the workgroup that arrived last.
__device__ __forceinline__ bool publish_and_check_last(double *partial, double s, unsigned *ticket, int *last_sh)
{
    if (threadIdx.x == 0) {
        handoff_put(partial + blockIdx.x, s);
        *last_sh = ticket_is_last(ticket, gridDim.x) ? 1 : 0;
    }
    __syncthreads();
    return *last_sh != 0;
}

// mse: loss = mean (out - x)^2, g = 2 (out - x) / count;  l1: loss = mean |out - x|, g = sign(out - x) / count
__global__ __launch_bounds__(RED_THREADS) void recon_loss_grad_kernel(const float *__restrict__ out, const float *__restrict__ x,
                                                                       int64_t count, int64_t count_total, int l1,
                                                                       float *__restrict__ g, double *partial, unsigned *ticket,
                                                                       float *loss)
{
    __shared__ double sm[RED_THREADS / 64];
    __shared__ int last_sh;
    const float scale = (l1 ? 1.0f : 2.0f) / (float)count_total;
    double acc = 0.0;
    auto one = [&](float o, float t) {
        const float d = o - t;
        if (l1) { acc += (double)__builtin_fabsf(d); return d > 0.f ? scale : (d < 0.f ? -scale : 0.f); }
        acc += (double)d * (double)d;
        return d * scale;
    };
    // 16-byte accesses when the three arrays allow it (a lane's four elements in index order, so the workgroup's sum takes the
    // elements in a fixed order either way); the ragged end and unaligned callers by scalars
    const bool vec = (((uintptr_t)out | (uintptr_t)x | (uintptr_t)g) & 15) == 0;
    const int64_t quads = vec ? count / 4 : 0;
    for (int64_t q = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; q < quads; q += (int64_t)gridDim.x * RED_THREADS) {
        const float4 o = reinterpret_cast<const float4 *>(out)[q], t = reinterpret_cast<const float4 *>(x)[q];
        float4 r;
        r.x = one(o.x, t.x); r.y = one(o.y, t.y); r.z = one(o.z, t.z); r.w = one(o.w, t.w);
        if (g) reinterpret_cast<float4 *>(g)[q] = r;
    }
    for (int64_t i = quads * 4 + (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < count; i += (int64_t)gridDim.x * RED_THREADS) {
        const float r = one(out[i], x[i]);
        if (g) g[i] = r;
    }
    const double s = block_sum(acc, sm);
    if (!ticket) {
        if (threadIdx.x == 0) partial[blockIdx.x] = s;
        return;
    }
    if (!publish_and_check_last(partial, s, ticket, &last_sh)) return;
    const double total = sum_partials_ordered<true>(partial, gridDim.x, sm);
    if (threadIdx.x == 0) *loss = (float)(total / (double)count_total);
}

__global__ __launch_bounds__(RED_THREADS) void recon_loss_finish_kernel(const double *partial, int blocks, int64_t count, float *loss)
{
    __shared__ double sm[RED_THREADS / 64];
    const double s = sum_partials_ordered<false>(partial, blocks, sm);
    if (threadIdx.x == 0) *loss = (float)(s / (double)count);
}

__device__ __forceinline__ void norm_and_coef(double sumsq, float max_norm, float *norm_out)
{
    const float norm = (float)__builtin_sqrt(sumsq);
    float coef = max_norm / (norm + 1e-6f);
    coef = coef > 1.0f ? 1.0f : coef;
    norm_out[0] = norm;
    norm_out[1] = coef;
}

// 16-byte loads (the flat gradient buffer is 256-byte aligned; a ragged tail is read by scalars): 35 MB at 768-d in ~7 us
__global__ __launch_bounds__(RED_THREADS) void sumsq_kernel(const float *__restrict__ g, int64_t count, double *partial, unsigned *ticket,
                                                             float max_norm, float *norm_out)
{
    __shared__ double sm[RED_THREADS / 64];
    __shared__ int last_sh;
    double acc = 0.0;
    const int64_t count4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? count / 4 : 0;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < count4; i += (int64_t)gridDim.x * RED_THREADS) {
        const f32x4 v = g4[i];
        acc += (double)v[0] * (double)v[0];
        acc += (double)v[1] * (double)v[1];
        acc += (double)v[2] * (double)v[2];
        acc += (double)v[3] * (double)v[3];
    }
    for (int64_t i = count4 * 4 + (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < count; i += (int64_t)gridDim.x * RED_THREADS) {
        const double v = (double)g[i];
        acc += v * v;
    }
    const double s = block_sum(acc, sm);
    if (!ticket) {
        if (threadIdx.x == 0) partial[blockIdx.x] = s;
        return;
    }
    if (!publish_and_check_last(partial, s, ticket, &last_sh)) return;
    const double total = sum_partials_ordered<true>(partial, gridDim.x, sm);
    if (threadIdx.x == 0) norm_and_coef(total, max_norm, norm_out);
}

// norm_out[0] = ||g||_2, norm_out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(RED_THREADS) void grad_norm_finish_kernel(const double *partial, int blocks, float max_norm, float *norm_out)
{
    __shared__ double sm[RED_THREADS / 64];
    const double s = sum_partials_ordered<false>(partial, blocks, sm);
    if (threadIdx.x == 0) norm_and_coef(s, max_norm, norm_out);
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
    unsigned *ticket;        // with it the LAST workgroup to finish advances *step (no second launch)
    const unsigned char *skip;   // device flag (a sticky "loss was NaN"): when set, nothing is updated and *step stays
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
constexpr int ADAM_THREADS = 1024;
__global__ __launch_bounds__(ADAM_THREADS) void adamw_step_kernel(AdamParams a)
{
    const bool skip = a.skip && *a.skip;              // uniform over the launch
    const int64_t s = *a.step;                       // steps taken before this one
    const double lr = lr_at(a, s);
    const double t = (double)(s + 1);
    const double bc1 = 1.0 - pow(a.beta1, t);
    const double bc2_sqrt = sqrt(1.0 - pow(a.beta2, t));
    const float coef = a.clip ? a.clip[1] : 1.0f;
    const float step_size = (float)(lr / bc1);
    const float b2 = (float)a.beta2, one_m_b1 = (float)(1.0 - a.beta1), one_m_b2 = (float)(1.0 - a.beta2);
    const float epsf = (float)a.eps, bc2s = (float)bc2_sqrt;
    const float decay = (float)(lr * a.weight_decay), wd = (float)a.weight_decay;
    if (!skip) {
        // one element: returns the clipped gradient (what clip_grad_norm_ leaves in .grad), updates p, m, v in place
        auto one = [&](float &p, float g0, float &m, float &v) {
            const float gc = g0 * coef;
            float g = gc;
            if (a.weight_decay != 0.0) {
                if (a.decoupled) p -= decay * p;
                else g += p * wd;
            }
            m = m + one_m_b1 * (g - m);                  // lerp(m, g, 1 - beta1)
            v = b2 * v + one_m_b2 * g * g;
            const float denom = __builtin_sqrtf(v) / bc2s + epsf;
            p -= step_size * m / denom;
            return gc;
        };
        // 16-byte accesses (the four flat buffers are 256-byte aligned; a caller's unaligned or ragged end goes by scalars): a
        // quarter of the memory instructions of the dword form -- 8.9 M parameters x 8 streams were 1.1 M wave-level
        // instructions on the address path.  The clipped gradient is written back only when clipping changed it (coef != 1).
        const bool vec = (((uintptr_t)a.p | (uintptr_t)a.g | (uintptr_t)a.m | (uintptr_t)a.v) & 15) == 0;
        const bool write_g = coef != 1.0f;
        const int64_t quads = vec ? a.count / 4 : 0;
        for (int64_t q = (int64_t)blockIdx.x * ADAM_THREADS + threadIdx.x; q < quads; q += (int64_t)gridDim.x * ADAM_THREADS) {
            float4 p = reinterpret_cast<float4 *>(a.p)[q], m = reinterpret_cast<float4 *>(a.m)[q], v = reinterpret_cast<float4 *>(a.v)[q];
            const float4 g = reinterpret_cast<const float4 *>(a.g)[q];
            float4 gc;
            gc.x = one(p.x, g.x, m.x, v.x); gc.y = one(p.y, g.y, m.y, v.y); gc.z = one(p.z, g.z, m.z, v.z); gc.w = one(p.w, g.w, m.w, v.w);
            if (write_g) reinterpret_cast<float4 *>(a.g)[q] = gc;
            reinterpret_cast<float4 *>(a.p)[q] = p; reinterpret_cast<float4 *>(a.m)[q] = m; reinterpret_cast<float4 *>(a.v)[q] = v;
        }
        for (int64_t i = quads * 4 + (int64_t)blockIdx.x * ADAM_THREADS + threadIdx.x; i < a.count; i += (int64_t)gridDim.x * ADAM_THREADS) {
            float p = a.p[i], m = a.m[i], v = a.v[i];
            const float gc = one(p, a.g[i], m, v);
            if (write_g) a.g[i] = gc;
            a.p[i] = p; a.m[i] = m; a.v[i] = v;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.lr_out) *a.lr_out = (float)lr;
    }
    if (a.ticket) {
        // every workgroup read *step on entry; the one that finishes last -- after all the others have taken their tickets,
        // i.e. are past that read -- advances it
        __syncthreads();
        if (threadIdx.x == 0 && ticket_is_last(a.ticket, gridDim.x) && !skip) *a.step = s + 1;
    }
}

__global__ void step_advance_kernel(int64_t *step, const unsigned char *skip) { if (!(skip && *skip)) *step += 1; }

// The scalar tail of a step, one thread: level losses and their mean (vq.py:90-92, rq.py:53), total loss (rqvae.py:83),
// the trainer's running sums (trainer.py:122-123) and its NaN check (trainer.py:116) as a sticky device flag.
__global__ void step_losses_kernel(const double *sse, int L, double count, float beta, float qlw, const float *recon,
                                   float *out3, double *sums2, unsigned char *nan_flag, const int64_t *probe,
                                   unsigned char *probe_flag)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (probe && probe_flag && *probe < 0) *probe_flag = 1;     // a poisoned (-1) Sinkhorn assignment: the solver gave up
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
        const float mse = (float)(sse[l] / count);
        acc += mse + beta * mse;
    }
    const float rq_loss = acc / (float)L;
    const float rec = *recon;
    const float loss = rec + qlw * rq_loss;
    out3[0] = loss; out3[1] = rec; out3[2] = rq_loss;
    if (sums2) { sums2[0] += (double)loss; sums2[1] += (double)rec; }
    if (nan_flag && loss != loss) *nan_flag = 1;
}

// d loss / d z of the quantiser (quantize.py): (coef * (z - C0[idx0])) * weight + g_xq, coef = beta * 2/(L n e)
__global__ __launch_bounds__(256) void quantizer_input_grad_kernel(const float *__restrict__ z, const float *__restrict__ cb0,
                                                                   const int64_t *__restrict__ idx, int64_t idx_stride, int64_t n,
                                                                   int e, float coef, float weight, const float *__restrict__ g_xq,
                                                                   float *__restrict__ out)
{
    const int64_t total = n * e;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / e;
        const int k = (int)(i - row * e);
        const float t = z[i] - cb0[idx[row * idx_stride] * e + k];
        const float u = coef * t;
        const float v = u * weight;
        out[i] = v + g_xq[i];
    }
}

// The same with the column sums of `out` beside it (dbias of the encoder's last Linear, which has neither BatchNorm nor an
// activation behind it: its bias gradient is the column sum of the gradient reaching z): ONE workgroup, thread (row group, column),
// rows ascending per lane, row groups added in order through LDS.  For batch-sized n (n * e <= 64 K elements).
constexpr int QGB_THREADS = 1024;
__global__ __launch_bounds__(QGB_THREADS) void quantizer_input_grad_bias_kernel(const float *__restrict__ z, const float *__restrict__ cb0,
                                                                                const int64_t *__restrict__ idx, int64_t idx_stride, int n,
                                                                                int e, float coef, float weight,
                                                                                const float *__restrict__ g_xq, float *__restrict__ out,
                                                                                float *__restrict__ dbias)
{
    __shared__ float part[QGB_THREADS];
    const int c = threadIdx.x % e, rg = threadIdx.x / e, rgs = QGB_THREADS / e;      // e divides 1024 (16, 32, 64)
    float acc = 0.f;
    for (int r0 = rg; r0 < n; r0 += 4 * rgs) {                // four rows' loads in flight
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * rgs;
            const int rr = r < n ? r : rg;                    // (a valid row: rg < n whenever this loop runs)
            const int64_t i = (int64_t)rr * e + c;
            const float t = z[i] - cb0[idx[(int64_t)rr * idx_stride] * e + c];
            const float uu = coef * t;
            const float w = uu * weight;
            v[u] = w + g_xq[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * rgs;
            if (r < n) {
                out[(int64_t)r * e + c] = v[u];
                acc += v[u];
            }
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < e) {
        float s = part[threadIdx.x];
        for (int g = 1; g < rgs; ++g) s += part[g * e + threadIdx.x];
        dbias[threadIdx.x] = s;
    }
}

// dL/dC[k][:] = (scale * (count[k] * C[k][:] - sum[k][:])) * weight -- the closed form autograd derives from vq.py:90-92
// (SURVEY.md a9), in the order quantize.py evaluates it
__global__ __launch_bounds__(256) void codebook_grad_kernel(const float *__restrict__ count, const float *__restrict__ sum,
                                                            const float *__restrict__ cb, int K, int e, float scale, float weight,
                                                            float *__restrict__ grad)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K * e) return;
    const float t = count[i / e] * cb[i] - sum[i];
    grad[i] = (scale * t) * weight;
}

// ---------------------------------------------------------------- host side

// strip width by feature count: wide layers read 128-byte row segments; narrow ones take narrower strips so that more
// than a handful of CUs work (their whole input is a few hundred KB)
static int strip_cols(int F)
{
    static const int forced = [] { const char *e = getenv("LCREC_STRIP_COLS"); return e ? atoi(e) : 0; }();   // tuning only
    if (forced == 8 || forced == 16 || forced == 32) return forced;
    if (forced == -1) return F >= 4096 ? 32 : (F >= 2048 ? 16 : 8);
    // measured again after the step had become GPU-bound (batch 1024, ms per step at 768-d / 4096-d): this table's
    // predecessor (32 from 1024 columns, 16 from 256, else 8) 1.259 / 1.848, 8 everywhere 1.293 / 1.920, 32 everywhere
    // 1.354 / 1.929, 16 everywhere 1.241 / 1.839
    (void)F;
    return 16;
}
#define LCREC_STRIP_LAUNCH(KERN, F, stream, ...)                                                                      \
    do {                                                                                                              \
        const int cols_ = strip_cols(F);                                                                              \
        const dim3 grid_((unsigned)(((F) + cols_ - 1) / cols_));                                                      \
        if (cols_ == 32) hipLaunchKernelGGL(KERN<32>, grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__);               \
        else if (cols_ == 16) hipLaunchKernelGGL(KERN<16>, grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__);          \
        else hipLaunchKernelGGL(KERN<8>, grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__);                            \
    } while (0)

// the same for the kernels with a register-cached form: narrower strips when that lets the lane's rows fit (n <= 32 rows per
// lane: 1024 rows at 32 columns, 2048 at 16, 4096 at 8), the cached form whenever they do
#define LCREC_STRIP_LAUNCH_N(KERN, F, n, stream, ...)                                                                  \
    do {                                                                                                              \
        int cols_ = strip_cols(F);                                                                                    \
        while (cols_ > 8 && (int64_t)(n) > (int64_t)CR_MAXR * (CR_THREADS / cols_)) cols_ /= 2;                       \
        /* measured (768-d recipe): at 1024 rows the second pass hits L2 anyway and the cached form is ~1 % slower; at */ \
        /* 2048 rows the narrower strips + cached rows are ~3 % of the step faster                                    */ \
        static const int cached_min_ = [] { const char *e = getenv("LCREC_BN_CACHED_MIN"); return e ? atoi(e) : 1025; }();   /* tuning */ \
        const bool cached_ = (int64_t)(n) >= cached_min_ && (int64_t)(n) <= (int64_t)CR_MAXR * (CR_THREADS / cols_);  \
        const dim3 grid_((unsigned)(((F) + cols_ - 1) / cols_));                                                      \
        if (cols_ == 32) { if (cached_) hipLaunchKernelGGL((KERN<32, true>), grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__); else hipLaunchKernelGGL((KERN<32, false>), grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__); } \
        else if (cols_ == 16) { if (cached_) hipLaunchKernelGGL((KERN<16, true>), grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__); else hipLaunchKernelGGL((KERN<16, false>), grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__); } \
        else { if (cached_) hipLaunchKernelGGL((KERN<8, true>), grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__); else hipLaunchKernelGGL((KERN<8, false>), grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__); } \
    } while (0)

// The float4 strip kernels: strip width for (n, F), or 0 when the shape is not theirs (F not a multiple of 4, a pointer off
// 16-byte alignment, or more rows than rmax per lane at the narrowest strip) -- then the dword kernels above run.
static int strip4_cols(int64_t n, int F, int rmax, std::initializer_list<const void *> ptrs)
{
    static const int mode = [] { const char *e = getenv("LCREC_BN_V4"); return e ? atoi(e) : 1; }();   // 0 off; 4 / 8 / 16 force a strip width (tuning)
    if (!mode || (F & 3) || n * F >= ((int64_t)1 << 29)) return 0;       // 32-bit byte offsets inside the kernels
    for (const void *p : ptrs)
        if ((uintptr_t)p & 15) return 0;
    // 16 columns (64-byte row segments) where that makes at least 32 strips; narrower strips for narrower layers, whose whole
    // input is a few hundred KB that a handful of CUs would pull from memory one after the other (in the step the operands are
    // cold: written by a GEMM on other XCDs, or a whole forward pass ago).  32-column strips: measured no better than 16 once
    // the strips of an XCD are neighbours (tools/bn_probe.py).
    int cols = mode == 4 || mode == 8 || mode == 16 ? mode : (F >= 512 ? 16 : (F >= 256 ? 8 : 4));
    while (cols > 4 && n > (int64_t)rmax * (CR_THREADS / (cols / 4))) cols /= 2;
    return n <= (int64_t)rmax * (CR_THREADS / (cols / 4)) ? cols : 0;
}
// rows per lane: 1, 2, 4 or 8 by n
#define LCREC_STRIP4_R8(KERN, COLS_, MODE_, n, grid_, stream, ...)                                                     \
    do {                                                                                                              \
        const int64_t rgs_ = CR_THREADS / (COLS_ / 4);                                                                \
        if ((n) <= rgs_) hipLaunchKernelGGL((KERN<COLS_, 1, MODE_>), grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__); \
        else if ((n) <= 2 * rgs_) hipLaunchKernelGGL((KERN<COLS_, 2, MODE_>), grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__); \
        else if ((n) <= 4 * rgs_) hipLaunchKernelGGL((KERN<COLS_, 4, MODE_>), grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERN<COLS_, 8, MODE_>), grid_, dim3(CR_THREADS), 0, stream, __VA_ARGS__);             \
    } while (0)
#define LCREC_STRIP4_LAUNCH(KERN, MODE_, cols, n, F, stream, ...)                                                      \
    do {                                                                                                              \
        const dim3 grid_((unsigned)(((F) + (cols) - 1) / (cols)));                                                    \
        if ((cols) == 16) LCREC_STRIP4_R8(KERN, 16, MODE_, n, grid_, stream, __VA_ARGS__);                            \
        else if ((cols) == 8) LCREC_STRIP4_R8(KERN, 8, MODE_, n, grid_, stream, __VA_ARGS__);                         \
        else LCREC_STRIP4_R8(KERN, 4, MODE_, n, grid_, stream, __VA_ARGS__);                                          \
    } while (0)

int bn_relu_forward(const float *t, int64_t n, int F, const float *gamma, const float *beta, float eps, float momentum,
                    float *running_mean, float *running_var, float *y, float *mean_out, float *rstd_out, int relu,
                    hipStream_t stream)
{
    if (n == 0 || F == 0) return LCREC_OK;
    if (!t || !y || !mean_out || !rstd_out) return fail(LCREC_EINVAL, "bn_relu_forward: NULL pointer");
    if (n < 2) return fail(LCREC_EINVAL, "bn_relu_forward: training-mode BatchNorm needs more than 1 row (n=%lld)", (long long)n);
    if (n > (1 << 20) || F < 1) return fail(LCREC_EUNSUPPORTED, "bn_relu_forward: sized for training batches (n=%lld)", (long long)n);
    TraceScope trace(K_BN_FWD, stream);
    const int v4 = strip4_cols(n, F, 8, {t, y, gamma, beta, running_mean, running_var, mean_out, rstd_out});
    if (v4) {
        LCREC_STRIP4_LAUNCH(bn_relu_forward_v4_kernel, false, v4, n, F, stream, t, n, F, gamma, beta, eps, momentum, running_mean, running_var, y,
                            mean_out, rstd_out, relu);
        return check_launch("bn_relu_forward_v4_kernel");
    }
    LCREC_STRIP_LAUNCH_N(bn_relu_forward_kernel, F, n, stream, t, n, F, gamma, beta, eps, momentum, running_mean, running_var, y,
                         mean_out, rstd_out, relu);
    return check_launch("bn_relu_forward_kernel");
}

int bn_relu_backward(const float *gy, const float *t, const float *y, int64_t n, int F, const float *gamma, const float *mean,
                     const float *rstd, int relu, float *dt, float *dgamma, float *dbeta, float *dbias, const float *fold_scale,
                     const float *fold_shift, hipStream_t stream)
{
    if (n == 0 || F == 0) return LCREC_OK;
    if (!gy || !t || !mean || !rstd || !dt || (relu && !y && !fold_shift))
        return fail(LCREC_EINVAL, "bn_relu_backward: NULL pointer (with relu: y, or fold_scale and fold_shift, or fold_shift = beta alone)");
    if (n > (1 << 20) || F < 1) return fail(LCREC_EUNSUPPORTED, "bn_relu_backward: sized for training batches (n=%lld)", (long long)n);
    TraceScope trace(K_BN_BWD, stream);
    const int v4 = strip4_cols(n, F, 8, {gy, t, y, gamma, mean, rstd, dt, dgamma, dbeta, dbias, fold_scale, fold_shift});
    if (v4) {
        const Bn4Bwd p{gy, t, y, n, F, gamma, mean, rstd, relu, dt, dgamma, dbeta, dbias, fold_scale, fold_shift, nullptr, nullptr, 0.f};
        LCREC_STRIP4_LAUNCH(bn_relu_backward_v4_kernel, 0, v4, n, F, stream, p);
        return check_launch("bn_relu_backward_v4_kernel");
    }
    LCREC_STRIP_LAUNCH_N(bn_relu_backward_kernel, F, n, stream, gy, t, y, n, F, gamma, mean, rstd, relu, dt, dgamma, dbeta, dbias,
                         fold_scale, fold_shift);
    return check_launch("bn_relu_backward_kernel");
}

int bn_stats(const float *t, int64_t n, int F, float *mean_out, float *m2_out, hipStream_t stream)
{
    if (!t || !mean_out || !m2_out) return fail(LCREC_EINVAL, "bn_stats: NULL pointer");
    if (n < 1 || n > (1 << 20) || F < 1) return fail(LCREC_EUNSUPPORTED, "bn_stats: sized for training batches (n=%lld)", (long long)n);
    TraceScope trace(K_BN_FWD, stream);
    const int v4 = strip4_cols(n, F, 8, {t, mean_out, m2_out});
    if (v4) {
        LCREC_STRIP4_LAUNCH(bn_relu_forward_v4_kernel, true, v4, n, F, stream, t, n, F, (const float *)nullptr, (const float *)nullptr, 0.f, 0.f,
                            (float *)nullptr, (float *)nullptr, (float *)nullptr, mean_out, m2_out, 0);
        return check_launch("bn_relu_forward_v4_kernel<stats>");
    }
    LCREC_STRIP_LAUNCH(bn_stats_kernel, F, stream, t, n, F, mean_out, m2_out);
    return check_launch("bn_stats_kernel");
}

int bn_merge_stats(const float *rows, int world, int F, float eps, float momentum, float *mean_out, float *rstd_out,
                   float *running_mean, float *running_var, hipStream_t stream)
{
    if (!rows || !mean_out || !rstd_out) return fail(LCREC_EINVAL, "bn_merge_stats: NULL pointer");
    if (world < 1 || world > 4096 || F < 1) return fail(LCREC_EINVAL, "bn_merge_stats: bad shape (world=%d, features=%d)", world, F);
    TraceScope trace(K_BN_FWD, stream);
    hipLaunchKernelGGL(bn_merge_stats_kernel, dim3((F + 255) / 256), dim3(256), 0, stream, rows, world, F, eps, momentum, mean_out,
                       rstd_out, running_mean, running_var);
    return check_launch("bn_merge_stats_kernel");
}

int bn_relu_apply(const float *t, int64_t n, int F, const float *gamma, const float *beta, const float *mean, const float *rstd,
                  int relu, float *y, hipStream_t stream)
{
    if (n == 0 || F == 0) return LCREC_OK;
    if (!t || !mean || !rstd || !y) return fail(LCREC_EINVAL, "bn_relu_apply: NULL pointer");
    int64_t blocks = (n * F + 256 * 8 - 1) / (256 * 8);
    if (blocks > 4096) blocks = 4096;
    TraceScope trace(K_BN_FWD, stream);
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t, n, F, gamma, beta, mean, rstd, relu, y);
    return check_launch("bn_relu_apply_kernel");
}

int bn_backward_reduce(const float *gy, const float *t, const float *y, int64_t n, int F, const float *mean, const float *rstd,
                       int relu, float *sum_g, float *sum_gx, float *dbeta, float *dgamma, hipStream_t stream)
{
    if (!gy || !t || !mean || !rstd || !sum_g || !sum_gx || (relu && !y)) return fail(LCREC_EINVAL, "bn_backward_reduce: NULL pointer");
    if (n < 1 || n > (1 << 20) || F < 1) return fail(LCREC_EUNSUPPORTED, "bn_backward_reduce: sized for training batches");
    TraceScope trace(K_BN_BWD, stream);
    const int v4 = strip4_cols(n, F, 8, {gy, t, y, mean, rstd, sum_g, sum_gx, dbeta, dgamma});
    if (v4) {
        const Bn4Bwd p{gy, t, y, n, F, nullptr, mean, rstd, relu, nullptr, dgamma, dbeta, nullptr, nullptr, nullptr, sum_g, sum_gx, 0.f};
        LCREC_STRIP4_LAUNCH(bn_relu_backward_v4_kernel, 1, v4, n, F, stream, p);
        return check_launch("bn_relu_backward_v4_kernel<reduce>");
    }
    LCREC_STRIP_LAUNCH(bn_backward_reduce_kernel, F, stream, gy, t, y, n, F, mean, rstd, relu, sum_g, sum_gx, dbeta, dgamma);
    return check_launch("bn_backward_reduce_kernel");
}

int bn_backward_apply(const float *gy, const float *t, const float *y, int64_t n, int F, const float *gamma, const float *mean,
                      const float *rstd, int relu, const float *sum_g, const float *sum_gx, float n_total, float *dt, float *dbias,
                      hipStream_t stream)
{
    if (!gy || !t || !mean || !rstd || !sum_g || !sum_gx || !dt || (relu && !y)) return fail(LCREC_EINVAL, "bn_backward_apply: NULL pointer");
    if (n < 1 || n > (1 << 20) || F < 1 || !(n_total >= 1.0f)) return fail(LCREC_EUNSUPPORTED, "bn_backward_apply: sized for training batches");
    TraceScope trace(K_BN_BWD, stream);
    const int v4 = strip4_cols(n, F, 8, {gy, t, y, gamma, mean, rstd, sum_g, sum_gx, dt, dbias});
    if (v4) {
        const Bn4Bwd p{gy, t, y, n, F, gamma, mean, rstd, relu, dt, nullptr, nullptr, dbias, nullptr, nullptr,
                       const_cast<float *>(sum_g), const_cast<float *>(sum_gx), n_total};
        LCREC_STRIP4_LAUNCH(bn_relu_backward_v4_kernel, 2, v4, n, F, stream, p);
        return check_launch("bn_relu_backward_v4_kernel<apply>");
    }
    LCREC_STRIP_LAUNCH(bn_backward_apply_kernel, F, stream, gy, t, y, n, F, gamma, mean, rstd, relu, sum_g, sum_gx, n_total, dt, dbias);
    return check_launch("bn_backward_apply_kernel");
}

int relu_bias_backward(const float *gy, const float *y, int64_t n, int F, int relu, float *g_out, float *dbias, hipStream_t stream)
{
    if (n == 0 || F == 0) return LCREC_OK;
    if (!gy || (relu && !y)) return fail(LCREC_EINVAL, "relu_bias_backward: NULL pointer");
    if (n > (1 << 20) || F < 1) return fail(LCREC_EUNSUPPORTED, "relu_bias_backward: sized for training batches (n=%lld)", (long long)n);
    TraceScope trace(K_RELU_BIAS_BWD, stream);
    LCREC_STRIP_LAUNCH(relu_bias_backward_kernel, F, stream, gy, y, n, F, relu, g_out, dbias);
    return check_launch("relu_bias_backward_kernel");
}

static int red_blocks(int64_t count, int per_thread = 16)
{
    int64_t b = (count + (int64_t)RED_THREADS * per_thread - 1) / ((int64_t)RED_THREADS * per_thread);
    return (int)(b < 1 ? 1 : (b > RED_MAX_BLOCKS ? RED_MAX_BLOCKS : b));
}

size_t train_reduce_workspace() { return RED_MAX_BLOCKS * sizeof(double); }

int recon_loss_grad(const float *out, const float *x, int64_t count, int64_t count_total, int l1, float *g, float *loss,
                    void *workspace, size_t workspace_bytes, unsigned *ticket, hipStream_t stream)
{
    if (count_total == 0) count_total = count;
    if (count_total < count) return fail(LCREC_EINVAL, "recon_loss_grad: count_total < count");
    if (!out || !x || !loss) return fail(LCREC_EINVAL, "recon_loss_grad: NULL pointer");
    if (count < 1) return fail(LCREC_EINVAL, "recon_loss_grad: empty input");
    if (!workspace || workspace_bytes < train_reduce_workspace()) return fail(LCREC_EWORKSPACE, "recon_loss_grad: workspace too small");
    const int blocks = red_blocks(count, 4);          // one 16-byte access per lane and array: a batch's 786 k elements on 192 CUs, not 48
    TraceScope trace(K_LOSS, stream);
    hipLaunchKernelGGL(recon_loss_grad_kernel, dim3(blocks), dim3(RED_THREADS), 0, stream, out, x, count, count_total, l1, g, (double *)workspace,
                       ticket, loss);
    if (!ticket)
        hipLaunchKernelGGL(recon_loss_finish_kernel, dim3(1), dim3(RED_THREADS), 0, stream, (const double *)workspace, blocks, count_total, loss);
    return check_launch("recon_loss_grad_kernel");
}

int grad_norm_clip(const float *g, int64_t count, float max_norm, float *norm_out, void *workspace, size_t workspace_bytes,
                   unsigned *ticket, hipStream_t stream)
{
    if (!g || !norm_out) return fail(LCREC_EINVAL, "grad_norm_clip: NULL pointer");
    if (count < 1) return fail(LCREC_EINVAL, "grad_norm_clip: empty input");
    if (!workspace || workspace_bytes < train_reduce_workspace()) return fail(LCREC_EWORKSPACE, "grad_norm_clip: workspace too small");
    const int blocks = red_blocks(count);
    TraceScope trace(K_GRAD_NORM, stream);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(RED_THREADS), 0, stream, g, count, (double *)workspace, ticket, max_norm, norm_out);
    if (!ticket)
        hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(RED_THREADS), 0, stream, (const double *)workspace, blocks, max_norm, norm_out);
    return check_launch("grad_norm kernels");
}

int step_losses(const double *sse, int L, int64_t n, int e, float beta, float qlw, const float *recon, float *out3, double *sums2,
                unsigned char *nan_flag, const int64_t *probe, unsigned char *probe_flag, hipStream_t stream)
{
    if (!sse || !recon || !out3) return fail(LCREC_EINVAL, "step_losses: NULL pointer");
    if (L < 1 || L > LCREC_MAX_LEVELS || n < 1 || e < 1) return fail(LCREC_EINVAL, "step_losses: bad shape");
    TraceScope trace(K_LOSS, stream);
    hipLaunchKernelGGL(step_losses_kernel, dim3(1), dim3(64), 0, stream, sse, L, (double)n * (double)e, beta, qlw, recon, out3, sums2, nan_flag,
                       probe, probe_flag);
    return check_launch("step_losses_kernel");
}

int quantizer_input_grad(const float *z, const float *cb0, const int64_t *idx, int64_t idx_stride, int64_t n, int e, float coef,
                         float weight, const float *g_xq, float *out, hipStream_t stream)
{
    if (n == 0) return LCREC_OK;
    if (!z || !cb0 || !idx || !g_xq || !out) return fail(LCREC_EINVAL, "quantizer_input_grad: NULL pointer");
    if (n < 0 || e < 1) return fail(LCREC_EINVAL, "quantizer_input_grad: bad shape");
    int64_t blocks = (n * e + 256 * 4 - 1) / (256 * 4);
    if (blocks > 2048) blocks = 2048;
    TraceScope trace(K_APPLY_LEVEL, stream);
    hipLaunchKernelGGL(quantizer_input_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, z, cb0, idx, idx_stride, n, e, coef, weight,
                       g_xq, out);
    return check_launch("quantizer_input_grad_kernel");
}

int quantizer_input_grad_bias(const float *z, const float *cb0, const int64_t *idx, int64_t idx_stride, int64_t n, int e, float coef,
                              float weight, const float *g_xq, float *out, float *dbias, hipStream_t stream)
{
    if (!dbias) return quantizer_input_grad(z, cb0, idx, idx_stride, n, e, coef, weight, g_xq, out, stream);
    if (n == 0) return LCREC_OK;
    if (!z || !cb0 || !idx || !g_xq || !out) return fail(LCREC_EINVAL, "quantizer_input_grad_bias: NULL pointer");
    if (n < 0 || e < 1) return fail(LCREC_EINVAL, "quantizer_input_grad_bias: bad shape");
    if ((e == 16 || e == 32 || e == 64) && n * e <= 65536) {
        TraceScope trace(K_APPLY_LEVEL, stream);
        hipLaunchKernelGGL(quantizer_input_grad_bias_kernel, dim3(1), dim3(QGB_THREADS), 0, stream, z, cb0, idx, idx_stride, (int)n, e, coef,
                           weight, g_xq, out, dbias);
        return check_launch("quantizer_input_grad_bias_kernel");
    }
    // other shapes: the two launches this call stands for
    if (int rc = quantizer_input_grad(z, cb0, idx, idx_stride, n, e, coef, weight, g_xq, out, stream)) return rc;
    return relu_bias_backward(out, nullptr, n, e, 0, nullptr, dbias, stream);
}

int codebook_grad(const float *count, const float *sum, const float *cb, int K, int e, float scale, float weight, float *grad,
                  hipStream_t stream)
{
    if (!count || !sum || !cb || !grad) return fail(LCREC_EINVAL, "codebook_grad: NULL pointer");
    if (K < 1 || e < 1) return fail(LCREC_EINVAL, "codebook_grad: bad shape");
    TraceScope trace(K_CODE_STATS, stream);
    hipLaunchKernelGGL(codebook_grad_kernel, dim3((K * e + 255) / 256), dim3(256), 0, stream, count, sum, cb, K, e, scale, weight, grad);
    return check_launch("codebook_grad_kernel");
}

int adamw_step(float *p, float *g, float *m, float *v, int64_t count, const float *clip, int64_t *step, double base_lr,
               double beta1, double beta2, double eps, double weight_decay, int decoupled, int schedule, int64_t warmup_steps,
               int64_t total_steps, float *lr_out, unsigned *ticket, const unsigned char *skip, hipStream_t stream)
{
    if (!p || !g || !m || !v || !step) return fail(LCREC_EINVAL, "adamw_step: NULL pointer");
    if (count < 1) return fail(LCREC_EINVAL, "adamw_step: empty parameter buffer");
    if (schedule < -1 || schedule > 1) return fail(LCREC_EINVAL, "adamw_step: schedule %d (supported: -1 none, 0 constant, 1 linear)", schedule);
    AdamParams a = {p, g, m, v, count, clip, step, base_lr, beta1, beta2, eps, weight_decay, decoupled, schedule, warmup_steps,
                    total_steps, lr_out, ticket, skip};
    int64_t blocks = (count + ADAM_THREADS * 8 - 1) / (ADAM_THREADS * 8);
    if (blocks > TICKET_MAX_WORKGROUPS) blocks = TICKET_MAX_WORKGROUPS;
    TraceScope trace(K_ADAMW, stream);
    hipLaunchKernelGGL(adamw_step_kernel, dim3((unsigned)blocks), dim3(ADAM_THREADS), 0, stream, a);
    if (!ticket) hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, stream, step, skip);
    return check_launch("adamw_step_kernel");
}

}  // namespace lcrec
