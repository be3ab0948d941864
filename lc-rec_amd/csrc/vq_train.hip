// Quantiser kernels around the hard-assignment path: Sinkhorn ("uniform semantic")
// assignment, applying a given assignment to a level, the per-code segmented reduce
// that feeds the codebook gradient and the EMA update, and the EMA blend itself.
//
// Reference (paths relative to the reference root):
//   index/models/vq.py:51-61,76-83       centring + fp64 Sinkhorn + argmax
//   index/models/layers.py:85-108        sinkhorn_algorithm
//   index/models/vq.py:87-95, rq.py:47-48  gather, losses, STE, residual update
//   index_improve/models/vq.py:151-184   count / per-code sum / EMA / blend
//   autograd of vq.py:90-92: dL/dC[k] ~ n_k*C[k] - S_k
#include "common.h"

#include <stdlib.h>

#include <vector>

namespace lcrec {

// ------------------------------------------------------------------------------------------
// Distances d[i][j] = (xx_i + cc_j) - 2*dot_ij, fp32, same fma chains as the argmin kernel
// (vq.py:71-73).  One thread owns one code (its row in registers) and walks the item tile.
// ------------------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void load_code_row(const float *cb, int K, int j, float (&c)[E], float &cc)
{
    cc = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) c[k] = 0.f;
    if (j < K) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(cb + (size_t)j * E);
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            const f32x4 v = src[q];
            c[4 * q] = v[0]; c[4 * q + 1] = v[1]; c[4 * q + 2] = v[2]; c[4 * q + 3] = v[3];
        }
#pragma unroll
        for (int k = 0; k < E; ++k) cc = __builtin_fmaf(c[k], c[k], cc);
    }
}

// order-preserving float <-> uint map for atomic min/max
__device__ __forceinline__ unsigned f2ord(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

constexpr int DIST_ITEMS = 32;   // items per block in the distance kernel

constexpr int DIST_TICKET_MAX_BLOCKS = 4096;   // (a batch of 1024 rows at K = 256: 32 workgroups)    // (min, max) partials of the ticket form: 8 bytes per workgroup

// grid = (ceil(n/DIST_ITEMS), ceil(K/256)); d is [n][K]; minmax = {ord(min), ord(max)}: initialised by an earlier launch and
// reached by atomics -- or, with a ticket (include/lcrec.h), written by the last workgroup to arrive from every workgroup's
// partial, together with the solver's control words; the workgroups also arm the scaling solver's exchange slots then, so the
// whole set-up of a solve is this one launch.
template <int E>
__global__ __launch_bounds__(256) void vq_distance_kernel(const float *__restrict__ r, int64_t n,
                                                         const float *__restrict__ cb, int K,
                                                         float *__restrict__ d, unsigned *minmax, unsigned *ticket,
                                                         unsigned *mm_partial, unsigned long long *slots, int64_t slot_count)
{
    __shared__ float rs[DIST_ITEMS][E];
    __shared__ float xs[DIST_ITEMS];
    __shared__ unsigned red[2][4];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * DIST_ITEMS;
    const int j = blockIdx.y * 256 + tid;
    float c[E], cc;
    load_code_row<E>(cb, K, j, c, cc);
    for (int q = tid; q < DIST_ITEMS * E; q += 256) {
        const int64_t i = i0 + q / E;
        rs[q / E][q % E] = i < n ? r[i * E + q % E] : 0.f;
    }
    __syncthreads();
    if (tid < DIST_ITEMS) {
        float xx = 0.f;
#pragma unroll
        for (int k = 0; k < E; ++k) xx = __builtin_fmaf(rs[tid][k], rs[tid][k], xx);
        xs[tid] = xx;
    }
    __syncthreads();
    float lo = __builtin_inff(), hi = -__builtin_inff();
    // four items' chains side by side (each its own k-ascending fma chain, the same bits): one wave per SIMD here, and a single
    // chain of 32 dependent fmas per item left the ALU waiting on itself
    for (int t0 = 0; t0 < DIST_ITEMS && i0 + t0 < n; t0 += 4) {
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < E; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) dot[u] = __builtin_fmaf(rs[t0 + u][k], c[k], dot[u]);      // rows past n hold zeros
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + t0 + u;
            const float s = xs[t0 + u] + cc;
            const float dv = __builtin_fmaf(-2.0f, dot[u], s);
            if (j < K && i < n) {
                d[i * K + j] = dv;
                lo = dv < lo ? dv : lo;
                hi = dv > hi ? dv : hi;
            }
        }
    }
    if (minmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if ((tid & 63) == 0) { red[0][tid >> 6] = f2ord(lo); red[1][tid >> 6] = f2ord(hi); }
        __syncthreads();
        if (ticket) {
            const unsigned nblk = gridDim.x * gridDim.y, me = blockIdx.y * gridDim.x + blockIdx.x;
            for (int64_t i = (int64_t)me * 256 + tid; i < slot_count; i += (int64_t)nblk * 256) slots[i] = ~0ull;   // SKP_EMPTY
            __shared__ int last_sh;
            if (tid == 0) {
                unsigned a = red[0][0], b = red[1][0];
                for (int w = 1; w < 4; ++w) { a = red[0][w] < a ? red[0][w] : a; b = red[1][w] > b ? red[1][w] : b; }
                __hip_atomic_store(mm_partial + 2 * me, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mm_partial + 2 * me + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last_sh = ticket_is_last(ticket, nblk) ? 1 : 0;
            }
            __syncthreads();
            if (last_sh && tid < 64) {
                // the lanes of wave 0 fetch the partials side by side (min / max: any order gives the same result)
                unsigned a = 0xffffffffu, b = 0u;
                for (unsigned q = tid; q < nblk; q += 64) {
                    const unsigned a2 = __hip_atomic_load(mm_partial + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned b2 = __hip_atomic_load(mm_partial + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a = a2 < a ? a2 : a;
                    b = b2 > b ? b2 : b;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const unsigned a2 = __shfl_xor(a, o, 64), b2 = __shfl_xor(b, o, 64);
                    a = a2 < a ? a2 : a;
                    b = b2 > b ? b2 : b;
                }
                if (tid == 0) { minmax[0] = a; minmax[1] = b; }
                if (tid >= 2 && tid < 32) minmax[tid] = 0u;           // barrier counter, timeout flag; [8] done, [16..31] XCD arrival counters
            }
            return;
        }
        if (tid == 0) {
            unsigned a = red[0][0], b = red[1][0];
            for (int w = 1; w < 4; ++w) { a = red[0][w] < a ? red[0][w] : a; b = red[1][w] > b ? red[1][w] : b; }
            atomicMin(&minmax[0], a);
            atomicMax(&minmax[1], b);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Sinkhorn, large problem (one B x K matrix, training): Q lives in global memory (L2-resident
// at 2048 x 256 x 8 B = 4 MB); one launch per iteration.  A block owns SK_ROWS whole rows, so
// row sums are block-local; column sums travel between launches as per-block partials.
//   launch 0      : Q = exp(-centre(d)/eps), per-block partial of the grand total
//   launch t (1..iters): [t==1: Q /= total | t>1: Q /= colsum, Q /= K]; Q /= rowsum; Q /= B; emit column partials
//   final         : Q /= colsum; Q /= K; Q *= B; argmax per row (first maximum)
// ------------------------------------------------------------------------------------------
constexpr int SK_ROWS = 32;      // rows per block
constexpr int SK_THREADS = 512;  // 8 waves, 4 rows each
constexpr int SK_MAXC = 16;      // columns per lane: K <= 1024

struct SkBig {
    const float *d;       // [B][K]
    double *Q;            // [B][K]
    double *col_part;     // [2][nblk][K] ping-pong column partial sums
    double *tot_part;     // [nblk]
    const unsigned *minmax;
    int64_t B;
    int K, nblk;
    double eps;
};

// x / d exactly as a division, cheaply when d is a power of two: then 1/d is exact and x * (1/d) is the correctly
// rounded value of the same real number, i.e. the same bits as x / d (the reference divides by B and by K every
// iteration, layers.py:100-104; K = 256 or 1024 and the batch / most collision groups are powers of two).  An fp64
// division is ~15 instructions; these two were half of all divisions in the solver.
struct ExactDiv {
    double d, inv;
    bool pow2;
    __device__ __forceinline__ explicit ExactDiv(double v) : d(v), inv(1.0 / v)
    {
        const unsigned long long bits = __double_as_longlong(v);
        pow2 = v > 0.0 && (bits & 0x000fffffffffffffull) == 0;      // zero mantissa: v = 2^e
    }
    __device__ __forceinline__ double operator()(double x) const { return pow2 ? x * inv : x / d; }
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(SK_THREADS) void sk_init_kernel(SkBig p)
{
    __shared__ double wsum[SK_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // vq.py:52-60 in fp32: middle = (max+min)/2; amplitude = max - middle + 1e-5
    const float hi = ord2f(p.minmax[1]), lo = ord2f(p.minmax[0]);
    const float middle = (hi + lo) / 2.0f;
    const float amplitude = (hi - middle) + 1e-5f;
    double part = 0.0;
    for (int rr = wave; rr < SK_ROWS; rr += SK_THREADS / 64) {
        const int64_t row = (int64_t)blockIdx.x * SK_ROWS + rr;
        if (row >= p.B) break;
        for (int j = lane; j < p.K; j += 64) {
            const float cen = (p.d[row * p.K + j] - middle) / amplitude;
            const double q = exp(-(double)cen / p.eps);
            p.Q[row * p.K + j] = q;
            part += q;
        }
    }
    part = wave_sum(part);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < SK_THREADS / 64; ++w) s += wsum[w];
        p.tot_part[blockIdx.x] = s;
    }
}

// mode 0: first iteration (divide by the grand total first); mode 1: later iteration (column-normalise first)
__global__ __launch_bounds__(SK_THREADS) void sk_iter_kernel(SkBig p, int mode, int src)
{
    extern __shared__ __attribute__((aligned(16))) double sk_sm[];
    double *colsum = sk_sm;                 // [K]   (mode 1)
    double *colacc = sk_sm + p.K;           // [waves][K] partials of this block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVES = SK_THREADS / 64;
    double total = 0.0;
    if (mode == 0) {
        for (int b = 0; b < p.nblk; ++b) total += p.tot_part[b];
    } else {
        const double *cp = p.col_part + (size_t)src * p.nblk * p.K;
        for (int j = threadIdx.x; j < p.K; j += SK_THREADS) {
            double s = 0.0;
            for (int b = 0; b < p.nblk; ++b) s += cp[(size_t)b * p.K + j];
            colsum[j] = s;
        }
        __syncthreads();
    }
    double acc[SK_MAXC];
#pragma unroll
    for (int c = 0; c < SK_MAXC; ++c) acc[c] = 0.0;
    const double Bd = (double)p.B, Kd = (double)p.K;
    const ExactDiv divB(Bd), divK(Kd);
    for (int rr = wave; rr < SK_ROWS; rr += WAVES) {
        const int64_t row = (int64_t)blockIdx.x * SK_ROWS + rr;
        if (row >= p.B) break;
        double q[SK_MAXC];
        double rs = 0.0;
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            q[c] = 0.0;
            if (j < p.K) {
                double v = p.Q[row * p.K + j];
                if (mode == 0) v = v / total;
                else { v = v / colsum[j]; v = divK(v); }
                q[c] = v;
                rs += v;
            }
        }
        rs = wave_sum(rs);
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            if (j < p.K) {
                double v = q[c] / rs;
                v = divB(v);
                p.Q[row * p.K + j] = v;
                acc[c] += v;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < SK_MAXC; ++c) {
        const int j = lane + 64 * c;
        if (j < p.K) colacc[wave * p.K + j] = acc[c];
    }
    __syncthreads();
    double *out = p.col_part + ((size_t)(src ^ 1) * p.nblk + blockIdx.x) * p.K;
    for (int j = threadIdx.x; j < p.K; j += SK_THREADS) {
        double s = 0.0;
        for (int w = 0; w < WAVES; ++w) s += colacc[w * p.K + j];
        out[j] = s;
    }
}

__global__ __launch_bounds__(SK_THREADS) void sk_final_kernel(SkBig p, int src, int64_t *idx_out, int64_t idx_stride)
{
    extern __shared__ __attribute__((aligned(16))) double sk_sm[];
    double *colsum = sk_sm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *cp = p.col_part + (size_t)src * p.nblk * p.K;
    for (int j = threadIdx.x; j < p.K; j += SK_THREADS) {
        double s = 0.0;
        for (int b = 0; b < p.nblk; ++b) s += cp[(size_t)b * p.K + j];
        colsum[j] = s;
    }
    __syncthreads();
    const double Bd = (double)p.B, Kd = (double)p.K;
    const ExactDiv divB(Bd), divK(Kd);
    for (int rr = wave; rr < SK_ROWS; rr += SK_THREADS / 64) {
        const int64_t row = (int64_t)blockIdx.x * SK_ROWS + rr;
        if (row >= p.B) break;
        double best = -1.0;
        int bj = 0;
        for (int j = lane; j < p.K; j += 64) {
            double v = p.Q[row * p.K + j] / colsum[j];
            v = divK(v);
            v = v * Bd;
            if (v > best) { best = v; bj = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double b2 = __shfl_xor(best, o, 64);
            const int j2 = __shfl_xor(bj, o, 64);
            if (b2 > best || (b2 == best && j2 < bj)) { best = b2; bj = j2; }
        }
        if (lane == 0) idx_out[row * idx_stride] = bj;
    }
}

// ------------------------------------------------------------------------------------------
// Sinkhorn, many independent problems (collision groups of generate_indices.py:113-119): one
// workgroup per group.  The g x K fp64 matrix lives in LDS when it fits (g*K <= SKS_MAX doubles; launches
// are bucketed by size so that pairs and triples -- the bulk -- are not charged a 128 KB allocation),
// otherwise in this group's slice of a workspace slab in HBM/L2 (GLOBALQ): the few-hundred-row groups a
// collision-heavy round produces then run side by side instead of ~50 launches each, one after another.
// ------------------------------------------------------------------------------------------
constexpr int SKS_MAX = 16384;   // doubles of LDS for Q (128 KB)
constexpr int SKS_THREADS = 256;

template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double *scratch)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < THREADS / 64; ++w) s += scratch[w];
    return s;
}

template <int E, bool GLOBALQ, int THREADS>
__global__ __launch_bounds__(THREADS) void sk_small_kernel(const float *__restrict__ r, const float *__restrict__ cb,
                                                              int K, const int64_t *__restrict__ offs,
                                                              double eps, int iters, int64_t *idx_out,
                                                              int64_t idx_stride, double *qslab)
{
    extern __shared__ __attribute__((aligned(16))) double sks_sm[];
    __shared__ double scratch[THREADS / 64];
    __shared__ float fred[2][THREADS / 64];
    const int64_t i0 = offs[3 * blockIdx.x];           // offs holds (begin, end, slab offset in doubles) triples
    const int g = (int)(offs[3 * blockIdx.x + 1] - i0);
    if (g <= 0) return;
    // (a workgroup's global-memory traffic is ordered by __syncthreads just like its LDS traffic: all its
    // waves share one CU, hence one L1)
    double *Q = GLOBALQ ? qslab + offs[3 * blockIdx.x + 2] : sks_sm;   // [g][K]
    double *rsum = Q + (size_t)g * K;            // [g]
    double *csum = rsum + ((g + 1) & ~1);        // [K]
    float *dmat = reinterpret_cast<float *>(Q);  // fp32 distances alias the front half of Q first
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // distances, fp32 canonical chains; thread j owns code j (+256, ...)
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int j = tid; j < K; j += THREADS) {
        float c[E], cc;
        load_code_row<E>(cb, K, j, c, cc);
        for (int t = 0; t < g; ++t) {
            const float *ri = r + (i0 + t) * E;
            float xx = 0.f, dot = 0.f;
#pragma unroll
            for (int k = 0; k < E; ++k) xx = __builtin_fmaf(ri[k], ri[k], xx);
#pragma unroll
            for (int k = 0; k < E; ++k) dot = __builtin_fmaf(ri[k], c[k], dot);
            const float dv = __builtin_fmaf(-2.0f, dot, xx + cc);
            dmat[(size_t)t * K + j] = dv;
            lo = dv < lo ? dv : lo;
            hi = dv > hi ? dv : hi;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (lane == 0) { fred[0][wave] = lo; fred[1][wave] = hi; }
    __syncthreads();
    for (int w = 0; w < THREADS / 64; ++w) { lo = fred[0][w] < lo ? fred[0][w] : lo; hi = fred[1][w] > hi ? fred[1][w] : hi; }
    const float middle = (hi + lo) / 2.0f;
    const float amplitude = (hi - middle) + 1e-5f;

    // Q = exp(-centre(d)/eps): walk backwards so the fp64 store never overtakes an unread fp32 slot
    const int total_el = g * K;
    double part = 0.0;
    {
        // each thread converts a contiguous chunk, highest chunk first is not needed: read all into registers per pass
        // pass structure: chunks of THREADS elements from the END; element q (fp32 at byte 4q) -> fp64 at byte 8q >= 4q
        for (int base = ((total_el - 1) / THREADS) * THREADS; base >= 0; base -= THREADS) {
            const int q = base + tid;
            float dv = 0.f;
            if (q < total_el) dv = dmat[q];
            __syncthreads();
            if (q < total_el) {
                const float cen = (dv - middle) / amplitude;
                const double v = exp(-(double)cen / eps);
                Q[q] = v;
                part += v;
            }
            __syncthreads();
        }
    }
    const double total = block_sum<THREADS>(part, scratch);
    for (int q = tid; q < total_el; q += THREADS) Q[q] = Q[q] / total;
    __syncthreads();
    const double Bd = (double)g, Kd = (double)K;
    const ExactDiv divB(Bd), divK(Kd);
    for (int it = 0; it < iters; ++it) {
        for (int t = wave; t < g; t += THREADS / 64) {          // row sums (layers.py:99)
            double s = 0.0;
            for (int j = lane; j < K; j += 64) s += Q[(size_t)t * K + j];
            s = wave_sum(s);
            if (lane == 0) rsum[t] = s;
        }
        __syncthreads();
        // Eight rows at a time: the loads of a batch are issued together.  Written one row at a time, every load waits
        // for the store before it (the compiler cannot prove Q[t*K+j] and Q[(t+1)*K+j] distinct), and with Q in the
        // slab that made an iteration a chain of 2g L2 round trips (117 us at g = 256).  Same operations, same order.
        // (Measured and dropped for the slab kernel: four threads per column instead of one, and row sums four rows at
        // a time -- its launches did not get shorter: they last as long as their largest group, whose ~2 g K divisions
        // per iteration run on one CU, ~12 us per row.)
        for (int j = tid; j < K; j += THREADS) {                 // Q /= rowsum; Q /= B; column sums (:100-103)
            double *col = Q + j;
            double s = 0.0;
            int t = 0;
            for (; t + 8 <= g; t += 8) {
                double v[8], rs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { v[u] = col[(size_t)(t + u) * K]; rs[u] = rsum[t + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = divB(v[u] / rs[u]);
#pragma unroll
                for (int u = 0; u < 8; ++u) { col[(size_t)(t + u) * K] = v[u]; s += v[u]; }
            }
            for (; t < g; ++t) {
                double v = col[(size_t)t * K] / rsum[t];
                v = divB(v);
                col[(size_t)t * K] = v;
                s += v;
            }
            csum[j] = s;
            t = 0;
            for (; t + 8 <= g; t += 8) {                             // Q /= colsum; Q /= K (:103-104)
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(t + u) * K];
#pragma unroll
                for (int u = 0; u < 8; ++u) col[(size_t)(t + u) * K] = divK(v[u] / s);
            }
            for (; t < g; ++t) {
                double v = col[(size_t)t * K] / s;
                col[(size_t)t * K] = divK(v);
            }
        }
        __syncthreads();
    }
    for (int t = wave; t < g; t += THREADS / 64) {               // Q *= B; argmax (:107, vq.py:83)
        double best = -1.0;
        int bj = 0;
        for (int j = lane; j < K; j += 64) {
            const double v = Q[(size_t)t * K + j] * Bd;
            if (v > best) { best = v; bj = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double b2 = __shfl_xor(best, o, 64);
            const int j2 = __shfl_xor(bj, o, 64);
            if (b2 > best || (b2 == best && j2 < bj)) { best = b2; bj = j2; }
        }
        if (lane == 0) idx_out[(i0 + t) * idx_stride] = bj;
    }
}

// ------------------------------------------------------------------------------------------
// Apply a given assignment to one level (vq.py:87-95 + rq.py:47-48): gather, SSE, STE, residual.
// One wave-quarter per item: thread handles 4 consecutive dims.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void apply_level_kernel(const float *__restrict__ r_in, int64_t n, int e,
                                                         const float *__restrict__ cb, int K,
                                                         const int64_t *__restrict__ idx, int64_t idx_stride,
                                                         float *xq, int xq_accumulate, float *r_out,
                                                         double *sse_partial, unsigned *ticket, double *sse_out)
{
    __shared__ double wsum[4];
    const int per = e / 4;
    const int64_t total = n * per;
    double part = 0.0;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t i = q / per;
        const int k4 = (int)(q % per) * 4;
        int64_t j = idx[i * idx_stride];
        j = j < 0 ? 0 : (j >= K ? K - 1 : j);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(cb + j * e + k4);
        const f32x4 r = *reinterpret_cast<const f32x4 *>(r_in + i * e + k4);
        f32x4 xo = {0.f, 0.f, 0.f, 0.f}, ro;
        if (xq && xq_accumulate) xo = *reinterpret_cast<const f32x4 *>(xq + i * e + k4);
        float sse = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float tt = c[t] - r[t];
            sse = __builtin_fmaf(tt, tt, sse);
            const float s = r[t] + tt;
            xo[t] = xo[t] + s;
            ro[t] = r[t] - s;
        }
        part += (double)sse;
        if (xq) *reinterpret_cast<f32x4 *>(xq + i * e + k4) = xo;
        if (r_out) *reinterpret_cast<f32x4 *>(r_out + i * e + k4) = ro;
    }
    if (sse_partial) {
        part = wave_sum(part);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
        __syncthreads();
        __shared__ int last_sh;
        if (threadIdx.x == 0) {
            const double mine = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (!ticket) {
                sse_partial[blockIdx.x] = mine;
            } else {
                handoff_put(sse_partial + blockIdx.x, mine);
                last_sh = ticket_is_last(ticket, gridDim.x) ? 1 : 0;
            }
        }
        if (ticket) {
            // (include/lcrec.h, `ticket`) the last workgroup to arrive adds the partials, in sum_partials_kernel's order
            __syncthreads();
            if (last_sh && threadIdx.x < 64) {
                const double s = handoff_sum_ordered(sse_partial, (int)gridDim.x, 1);
                if (threadIdx.x == 0) *sse_out = s;
            }
        }
    }
}

__global__ void sum_partials_kernel(const double *partial, int count, double *out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < count; ++b) s += partial[b];
        *out = s;
    }
}

// ------------------------------------------------------------------------------------------
// Per-code statistics: count[k] = #{i: idx_i = k}, sum[k][:] = sum of resid_i over those items,
// added in ITEM ORDER in fp32 -- the order of the CPU index_add_ the reference's EMA path uses
// (index_improve/models/vq.py:151-167), so the result is bit-identical to the CPU and
// independent of launch geometry.  Thread (code, dim) scans all items; items stream through LDS.
// ------------------------------------------------------------------------------------------
constexpr int CS_CHUNK = 256;

template <int E>
__global__ __launch_bounds__(256) void code_stats_kernel(const int64_t *__restrict__ idx, int64_t idx_stride,
                                                        const float *__restrict__ resid, int64_t n, int K,
                                                        float *__restrict__ count, float *__restrict__ sum)
{
    constexpr int CODES = 256 / E;
    __shared__ int sidx[CS_CHUNK];
    __shared__ float sres[CS_CHUNK][E];
    const int tid = threadIdx.x;
    const int d = tid % E;
    const int k = blockIdx.x * CODES + tid / E;
    float acc = 0.f, cnt = 0.f;
    for (int64_t i0 = 0; i0 < n; i0 += CS_CHUNK) {
        const int m = (int)((n - i0 < CS_CHUNK) ? n - i0 : CS_CHUNK);
        __syncthreads();
        for (int q = tid; q < m; q += 256) sidx[q] = (int)idx[(i0 + q) * idx_stride];
        for (int q = tid; q < m * E; q += 256) sres[q / E][q % E] = resid[i0 * E + q];
        __syncthreads();
        for (int t = 0; t < m; ++t) {
            if (sidx[t] == k) {
                acc = acc + sres[t][d];
                cnt = cnt + 1.0f;
            }
        }
    }
    if (k < K) {
        sum[(size_t)k * E + d] = acc;
        if (d == 0) count[k] = cnt;
    }
}

// index_improve/models/vq.py:155-184.  decay/alpha/keep are the reference's python-double rates
// rounded to fp32 by the caller (alpha = 1-decay, keep = 1-(1-decay)).
__global__ void ema_update_kernel(float *ema_count, float *ema_sum, float *codebook, const float *count,
                                  const float *sum, int K, int e, float decay, float alpha, float keep, float eps,
                                  const unsigned char *skip)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)K * e) return;
    if (skip && *skip) return;                  // a sticky "loss was NaN": the statistics and the codebook keep their last good state
    const int k = (int)(q / e);
    const float en = __builtin_fmaf(count[k], alpha, ema_count[k] * decay);
    const float ew = __builtin_fmaf(sum[q], alpha, ema_sum[q] * decay);
    ema_sum[q] = ew;
    if (en > eps) {
        const float nw = ew / (en + eps);
        const float a = codebook[q] * keep;
        const float b = nw * alpha;
        codebook[q] = a + b;
    }
    // the e threads of a code sit in one wavefront (64 % e == 0), so every lane has read
    // ema_count[k] above before lane d==0 overwrites it here
    if (q % e == 0) ema_count[k] = en;
}

// ---------------------------------------------------------------- host launchers

int vq_distances(const float *r, int64_t n, int e, const float *cb, int K, float *d, unsigned *minmax,
                 hipStream_t stream, unsigned *ticket = nullptr, unsigned *mm_partial = nullptr, unsigned long long *slots = nullptr,
                 int64_t slot_count = 0)
{
    dim3 grid((unsigned)((n + DIST_ITEMS - 1) / DIST_ITEMS), (unsigned)((K + 255) / 256));
    TraceScope trace(K_VQ_DISTANCE, stream);
    if (e == 16) hipLaunchKernelGGL(vq_distance_kernel<16>, grid, dim3(256), 0, stream, r, n, cb, K, d, minmax, ticket, mm_partial, slots, slot_count);
    else if (e == 32) hipLaunchKernelGGL(vq_distance_kernel<32>, grid, dim3(256), 0, stream, r, n, cb, K, d, minmax, ticket, mm_partial, slots, slot_count);
    else if (e == 64) hipLaunchKernelGGL(vq_distance_kernel<64>, grid, dim3(256), 0, stream, r, n, cb, K, d, minmax, ticket, mm_partial, slots, slot_count);
    else return fail(LCREC_EUNSUPPORTED, "vq_distances: e_dim=%d (supported: 16, 32, 64)", e);
    return check_launch("vq_distance_kernel");
}

static size_t sk_big_bytes(int64_t B, int K)
{
    const int64_t nblk = (B + SK_ROWS - 1) / SK_ROWS;
    return align_up((size_t)B * K * sizeof(float), 256) + align_up((size_t)B * K * sizeof(double), 256) +
           align_up((size_t)2 * nblk * K * sizeof(double), 256) + align_up((size_t)nblk * sizeof(double), 256) + 256 +
           (size_t)DIST_TICKET_MAX_BLOCKS * 2 * sizeof(unsigned);          // control words; (min, max) partials of the ticket form
}

// Size classes of a group of sz rows (see sinkhorn_assign): 0..3 = Q in LDS, allocation 16 / 32 / 64 / 128 KB (so that
// pairs and triples -- the bulk of a collision round -- run eight workgroups to a CU, and a 9-row group is not charged
// a 64-row allocation), SK_SLAB = Q in the workspace slab, one 1024-thread workgroup per group, SK_BATCH = batch-sized
// (persistent / multi-launch solver).
constexpr int64_t SKS_TINY = 2048;              // doubles: groups of <= 8 items at K = 256
constexpr int64_t SKS_SLAB_MAX_ROWS = 4096;     // a larger group is a training-batch-sized problem
constexpr int SK_SLAB = 4, SK_BATCH = 5;

static inline size_t sk_group_doubles(int64_t g, int K) { return (size_t)g * K + (size_t)((g + 1) & ~(int64_t)1) + K; }

static inline int sk_class(int64_t sz, int K)
{
    const int64_t q = sz * K;
    if (q <= SKS_TINY) return 0;
    if (q <= 2 * SKS_TINY) return 1;
    if (q <= 4 * SKS_TINY) return 2;
    if (q <= SKS_MAX) return 3;
    return sz <= SKS_SLAB_MAX_ROWS ? SK_SLAB : SK_BATCH;
}

// slab_ok: whether the mid-sized groups (class SK_SLAB) of this call run side by side, one workgroup each, or one
// after another as batch-sized problems.  One workgroup needs about 12 us per row at K = 256 for 50 iterations
// (measured alone: 64 rows 0.9 ms, 256 rows 3.0 ms, 400 rows 4.9 ms -- K of its 1024 threads do the divisions of a
// column); a batch-sized solve spreads the rows over 64-128 CUs and costs 0.4-0.55 ms whatever its size.  So a
// training step's single group of 256 or 2048 rows must take the batch path, and a collision round's hundred groups
// of 65..3000 rows must not (a few ms side by side, not 50+ ms in a row).
struct SkPlan { int64_t slab_doubles; int64_t biggest; int n_slab; bool slab_ok; };

static SkPlan sk_plan(int K, const int64_t *offs, int G)
{
    SkPlan p = {0, 0, 0, false};
    int64_t mid_max = 0, mid_biggest = 0;
    int n_mid = 0;
    for (int g = 0; g < G; ++g) {
        const int64_t sz = offs[g + 1] - offs[g];
        if (sz <= 0) continue;
        const int cls = sk_class(sz, K);
        if (cls == SK_SLAB) { p.slab_doubles += (int64_t)sk_group_doubles(sz, K); ++n_mid; if (sz > mid_max) mid_max = sz; }
        if (cls >= SK_SLAB && sz > mid_biggest) mid_biggest = sz;
        if (cls == SK_BATCH && sz > p.biggest) p.biggest = sz;
    }
    const double side_by_side_us = (double)mid_max * 12.0 * (K / 256.0) * ((n_mid + 255) / 256);
    p.slab_ok = n_mid > 0 && side_by_side_us <= 500.0 * n_mid;
    if (p.slab_ok) {
        p.n_slab = n_mid;
    } else {
        p.slab_doubles = 0;
        p.biggest = mid_biggest;            // the mid-sized groups go through the batch path too
    }
    return p;
}

size_t sinkhorn_workspace(int64_t n, int K, const int64_t *offs, int G)
{
    (void)n;
    const SkPlan p = sk_plan(K, offs, G);
    return align_up((size_t)3 * G * sizeof(int64_t), 256) + align_up((size_t)p.slab_doubles * sizeof(double), 256) +
           (p.biggest ? sk_big_bytes(p.biggest, K) : 0);
}

// ------------------------------------------------------------------------------------------
// Sinkhorn, training batch, ONE launch: every workgroup keeps its rows of Q in REGISTERS for the
// whole solve (a wave owns RW rows, a lane CPL = K/64 columns of each: 32 VGPRs of fp64), row sums
// are wave reductions, and only the per-workgroup column partials ([nblk][K] doubles) cross
// workgroups -- through global memory behind a grid barrier, once per iteration.  Same divisions in
// the same order as the multi-launch path, so the same bits.
//
// Exchange of the column sums, once per iteration, WITHOUT a barrier: the values are their own flags.  Every slot
// of the exchange buffers starts as a sentinel (all-ones, a NaN no arithmetic produces); writers use agent-scope relaxed
// atomic stores, readers agent-scope atomic loads, re-reading a slot until it is not the sentinel.  Two hand-overs:
//   A  workgroup X stores its K partials to part[it % 3][X][.];
//   B  the OWNER of column j (workgroup j % nblk) reads part[it % 3][0..nblk)[j], adds them in workgroup order (the
//      same order, hence the same bits, as before) and stores fin[it % 3][j];
//   C  every workgroup reads the K finished sums fin[it % 3][.].
// That is two store -> load trips through memory and ~4 KB of traffic per workgroup, against store, release fence,
// counter add, poll, acquire fence and 128 KB of loads per workgroup for the counter barrier this replaces (in-kernel
// stamps, tools/sk_stamp_probe.py: 31.7 k cycles per iteration, 9.9 k of them arithmetic).
// Re-arming: at the START of iteration `it` a workgroup resets its own slots of part[(it + 1) % 3] and fin[(it + 1) % 3]
// and waits for those stores (vmcnt) before it stores anything of `it`.  Those buffers last carried iteration it - 2,
// and whoever is in iteration `it` has read every sum of it - 1, each built from every workgroup's partials of it - 1,
// which each of them stored after it had finished reading it - 2: nobody reads those slots any more.  And whoever
// polls them for it + 1 has read sums of `it` that were built from this workgroup's partials of `it`, stored after
// the reset had reached memory: nobody sees a stale value.  (With two buffers the first argument fails.)
//
// The grand total before the first iteration still uses a grid barrier (guide: Guideline 16, counter form), which also
// publishes the sentinels: monotonic agent-scope counter zeroed by a memset
// node before the launch; lane 0 of each workgroup release-fences, adds 1, polls relaxed with
// s_sleep until all workgroups of this phase have arrived, acquire-fences, then the workgroup's
// barrier releases the other waves.  Every workgroup must be resident: the launcher only takes this
// path for <= SKP_MAX_BLOCKS workgroups (one per CU, half the chip) and every spin is bounded -- on
// timeout the kernel sets *flag and finishes (result invalid, reported by the host), it never hangs.
// ------------------------------------------------------------------------------------------
#ifdef LCREC_GEMM_STAMP
// diagnostic builds (make STAMP=1): cycles workgroup 0 spends in each part of an iteration, summed over iterations >= 1
__device__ unsigned long long g_sk_stamps[8];
#define SK_STAMP(slot)                                                                             \
    do {                                                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                 \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
            g_sk_stamps[slot] += t_ - sk_t_;                                                       \
            sk_t_ = t_;                                                                            \
        }                                                                                          \
    } while (0)
#define SK_STAMP_DECL unsigned long long sk_t_ = 0; if (blockIdx.x == 0 && threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sk_t_)::"memory")
#else
#define SK_STAMP(slot) do { } while (0)
#define SK_STAMP_DECL do { } while (0)
#endif
constexpr int SKP_THREADS = 512;
constexpr int SKP_WAVES = SKP_THREADS / 64;
constexpr int SKP_MAX_BLOCKS = 128;
// One poll = s_sleep 1 + an L2 load, 0.3-1 us: the limit is ~0.1 s of waiting for ONE hand-over.  A spinning thread also
// looks at *flag every 64 polls, so once any thread anywhere has timed out every other wait ends within microseconds and
// the whole grid drains in about one limit, not one limit per remaining hand-over (iters x 2 per thread).
constexpr unsigned SKP_SPIN_LIMIT = 200u * 1000u;
constexpr unsigned long long SKP_EMPTY = ~0ull;               // exchange slot not written yet

__device__ __forceinline__ void skp_put(double *slot, unsigned long long bits)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long skp_peek(const double *slot)
{
    return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the slot's value once it has one.  Bounded: on timeout, or when another thread's timeout is seen in *flag, *flag is
// set, `gave_up` becomes true and whatever is there is returned (the caller abandons the solve; outputs are poisoned).
__device__ __forceinline__ double skp_take(const double *slot, unsigned long long first, unsigned *flag, bool &gave_up)
{
    unsigned long long v = first;
    unsigned spins = 0;
    while (v == SKP_EMPTY) {
        __builtin_amdgcn_s_sleep(1);
        v = skp_peek(slot);
        ++spins;
        if (v == SKP_EMPTY && ((spins & 63u) == 0u) &&
            (spins > SKP_SPIN_LIMIT || __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gave_up = true;
            break;
        }
    }
    return __builtin_bit_cast(double, v);
}

struct SkPersist {
    const float *d;        // [B][K] fp32 distances
    double *col_part;      // [3][nblk][K] partials, then [3][K] finished sums
    double *tot_part;      // [nblk]
    const unsigned *minmax;
    unsigned *counter;     // grid-barrier arrivals (zeroed before launch)
    unsigned *flag;        // set to 1 on barrier timeout
    int64_t B;
    int K, nblk, iters;
    double eps;
    int64_t *idx_out;
    int64_t idx_stride;
};

#ifdef LCREC_GEMM_STAMP
#define SKP_BARRIER(c, t, f) skp_grid_barrier(c, t, f, sk_t_)
__device__ __forceinline__ void skp_grid_barrier(unsigned *counter, unsigned target, unsigned *flag, unsigned long long &sk_t_)
#else
#define SKP_BARRIER(c, t, f) skp_grid_barrier(c, t, f)
__device__ __forceinline__ void skp_grid_barrier(unsigned *counter, unsigned target, unsigned *flag)
#endif
{
    __syncthreads();
    SK_STAMP(2);
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SK_STAMP(3);
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if (((spins & 63u) == 0u) &&
                (spins > SKP_SPIN_LIMIT || __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        SK_STAMP(4);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SK_STAMP(5);
    }
    __syncthreads();
}

template <int CPL, int RW>
__global__ __launch_bounds__(SKP_THREADS) void sk_persistent_kernel(SkPersist p)
{
    constexpr int ROWS = RW * SKP_WAVES;             // rows per workgroup
    extern __shared__ __attribute__((aligned(16))) double skp_sm[];
    double *colsum = skp_sm;                          // [K]
    double *colacc = skp_sm + p.K;                    // [SKP_WAVES][K] (+ nblk: also the owner's gather buffer)
    __shared__ double wsum[SKP_WAVES];
    __shared__ double total_sh;
    __shared__ int abort_sh;                          // a thread of this workgroup gave up a wait: leave the iteration loop
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = p.K;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS + wave * RW;
    if (threadIdx.x == 0) abort_sh = 0;
    bool gave_up = false;
    const double Bd = (double)p.B, Kd = (double)K;
    const ExactDiv divB(Bd), divK(Kd);
    unsigned phase = 0;

    const float hi = ord2f(p.minmax[1]), lo = ord2f(p.minmax[0]);
    const float middle = (hi + lo) / 2.0f;
    const float amplitude = (hi - middle) + 1e-5f;

    double q[RW][CPL];
    double part = 0.0;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            q[r][c] = 0.0;
            if (row0 + r < p.B && j < K) {
                const float cen = (p.d[(row0 + r) * K + j] - middle) / amplitude;
                q[r][c] = exp(-(double)cen / p.eps);
                part += q[r][c];
            }
        }
    }
    part = wave_sum(part);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < SKP_WAVES; ++w) s += wsum[w];
        p.tot_part[blockIdx.x] = s;
    }
    double *const fin_base = p.col_part + (size_t)3 * p.nblk * K;                              // [3][K]
    const int owned = (int)blockIdx.x < K ? (K - (int)blockIdx.x + p.nblk - 1) / p.nblk : 0;   // columns X, X + nblk, ...
    for (int j = threadIdx.x; j < 3 * K; j += SKP_THREADS)       // this workgroup's slot in each of the three buffers
        skp_put(p.col_part + ((size_t)(j / K) * p.nblk + blockIdx.x) * K + j % K, SKP_EMPTY);
    for (int c = threadIdx.x; c < 3 * owned; c += SKP_THREADS)   // and the sums it owns
        skp_put(fin_base + (size_t)(c / owned) * K + blockIdx.x + (c % owned) * p.nblk, SKP_EMPTY);
    SK_STAMP_DECL;
    SKP_BARRIER(p.counter, (++phase) * (unsigned)p.nblk, p.flag);
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < p.nblk; ++b) s += p.tot_part[b];
        total_sh = s;
        if (__hip_atomic_load(p.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) abort_sh = 1;   // barrier timed out
    }
    __syncthreads();
    const double total = total_sh;

    for (int it = abort_sh ? p.iters : 0; it < p.iters; ++it) {
        SK_STAMP(7);
        if (it > 0) {                                            // re-arm the buffer of the NEXT iteration (see above)
            double *arm = p.col_part + ((size_t)((it + 1) % 3) * p.nblk + blockIdx.x) * K;
            for (int j = threadIdx.x; j < K; j += SKP_THREADS) skp_put(arm + j, SKP_EMPTY);
            double *arm_fin = fin_base + (size_t)((it + 1) % 3) * K + blockIdx.x;
            for (int c = threadIdx.x; c < owned; c += SKP_THREADS) skp_put(arm_fin + c * p.nblk, SKP_EMPTY);
        }
        double acc[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = 0.0;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            double rs = 0.0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = lane + 64 * c;
                if (row0 + r < p.B && j < K) {
                    double v = q[r][c];
                    if (it == 0) v = v / total;
                    else { v = v / colsum[j]; v = divK(v); }
                    q[r][c] = v;
                    rs += v;
                }
            }
            rs = wave_sum(rs);
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = lane + 64 * c;
                if (row0 + r < p.B && j < K) {
                    double v = q[r][c] / rs;
                    v = divB(v);
                    q[r][c] = v;
                    acc[c] += v;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            if (j < K) colacc[wave * K + j] = acc[c];
        }
        __syncthreads();
        SK_STAMP(0);
        double *out = p.col_part + ((size_t)(it % 3) * p.nblk + blockIdx.x) * K;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the re-arming stores have reached memory
        for (int j = threadIdx.x; j < K; j += SKP_THREADS) {
            double s = 0.0;
            for (int w = 0; w < SKP_WAVES; ++w) s += colacc[w * K + j];
            skp_put(out + j, __builtin_bit_cast(unsigned long long, s));
        }
        __syncthreads();                                         // colacc has been read: it is the gather buffer now
        SK_STAMP(1);
        // B: the sums this workgroup owns, partials added in workgroup order
        const double *pb = p.col_part + (size_t)(it % 3) * p.nblk * K + blockIdx.x;
        for (int i = threadIdx.x; i < owned * p.nblk; i += SKP_THREADS) {
            const double *slot = pb + (size_t)(i % p.nblk) * K + (i / p.nblk) * p.nblk;
            colacc[i] = skp_take(slot, skp_peek(slot), p.flag, gave_up);
        }
        __syncthreads();
        SK_STAMP(2);
        double *fin = fin_base + (size_t)(it % 3) * K;
        for (int c = threadIdx.x; c < owned; c += SKP_THREADS) {
            const double *g = colacc + c * p.nblk;
            double s = 0.0;
            int b = 0;
            for (; b + 16 <= p.nblk; b += 16) {                  // 16 LDS reads in flight, then the adds in workgroup order
                double v[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = g[b + t];
#pragma unroll
                for (int t = 0; t < 16; ++t) s += v[t];
            }
            for (; b < p.nblk; ++b) s += g[b];
            skp_put(fin + blockIdx.x + c * p.nblk, __builtin_bit_cast(unsigned long long, s));
        }
        SK_STAMP(3);
        // C: all K sums
        for (int j = threadIdx.x; j < K; j += SKP_THREADS) colsum[j] = skp_take(fin + j, skp_peek(fin + j), p.flag, gave_up);
        if (gave_up) abort_sh = 1;
        __syncthreads();
        SK_STAMP(6);
        if (abort_sh) break;                                     // uniform: read after the barrier every thread passed
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        double best = -1.0;
        int bj = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            if (row0 + r < p.B && j < K) {
                double v = q[r][c] / colsum[j];
                v = divK(v);
                v = v * Bd;
                if (v > best) { best = v; bj = j; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double b2 = __shfl_xor(best, o, 64);
            const int j2 = __shfl_xor(bj, o, 64);
            if (b2 > best || (b2 == best && j2 < bj)) { best = b2; bj = j2; }
        }
        // a barrier timeout anywhere poisons the output (-1) so the caller fails loudly instead of training on garbage
        if (lane == 0 && row0 + r < p.B) {
            const unsigned bad = __hip_atomic_load(p.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.idx_out[(row0 + r) * p.idx_stride] = bad ? -1 : bj;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same solve in SCALING form.  The reference's loop (layers.py:85-108) only ever multiplies whole rows and whole
// columns of Q = exp(-d/eps): after any number of steps Q_ij = a_i E_ij b_j, and the two normalisations of an iteration are
//     a_i = 1 / (B * sum_j E_ij b_j)          (Q /= Q.sum(1); Q /= B  -- the old a_i cancels, and so does the grand total)
//     b_j = 1 / (K * sum_i a_i E_ij)          (Q /= Q.sum(0); Q /= K  -- the old b_j cancels)
// with the final argmax over E_ij b_j (a_i and the trailing `Q *= B` are per-row constants).  E stays in registers untouched;
// an iteration is two fused multiply-add sweeps and B + K divisions instead of 2 B K of them (16 per lane and iteration
// before, 6.5 k of an iteration's 16-19 k cycles), no grand total and so no grid barrier, and with the arithmetic gone the rows
// fit on 16 workgroups instead of 64-128, which makes the ONE-hop exchange cheap: every workgroup publishes its K column
// partials (values are their own flags, three rotating buffers, exactly as above) and every workgroup adds all nblk of
// them, in workgroup order, itself -- one store -> load trip through memory per iteration instead of two.
// Values differ from the in-place form by rounding only (~1e-15 relative after 50 iterations); assignments can differ
// where the two best entries of a row agree to that precision -- the tests' margin is 1e-9.  Exchange slots are armed by a
// launch before this kernel (sk_ctrl_init_kernel, with the control words), since nothing inside it orders "armed" before the first poll.
// Measured, 1024 x 256, 50 iterations (rocprofv3 / in-kernel stamps, tools/sk_stamp_probe.py): in-place form 342 us, 16.4 k
// cycles per iteration on 64 workgroups; this form 8.4 k cycles on 16 workgroups (175 us): the hand-over 3.1 k, row sums 1.6 k,
// scales + column accumulation 1.0 k, publish 1.0 k, column scales 0.9 k, barriers and loop 0.9 k.  8 / 32 workgroups: the same
// within 3 % (fewer partials to poll against more rows per wave).  The collision-group kernels keep the in-place form: their
// assignments go into the index file, which is compared byte for byte with the reference's.
// ------------------------------------------------------------------------------------------
struct SkScale {
    const float *d;        // [B][K] fp32 distances
    double *part;          // [3][nblk][K] column partials
    const unsigned *minmax;
    unsigned *flag;        // set to 1 when a wait timed out
    int64_t B;
    int K, nblk, iters;
    double eps;
    int64_t *idx_out;
    int64_t idx_stride;
    int replicas;          // > 1: the XCD-local form (below); part then holds `replicas` sets of buffers
    unsigned *ranks;       // [16] arrival counters per XCC_ID, zero at launch
    unsigned *done;        // set by the first workgroup to finish a solve
};

// ---- XCD-local exchange (replicas > 1).  The hand-over above is an agent-scope store (sc1: written through the XCD's L2 to the
// memory side, because the eight L2s are not coherent with each other) and an agent-scope load that fetches it from there:
// ~1 500 cycles per hop between XCDs, ~1 200 even when both workgroups sit on the same XCD.  Between two CUs of ONE XCD a plain
// store (it stops in the shared L2) and an sc1 load (misses the CU's L1, hits that L2) take ~650 (tools/xcd_pingpong.hip: round
// trips of 1 300 cycles against 2 400 / 3 050).  Workgroups are dealt to the XCDs round-robin, so a launch of 8 x nblk workgroups
// puts nblk on each XCD -- and the solve is small enough to be done EIGHT times side by side on CUs that idle anyway: every XCD
// runs the whole problem among its own workgroups, exchanging through its own L2, and all of them write the same assignments.
// Nothing depends on the round-robin: a workgroup reads its XCC_ID, takes a rank from that XCD's arrival counter, and leaves if
// the rank is beyond nblk; an XCD that got fewer than nblk never finishes an iteration, and its workgroups leave quietly once
// any complete set has set `done` (they poll it while they wait).  Some XCD always gets at least nblk of 8 x nblk.
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u; }      // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ void skp_put_local(double *slot, unsigned long long bits)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Cross-lane moves of a double without the LDS crossbar.  row_partner<LEV>: the value of the lane this one is paired with at
// level LEV inside its row of 16 lanes -- an involution that flips lane bit LEV (3: row_mirror, 2: row_half_mirror, 1 / 0:
// quad permutes), which is all a sum reduction needs.  swap16_sum / swap32_sum: v + (v of the lane 16 / 32 away), by the gfx950
// v_permlane16_swap / v_permlane32_swap.
template <int LEV>
__device__ __forceinline__ double row_partner(double v)
{
    constexpr int CTRL = LEV == 3 ? 0x140 : LEV == 2 ? 0x141 : LEV == 1 ? 0x4E : 0xB1;
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double swap16_sum(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u32x2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double swap32_sum(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u32x2 l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
// f(level 3, RW/2), f(level 2, RW/4), ...: the LOG halving levels of the row-sum reduction, as compile-time constants
template <int N> struct SkInt { static constexpr int value = N; };
template <int LOG, int I = 0, class F>
__device__ __forceinline__ void static_for_levels(F &&f)
{
    if constexpr (I < LOG) {
        f(SkInt<3 - I>{}, SkInt<((1 << LOG) >> (I + 1))>{});
        static_for_levels<LOG, I + 1>(f);
    }
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane)      // wave-uniform result (two v_readlane_b32)
{
    const unsigned long long bits = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)bits, src_lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(bits >> 32), src_lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

template <int CPL, int RW>
__global__ __launch_bounds__(SKP_THREADS) void sk_scaling_kernel(SkScale p)
{
    constexpr int ROWS = RW * SKP_WAVES;             // rows per workgroup
    extern __shared__ __attribute__((aligned(16))) double sks_sm[];
    const int K = p.K;
    double *colacc = sks_sm;                          // [SKP_WAVES][K]; after the publish: the gather buffer [split][K]
    double *bsh = sks_sm + (size_t)SKP_WAVES * K;     // [K] the new column scales
    __shared__ int abort_sh, bid_sh, set_sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool local = p.replicas > 1;
    if (threadIdx.x == 0) {
        abort_sh = 0;
        bid_sh = (int)blockIdx.x;
        set_sh = 0;
        if (local) {
            const unsigned x = xcc_id();
            set_sh = (int)x;
            bid_sh = (int)__hip_atomic_fetch_add(p.ranks + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const int bid = bid_sh;
    if (bid >= p.nblk || set_sh >= p.replicas) return;        // more than nblk on this XCD: not needed
    double *const part = p.part + (size_t)set_sh * 3 * p.nblk * p.K;
    const int64_t row0 = (int64_t)bid * ROWS + wave * RW;
    bool gave_up = false, quiet = false;
    const double Bd = (double)p.B, Kd = (double)K;
    const float hi = ord2f(p.minmax[1]), lo = ord2f(p.minmax[0]);
    const float middle = (hi + lo) / 2.0f;
    const float amplitude = (hi - middle) + 1e-5f;

    double e[RW][CPL], b[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) b[c] = 1.0;
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            e[r][c] = 0.0;                            // rows past B and columns past K: zeros, they add nothing anywhere
            if (row0 + r < p.B && j < K) {
                const float cen = (p.d[(row0 + r) * K + j] - middle) / amplitude;
                e[r][c] = exp(-(double)cen / p.eps);
            }
        }
    const int split = K >= SKP_THREADS ? 1 : SKP_THREADS / K;        // threads per column in the gather (K a multiple of 64)
    const int chunk = (p.nblk + split - 1) / split;
    constexpr int LOG_RW = RW == 16 ? 4 : RW == 8 ? 3 : RW == 4 ? 2 : RW == 2 ? 1 : 0, LOWBITS = 4 - LOG_RW;
    static_assert((1 << LOG_RW) == RW, "RW is a power of two");
    const int my_row = (lane & 15) >> LOWBITS;       // the row whose sum this lane ends up holding (see below)
    __syncthreads();
    SK_STAMP_DECL;

    for (int it = 0; it < p.iters; ++it) {
        SK_STAMP(7);
        if (it >= 2) {                                           // re-arm the buffer of the NEXT iteration (it carried it - 2)
            double *arm = part + ((size_t)((it + 1) % 3) * p.nblk + bid) * K;
            for (int j = threadIdx.x; j < K; j += SKP_THREADS) {
                if (local) skp_put_local(arm + j, SKP_EMPTY);
                else skp_put(arm + j, SKP_EMPTY);
            }
        }
        // row scales a_i = 1 / (B * sum_j E_ij b_j).  The RW row sums of a wave are reduced TOGETHER, in registers: inside each
        // row of 16 lanes LOG_RW halving exchanges (lanes with the level's bit set keep the upper half of the values, the
        // others the lower half: RW - 1 exchanges in all instead of 4 RW) leave lane l with the sum of value (l & 15) >> (4 -
        // LOG_RW) over some of its row's lanes, plain butterflies finish the row, and two swaps across the four rows finish the
        // wave.  All of it DPP row operations and the gfx950 permlane swaps: a ds_bpermute (what __shfl_xor compiles to) is a
        // ~250-cycle round trip through the LDS crossbar, and the first form of this loop was a chain of them -- 3 100 of an
        // iteration's 10 400 cycles at 8 rows per wave, 9 200 at 16.  Then ONE division sequence serves all rows (lane = row) and
        // the scales come back as wave-uniform values by readlane.
        double s[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            s[r] = 0.0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) s[r] = fma(e[r][c], b[c], s[r]);
        }
        static_for_levels<LOG_RW>([&](auto lev_c, auto n_c) {                // halving levels: lane bit 3, 2, ...
            constexpr int LEV = decltype(lev_c)::value, n = decltype(n_c)::value;
            const bool upper = (lane & (1 << LEV)) != 0;
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const double keep = upper ? s[i + n] : s[i];
                const double send = upper ? s[i] : s[i + n];
                s[i] = keep + row_partner<LEV>(send);
            }
        });
        if constexpr (LOG_RW < 4) s[0] += row_partner<3 - LOG_RW>(s[0]);     // the row's remaining levels: plain butterflies
        if constexpr (LOG_RW < 3) s[0] += row_partner<2 - LOG_RW>(s[0]);
        if constexpr (LOG_RW < 2) s[0] += row_partner<1 - LOG_RW>(s[0]);
        s[0] = swap16_sum(s[0]);
        s[0] = swap32_sum(s[0]);
        SK_STAMP(0);
        const double a_mine = row0 + my_row < p.B ? 1.0 / (Bd * s[0]) : 0.0;
        double acc[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = 0.0;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const double a = readlane_f64(a_mine, r << LOWBITS);
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = fma(a, e[r][c], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            if (j < K) colacc[wave * K + j] = acc[c];
        }
        __syncthreads();
        SK_STAMP(1);
        double *out = part + ((size_t)(it % 3) * p.nblk + bid) * K;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the re-arming stores have reached memory
        for (int j = threadIdx.x; j < K; j += SKP_THREADS) {
            double t = 0.0;
            for (int w = 0; w < SKP_WAVES; ++w) t += colacc[w * K + j];
            if (local) skp_put_local(out + j, __builtin_bit_cast(unsigned long long, t));
            else skp_put(out + j, __builtin_bit_cast(unsigned long long, t));
        }
        __syncthreads();                                         // colacc has been read: it is the gather buffer now
        SK_STAMP(2);
        // every workgroup's partial of every column, added in workgroup order (chunk by chunk when several threads share a column)
        const double *src = part + (size_t)(it % 3) * p.nblk * K;
        for (int i = threadIdx.x; i < K * split; i += SKP_THREADS) {
            const int j = i % K, h = i / K;
            const int b0 = h * chunk, b1 = b0 + chunk < p.nblk ? b0 + chunk : p.nblk;
            double t = 0.0;
            for (int bb = b0; bb < b1; bb += 8) {                // eight polls in flight, then the adds in order
                // Every round re-polls ALL slots that were still empty, side by side.  (First form: after the eight first polls,
                // each empty slot was waited for on its own, one load in flight -- the workgroups publish at about the same time,
                // so all eight are usually empty at the first poll, and the second slot's wait only started when the first's
                // ended: up to eight dependent round trips where one or two do.)
                unsigned long long v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = bb + u < b1 ? skp_peek(src + (size_t)(bb + u) * K + j) : 0ull;
                unsigned spins = 0;
                for (;;) {
                    bool pending = false;
#pragma unroll
                    for (int u = 0; u < 8; ++u) pending = pending || v[u] == SKP_EMPTY;
                    if (!pending) break;
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (v[u] == SKP_EMPTY) v[u] = skp_peek(src + (size_t)(bb + u) * K + j);
                    if ((++spins & 63u) == 0u) {
                        if (local && __hip_atomic_load(p.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { gave_up = true; quiet = true; break; }
                        if (spins > SKP_SPIN_LIMIT || __hip_atomic_load(p.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                            __hip_atomic_store(p.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            gave_up = true;
                            break;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (bb + u < b1) t += __builtin_bit_cast(double, v[u]);
            }
            colacc[h * K + j] = t;
        }
        if (gave_up) abort_sh = quiet ? 2 : 1;
        SK_STAMP(3);
        __syncthreads();
        SK_STAMP(4);
        for (int j = threadIdx.x; j < K; j += SKP_THREADS) {
            double t = 0.0;
            for (int h = 0; h < split; ++h) t += colacc[h * K + j];
            bsh[j] = 1.0 / (Kd * t);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            if (j < K) b[c] = bsh[j];
        }
        SK_STAMP(5);
        if (abort_sh) break;                                     // uniform: read after a barrier every thread passed
    }
    if (abort_sh == 2) return;                                   // another XCD's set finished the solve: it writes the assignments
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        double best = -1.0;
        int bj = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            if (row0 + r < p.B && j < K) {
                const double v = e[r][c] * b[c];
                if (v > best) { best = v; bj = j; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double b2 = __shfl_xor(best, o, 64);
            const int j2 = __shfl_xor(bj, o, 64);
            if (b2 > best || (b2 == best && j2 < bj)) { best = b2; bj = j2; }
        }
        if (lane == 0 && row0 + r < p.B) {
            const unsigned bad = __hip_atomic_load(p.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.idx_out[(row0 + r) * p.idx_stride] = bad ? -1 : bj;
        }
    }
    // (a workgroup that got here has seen every member of its set publish the last iteration: the set is complete and all of it
    // will write its rows; incomplete sets may stop waiting)
    if (local && threadIdx.x == 0) __hip_atomic_store(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static size_t sks_lds_bytes(int K) { return (size_t)(SKP_WAVES + 1) * K * sizeof(double); }

template <int CPL, int RW>
static bool launch_sks(const SkScale &p, hipStream_t stream)
{
    // every workgroup must be resident (the hand-overs assume it): at most 64 of them, one per CU is always possible on an
    // idle device; other work holding CUs is what the bounded spins are for
    const size_t lds = sks_lds_bytes(p.K);
    auto kern = sk_scaling_kernel<CPL, RW>;
    if (lds > 48 * 1024) {
        static thread_local size_t granted = 0;
        if (granted < lds) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            granted = lds;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nblk * (p.replicas > 1 ? p.replicas : 1))), dim3(SKP_THREADS), lds, stream, p);
    return true;
}

template <int CPL>
static bool launch_sks_rw(const SkScale &p, int rw, hipStream_t stream)
{
    if constexpr (CPL > 1 && CPL <= 4) { if (rw == 16) return launch_sks<CPL, 16>(p, stream); }
    if constexpr (CPL <= 8) { if (rw == 8) return launch_sks<CPL, 8>(p, stream); }
    if (rw == 4) return launch_sks<CPL, 4>(p, stream);
    return launch_sks<CPL, 2>(p, stream);
}

static size_t skp_lds_bytes(int K, int nblk) { return ((size_t)(1 + SKP_WAVES) * K + nblk) * sizeof(double); }

// Whether all `nblk` workgroups of sk_persistent_kernel<CPL, RW> can be resident at once on the current device: the
// kernel's hand-overs assume it (a workgroup that has not started cannot publish its sums).  Occupancy per CU for the
// actual LDS size, as the runtime computes it from the kernel's registers and LDS, times the CU count; cached per
// (variant, K, nblk).  Other work on the device can still hold CUs -- that case is what the bounded spins are for.
template <int CPL, int RW>
static bool skp_fits(int K, int nblk)
{
    static thread_local int cached_K = -1, cached_nblk = -1;
    static thread_local bool cached = false;
    if (K == cached_K && nblk == cached_nblk) return cached;
    const size_t lds = skp_lds_bytes(K, nblk);
    auto kern = sk_persistent_kernel<CPL, RW>;
    bool ok = true;
    if (lds > 48 * 1024)
        ok = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    int per_cu = 0, cus = 0, dev = 0;
    ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, SKP_THREADS, lds) == hipSuccess &&
         hipGetDevice(&dev) == hipSuccess &&
         hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess;
    (void)hipGetLastError();
    cached = ok && (int64_t)per_cu * cus >= nblk;
    cached_K = K;
    cached_nblk = nblk;
    return cached;
}

template <int CPL, int RW>
static bool launch_skp(const SkPersist &p, hipStream_t stream)
{
    if (!skp_fits<CPL, RW>(p.K, p.nblk)) return false;           // caller falls back to the multi-launch solver
    const size_t lds = skp_lds_bytes(p.K, p.nblk);
    hipLaunchKernelGGL((sk_persistent_kernel<CPL, RW>), dim3((unsigned)p.nblk), dim3(SKP_THREADS), lds, stream, p);
    return true;
}

// control words {ord(min) = 0xffffffff, ord(max) = 0, -, -, barrier counter = 0, timeout flag = 0} and, for the scaling-form
// solver, the sentinels of its exchange slots (`count` doubles; nothing inside that kernel orders "armed" before the first poll)
__global__ __launch_bounds__(256) void sk_ctrl_init_kernel(unsigned *ctrl, double *slots, int64_t count)
{
    if (blockIdx.x == 0 && threadIdx.x < 32) ctrl[threadIdx.x] = threadIdx.x == 0 ? 0xffffffffu : 0u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
        reinterpret_cast<unsigned long long *>(slots)[i] = SKP_EMPTY;
}

static int sinkhorn_big(const float *r, int64_t B, int e, const float *cb, int K, double eps, int iters,
                        int64_t *idx_out, int64_t idx_stride, char *ws, unsigned *ticket, hipStream_t stream)
{
    if (K > 64 * SK_MAXC) return fail(LCREC_EUNSUPPORTED, "sinkhorn: K=%d > %d", K, 64 * SK_MAXC);
    SkBig p;
    const int64_t nblk = (B + SK_ROWS - 1) / SK_ROWS;
    float *d = reinterpret_cast<float *>(ws);
    ws += align_up((size_t)B * K * sizeof(float), 256);
    p.Q = reinterpret_cast<double *>(ws);
    ws += align_up((size_t)B * K * sizeof(double), 256);
    p.col_part = reinterpret_cast<double *>(ws);
    ws += align_up((size_t)2 * nblk * K * sizeof(double), 256);
    p.tot_part = reinterpret_cast<double *>(ws);
    ws += align_up((size_t)nblk * sizeof(double), 256);
    unsigned *minmax = reinterpret_cast<unsigned *>(ws);
    p.d = d; p.minmax = minmax; p.B = B; p.K = K; p.nblk = (int)nblk; p.eps = eps;
    // one launch, scaling form (sk_scaling_kernel): rows per workgroup 16 / 32 / 64 / 128 (RW = 2 .. 16 rows per wave, at most
    // 64 doubles of E per lane), the smallest that needs no more than 16 workgroups -- fewer partials to add, and the arithmetic
    // no longer wants more CUs.  LCREC_SINKHORN_SCALING=0: the in-place form below.
    static const bool allow_scaling = [] { const char *e = getenv("LCREC_SINKHORN_SCALING"); return !e || atoi(e) != 0; }();
    static const int nblk_target = [] { const char *e = getenv("LCREC_SK_BLOCKS"); return e ? atoi(e) : 16; }();
    const int cpl_s = K / 64, cpl_t = cpl_s <= 1 ? 1 : cpl_s <= 2 ? 2 : cpl_s <= 4 ? 4 : cpl_s <= 8 ? 8 : 16;
    const int rw_cap = cpl_t == 1 ? 8 : cpl_t <= 4 ? 16 : (cpl_t <= 8 ? 8 : 4);       // (<1, 16> spills: not instantiated)
    int rw = 2;
    while (rw < rw_cap && (B + 8 * rw - 1) / (8 * rw) > nblk_target) rw *= 2;
    const int64_t nblk_s = (B + 8 * rw - 1) / (8 * rw);
    // (the exchange buffers live in the Q region, [B][K] doubles, which this path does not use otherwise)
    const bool scaling = allow_scaling && iters >= 1 && K % 64 == 0 && K <= 1024 && nblk_s <= 64 && 3 * nblk_s <= B;
    // control words (and the scaling form's slot sentinels) by a kernel, not by memset nodes: inside a captured hipGraph
    // (engine.py) the two memsets were seen to take effect late -- replays found the previous solve's flag / a counter reset
    // under a running barrier
    // With a ticket the distance launch does all of that itself (its last workgroup writes the control words).
    // XCD-local exchange (sk_scaling_kernel): eight sets of buffers when they fit the region.  LCREC_SK_LOCAL=0: one set, agent scope.
    static const bool allow_local = [] { const char *e = getenv("LCREC_SK_LOCAL"); return !e || atoi(e) != 0; }();
    const int replicas = scaling && allow_local && (int64_t)8 * 3 * nblk_s <= B && 8 * nblk_s <= 256 ? 8 : 1;
    const int64_t slot_count = scaling ? (int64_t)replicas * 3 * nblk_s * K : (int64_t)0;
    const int64_t dist_blocks = ((B + DIST_ITEMS - 1) / DIST_ITEMS) * ((K + 255) / 256);
    int rc;
    if (ticket && dist_blocks <= DIST_TICKET_MAX_BLOCKS) {
        rc = vq_distances(r, B, e, cb, K, d, minmax, stream, ticket, minmax + 64, reinterpret_cast<unsigned long long *>(p.Q), slot_count);
    } else {
        hipLaunchKernelGGL(sk_ctrl_init_kernel, dim3(scaling ? 64 : 1), dim3(256), 0, stream, minmax, p.Q, slot_count);
        if (int rc0 = check_launch("sk_ctrl_init_kernel")) return rc0;
        rc = vq_distances(r, B, e, cb, K, d, minmax, stream);
    }
    if (rc) return rc;
    if (iters == 0) return fail(LCREC_EUNSUPPORTED, "sinkhorn: iters must be >= 1");
    TraceScope trace(K_SINKHORN, stream);

    if (scaling) {
        SkScale q;
        q.d = d; q.part = p.Q; q.minmax = minmax; q.flag = minmax + 5;
        q.B = B; q.K = K; q.nblk = (int)nblk_s; q.iters = iters; q.eps = eps;
        q.idx_out = idx_out; q.idx_stride = idx_stride;
        q.replicas = replicas; q.done = minmax + 8; q.ranks = minmax + 16;
        bool launched;
        if (cpl_t == 1) launched = launch_sks_rw<1>(q, rw, stream);
        else if (cpl_t == 2) launched = launch_sks_rw<2>(q, rw, stream);
        else if (cpl_t == 4) launched = launch_sks_rw<4>(q, rw, stream);
        else if (cpl_t == 8) launched = launch_sks_rw<8>(q, rw, stream);
        else launched = launch_sks_rw<16>(q, rw, stream);
        if (launched) return check_launch("sk_scaling_kernel");
    }
    // one-launch register-resident path when every workgroup can be resident (see sk_persistent_kernel)
    const int cpl = (K + 63) / 64;
    // K = 256 (the reference's codebook size): two rows per wave instead of four when that still fits 128 workgroups
    // (B <= 2048) -- twice the CUs on the divisions, 22.2 k -> 18.9 k cycles per iteration.  LCREC_SK_RW=4 disables.
    static const int rw_env = [] { const char *e = getenv("LCREC_SK_RW"); return e ? atoi(e) : 2; }();
    const bool rw1 = rw_env == 1 && cpl == 4 && (B + 7) / 8 <= SKP_MAX_BLOCKS;           // LCREC_SK_RW=1 (tuning): 8 rows per workgroup
    const bool rw2 = !rw1 && rw_env <= 2 && cpl == 4 && (B + 15) / 16 <= SKP_MAX_BLOCKS;
    const int rows_p = rw1 ? 8 : rw2 ? 16 : cpl <= 4 ? 32 : (cpl <= 8 ? 16 : 8);
    const int64_t nblk_p = (B + rows_p - 1) / rows_p;
    static const bool allow_persistent = [] { const char *e = getenv("LCREC_SINKHORN_PERSISTENT"); return !e || atoi(e) != 0; }();
    if (allow_persistent && nblk_p <= SKP_MAX_BLOCKS && nblk_p <= nblk * 4 && (3 * (int64_t)K + 1) * nblk_p + 3 * (int64_t)K <= B * (int64_t)K) {
        SkPersist q;
        // the Q region ([B][K] doubles) is unused on this path and holds the exchange buffers:
        // 3*nblk_p*K partials + 3*K sums + nblk_p totals (checked above)
        q.d = d; q.col_part = p.Q; q.tot_part = p.Q + (size_t)3 * nblk_p * K + (size_t)3 * K; q.minmax = minmax;
        q.counter = minmax + 4; q.flag = minmax + 5;
        q.B = B; q.K = K; q.nblk = (int)nblk_p; q.iters = iters; q.eps = eps;
        q.idx_out = idx_out; q.idx_stride = idx_stride;
        bool launched;
        if (cpl <= 1) launched = launch_skp<1, 4>(q, stream);
        else if (cpl <= 2) launched = launch_skp<2, 4>(q, stream);
        else if (rw1) launched = launch_skp<4, 1>(q, stream);
        else if (rw2) launched = launch_skp<4, 2>(q, stream);
        else if (cpl <= 4) launched = launch_skp<4, 4>(q, stream);
        else if (cpl <= 8) launched = launch_skp<8, 2>(q, stream);
        else launched = launch_skp<16, 1>(q, stream);
        if (launched) return check_launch("sk_persistent_kernel");
    }
    hipLaunchKernelGGL(sk_init_kernel, dim3((unsigned)nblk), dim3(SK_THREADS), 0, stream, p);
    const size_t lds_iter = (size_t)(1 + SK_THREADS / 64) * K * sizeof(double);
    int src = 0;
    for (int t = 0; t < iters; ++t) {
        hipLaunchKernelGGL(sk_iter_kernel, dim3((unsigned)nblk), dim3(SK_THREADS), lds_iter, stream, p, t == 0 ? 0 : 1, src);
        src ^= 1;
    }
    if (iters == 0) return fail(LCREC_EUNSUPPORTED, "sinkhorn: iters must be >= 1");
    hipLaunchKernelGGL(sk_final_kernel, dim3((unsigned)nblk), dim3(SK_THREADS), (size_t)K * sizeof(double), stream, p, src,
                       idx_out, idx_stride);
    return check_launch("sinkhorn kernels");
}

template <int E, bool GLOBALQ>
static int launch_sk_small(const float *r, const float *cb, int K, const int64_t *triples_dev, int G, int maxg, double eps,
                           int iters, int64_t *idx_out, int64_t idx_stride, double *qslab, hipStream_t stream)
{
    constexpr int THREADS = GLOBALQ ? 1024 : SKS_THREADS;
    const size_t lds = GLOBALQ ? 0 : sk_group_doubles(maxg, K) * sizeof(double);
    auto kern = sk_small_kernel<E, GLOBALQ, THREADS>;
    if (lds > 48 * 1024) {
        hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (he != hipSuccess) return fail(LCREC_EHIP, "sinkhorn: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(he));
    }
    TraceScope trace(GLOBALQ ? K_SINKHORN_SLAB : (int64_t)maxg * K <= SKS_TINY ? K_SINKHORN_TINY : K_SINKHORN_SMALL, stream);
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(THREADS), lds, stream, r, cb, K, triples_dev, eps, iters, idx_out,
                       idx_stride, qslab);
    return check_launch("sk_small_kernel");
}

template <bool GLOBALQ>
static int launch_sk_small_e(int e, const float *r, const float *cb, int K, const int64_t *triples_dev, int G, int maxg,
                             double eps, int iters, int64_t *idx_out, int64_t idx_stride, double *qslab, hipStream_t stream)
{
    if (e == 16) return launch_sk_small<16, GLOBALQ>(r, cb, K, triples_dev, G, maxg, eps, iters, idx_out, idx_stride, qslab, stream);
    if (e == 32) return launch_sk_small<32, GLOBALQ>(r, cb, K, triples_dev, G, maxg, eps, iters, idx_out, idx_stride, qslab, stream);
    return launch_sk_small<64, GLOBALQ>(r, cb, K, triples_dev, G, maxg, eps, iters, idx_out, idx_stride, qslab, stream);
}

struct SmallTable { static constexpr size_t CAP = 12; int64_t v[CAP]; };
__global__ void write_table_kernel(int64_t *dst, SmallTable t, int count)
{
    if ((int)threadIdx.x < count) dst[threadIdx.x] = t.v[threadIdx.x];
}

int sinkhorn_assign(const float *r, int64_t n, int e, const float *cb, int K, const int64_t *offs, int G, double eps,
                    int iters, int64_t *idx_out, int64_t idx_stride, void *workspace, size_t workspace_bytes,
                    lcrec_context *ctx, unsigned *ticket, hipStream_t stream)
{
    if (n == 0 || G == 0) return LCREC_OK;
    if (!r || !cb || !offs || !idx_out) return fail(LCREC_EINVAL, "sinkhorn_assign: NULL pointer");
    if (e != 16 && e != 32 && e != 64) return fail(LCREC_EUNSUPPORTED, "sinkhorn_assign: e_dim=%d (supported: 16, 32, 64)", e);
    if (G < 0 || K < 1 || iters < 1 || !(eps > 0)) return fail(LCREC_EINVAL, "sinkhorn_assign: bad G/K/iters/eps");
    if (G == 0) return LCREC_OK;
    if (int rc = check_context(ctx, "sinkhorn_assign")) return rc;
    if (offs[0] < 0 || offs[G] > n) return fail(LCREC_EINVAL, "sinkhorn_assign: group offsets outside [0, n]");
    for (int g = 0; g < G; ++g)
        if (offs[g + 1] < offs[g]) return fail(LCREC_EINVAL, "sinkhorn_assign: group offsets not ascending");
    const size_t need = sinkhorn_workspace(n, K, offs, G);
    if (!workspace || workspace_bytes < need)
        return fail(LCREC_EWORKSPACE, "sinkhorn_assign: workspace %zu B < required %zu B", workspace_bytes, need);
    const SkPlan plan = sk_plan(K, offs, G);
    char *ws = reinterpret_cast<char *>(workspace);
    int64_t *triples_dev = reinterpret_cast<int64_t *>(ws);
    ws += align_up((size_t)3 * G * sizeof(int64_t), 256);
    double *qslab = reinterpret_cast<double *>(ws);
    ws += align_up((size_t)plan.slab_doubles * sizeof(double), 256);

    // Size classes, one launch each (see sk_class): four LDS allocations, then the slab; anything larger is a
    // batch-sized problem (below).
    constexpr int NCLS = SK_SLAB + 1;
    std::vector<int64_t> triples;
    triples.reserve((size_t)3 * G);
    int count[NCLS] = {0}, maxg[NCLS] = {0};
    for (int cls = 0; cls < NCLS; ++cls) {
        int64_t slab = 0;
        for (int g = 0; g < G; ++g) {
            const int64_t sz = offs[g + 1] - offs[g];
            if (sz <= 0 || sk_class(sz, K) != cls || (cls == SK_SLAB && !plan.slab_ok)) continue;
            triples.push_back(offs[g]);
            triples.push_back(offs[g + 1]);
            triples.push_back(cls == SK_SLAB ? slab : 0);
            if (cls == SK_SLAB) slab += (int64_t)sk_group_doubles(sz, K);
            ++count[cls];
            if (sz > maxg[cls]) maxg[cls] = (int)sz;
        }
    }
    if (!triples.empty()) {
        // Group table -> device.  With a context it goes through the context's pinned ring (the copy reads pinned memory
        // asynchronously; the slot is reused RING calls later), so the call returns without waiting; without one the
        // table is a host temporary and the copy has to be waited for.
        const size_t tbytes = sizeof(int64_t) * triples.size();
        hipError_t he;
        if (triples.size() <= SmallTable::CAP) {
            // a handful of groups (a small training batch is ONE): the table travels as a kernel argument -- no host
            // buffer whose lifetime matters, no wait, and safe to capture in a hipGraph (a captured copy would re-read a
            // host address on every replay)
            SmallTable tb;
            for (size_t i = 0; i < triples.size(); ++i) tb.v[i] = triples[i];
            hipLaunchKernelGGL(write_table_kernel, dim3(1), dim3(64), 0, stream, triples_dev, tb, (int)triples.size());
            he = hipGetLastError();
        } else if (ctx) {
            int slot = -1;
            void *pin = ctx->ring_acquire(tbytes, &slot);
            if (!pin) return LCREC_EHIP;
            memcpy(pin, triples.data(), tbytes);
            he = hipMemcpyAsync(triples_dev, pin, tbytes, hipMemcpyHostToDevice, stream);
            ctx->ring_release(slot, stream);
        } else {
            he = hipMemcpyAsync(triples_dev, triples.data(), tbytes, hipMemcpyHostToDevice, stream);
            if (he == hipSuccess) he = hipStreamSynchronize(stream);
        }
        if (he != hipSuccess) return fail(LCREC_EHIP, "sinkhorn_assign: %s", hipGetErrorString(he));
        // The size classes are independent (disjoint rows of idx_out).  The slab launch is a hundred or so long-running
        // workgroups; with a context it goes to helper stream 0, forked from `stream` (and joined back on leaving this
        // block), so the tens of thousands of short workgroups of the LDS classes fill the CUs it leaves idle.
        // Likewise the three larger LDS classes (thousands of mid-length workgroups, launches with long tails) go to
        // helper stream 1 beside the pair/triple class on the caller's: 20 rounds at 1 M items 148 -> 130-140 ms.
        const int n_lds = count[0] + count[1] + count[2] + count[3];
        const bool fork_slab = ctx && count[SK_SLAB] > 0 && n_lds > 0;
        const bool fork_mid = ctx && count[0] > 0 && (count[1] + count[2] + count[3]) > 0;
        if (fork_slab || fork_mid) {
            if (int rc = ctx->ensure_streams()) return rc;
        }
        ForkJoin fj(ctx, stream, (fork_slab ? 1u : 0u) | (fork_mid ? 2u : 0u));
        const int64_t *t = triples_dev;
        for (int cls = 0; cls < NCLS; ++cls) {
            if (!count[cls]) continue;
            hipStream_t on = cls == SK_SLAB ? fj.on(0) : (cls > 0 ? fj.on(1) : stream);
            int rc = cls == SK_SLAB ? launch_sk_small_e<true>(e, r, cb, K, t, count[cls], maxg[cls], eps, iters, idx_out, idx_stride, qslab, on)
                                    : launch_sk_small_e<false>(e, r, cb, K, t, count[cls], maxg[cls], eps, iters, idx_out, idx_stride, nullptr, on);
            if (rc) return rc;
            t += (size_t)3 * count[cls];
        }
    }
    // larger problems (a training batch) go through the multi-launch path, one at a time
    for (int g = 0; g < G; ++g) {
        const int64_t sz = offs[g + 1] - offs[g];
        const int cls = sz > 0 ? sk_class(sz, K) : -1;
        if (cls == SK_BATCH || (cls == SK_SLAB && !plan.slab_ok)) {
            int rc = sinkhorn_big(r + offs[g] * e, sz, e, cb, K, eps, iters, idx_out + offs[g] * idx_stride, idx_stride, ws, ticket, stream);
            if (rc) return rc;
        }
    }
    return LCREC_OK;
}

int apply_level(const float *r_in, int64_t n, int e, const float *cb, int K, const int64_t *idx, int64_t idx_stride,
                float *xq, int xq_accumulate, float *r_out, double *sse_out, void *workspace, size_t workspace_bytes,
                unsigned *ticket, hipStream_t stream)
{
    if (n == 0) return LCREC_OK;
    if (!r_in || !cb || !idx) return fail(LCREC_EINVAL, "rq_apply_level: NULL pointer");
    if (e % 4 || e <= 0 || K < 1 || n < 0) return fail(LCREC_EINVAL, "rq_apply_level: bad shape");
    if (n == 0) return LCREC_OK;
    int64_t blocks = (n * (e / 4) + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (sse_out && blocks > TICKET_MAX_WORKGROUPS) blocks = TICKET_MAX_WORKGROUPS;   // (both forms: the sum's grouping follows the grid)
    double *partial = nullptr;
    if (sse_out) {
        if (!workspace || workspace_bytes < 1024 * sizeof(double))
            return fail(LCREC_EWORKSPACE, "rq_apply_level: workspace %zu B < required %zu B", workspace_bytes, 1024 * sizeof(double));
        partial = reinterpret_cast<double *>(workspace);
    }
    TraceScope trace(K_APPLY_LEVEL, stream);
    hipLaunchKernelGGL(apply_level_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, r_in, n, e, cb, K, idx, idx_stride, xq,
                       xq_accumulate, r_out, partial, sse_out ? ticket : nullptr, sse_out);
    if (sse_out && !ticket) hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, stream, partial, (int)blocks, sse_out);
    return check_launch("apply_level_kernel");
}

// Batch-sized problems (n <= CSS_MAX_N, the training case): ONE workgroup does a stable counting sort
// of the items by code entirely in LDS, then thread (code, dim) adds only ITS items, still in item
// order -- same bits as the streaming kernel and as the CPU, ~n/K additions per thread instead of n.
constexpr int CSS_MAX_N = 8192;
constexpr int CSS_MAX_K = 1024;
constexpr int CSS_THREADS = 256;

constexpr int CSS_CHUNK = 256;          // rows staged in LDS per pass of the summation

// `bx` = which CSS_THREADS / E codes this workgroup sums.  cb / grad (both or neither): also write the codebook
// gradient (scale * (count[k] * C[k] - sum[k])) * weight of those codes (lcrec_codebook_grad, fused).
template <int E>
__device__ __forceinline__ void code_stats_sorted_body(const int64_t *__restrict__ idx, int64_t idx_stride,
                                                       const float *__restrict__ resid, int n, int K,
                                                       float *__restrict__ count, float *__restrict__ sum, unsigned bx,
                                                       const float *__restrict__ cb, float *__restrict__ grad, float scale, float weight)
{
    __shared__ __attribute__((aligned(16))) unsigned short skey[CSS_MAX_N];     // code of item i
    __shared__ unsigned short order[CSS_MAX_N];                                 // items sorted by (code, item)
    __shared__ int start[CSS_MAX_K + 1];
    __shared__ int cursor[CSS_MAX_K];
    __shared__ __attribute__((aligned(16))) float rows[CSS_CHUNK * E];          // the rows of one chunk, in sorted order
    const int tid = threadIdx.x;
    for (int k = tid; k < K; k += CSS_THREADS) cursor[k] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 8 * CSS_THREADS) {            // eight index loads in flight per thread
        int kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * CSS_THREADS + tid;
            kk[u] = i < n ? (int)idx[(int64_t)i * idx_stride] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * CSS_THREADS + tid;
            if (i < n) {
                const int k = kk[u] < 0 ? 0 : (kk[u] >= K ? K - 1 : kk[u]);
                skey[i] = (unsigned short)k;
                atomicAdd(&cursor[k], 1);    // integer histogram: the totals do not depend on arrival order
            }
        }
    }
    __syncthreads();
    if (tid == 0) {                          // K <= 1024: a serial exclusive scan is ~1 us
        int run = 0;
        for (int k = 0; k < K; ++k) { start[k] = run; run += cursor[k]; }
        start[K] = run;
    }
    __syncthreads();
    if (bx == 0)
        for (int k = tid; k < K; k += CSS_THREADS) count[k] = (float)cursor[k];
    // This workgroup sums CSS_THREADS / E codes: [k_lo, k_hi).  Stable placement of THEIR items only, by all threads:
    // thread (code c, segment g) owns items [g*seg, (g+1)*seg) -- it counts its matches, the segment counts of a code
    // are prefix-summed, then it appends its matches (item order inside a segment, segments in order).
    constexpr int CPB = CSS_THREADS / E;
    __shared__ int segcnt[CPB][E + 1];
    const int k_lo = bx * CPB, k_hi = k_lo + CPB < K ? k_lo + CPB : K;
    {
        const int c = tid / E, g = tid % E;
        const unsigned k = (unsigned)(k_lo + c);
        const int seg = ((n + E - 1) / E + 7) & ~7;
        const int i_lo = g * seg < n ? g * seg : n, i_hi = (g + 1) * seg < n ? (g + 1) * seg : n;
        auto walk = [&](auto &&hit) {
            int i = i_lo;
            for (; i + 8 <= i_hi; i += 8) {
                const uint4 q = *reinterpret_cast<const uint4 *>(&skey[i]);
                const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    if ((w[h] & 0xffffu) == k) hit(i + 2 * h);
                    if ((w[h] >> 16) == k) hit(i + 2 * h + 1);
                }
            }
            for (; i < i_hi; ++i)
                if (skey[i] == k) hit(i);
        };
        int cnt = 0;
        if (k_lo + c < K) walk([&](int) { ++cnt; });
        segcnt[c][g] = cnt;
        __syncthreads();
        if (k_lo + c < K) {
            int pos = start[k];
            for (int t = 0; t < g; ++t) pos += segcnt[c][t];
            walk([&](int i) { order[pos++] = (unsigned short)i; });
        }
    }
    __syncthreads();
    // Per-code sums in item order (bit-identical to the CPU's sequential index_add_).  A code can own most of the
    // batch (early training), so the chain must not wait on memory: the workgroup's rows are first gathered, in
    // sorted order and with every load independent, into LDS a chunk at a time; the chains then read LDS.
    const int p_lo = start[k_lo], p_hi = start[k_hi];
    const int kq = tid / E, d = tid % E;
    const int my_lo = k_lo + kq < K ? start[k_lo + kq] : p_hi, my_hi = k_lo + kq < K ? start[k_lo + kq + 1] : p_hi;
    float acc = 0.f;
    for (int c0 = p_lo; c0 < p_hi; c0 += CSS_CHUNK) {
        const int c1 = c0 + CSS_CHUNK < p_hi ? c0 + CSS_CHUNK : p_hi;
        constexpr int LPR = E / 4;                                   // 16-byte pieces per row
        for (int q = tid; q < (c1 - c0) * LPR; q += CSS_THREADS) {
            const int r = q / LPR, piece = q % LPR;
            *reinterpret_cast<f32x4 *>(&rows[r * E + piece * 4]) =
                *reinterpret_cast<const f32x4 *>(resid + (int64_t)order[c0 + r] * E + piece * 4);
        }
        __syncthreads();
        const int a = my_lo > c0 ? my_lo : c0, b = my_hi < c1 ? my_hi : c1;
#pragma unroll 8
        for (int p = a; p < b; ++p) acc = acc + rows[(p - c0) * E + d];      // the LDS reads run ahead of the add chain
        __syncthreads();
    }
    if (k_lo + kq < K) {
        const size_t o = (size_t)(k_lo + kq) * E + d;
        sum[o] = acc;
        if (grad) {
            const float t = (float)cursor[k_lo + kq] * cb[o] - acc;
            grad[o] = (scale * t) * weight;
        }
    }
}

template <int E>
__global__ __launch_bounds__(CSS_THREADS) void code_stats_sorted_kernel(const int64_t *__restrict__ idx, int64_t idx_stride,
                                                                       const float *__restrict__ resid, int n, int K,
                                                                       float *__restrict__ count, float *__restrict__ sum)
{
    code_stats_sorted_body<E>(idx, idx_stride, resid, n, K, count, sum, blockIdx.x, nullptr, nullptr, 0.f, 0.f);
}

// all levels of a quantiser in one launch: blockIdx.y = level
struct CsLevels {
    const int64_t *idx;          // [n][L]
    int L, n;
    float scale, weight;
    const float *resid[LCREC_MAX_LEVELS], *cb[LCREC_MAX_LEVELS];
    float *count[LCREC_MAX_LEVELS], *sum[LCREC_MAX_LEVELS], *grad[LCREC_MAX_LEVELS];
    int K[LCREC_MAX_LEVELS];
};

template <int E>
__global__ __launch_bounds__(CSS_THREADS) void code_stats_levels_kernel(CsLevels g)
{
    const int l = blockIdx.y;
    if ((int)(blockIdx.x * (CSS_THREADS / E)) >= g.K[l]) return;
    code_stats_sorted_body<E>(g.idx + l, g.L, g.resid[l], g.n, g.K[l], g.count[l], g.sum[l], blockIdx.x, g.cb[l], g.grad[l], g.scale,
                              g.weight);
}

int code_stats(const int64_t *idx, int64_t idx_stride, const float *resid, int64_t n, int e, int K, float *count,
               float *sum, hipStream_t stream)
{
    if ((n > 0 && (!idx || !resid)) || !count || !sum) return fail(LCREC_EINVAL, "code_stats: NULL pointer");
    if (n < 0 || K < 1) return fail(LCREC_EINVAL, "code_stats: bad shape");
    TraceScope trace(K_CODE_STATS, stream);
    if (n > 0 && n <= CSS_MAX_N && K <= CSS_MAX_K && (e == 16 || e == 32 || e == 64)) {
        if (e == 16) hipLaunchKernelGGL(code_stats_sorted_kernel<16>, dim3((unsigned)((K * e + CSS_THREADS - 1) / CSS_THREADS)), dim3(CSS_THREADS), 0, stream, idx, idx_stride, resid, (int)n, K, count, sum);
        else if (e == 32) hipLaunchKernelGGL(code_stats_sorted_kernel<32>, dim3((unsigned)((K * e + CSS_THREADS - 1) / CSS_THREADS)), dim3(CSS_THREADS), 0, stream, idx, idx_stride, resid, (int)n, K, count, sum);
        else hipLaunchKernelGGL(code_stats_sorted_kernel<64>, dim3((unsigned)((K * e + CSS_THREADS - 1) / CSS_THREADS)), dim3(CSS_THREADS), 0, stream, idx, idx_stride, resid, (int)n, K, count, sum);
        return check_launch("code_stats_sorted_kernel");
    }
    if (e == 16) hipLaunchKernelGGL(code_stats_kernel<16>, dim3((K + 15) / 16), dim3(256), 0, stream, idx, idx_stride, resid, n, K, count, sum);
    else if (e == 32) hipLaunchKernelGGL(code_stats_kernel<32>, dim3((K + 7) / 8), dim3(256), 0, stream, idx, idx_stride, resid, n, K, count, sum);
    else if (e == 64) hipLaunchKernelGGL(code_stats_kernel<64>, dim3((K + 3) / 4), dim3(256), 0, stream, idx, idx_stride, resid, n, K, count, sum);
    else return fail(LCREC_EUNSUPPORTED, "code_stats: e_dim=%d (supported: 16, 32, 64)", e);
    return check_launch("code_stats_kernel");
}

int code_stats_levels(const int64_t *idx, const float *const *resid, int64_t n, int e, const int *K, int L, float *const *count,
                      float *const *sum, const float *const *cb, float *const *grad, float scale, float weight, hipStream_t stream)
{
    if (!idx || !resid || !K || !count || !sum) return fail(LCREC_EINVAL, "code_stats_levels: NULL pointer");
    if (L < 1 || L > LCREC_MAX_LEVELS || n < 1) return fail(LCREC_EINVAL, "code_stats_levels: bad L=%d or n=%lld", L, (long long)n);
    if ((cb == nullptr) != (grad == nullptr)) return fail(LCREC_EINVAL, "code_stats_levels: codebooks and grad_out go together");
    int kmax = 0;
    bool fused = n <= CSS_MAX_N && (e == 16 || e == 32 || e == 64);
    for (int l = 0; l < L; ++l) {
        if (K[l] < 1 || !resid[l] || !count[l] || !sum[l] || (cb && (!cb[l] || !grad[l])))
            return fail(LCREC_EINVAL, "code_stats_levels: level %d: bad K or NULL pointer", l);
        kmax = K[l] > kmax ? K[l] : kmax;
        fused = fused && K[l] <= CSS_MAX_K;
    }
    if (!fused) {                                   // sizes beyond the one-workgroup sort: level by level
        for (int l = 0; l < L; ++l) {
            int rc = code_stats(idx + l, L, resid[l], n, e, K[l], count[l], sum[l], stream);
            if (!rc && cb) rc = codebook_grad(count[l], sum[l], cb[l], K[l], e, scale, weight, grad[l], stream);
            if (rc) return rc;
        }
        return LCREC_OK;
    }
    CsLevels g = {};
    g.idx = idx; g.L = L; g.n = (int)n; g.scale = scale; g.weight = weight;
    for (int l = 0; l < L; ++l) {
        g.resid[l] = resid[l]; g.count[l] = count[l]; g.sum[l] = sum[l]; g.K[l] = K[l];
        g.cb[l] = cb ? cb[l] : nullptr; g.grad[l] = grad ? grad[l] : nullptr;
    }
    TraceScope trace(K_CODE_STATS, stream);
    const dim3 grid((unsigned)((kmax * e + CSS_THREADS - 1) / CSS_THREADS), (unsigned)L);
    if (e == 16) hipLaunchKernelGGL(code_stats_levels_kernel<16>, grid, dim3(CSS_THREADS), 0, stream, g);
    else if (e == 32) hipLaunchKernelGGL(code_stats_levels_kernel<32>, grid, dim3(CSS_THREADS), 0, stream, g);
    else hipLaunchKernelGGL(code_stats_levels_kernel<64>, grid, dim3(CSS_THREADS), 0, stream, g);
    return check_launch("code_stats_levels_kernel");
}

int ema_update(float *ema_count, float *ema_sum, float *codebook, const float *count, const float *sum, int K, int e,
               float decay, float alpha, float keep, float eps, const unsigned char *skip, hipStream_t stream)
{
    if (!ema_count || !ema_sum || !codebook || !count || !sum) return fail(LCREC_EINVAL, "ema_update: NULL pointer");
    if (e != 16 && e != 32 && e != 64) return fail(LCREC_EUNSUPPORTED, "ema_update: e_dim=%d (supported: 16, 32, 64)", e);
    const int64_t total = (int64_t)K * e;
    TraceScope trace(K_EMA_UPDATE, stream);
    hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ema_count, ema_sum, codebook,
                       count, sum, K, e, decay, alpha, keep, eps, skip);
    return check_launch("ema_update_kernel");
}

}  // namespace lcrec

#ifdef LCREC_GEMM_STAMP
extern "C" __attribute__((visibility("default"))) int lcrec_debug_sk_stamps(unsigned long long *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(lcrec::g_sk_stamps), sizeof(unsigned long long) * 8);
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(lcrec::g_sk_stamps), z, sizeof z);
    }
    return (int)e;
}
#endif
