// L-level residual quantiser, hard (argmin) assignment, for gfx950.
//
// Replaces ResidualVectorQuantizer.forward (reference index/models/rq.py:39-55)
// over VectorQuantizer.forward with use_sk=False (index/models/vq.py:63-99):
// distance (:71-73), argmin (:75), gather (:87), losses (:90-92), STE (:95),
// residual update (rq.py:47-48) -- all L levels in one launch when the
// codebooks fit in LDS, with the residual living in registers throughout.
//
// Mapping onto the hardware
//   * Codebooks (and their squared norms) are staged once per workgroup into
//     LDS; workgroups are persistent and stride over 64-item tiles per wave.
//   * The -2 x.C term is v_mfma_f32_32x32x2_f32 with the CODEBOOK as the A
//     operand and the residuals as B, so the 32x32 result tile has the item on
//     the lane and 16 codes in the lane's registers: the argmin over codes is a
//     register-local scan plus ONE cross-half exchange (v_permlane32_swap).
//   * One lane owns one item (its E residual floats live in its VGPRs).  The
//     B operands of the two 32-item groups of a wave are produced from those
//     registers by v_permlane32_swap: swap(r[2s], r[2s+1]) yields, in one
//     instruction, {r_lo[2s] | r_lo[2s+1]} for items 0-31 and
//     {r_hi[2s] | r_hi[2s+1]} for items 32-63 -- exactly the k = 2s / 2s+1
//     split a 32x32x2 MFMA wants.
//   * LDS rows store a code de-interleaved ([even k | odd k]) and padded by 4
//     floats: a lane's A operand for all E/2 MFMA steps of a 32-code block is
//     E/8 ds_read_b128, conflict-free (row stride E+4 floats puts 16
//     consecutive rows on 16 distinct 4-bank slots).
//
// Arithmetic contract = oracle/lcrec_oracle.c (assign_one, lcrec_oracle_rq_assign):
//   xx = chain_k r[k]^2;  cc[j] = chain_k C[j][k]^2;  dot = chain_k r[k]*C[j][k]  (fp32 fma chains, k ascending)
//   d[j] = (xx + cc[j]) - 2*dot;  argmin takes the first minimum;
//   t = c - r;  s = r + t;  x_q += s;  r -= s.
#include "common.h"

namespace lcrec {

struct RqParams {
    const float *z_in;     // [n][E] residual entering level l0
    const float *cb;       // all codebooks (level l starts at float offset cb_off[l])
    int64_t n;
    int l0, l1, L;
    int K[LCREC_MAX_LEVELS];        // codes per level (absolute level index)
    int64_t cb_off[LCREC_MAX_LEVELS];  // float offset of a level's [K][E] rows in cb
    int row_off[LCREC_MAX_LEVELS];  // first LDS row of a level; a level occupies K rounded up to 32 rows
    int rows;                       // total (padded) rows staged by this launch
    int xq_accumulate;              // xq holds an initial value to add to (else starts at 0)
    int64_t *idx_out;      // [n][L] with row stride idx_stride (>= L) elements
    int64_t idx_stride;
    unsigned *ticket;      // with it the last workgroup to arrive adds the SSE partials (include/lcrec.h); else a second launch
    double *sse_out;       // [L] (ticket form)
    float *xq;             // [n][E] in/out (read when l0 > 0), or NULL
    float *resid_next;     // [n][E] residual after level l1-1, or NULL
    float *resid_levels;   // [L+1][n][E]: entry l = residual entering level l, entry L = final residual; or NULL
    double *sse_partial;   // [gridDim.x][L], or NULL
    float *margin;         // [n][L]: second smallest distance minus the smallest, per level; or NULL
    uint32_t *neartie;     // [n]: bit l set when margin_l <= tie_tau * (xx_l + cc_l[idx_l]); or NULL
    float tie_tau;
    int split;             // batch-sized inputs: one 64-item tile per WORKGROUP, a level's code blocks dealt over its waves
};

__device__ __forceinline__ void swap32(float a, float b, float &lo_pair, float &hi_pair)
{
    // lanes 32-63 of `a` exchange with lanes 0-31 of `b`:
    //   lo_pair = {a.lo | b.lo},  hi_pair = {a.hi | b.hi}
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    lo_pair = __uint_as_float(r[0]);
    hi_pair = __uint_as_float(r[1]);
}

// WANT_MARGIN: also track the second smallest distance (the near-tie audit of SURVEY.md section 8b / section 7 hard
// part 1: rows whose two best codes are closer than the rounding noise of vq.py:71-73 are the only ones on which the
// reference's own CPU arithmetic can pick a different code).  Same value as the oracle's sequential scan: the minimum
// over all codes but the winner -- an order-independent quantity, so the per-half partials merge exactly.
template <int E, int THREADS, bool WANT_XQ, bool WANT_MARGIN>
__global__ __launch_bounds__(THREADS) void rq_assign_kernel(RqParams p)
{
    constexpr int S = E + 4;   // padded LDS row (floats)
    constexpr int H = E / 2;   // MFMA steps per 32-code block
    constexpr int WAVES = THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *cbs = smem;                       // [rows][S]
    float *ccs = smem + (size_t)p.rows * S;  // [rows]
    double *wave_sse = reinterpret_cast<double *>(ccs + ((p.rows + 3) & ~3));  // [WAVES][L]
    // (split) per-level hand-over of the waves' partial argmins: [2 parities][WAVES][64 lanes] x {distance, index, second}
    float *ex = reinterpret_cast<float *>(wave_sse + (size_t)WAVES * p.L);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;

    // ---- stage codebooks: global [row][E] -> LDS [row][even k | odd k | pad]
    // (rows K..roundup32(K) of a level are padding: zero vector, norm +inf, never the argmin)
    for (int l = p.l0; l < p.l1; ++l) {
      const int kpad = (p.K[l] + 31) & ~31;
      const float *lsrc = p.cb + p.cb_off[l];
      // Four trips' loads issued before the first LDS write, unconditionally (padding rows read row 0 and are zeroed): a
      // batch-sized launch is one tile per workgroup, and the 12 trips of this loop at 3 x 256 codes were 12 dependent memory round
      // trips before the first MFMA.
      const int total = kpad * (E / 8);
      for (int q0 = tid; q0 < total; q0 += 4 * THREADS) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * THREADS;
            const int lrow = q < total ? q / (E / 8) : 0, g = q % (E / 8);
            const f32x4 *src = reinterpret_cast<const f32x4 *>(lsrc + (size_t)(lrow < p.K[l] ? lrow : 0) * E + g * 8);
            a[u] = src[0];
            b[u] = src[1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * THREADS;
            if (q >= total) break;
            const int lrow = q / (E / 8), g = q % (E / 8);
            const int row = p.row_off[l] + lrow;
            const bool real = lrow < p.K[l];
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 av = real ? a[u] : zero, bv = real ? b[u] : zero;
            f32x4 ev = {av[0], av[2], bv[0], bv[2]};
            f32x4 od = {av[1], av[3], bv[1], bv[3]};
            *reinterpret_cast<f32x4 *>(cbs + row * S + g * 4) = ev;
            *reinterpret_cast<f32x4 *>(cbs + row * S + H + g * 4) = od;
        }
      }
    }
    for (int q = tid; q < WAVES * p.L; q += THREADS) wave_sse[q] = 0.0;
    __syncthreads();
    for (int row = tid; row < p.rows; row += THREADS) {
        const float *cr = cbs + row * S;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) {
            a = __builtin_fmaf(cr[k], cr[k], a);          // k even = 2k'
            a = __builtin_fmaf(cr[H + k], cr[H + k], a);  // k odd  = 2k'+1
        }
        ccs[row] = a;
    }
    __syncthreads();
    for (int l = p.l0; l < p.l1; ++l)   // padding rows
        for (int row = p.row_off[l] + p.K[l] + tid; row < p.row_off[l] + ((p.K[l] + 31) & ~31); row += THREADS)
            ccs[row] = __builtin_inff();
    __syncthreads();

    // Split form (batch-sized inputs, a training step's 1-4 k rows): with one tile per wave a 1024-row batch is 16 waves
    // on 16 SIMDs, each walking all of a level's code blocks (8 x 32 MFMAs at 256 codes) level after level -- 43 us of a
    // 1.2 ms step.  Here every wave of a workgroup holds the SAME 64 items and takes a contiguous share of the level's code
    // blocks; the partial (best, index, second) of each wave meet in LDS and every wave applies the same winner, so the
    // residuals stay identical in all of them.  Ties still go to the lowest code (lower waves hold lower codes).  Only wave 0
    // writes results and accumulates the sums of squares.
    const bool split = p.split != 0;
    const bool writer = !split || wave == 0;
    const int64_t tiles = (p.n + 63) / 64;
    const int64_t gw = split ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * WAVES + wave;
    const int64_t GW = split ? (int64_t)gridDim.x : (int64_t)gridDim.x * WAVES;

    for (int64_t tile = gw; tile < tiles; tile += GW) {
        const int64_t item = tile * 64 + lane;
        const bool valid = item < p.n;

        float r[E], xq[WANT_XQ ? E : 1];
        if (valid) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(p.z_in + item * E);
#pragma unroll
            for (int q = 0; q < E / 4; ++q) {
                const f32x4 v = src[q];
                r[4 * q] = v[0]; r[4 * q + 1] = v[1]; r[4 * q + 2] = v[2]; r[4 * q + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) r[k] = 0.f;
        }
        if (WANT_XQ) {
            if ((p.l0 > 0 || p.xq_accumulate) && valid) {
                const f32x4 *src = reinterpret_cast<const f32x4 *>(p.xq + item * E);
#pragma unroll
                for (int q = 0; q < E / 4; ++q) {
                    const f32x4 v = src[q];
                    xq[4 * q] = v[0]; xq[4 * q + 1] = v[1]; xq[4 * q + 2] = v[2]; xq[4 * q + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int k = 0; k < (WANT_XQ ? E : 1); ++k) xq[k] = 0.f;
            }
        }

        uint32_t tie_bits = 0;
        if (WANT_MARGIN && p.neartie && p.l0 > 0 && valid) tie_bits = p.neartie[item];   // levels of earlier launches

        for (int l = p.l0; l < p.l1; ++l) {
            const int ro = p.row_off[l];
            const int nblk = (p.K[l] + 31) >> 5;

            if (p.resid_levels && valid && writer) {
                f32x4 *dst = reinterpret_cast<f32x4 *>(p.resid_levels + ((size_t)l * p.n + item) * E);
#pragma unroll
                for (int q = 0; q < E / 4; ++q) {
                    f32x4 v = {r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
                    dst[q] = v;
                }
            }

            // xx = chain_k r[k]^2 for the lane's own item, then handed to both halves
            float xx = 0.f;
#pragma unroll
            for (int k = 0; k < E; ++k) xx = __builtin_fmaf(r[k], r[k], xx);
            float xx0, xx1;
            swap32(xx, xx, xx0, xx1);   // xx0 = {xx.lo | xx.lo}: items 0-31; xx1: items 32-63

            float b0[H], b1[H];
#pragma unroll
            for (int s = 0; s < H; ++s) swap32(r[2 * s], r[2 * s + 1], b0[s], b1[s]);

            float best0 = __builtin_inff(), best1 = __builtin_inff();
            float sec0 = __builtin_inff(), sec1 = __builtin_inff();
            int bi0 = 0, bi1 = 0;

            const int per = split ? (nblk + WAVES - 1) / WAVES : nblk;
            const int b_lo = split ? wave * per : 0, b_hi = b_lo + per < nblk ? b_lo + per : nblk;
            for (int b = b_lo; b < b_hi; ++b) {
                const float *arow = cbs + (ro + b * 32 + c) * S + h * H;
                float af[H];
#pragma unroll
                for (int q = 0; q < H / 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(arow + 4 * q);
                    af[4 * q] = v[0]; af[4 * q + 1] = v[1]; af[4 * q + 2] = v[2]; af[4 * q + 3] = v[3];
                }
                float ccv[16];
                const float *ccb = ccs + ro + b * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(ccb + 8 * q);
                    ccv[4 * q] = v[0]; ccv[4 * q + 1] = v[1]; ccv[4 * q + 2] = v[2]; ccv[4 * q + 3] = v[3];
                }
                f32x16 acc0, acc1;
#pragma unroll
                for (int t = 0; t < 16; ++t) { acc0[t] = 0.f; acc1[t] = 0.f; }
#pragma unroll
                for (int s = 0; s < H; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], b0[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], b1[s], acc1, 0, 0, 0);
                }
                // register t of the tile is code (t&3) + 8*(t>>2) + 4*h of this block: ascending in t
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int code = b * 32 + (t & 3) + 8 * (t >> 2) + 4 * h;
                    const float t0 = xx0 + ccv[t];
                    const float d0 = t0 - 2.0f * acc0[t];
                    const bool lt0 = d0 < best0;
                    if (WANT_MARGIN) sec0 = lt0 ? best0 : (d0 < sec0 ? d0 : sec0);
                    best0 = lt0 ? d0 : best0;
                    bi0 = lt0 ? code : bi0;
                    const float t1 = xx1 + ccv[t];
                    const float d1 = t1 - 2.0f * acc1[t];
                    const bool lt1 = d1 < best1;
                    if (WANT_MARGIN) sec1 = lt1 ? best1 : (d1 < sec1 ? d1 : sec1);
                    best1 = lt1 ? d1 : best1;
                    bi1 = lt1 ? code : bi1;
                }
            }

            // bring both half-partials of the lane's own item home:
            //   dA/iA = partial over the h=0 codes, dB/iB = partial over the h=1 codes
            float dA, dB, fA, fB;
            swap32(best0, best1, dA, dB);
            swap32(__int_as_float(bi0), __int_as_float(bi1), fA, fB);
            const int iA = __float_as_int(fA), iB = __float_as_int(fB);
            const bool takeB = (dB < dA) || (dB == dA && iB < iA);
            int bi = takeB ? iB : iA;
            float win = takeB ? dB : dA, second = __builtin_inff();
            if (WANT_MARGIN) {
                float sA, sB;
                swap32(sec0, sec1, sA, sB);
                const float lose = takeB ? dA : dB, wsec = takeB ? sB : sA;     // the loser's best, the winner's second
                second = lose < wsec ? lose : wsec;
            }
            if (split) {
                // this wave's (best, index, second) over ITS code blocks -> LDS; then every wave merges all of them in wave
                // order.  Parity buffers: a wave may be a level ahead of another, never two (the barrier below).
                float *mine = ex + ((size_t)((l & 1) * WAVES + wave) * 64 + lane) * 3;
                mine[0] = win; mine[1] = __int_as_float(bi); mine[2] = second;
                __syncthreads();
                const float *all = ex + ((size_t)(l & 1) * WAVES * 64 + lane) * 3;
                win = all[0]; bi = __float_as_int(all[1]); second = all[2];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) {
                    const float dw = all[(size_t)w * 64 * 3], sw = all[(size_t)w * 64 * 3 + 2];
                    const int iw = __float_as_int(all[(size_t)w * 64 * 3 + 1]);
                    if (dw < win || (dw == win && iw < bi)) {            // the new winner's second, or the old winner
                        second = win < sw ? win : sw;
                        win = dw; bi = iw;
                    } else {
                        second = dw < second ? dw : second;
                    }
                }
            }

            if (valid && writer) p.idx_out[item * p.idx_stride + l] = (int64_t)bi;
            if (WANT_MARGIN) {
                const float margin = second - win;
                if (p.margin && valid && writer) p.margin[item * p.L + l] = margin;
                const float scale = xx + ccs[ro + bi];
                if (margin <= p.tie_tau * scale) tie_bits |= 1u << l;
            }

            // gather the winning code (LDS row is [even k | odd k])
            const float *crow = cbs + (ro + bi) * S;
            float cv[E];
#pragma unroll
            for (int q = 0; q < H / 4; ++q) {
                const f32x4 ev = *reinterpret_cast<const f32x4 *>(crow + 4 * q);
                const f32x4 od = *reinterpret_cast<const f32x4 *>(crow + H + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    cv[2 * (4 * q + t)] = ev[t];
                    cv[2 * (4 * q + t) + 1] = od[t];
                }
            }
            float sse = 0.f;
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const float t = cv[k] - r[k];
                sse = __builtin_fmaf(t, t, sse);
                const float s = r[k] + t;
                if (WANT_XQ) xq[k] = xq[k] + s;
                r[k] = r[k] - s;
            }
            if (p.sse_partial && writer) {
                double v = valid ? (double)sse : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
                if (lane == 0) wave_sse[wave * p.L + l] += v;
            }
        }

        if (WANT_MARGIN && p.neartie && valid && writer) p.neartie[item] = tie_bits;
        if (valid && writer) {
            if (WANT_XQ) {
                f32x4 *dst = reinterpret_cast<f32x4 *>(p.xq + item * E);
#pragma unroll
                for (int q = 0; q < E / 4; ++q) {
                    f32x4 v = {xq[4 * q], xq[4 * q + 1], xq[4 * q + 2], xq[4 * q + 3]};
                    dst[q] = v;
                }
            }
            if (p.resid_next) {
                f32x4 *dst = reinterpret_cast<f32x4 *>(p.resid_next + item * E);
#pragma unroll
                for (int q = 0; q < E / 4; ++q) {
                    f32x4 v = {r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
                    dst[q] = v;
                }
            }
            if (p.resid_levels && p.l1 == p.L) {
                f32x4 *dst = reinterpret_cast<f32x4 *>(p.resid_levels + ((size_t)p.L * p.n + item) * E);
#pragma unroll
                for (int q = 0; q < E / 4; ++q) {
                    f32x4 v = {r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
                    dst[q] = v;
                }
            }
        }
    }

    if (p.sse_partial) {
        __syncthreads();
        if (!p.ticket) {
            for (int l = p.l0 + tid; l < p.l1; l += THREADS) {
                double v = 0.0;
                for (int w = 0; w < WAVES; ++w) v += wave_sse[w * p.L + l];
                p.sse_partial[(size_t)blockIdx.x * p.L + l] = v;
            }
            return;
        }
        // ticket form: thread 0 publishes all of the workgroup's level sums (L <= 16) and takes the ticket; in the last workgroup
        // to arrive wave 0 adds every workgroup's partials in workgroup order -- rq_sse_finalize_kernel's sums, its bits
        __shared__ int last_sh;
        if (tid == 0) {
            for (int l = p.l0; l < p.l1; ++l) {
                double v = 0.0;
                for (int w = 0; w < WAVES; ++w) v += wave_sse[w * p.L + l];
                handoff_put(p.sse_partial + (size_t)blockIdx.x * p.L + l, v);
            }
            last_sh = ticket_is_last(p.ticket, gridDim.x) ? 1 : 0;
        }
        __syncthreads();
        if (last_sh && tid < 64) {
            for (int l = p.l0; l < p.l1; ++l) {
                const double v = handoff_sum_ordered(p.sse_partial + l, (int)gridDim.x, p.L);
                if (tid == 0) p.sse_out[l] = v;
            }
        }
    }
}

// sse_out[l] = sum over blocks (fixed order) of the per-block partials
__global__ void rq_sse_finalize_kernel(const double *partial, int blocks, int L, int l0, int l1, double *sse_out)
{
    const int l = l0 + threadIdx.x;
    if (l >= l1) return;
    double v = 0.0;
    for (int b = 0; b < blocks; ++b) v += partial[(size_t)b * L + l];
    sse_out[l] = v;
}

// ---------------------------------------------------------------- host side

constexpr size_t LDS_BUDGET = 160 * 1024;
constexpr int MAX_GRID = 256;   // one persistent workgroup per CU

static size_t lds_bytes(int rows, int E, int L, int waves, bool split = false)
{
    return ((size_t)rows * (E + 4) + ((rows + 3) & ~3)) * sizeof(float) + (size_t)waves * L * sizeof(double) +
           (split ? (size_t)2 * waves * 64 * 3 * sizeof(float) : 0);      // the split form's hand-over buffers
}

static int threads_for(int e, int64_t n)
{
    if (e == 64) return 256;             // 512-register budget per lane
    return n >= 256 * 512 ? 512 : 256;   // small inputs: more, smaller workgroups
}

static int grid_for(int64_t n, int threads)
{
    const int64_t per_block = threads;   // 64 items per wave per iteration
    int64_t g = (n + per_block - 1) / per_block;
    if (g > MAX_GRID) g = MAX_GRID;
    if (g < 1) g = 1;
    return (int)g;
}

size_t rq_assign_workspace(int64_t n, int e, const int *K, int L)
{
    (void)K;
    // [n][e] ping-pong residual for multi-launch configurations + SSE partials
    size_t resid = align_up((size_t)(n > 0 ? n : 1) * e * sizeof(float), 256);
    size_t part = align_up((size_t)MAX_GRID * L * sizeof(double), 256);
    return 2 * resid + part;
}

template <int E, int THREADS, bool WANT_XQ, bool WANT_MARGIN>
static int launch_one(const RqParams &p, int grid, size_t lds, hipStream_t stream)
{
    auto kern = rq_assign_kernel<E, THREADS, WANT_XQ, WANT_MARGIN>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(LCREC_EHIP, "rq_assign: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    TraceScope trace(K_RQ_ASSIGN, stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, stream, p);
    return check_launch("rq_assign_kernel");
}

template <int E, int THREADS>
static int dispatch2(const RqParams &p, int grid, size_t lds, bool want_xq, hipStream_t stream)
{
    const bool want_margin = p.margin || p.neartie;
    if (want_margin) return want_xq ? launch_one<E, THREADS, true, true>(p, grid, lds, stream)
                                    : launch_one<E, THREADS, false, true>(p, grid, lds, stream);
    return want_xq ? launch_one<E, THREADS, true, false>(p, grid, lds, stream)
                   : launch_one<E, THREADS, false, false>(p, grid, lds, stream);
}

template <int E>
static int dispatch(const RqParams &p, int threads, int grid, size_t lds, bool want_xq, hipStream_t stream)
{
    if (threads == 512) {
        if constexpr (E == 64) return fail(LCREC_EUNSUPPORTED, "rq_assign: e=64 runs 256-thread workgroups");
        else return dispatch2<E, 512>(p, grid, lds, want_xq, stream);
    }
    return dispatch2<E, 256>(p, grid, lds, want_xq, stream);
}

int rq_assign(const float *z, int64_t n, int e, const float *codebooks, const int *K, int L,
              int64_t *idx_out, int64_t idx_stride, float *xq_out, int xq_accumulate, double *sse_out, float *resid_out,
              float *margin_out, uint32_t *neartie_out, float tie_tau,
              void *workspace, size_t workspace_bytes, unsigned *ticket, hipStream_t stream)
{
    if (n == 0 && K && L >= 1 && L <= LCREC_MAX_LEVELS) return LCREC_OK;   // empty batch
    if (!z || !codebooks || !K || !idx_out) return fail(LCREC_EINVAL, "rq_assign: NULL pointer");
    if (n < 0 || L < 1 || L > LCREC_MAX_LEVELS) return fail(LCREC_EINVAL, "rq_assign: bad n=%lld or L=%d", (long long)n, L);
    if (idx_stride == 0) idx_stride = L;
    if (idx_stride < L) return fail(LCREC_EINVAL, "rq_assign: idx_stride %lld < L=%d", (long long)idx_stride, L);
    if (neartie_out && !(tie_tau >= 0.0f)) return fail(LCREC_EINVAL, "rq_assign: tie_tau must be >= 0");
    if (e != 16 && e != 32 && e != 64) return fail(LCREC_EUNSUPPORTED, "rq_assign: e_dim=%d (supported: 16, 32, 64)", e);
    if (((uintptr_t)z | (uintptr_t)codebooks | (uintptr_t)xq_out | (uintptr_t)resid_out) & 15)
        return fail(LCREC_EINVAL, "rq_assign: buffers must be 16-byte aligned");
    // batch-sized inputs with enough code blocks to deal out: the split form (see the kernel).  Up to 512 tiles of 64 items: below
    // that the one-tile-per-wave form leaves most CUs without a wave while each wave walks its tile's levels alone (Games' 16 859
    // items: 264 tiles on 66 workgroups, 55 us; LCREC_RQ_SPLIT_TILES, tuning)
    static const int64_t split_tiles = [] { const char *v = getenv("LCREC_RQ_SPLIT_TILES"); return (int64_t)(v ? atoi(v) : 512); }();
    int max_k = 0;
    for (int l = 0; l < L; ++l) max_k = K[l] > max_k ? K[l] : max_k;
    static const bool allow_split = [] { const char *v = getenv("LCREC_RQ_SPLIT"); return !v || atoi(v) != 0; }();
    bool split = allow_split && n > 0 && (n + 63) / 64 <= split_tiles && max_k >= 128;
    for (int l = 0; split && l < L; ++l)       // (its hand-over buffers must not push a level out of LDS that fits without them)
        if (K[l] > 0 && lds_bytes((K[l] + 31) & ~31, e, L, 4, true) > LDS_BUDGET) split = false;
    const int threads = split ? 256 : threads_for(e, n);
    const int waves = threads / 64;
    for (int l = 0; l < L; ++l) {
        if (K[l] <= 0) return fail(LCREC_EINVAL, "rq_assign: K[%d]=%d", l, K[l]);
        if (lds_bytes((K[l] + 31) & ~31, e, L, waves, split) > LDS_BUDGET)
            return fail(LCREC_EUNSUPPORTED, "rq_assign: level %d (K=%d, e=%d) does not fit in 160 KB of LDS", l, K[l], e);
    }
    if (n == 0) return LCREC_OK;
    if (workspace_bytes < rq_assign_workspace(n, e, K, L) || !workspace)
        return fail(LCREC_EWORKSPACE, "rq_assign: workspace %zu B < required %zu B", workspace_bytes, rq_assign_workspace(n, e, K, L));

    const size_t resid_bytes = align_up((size_t)n * e * sizeof(float), 256);
    float *ping = reinterpret_cast<float *>(workspace);
    float *pong = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + resid_bytes);
    double *partial = reinterpret_cast<double *>(reinterpret_cast<char *>(workspace) + 2 * resid_bytes);
    // (split: the staged codebooks leave room for one workgroup per CU, so more than MAX_GRID of them would run in rounds and
    // stage again; the tile loop takes the rest)
    const int grid = split ? (int)((n + 63) / 64 < MAX_GRID ? (n + 63) / 64 : MAX_GRID) : grid_for(n, threads);

    // Greedily pack consecutive levels into launches whose codebooks fit in LDS.
    int64_t cb_offs[LCREC_MAX_LEVELS];
    {
        int64_t o = 0;
        for (int l = 0; l < L; ++l) { cb_offs[l] = o; o += (int64_t)K[l] * e; }
    }
    const float *zin = z;
    int l0 = 0;
    while (l0 < L) {
        RqParams p = {};
        int rows = 0, l1 = l0;
        while (l1 < L && lds_bytes(rows + ((K[l1] + 31) & ~31), e, L, waves, split) <= LDS_BUDGET) {
            p.row_off[l1] = rows;
            rows += (K[l1] + 31) & ~31;
            ++l1;
        }
        for (int l = 0; l < L; ++l) { p.K[l] = K[l]; p.cb_off[l] = cb_offs[l]; }
        p.xq_accumulate = xq_accumulate;
        p.z_in = zin;
        p.cb = codebooks;
        p.n = n; p.l0 = l0; p.l1 = l1; p.L = L; p.rows = rows;
        p.idx_out = idx_out;
        p.idx_stride = idx_stride;
        p.ticket = sse_out ? ticket : nullptr;
        p.sse_out = sse_out;
        p.xq = xq_out;
        p.resid_levels = resid_out;
        p.sse_partial = sse_out ? partial : nullptr;
        p.margin = margin_out;
        p.neartie = neartie_out;
        p.tie_tau = tie_tau;
        p.split = split ? 1 : 0;
        float *next = nullptr;
        if (l1 < L) next = (zin == ping) ? pong : ping;
        p.resid_next = next;
        const size_t lds = lds_bytes(rows, e, L, waves, split);
        int rc;
        if (e == 16) rc = dispatch<16>(p, threads, grid, lds, xq_out != nullptr, stream);
        else if (e == 32) rc = dispatch<32>(p, threads, grid, lds, xq_out != nullptr, stream);
        else rc = dispatch<64>(p, threads, grid, lds, xq_out != nullptr, stream);
        if (rc) return rc;
        if (sse_out && !ticket) {
            TraceScope trace(K_RQ_SSE_FINALIZE, stream);
            hipLaunchKernelGGL(rq_sse_finalize_kernel, dim3(1), dim3(64), 0, stream, partial, grid, L, l0, l1, sse_out);
            rc = check_launch("rq_sse_finalize_kernel");
            if (rc) return rc;
        }
        zin = next;
        l0 = l1;
    }
    return LCREC_OK;
}

}  // namespace lcrec
