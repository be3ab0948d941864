// Fused Linear (+ eval BatchNorm affine) (+ ReLU) for the RQ-VAE encoder on
// gfx950: y = epi(x @ W^T), fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
//
// Replaces one Linear/[BatchNorm1d]/[ReLU] group of MLPLayers.forward
// (reference index/models/layers.py:18-30,42).  On the reference's CPU path
// this chain is 94 % of get_indices' time (unfused clamp_min 54 %, bias copy
// 26 %, addmm 14 %); here bias, BN affine and ReLU are
// the epilogue of the MFMA tile.
//
// Arithmetic contract (oracle/lcrec_oracle.c, linear_rows): each output is ONE
// fp32 fma chain over k ascending from 0.  v_mfma_f32_32x32x2_f32 computes
// D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) with k0 = lanes 0-31, k1 = lanes 32-63,
// so feeding it k = 2s (low half) and 2s+1 (high half) for s ascending, into one
// accumulator, reproduces the chain exactly.  The forward product has no split-K; the backward product
// dW = dY^T X is an ordered sum of a few such chains over runs of the batch (linear_backward, below).
//
// Two kernels, one arithmetic (the dispatch is at the bottom of the file):
//   linear_fwd_kernel<WM,WN,TM,TN,...>  256 threads = 4 waves, block tile (WM*TM*32) x (WN*TN*32): 128x128 for
//                                   launches that do not fill the chip in whole rounds of the big tile, 64x64
//                                   for batch-sized ones, 128x64 / 128x32 for the narrow tail layers; several
//                                   workgroups per CU hide each other's stalls; also the k-major-operand
//                                   (backward) instantiations;
//   linear_fwd_pp2_kernel           512 threads, 256x128 tile, K % 32 == 0: the two waves of a SIMD alternate
//                                   between "64 MFMAs" and "stage the next tile" (see there); 92 % of the fp32
//                                   MFMA peak over the encoder's wide layers.
// Common: K step 32.  Operand tiles go global -> registers -> LDS (the next tile's loads are issued
// before the current tile's MFMAs).  LDS rows hold a 32-wide K slice with k de-interleaved inside
// each group of 8 ([k0 k2 k4 k6 | k1 k3 k5 k7]) so that one ds_read_b128 per lane yields the lane's
// operand for four consecutive MFMAs; rows are padded to 36 floats, which makes the ds_read_b128
// pattern bank-conflict free (16 consecutive rows cover 16 distinct 4-bank slots).
#include "common.h"

#include <stdlib.h>

namespace lcrec {

// In-kernel cycle stamps (diagnostic builds only: make STAMP=1, then lcrec_debug_gemm_stamps()).
#ifdef LCREC_GEMM_STAMP
__device__ unsigned long long g_stamps[8][64][4];
__device__ unsigned long long g_marks[8][4];      // kernel entry, prologue done, K loop done, epilogue done
#define LCREC_MARK(slot)                                                                           \
    do {                                                                                           \
        if (blockIdx.x == 8 && lane == 0) {                                                        \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
            g_marks[wave][slot] = t_;                                                              \
        }                                                                                          \
    } while (0)
#define LCREC_STAMP(slot)                                                                          \
    do {                                                                                           \
        if (blockIdx.x == 8 && lane == 0 && 2 * u + half < 64) {                                   \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
            g_stamps[wave][2 * u + half][slot] = t_;                                               \
        }                                                                                          \
    } while (0)
// generic (non ping-pong) kernel: workgroup with tile number 9, lane 0 of each wave, first 64 K-tiles, 6 points per K-tile
__device__ unsigned long long g_gen_stamps[4][64][6];
#define LCREC_GSTAMP(slot)                                                                         \
    do {                                                                                           \
        if (bid == 9 && split == 0 && lane == 0 && kt - kt0 < 64) {                                \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
            g_gen_stamps[wave][kt - kt0][slot] = t_;                                               \
        }                                                                                          \
    } while (0)
// whole-kernel marks of the same workgroup (entry, K loop start, K loop end, epilogue end) in the unused K-tile row 60
#define LCREC_GMARK(slot)                                                                          \
    do {                                                                                           \
        if (bid == 9 && split == 0 && lane == 0) {                                                 \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
            g_gen_stamps[wave][60][slot] = t_;                                                     \
        }                                                                                          \
    } while (0)
#else
#define LCREC_STAMP(slot) do { } while (0)
#define LCREC_MARK(slot) do { } while (0)
#define LCREC_GSTAMP(slot) do { } while (0)
#define LCREC_GMARK(slot) do { } while (0)
#endif

template <int N>
struct IntC { static constexpr int value = N; };

// f(IntC<0>{}) ... f(IntC<N-1>{}): a loop whose index is a compile-time constant in the body
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<N, I + 1>(f);
    }
}

constexpr int BK = 32;   // K slice per step
constexpr int LDK = 36;  // padded LDS row length in floats
#ifndef LCREC_GEMM_RING
#define LCREC_GEMM_RING 4   // register sets of the 64 x 64 kernel's global prefetch ring (see linear_tile_body)
#endif

template <int ROWS>
struct StageRegs {
    static constexpr int ITERS = (ROWS * 4 + 255) / 256;
    f32x4 v[ITERS][2];
};

// Load ROWS x 32 floats of a [rows_total][K] row-major matrix, starting at
// (row0, k0), into registers: thread p handles row p/4, k-group p%4 (8 floats).
template <int ROWS>
__device__ __forceinline__ void stage_load(StageRegs<ROWS> &r, const float *__restrict__ src,
                                           int64_t row0, int64_t rows_total, int K, int k0, int tid)
{
#pragma unroll
    for (int it = 0; it < StageRegs<ROWS>::ITERS; ++it) {
        const int p = tid + it * 256;
        const int row = p >> 2, kg = p & 3;
        const int64_t grow = row0 + row;
        const int k = k0 + kg * 8;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
        if (row < ROWS && grow < rows_total && k < K) {
            const f32x4 *g = reinterpret_cast<const f32x4 *>(src + grow * (int64_t)K + k);
            a = g[0];
            b = g[1];
        }
        r.v[it][0] = a;
        r.v[it][1] = b;
    }
}

template <int ROWS>
__device__ __forceinline__ void stage_store(const StageRegs<ROWS> &r, float *lds, int tid)
{
#pragma unroll
    for (int it = 0; it < StageRegs<ROWS>::ITERS; ++it) {
        const int p = tid + it * 256;
        const int row = p >> 2, kg = p & 3;
        if (row < ROWS) {
            const f32x4 a = r.v[it][0], b = r.v[it][1];
            f32x4 ev = {a[0], a[2], b[0], b[2]};
            f32x4 od = {a[1], a[3], b[1], b[3]};
            f32x4 *d = reinterpret_cast<f32x4 *>(lds + row * LDK + kg * 8);
            d[0] = ev;
            d[1] = od;
        }
    }
}

// Epilogue of one 32x32 accumulator tile (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) +
// 4*(lane>>5)): bias / folded BatchNorm / ReLU in registers, then a transpose through a wave-private
// 32 x LDK patch of LDS so that the global stores are 16 B per lane -- eight lanes write one whole
// 128-B row segment -- instead of 16 one-dword stores per lane.  The store tail of an MFMA tile is
// issue-bound, not bandwidth-bound, so a quarter of the store instructions is a quarter of the tail.
__device__ __forceinline__ void store_tile_32x32(const f32x16 &acc, float *stg, int lane, float *__restrict__ C,
                                                 int64_t row0, int64_t M, int col0, int N,
                                                 const float *__restrict__ bias, const float *__restrict__ bn_scale,
                                                 const float *__restrict__ bn_shift, int relu)
{
    const int c = lane & 31, h = lane >> 5;
    const int col = col0 + c;
    const bool has_bn = bn_scale != nullptr;
    float bj = 0.f, sc = 1.f, sh = 0.f;
    if (col < N) {
        bj = bias ? bias[col] : 0.f;
        if (has_bn) { sc = bn_scale[col]; sh = bn_shift[col]; }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float t = acc[r] + bj;
        if (has_bn) t = __builtin_fmaf(t, sc, sh);
        if (relu) t = (t > 0.f) ? t : 0.f;
        stg[((r & 3) + 8 * (r >> 2) + 4 * h) * LDK + c] = t;
    }
    __builtin_amdgcn_wave_barrier();          // LDS ops of one wave execute in order; this pins the compiler
    const bool vec = (N & 3) == 0;
    const int c4 = (lane & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (lane >> 3) + 8 * p;
        const int64_t grow = row0 + row;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(stg + row * LDK + c4);
        if (grow < M) {
            float *dst = C + grow * (int64_t)N + col0 + c4;
            if (vec && col0 + c4 + 3 < N) {
                *reinterpret_cast<f32x4 *>(dst) = v;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (col0 + c4 + t < N) dst[t] = v[t];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- VALU-free operand staging (see linear_fwd_pp2_kernel for why it matters): LDS byte address of a
// __shared__ pointer, the k de-interleave done by the LDS unit, and a raw buffer descriptor over the valid
// rows of an operand tile (rows past the matrix read as 0.0f, an empty extent touches no memory).
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

// a = k..k+3, b = k+4..k+7 of one row -> the row's group of 8 in LDS: [k0 k2 k4 k6 | k1 k3 k5 k7]
__device__ __forceinline__ void lds_store_deint8(uint32_t addr, const f32x4 &a, const f32x4 &b)
{
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:4" ::"v"(addr), "v"(a[0]), "v"(a[1]) : "memory");
    asm volatile("ds_write2_b32 %0, %1, %2 offset0:1 offset1:5" ::"v"(addr), "v"(a[2]), "v"(a[3]) : "memory");
    asm volatile("ds_write2_b32 %0, %1, %2 offset0:2 offset1:6" ::"v"(addr), "v"(b[0]), "v"(b[1]) : "memory");
    asm volatile("ds_write2_b32 %0, %1, %2 offset0:3 offset1:7" ::"v"(addr), "v"(b[2]), "v"(b[3]) : "memory");
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float *base, int64_t first_row, int64_t total_rows,
                                                            int tile_rows, int K)
{
    int64_t rows = total_rows - first_row;
    rows = rows < 0 ? 0 : (rows > tile_rows ? tile_rows : rows);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + first_row * (int64_t)K), 0, (int)(rows * K * 4),
                                             0x00020000);
}

__device__ __forceinline__ f32x4 buffer_load_f32x4(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// What a TRAINING step adds to a tile (lcrec_linear_bn_forward, lcrec_linear_backward_weights with input transforms):
//   PRO = 1  the row-major A operand is u = max(a * in_scale[k] + in_shift[k], in_lo) on its way into LDS -- the BatchNorm
//            affine + ReLU of the layer that produced `a`, which therefore never has to be written out (layers.py:25-30);
//   PRO = 2  the same for the k-major "W" operand of dW = dY^T X (X is the layer input; scale / shift run along its
//            columns), rows past the batch kept zero;
//   STATS    per-column batch statistics of the output t = acc + bias in the epilogue: every tile publishes, per column,
//            (pivot, sum (t - pivot), sum (t - pivot)^2) over its valid rows (pivot = the tile's first row: a sample of the
//            column, so the sums lose a bit or two, not the digits E[t^2] - mean^2 loses); the LAST tile of a column strip to
//            arrive (ticket per strip, include/lcrec.h) merges the row tiles' records in tile order (Chan et al.) and
//            writes mean, rstd, the folded scale / shift of this BatchNorm and its running statistics.
struct TileExtras {
    const float *in_scale, *in_shift;   // PRO: [K] (PRO 1) or [N] (PRO 2)
    float in_lo;                        // 0 (ReLU) or -inf
    float *stat_partial;                // STATS: [bm_blocks][3][N]
    unsigned *tickets;                  // STATS: [bn_blocks], zero before and after
    const float *gamma, *beta;          // STATS: [N] (NULL = 1 / 0)
    float eps, momentum;
    float *running_mean, *running_var;  // STATS: updated in place, or NULL
    float *mean_out, *rstd_out, *scale_out, *shift_out;   // STATS: [N]
};

// TA / TB: the A / W operand is given K-MAJOR ([K][M] resp. [K][N] row-major) instead of [M][K] / [N][K].  These are
// the operand shapes of the two backward products of a Linear layer -- dX = dY W reads W [out][in] as the
// k-major "W" of an [n][in] output, dW = dY^T X reads dY [n][out] and X [n][in] both k-major -- so no transposed
// copy of W, dY or X is ever made.  The transposition happens in the LDS write (ds_write2_b32 places the four
// rows of a 16-byte global load); chain order over k is unchanged.  Requires FAST.
// The tile body; `bid` / `split` are the workgroup's tile number and K-run (blockIdx.x / blockIdx.y of a plain launch, or
// what a grouped launch derives from its problem table).
template <int WAVES_M, int WAVES_N, int TM, int TN, bool FAST, bool TA = false, bool TB = false, int PRO = 0, bool STATS = false>
__device__ __forceinline__ void linear_tile_body(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    const float *__restrict__ bn_scale, const float *__restrict__ bn_shift, float *__restrict__ C,
    int64_t M, int N, int K, int relu, int bn_blocks, int bm_blocks, int tune, int kt_per_split, int64_t split_stride,
    unsigned bid, unsigned split, const TileExtras &ex = TileExtras{})
{
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    static_assert(PRO == 0 || (PRO == 1 && !TA) || (PRO == 2 && TB), "PRO 1: row-major A; PRO 2: k-major W");
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    // one allocation: the epilogue stages 4 x 32 output rows from its start, which is more than As when BM = 64
    static_assert(BM + BN >= 128, "the epilogue needs 128 staging rows");
    // DB (the one-accumulator tiles, 64 x 64 and 128 x 32, of batch-sized launches -- about one workgroup per CU, one wave per
    // SIMD, nothing else to run under a stall): two LDS buffers, two fragment register sets, ONE barrier per K-tile, and every
    // load, LDS store and fragment read of the NEXT K-tiles issued one per gap between the current K-tile's 16 MFMAs
    // (k_tile_rb below).  History of this path at 1024 x 2048 -> 1024, us per launch: 62 (one buffer, loads behind a branch)
    // -> 49 (prefetch distance 2, second LDS buffer) -> 38.7 (this form; the MFMA pipe alone needs 27.3): what the in-kernel
    // stamps (tools/rb_stamp_probe.py) and leave-one-out builds showed on the way is noted at each piece.
    constexpr bool DB = FAST && (TM * TN == 1 || (TM * TN == 2 && WAVES_M == 2));
    static_assert((PRO == 0 && !STATS) || DB, "the training-step extras live in the register-buffered K-tile");
    constexpr int BUF = (BM + BN) * LDK;
    __shared__ __attribute__((aligned(16))) float smem[BUF * (DB ? 2 : 1)];
    float *const As = smem, *const Ws = smem + BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // Block -> tile.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD
    // and its L2), so row panel p is given to XCD p%8 and all column blocks of a panel run there:
    // the activation panel is fetched into ONE L2 instead of up to eight.  Placement only affects speed.
    int64_t bm;
    int bn;
    if (tune & 1) {
        const int xcd = bid & 7, j = bid >> 3;
        const int panels = (bm_blocks - xcd + 7) >> 3;
        if (j >= panels * bn_blocks) return;
        bm = (int64_t)(j / bn_blocks) * 8 + xcd;
        bn = j % bn_blocks;
    } else {
        bm = bid / bn_blocks;
        bn = bid % bn_blocks;
    }
    const int64_t m0 = bm * BM;
    const int n0 = bn * BN;
    // (Measured and dropped: static per-workgroup s_setprio levels cost 4 %, start-up staggering of
    // co-resident workgroups changed nothing, double-buffered LDS with one barrier per K-tile lost 5-10 % at
    // Games-sized launches of the multi-accumulator tiles -- the second buffer costs a co-resident workgroup -- and
    // 128 x 64 / 64 x 128 tiles were slower than 64 x 64 at batch-sized launches on all but the widest layer.  A deeper
    // global prefetch ring changes nothing: 2 / 4 / 6 register sets 47.9 / 47.2 / 48.0 us before the loads were spread.)

    LCREC_GMARK(0);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // (non-DB tiles) two register sets: the global loads of K-tile kt+2 are issued before K-tile kt's MFMAs (prefetch
    // distance 2).  rocprofv3 counters on the batch-sized launches (1024 x 2048 -> 1024: 256 workgroups, one per CU) had shown
    // MFMA busy 48 % of the kernel and 55 % of every wave's cycles in s_waitcnt, four fifths of that on vmcnt.
    StageRegs<BM> ra, ra2;
    StageRegs<BN> rw, rw2;
    // split K (backward dW only, where K is the batch and the output is small): workgroup (tile, blockIdx.y) runs the
    // K-tiles [kt0, nk) of its split and writes a partial result at C + blockIdx.y * split_stride; a second kernel adds
    // the partials in split order.  Forward launches pass kt_per_split = all K-tiles, gridDim.y = 1.
    const int nk_all = (K + BK - 1) / BK;
    const int kt0 = split * kt_per_split;
    const int nk = kt0 + kt_per_split < nk_all ? kt0 + kt_per_split : nk_all;
    C += split * split_stride;

    static_assert(FAST || (!TA && !TB), "k-major operands need the buffer-load path");
    // FAST (K % 32 == 0): buffer loads + ds_write2_b32 -- no VALU in the staging path; thread p of a pass covers
    // row p/4, k-group p%4; threads beyond a narrow tile get an out-of-range offset (reads 0, stores nothing)
    // k-major operand S [K][R], tile rows r0..: the descriptor starts at column r0 and ends with the last valid
    // column of row K-1, so k >= K reads 0.0f by the range check; columns past R are masked per thread below
    auto kmajor_rsrc = [&](const float *base, int64_t r0, int64_t R, int tile_rows) {
        int64_t cols = R - r0;
        cols = cols < 0 ? 0 : (cols > tile_rows ? tile_rows : cols);
        const int64_t bytes = cols > 0 ? ((int64_t)(K - 1) * R + cols) * 4 : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + r0), 0, (int)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t a_rsrc = TA ? kmajor_rsrc(A, m0, M, BM) : tile_rsrc(A, m0, M, BM, K);
    const __amdgpu_buffer_rsrc_t w_rsrc = TB ? kmajor_rsrc(W, n0, N, BN) : tile_rsrc(W, n0, N, BN, K);
    // k-major staging: thread p of a pass loads the 16 bytes at (k = p / (ROWS/4), rows 4*(p % (ROWS/4)) .. +3) and
    // writes them to four LDS rows at k's de-interleaved slot (k steps of 8+ between passes keep the slot's low bits)
    // `live` (uniform): a K-tile past the end of this workgroup's run is still "loaded" -- with a scalar offset beyond the
    // descriptor's extent, so the range check returns zeros without touching memory.  The loads must be unconditional:
    // behind a branch, the compiler's s_waitcnt insertion merges the two paths and makes the LDS store of K-tile kt+1 wait
    // for vmcnt(0), i.e. for the loads of kt+2 just issued -- which is what made prefetch distance 2 worthless at first.
    constexpr unsigned SOFF_OUT = 0x7fffff00u;
    auto kmajor_load = [&](auto &r, __amdgpu_buffer_rsrc_t rs, auto rows_c, int64_t R, int64_t r0, int kt, bool live) {
        constexpr int ROWS = decltype(rows_c)::value, Q = ROWS / 4, KSTEP = 256 / Q, NL = ROWS / 32;
        const int kk = tid / Q, r4 = tid % Q;
        const bool ok = tid < Q * 32 && r0 + r4 * 4 < R;
        const int vo = ok ? (int)((kk * R + r4 * 4) * 4) : 0x7fffff00;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const unsigned so = live ? (unsigned)(((int64_t)kt * BK + j * KSTEP) * R * 4) : SOFF_OUT;
            r.v[j >> 1][j & 1] = buffer_load_f32x4(rs, vo, (int)so);
        }
    };
    auto kmajor_store = [&](const auto &r, float *lds, auto rows_c) {
        constexpr int ROWS = decltype(rows_c)::value, Q = ROWS / 4, KSTEP = 256 / Q, NL = ROWS / 32;
        const int kk = tid / Q, r4 = tid % Q;
        if (tid < Q * 32) {
            const int slot = (kk & ~7) + ((kk & 1) << 2) + ((kk >> 1) & 3);
            const uint32_t addr = lds_addr(lds) + (uint32_t)((r4 * 4 * LDK + slot) * 4);
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const f32x4 v = r.v[j >> 1][j & 1];
                const uint32_t a = addr + j * KSTEP * 4;
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:36" ::"v"(a), "v"(v[0]), "v"(v[1]) : "memory");
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:72 offset1:108" ::"v"(a), "v"(v[2]), "v"(v[3]) : "memory");
            }
        }
    };
    const int t_g = ((tid >> 2) * K + (tid & 3) * 8) * 4;
    const uint32_t t_s = (uint32_t)(((tid >> 2) * LDK + (tid & 3) * 8) * 4);
    // the ring's form of the same (load_piece / store_piece): byte offset of the thread's 16 bytes in a row's first 64, and
    // where its first value goes in the row's LDS image
    const int t_q = ((tid >> 2) * K) * 4 + (tid & 3) * 16;
    const uint32_t t_w = (uint32_t)(((tid >> 2) * LDK + ((tid & 3) >> 1) * 8 + ((tid & 3) & 1) * 2) * 4);
    auto fast_load = [&](auto &r, __amdgpu_buffer_rsrc_t rs, int rows, int kt, bool live) {
        constexpr int IT = sizeof(r.v) / sizeof(r.v[0]);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const bool in_tile = (tid >> 2) + it * 64 < rows;
            const int vo = in_tile ? t_g : 0x7fffff00;
            const unsigned so = live ? (unsigned)(kt * (BK * 4) + it * 64 * K * 4) : SOFF_OUT;   // (0x7fffff00 twice still < 2^32)
            r.v[it][0] = buffer_load_f32x4(rs, vo, (int)so);
            r.v[it][1] = buffer_load_f32x4(rs, vo + 16, (int)so);
        }
    };
    auto fast_store = [&](const auto &r, float *lds, int rows) {
        constexpr int IT = sizeof(r.v) / sizeof(r.v[0]);
#pragma unroll
        for (int it = 0; it < IT; ++it)
            if ((tid >> 2) + it * 64 < rows) lds_store_deint8(lds_addr(lds) + t_s + it * 64 * LDK * 4, r.v[it][0], r.v[it][1]);
    };
    auto stage_in = [&](int kt, StageRegs<BM> &ra_, StageRegs<BN> &rw_, bool live) {
        if constexpr (FAST) {
            if constexpr (TA) kmajor_load(ra_, a_rsrc, IntC<BM>{}, M, m0, kt, live);
            else fast_load(ra_, a_rsrc, BM, kt, live);
            if constexpr (TB) kmajor_load(rw_, w_rsrc, IntC<BN>{}, N, n0, kt, live);
            else fast_load(rw_, w_rsrc, BN, kt, live);
        } else if (live) {
            stage_load<BM>(ra_, A, m0, M, K, kt * BK, tid);
            stage_load<BN>(rw_, W, n0, N, K, kt * BK, tid);
        }
    };
    auto stage_out = [&](const StageRegs<BM> &ra_, const StageRegs<BN> &rw_) {
        if constexpr (FAST) {
            if constexpr (TA) kmajor_store(ra_, As, IntC<BM>{});
            else fast_store(ra_, As, BM);
            if constexpr (TB) kmajor_store(rw_, Ws, IntC<BN>{});
            else fast_store(rw_, Ws, BN);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the asm stores are invisible to the compiler's counters
        } else {
            stage_store<BM>(ra_, As, tid);
            stage_store<BN>(rw_, Ws, tid);
        }
    };

    if constexpr (!DB) {
        if (kt0 < nk) {
            stage_in(kt0, ra, rw, true);
            stage_out(ra, rw);
            stage_in(kt0 + 1, ra2, rw2, kt0 + 1 < nk);    // K-tile kt0+1 waits in the second set
        }
        __syncthreads();
    }

    const float *a_base = As + (wm * TM * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
    const float *w_base = Ws + (wn * TN * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;

    // one K-tile: loads of kt+2 into the set `ra_free` (K-tile kt left it for LDS an iteration ago), MFMAs of kt from LDS,
    // then K-tile kt+1 -- loaded a whole iteration ago into `ra_next` -- goes to LDS
    auto k_tile = [&](int kt, StageRegs<BM> &ra_free, StageRegs<BN> &rw_free, const StageRegs<BM> &ra_next, const StageRegs<BN> &rw_next) {
        LCREC_GSTAMP(0);
        stage_in(kt + 2, ra_free, rw_free, kt + 2 < nk);
        LCREC_GSTAMP(1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4 *>(a_base + i * 32 * LDK + g * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                wf[j] = *reinterpret_cast<const f32x4 *>(w_base + j * 32 * LDK + g * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], wf[j][q],
                                                                         acc[i][j], 0, 0, 0);
        }
        LCREC_GSTAMP(2);
        __syncthreads();
        LCREC_GSTAMP(3);
        if (kt + 1 < nk) {
            stage_out(ra_next, rw_next);
            LCREC_GSTAMP(4);
            __syncthreads();
        }
        LCREC_GSTAMP(5);
    };
    // DB form (the 64 x 64 tile): the operand fragments of K-tile kt+1 are read from LDS into a SECOND register set while
    // K-tile kt's 16 MFMAs run from the first, and K-tile kt+2 (global loads issued two iterations ago) is stored into the
    // buffer K-tile kt was read from -- one LDS instruction after each MFMA, so the in-order wave never queues more LDS
    // work than fits in an MFMA's 64 cycles.  One barrier per K-tile; nothing but the barrier itself is exposed:
    //   iteration kt:   global loads kt+RING+1 -> free set | MFMA(kt) from F[kt&1] | LDS reads kt+1 -> F[~kt&1] from buf[~kt&1]
    //                   | LDS stores kt+2 -> buf[kt&1] | wait, barrier
    // (first form: fragments read at the top of the iteration -- an exposed LDS round trip per K-tile with one workgroup
    // per CU -- and the stores in one block after the fourth MFMA, where the in-order wave stalled on the LDS queue: 1 770
    // cycles per K-tile against the MFMA pipe's 1 024.)  Stores and loads are unconditional (a K-tile past the end is
    // zeros from an out-of-range load; nobody multiplies it), so there is no branch for the waitcnt bookkeeping to merge.
    struct Frag { f32x4 a[TM][4], w[TN][4]; };
    // PRO 1: scale / shift of the K-tile whose A rows go to LDS in the NEXT iteration, per half of the thread's 32-byte share
    // (k = 32 kt + 16 half + 4 (tid & 3) ..+3: the same for all of the thread's rows); two sets, alternating with the K-tile
    struct ProRegs { f32x4 s[2], h[2]; };
    ProRegs pro[2];
    const __amdgpu_buffer_rsrc_t ps_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ex.in_scale), 0, PRO == 1 ? K * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ph_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ex.in_shift), 0, PRO == 1 ? K * 4 : 0, 0x00020000);
    auto pro_load = [&](ProRegs &q, int kt, auto half_c) {
        constexpr int half = decltype(half_c)::value;
        if constexpr (PRO == 1) {
            const int vo = (tid & 3) * 16 + half * 64;
            q.s[half] = buffer_load_f32x4(ps_rsrc, vo, kt * (BK * 4));        // (past K: zeros -- u = max(0, lo), never multiplied)
            q.h[half] = buffer_load_f32x4(ph_rsrc, vo, kt * (BK * 4));
        }
    };
    // PRO 2: the thread's four columns of the k-major operand never change: scale / shift once, in registers
    f32x4 pcs = {1.f, 1.f, 1.f, 1.f}, pch = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PRO == 2) {
        constexpr int Q = BN / 4;
        const int r4 = tid % Q;
        if (ex.in_scale && n0 + r4 * 4 < N) {
            pcs = *reinterpret_cast<const f32x4 *>(ex.in_scale + n0 + r4 * 4);
            pch = *reinterpret_cast<const f32x4 *>(ex.in_shift + n0 + r4 * 4);
        }
    }
    const float pro_lo = ex.in_lo;
    // A K-MAJOR operand (TA / TB: the backward products) keeps its LDS image k-major too in this form: [32 k][ROWS], no
    // padding.  Its global loads are 16 bytes = four rows of one k, so a K-tile goes to LDS with one conflict-free
    // ds_write_b128 per load (row-major image: four rows 36 floats apart per load, 16 r4 + 4 i + slot mod 32 -- the 32 lanes
    // of a store half on FOUR banks, 8 cycles per dword where one would do: ~1 000 LDS cycles per K-tile when both operands
    // are k-major, as in dW); a fragment is then four 4-byte reads at (k = 8g + 2q + lane/32, row): consecutive lanes,
    // consecutive dwords, and the compiler pairs them into ds_read2st64_b32.  No de-interleaving: k is the address.
    const float *ak_base = As + (lane >> 5) * BM + wm * TM * 32 + (lane & 31);
    const float *wk_base = Ws + (lane >> 5) * BN + wn * TN * 32 + (lane & 31);
    constexpr int NREAD = 4 * (TM + TN);              // fragment reads per K-tile: 32 rows x 8 k each, A's first
    auto frag_read = [&](Frag &f, auto slot_c, int buf) {
        constexpr int slot = decltype(slot_c)::value, g = slot & 3;
        if constexpr (slot < 4 * TM) {
            constexpr int i = slot >> 2;
            if constexpr (TA) {
#pragma unroll
                for (int q = 0; q < 4; ++q) f.a[i][g][q] = ak_base[buf * BUF + (8 * g + 2 * q) * BM + i * 32];
            } else {
                f.a[i][g] = *reinterpret_cast<const f32x4 *>(a_base + buf * BUF + i * 32 * LDK + g * 8);
            }
        } else {
            constexpr int j = (slot - 4 * TM) >> 2;
            if constexpr (TB) {
#pragma unroll
                for (int q = 0; q < 4; ++q) f.w[j][g][q] = wk_base[buf * BUF + (8 * g + 2 * q) * BN + j * 32];
            } else {
                f.w[j][g] = *reinterpret_cast<const f32x4 *>(w_base + buf * BUF + j * 32 * LDK + g * 8);
            }
        }
    };
    // quarter `piece` of a K-tile's LDS stores: operand A (0, 1) or W (2, 3), first or second half of the thread's share
    auto store_piece = [&](const StageRegs<BM> &ra_, const StageRegs<BN> &rw_, int buf, auto piece_c, int kt_s = 0,
                           const ProRegs *q = nullptr) {
        constexpr int piece = decltype(piece_c)::value, half = piece & 1;
        auto one = [&](const auto &r, float *lds, auto rows_c, auto kmajor_c, auto pro_c) {
            constexpr int ROWS = decltype(rows_c)::value;
            constexpr int pro_kind = decltype(pro_c)::value;
            if constexpr (decltype(kmajor_c)::value) {
                constexpr int Q = ROWS / 4, KSTEP = 256 / Q, NL = ROWS / 32;      // (Q * 32 >= 256: every thread has a share)
                const int kk = tid / Q, r4 = tid % Q;
                const uint32_t addr = lds_addr(lds) + (uint32_t)((kk * ROWS + r4 * 4) * 4);
#pragma unroll
                for (int j = half; j < NL; j += 2) {
                    f32x4 v = r.v[j >> 1][j & 1];
                    if constexpr (pro_kind == 2) {
                        // this load's k (a batch row) is kt_s * 32 + j * KSTEP + kk; rows past the batch stay zero
                        const bool in_batch = kt_s * BK + j * KSTEP + kk < K;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            float u = __builtin_fmaf(v[t], pcs[t], pch[t]);
                            u = u > pro_lo ? u : pro_lo;
                            v[t] = in_batch ? u : 0.f;
                        }
                    }
                    asm volatile("ds_write_b128 %0, %1" ::"v"(addr + (uint32_t)(j * KSTEP * ROWS * 4)), "v"(v) : "memory");
                }
            } else {
                constexpr int IT = sizeof(r.v) / sizeof(r.v[0]);
#pragma unroll
                for (int it = 0; it < IT; ++it)
                    if (ROWS % 64 == 0 || (tid >> 2) + it * 64 < ROWS) {          // (a 64-row multiple: every thread has a share)
                        // k = 16*half + 4c + i (c = tid & 3) -> group of eight 2*half + c/2, slot (k&1)*4 + (k%8)/2: i = 0, 2 are
                        // neighbours, i = 1, 3 four slots further.  Banks: 4*row + 8*(c/2) + 2*(c&1) (+1) -- the 32 lanes of a
                        // store half spread two-deep over 16 banks, which a ds_write2_b32's transfer time covers: no conflicts,
                        // no v_mov (the first form's four ds_write2_b32 per eight floats landed on 8 banks each: 4x)
                        const uint32_t a = lds_addr(lds) + t_w + (uint32_t)((it * 64 * LDK + half * 16) * 4);
                        f32x4 v = r.v[it][half];
                        if constexpr (pro_kind == 1) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const float u = __builtin_fmaf(v[t], q->s[half][t], q->h[half][t]);
                                v[t] = u > pro_lo ? u : pro_lo;
                            }
                        }
                        asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(a), "v"(v[0]), "v"(v[2]) : "memory");
                        asm volatile("ds_write2_b32 %0, %1, %2 offset0:4 offset1:5" ::"v"(a), "v"(v[1]), "v"(v[3]) : "memory");
                    }
            }
        };
        if constexpr (piece < 2) one(ra_, As + buf * BUF, IntC<BM>{}, IntC<TA ? 1 : 0>{}, IntC<PRO == 1 ? 1 : 0>{});
        else one(rw_, Ws + buf * BUF, IntC<BN>{}, IntC<TB ? 1 : 0>{}, IntC<PRO == 2 ? 2 : 0>{});
    };
    // quarter `piece` of a K-tile's global loads, split like the stores: operand A (0, 1) or W (2, 3), first or second 16 bytes
    // of each of the thread's 32-byte shares.  Issued one per MFMA gap: at the top of the iteration the four waves' sixteen
    // 1-KB loads reached the texture-address unit (64 B per clock) together and every wave sat in its issue for 150-300 cycles.
    auto load_piece = [&](StageRegs<BM> &ra_, StageRegs<BN> &rw_, int kt, bool live, auto piece_c) {
        constexpr int piece = decltype(piece_c)::value, half = piece & 1;
        auto one = [&](auto &r, __amdgpu_buffer_rsrc_t rs, auto rows_c, auto kmajor_c, int64_t R, int64_t r0) {
            constexpr int ROWS = decltype(rows_c)::value;
            if constexpr (decltype(kmajor_c)::value) {
                constexpr int Q = ROWS / 4, KSTEP = 256 / Q, NL = ROWS / 32;
                const int kk = tid / Q, r4 = tid % Q;
                const bool ok = tid < Q * 32 && r0 + r4 * 4 < R;
                const int vo = ok ? (int)((kk * R + r4 * 4) * 4) : 0x7fffff00;
#pragma unroll
                for (int j = half; j < NL; j += 2) {
                    const unsigned so = live ? (unsigned)(((int64_t)kt * BK + j * KSTEP) * R * 4) : SOFF_OUT;
                    r.v[j >> 1][j & 1] = buffer_load_f32x4(rs, vo, (int)so);
                }
            } else {
                constexpr int IT = sizeof(r.v) / sizeof(r.v[0]);
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    // the four lanes of a row take 64 CONTIGUOUS bytes per instruction (k = 16*half + 4*(tid&3) ..+3), not four
                    // 16-byte pieces 32 bytes apart as the prologue's two-loads-per-thread form does
                    const bool in_tile = (tid >> 2) + it * 64 < ROWS;
                    const int vo = in_tile ? t_q : 0x7fffff00;
                    const unsigned so = live ? (unsigned)(kt * (BK * 4) + it * 64 * K * 4) : SOFF_OUT;
                    r.v[it][half] = buffer_load_f32x4(rs, vo + half * 64, (int)so);
                }
            }
        };
        if constexpr (piece < 2) one(ra_, a_rsrc, IntC<BM>{}, IntC<TA ? 1 : 0>{}, M, m0);
        else one(rw_, w_rsrc, IntC<BN>{}, IntC<TB ? 1 : 0>{}, N, n0);
    };
    auto k_tile_rb = [&](int kt, StageRegs<BM> &ra_free, StageRegs<BN> &rw_free, const StageRegs<BM> &ra_next, const StageRegs<BN> &rw_next,
                         const Frag &fc, Frag &fn, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value;
        LCREC_GSTAMP(0);
        const bool live = kt + LCREC_GEMM_RING + 1 < nk;
        LCREC_GSTAMP(1);
        static_for<16 * TM * TN>([&](auto s_c) {
            // MFMA s of the K-tile: k-step (g, q) outermost, then the wave's accumulators -- each accumulator's chain runs over k
            // ascending, and with two of them consecutive MFMAs are independent
            constexpr int s = decltype(s_c)::value, j = s % TN, i = (s / TN) % TM, kq = s / (TN * TM), g = kq >> 2, q = kq & 3;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fc.a[i][g][q], fc.w[j][g][q], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // gap s carries: fragment read s (s < NREAD); load quarter s (s < 4); store quarter (s - 5) / 2 (s = 5, 7, 9, 11) -- the
            // last LDS operation leaves at least four MFMAs before the wait.  Measured at 1024 x 2048 -> 1024 (64 x 64 tile):
            // 38.7 us; loads in gaps 0, 2, 4, 6 and stores in 8 .. 14: 40.1; stores first (1 .. 7) and loads last (9 .. 15): 40.0.
            // Leaving out, one at a time, the reads / the barrier / the loads (wrong results, timing only): 36.9 / 38.3 / 36.1.
            if constexpr (s < NREAD) frag_read(fn, IntC<s>{}, cur ^ 1);
            if constexpr (s < 4) load_piece(ra_free, rw_free, kt + LCREC_GEMM_RING + 1, live, IntC<s>{});
            // (PRO 1) scale / shift of K-tile kt+3, whose A rows iteration kt+1 stores: two 16-byte loads in each of gaps 12 and 13,
            // half an iteration ahead of their use, into the set the previous iteration's stores have finished with
            if constexpr (PRO == 1 && (s == 12 || s == 13)) pro_load(pro[cur ^ 1], kt + 3, IntC<s - 12>{});
            if constexpr (s >= 5 && s <= 11 && (s & 1) == 1) store_piece(ra_next, rw_next, cur, IntC<(s - 5) / 2>{}, kt + 2, &pro[cur]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s == 7) LCREC_GSTAMP(2);
        });
        LCREC_GSTAMP(3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads and its stores
        LCREC_GSTAMP(4);
        __syncthreads();
        LCREC_GSTAMP(5);
    };
    if constexpr (DB) {
        // Global prefetch: a ring of RING register sets; K-tile kt0+j lives in set j % RING, is loaded RING-1 iterations
        // before it is stored to LDS, i.e. RING-1 K-tiles (16 KB each) are in flight per workgroup.  (The depth turned out
        // not to matter -- see above; 4 leaves slack for a slow first touch at no cost but 32 registers.)
        constexpr int RING = LCREC_GEMM_RING;
        static_assert(RING >= 2 && RING % 2 == 0, "the fragment sets alternate with the parity of the ring slot");
        Frag fa, fb;
        StageRegs<BM> sa[RING];
        StageRegs<BN> sw[RING];
        if (kt0 < nk) {
            // K-tiles 0 and 1 through sets 0 and 1 into the two buffers, then the ring is filled: K-tiles 2 .. RING-1 into their
            // own sets, K-tile RING into set 0 (whose K-tile is in LDS by then); iteration 0 loads K-tile RING+1 into set 1
            static_for<4>([&](auto p_c) { load_piece(sa[0], sw[0], kt0, true, p_c); });
            static_for<4>([&](auto p_c) { load_piece(sa[1], sw[1], kt0 + 1, kt0 + 1 < nk, p_c); });
            static_for<RING - 2>([&](auto j_c) {
                constexpr int j = decltype(j_c)::value + 2;
                static_for<4>([&](auto p_c) { load_piece(sa[j], sw[j], kt0 + j, kt0 + j < nk, p_c); });
            });
            if constexpr (PRO == 1) {
                static_for<2>([&](auto h_c) { pro_load(pro[0], kt0, h_c); pro_load(pro[1], kt0 + 1, h_c); });
            }
            static_for<4>([&](auto p_c) { store_piece(sa[0], sw[0], 0, p_c, kt0, &pro[0]); });
            static_for<4>([&](auto p_c) { store_piece(sa[1], sw[1], 1, p_c, kt0 + 1, &pro[1]); });
            if constexpr (PRO == 1) {
                // iteration 0 stores K-tile 2 with set 0 and loads K-tile 3's into set 1
                static_for<2>([&](auto h_c) { pro_load(pro[0], kt0 + 2, h_c); });
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for<4>([&](auto p_c) { load_piece(sa[0], sw[0], kt0 + RING, kt0 + RING < nk, p_c); });
            __syncthreads();
            static_for<NREAD>([&](auto s_c) { frag_read(fa, s_c, 0); });
            // iteration 0 stores K-tile 2 into buffer 0: not before every wave has its fragments of K-tile 0 out of it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
        LCREC_GMARK(1);
        for (int kt = kt0; kt < nk; kt += RING) {
            static_for<RING>([&](auto u_c) {
                constexpr int u = decltype(u_c)::value;
                if (u == 0 || kt + u < nk) {
                    if constexpr (u % 2 == 0) k_tile_rb(kt + u, sa[(u + 1) % RING], sw[(u + 1) % RING], sa[(u + 2) % RING], sw[(u + 2) % RING], fa, fb, IntC<0>{});
                    else k_tile_rb(kt + u, sa[(u + 1) % RING], sw[(u + 1) % RING], sa[(u + 2) % RING], sw[(u + 2) % RING], fb, fa, IntC<1>{});
                }
            });
        }
    } else {
        for (int kt = kt0; kt < nk; kt += 2) {              // unrolled by two so that both register sets are static
            k_tile(kt, ra, rw, ra2, rw2);
            if (kt + 1 < nk) k_tile(kt + 1, ra2, rw2, ra, rw);
        }
    }

    LCREC_GMARK(2);
    // epilogue (every wave passed the loop's last barrier after its final LDS operand read, so the
    // activation tile's LDS can be reused: 32 rows per wave)
    // ---- STATS, first half (before the tile's stores): batch statistics of t = acc + bias, per column (see TileExtras).  LDS: the
    // second K-tile buffer, which nobody reads any more (the stores' staging patches are the first buffer).
    constexpr int RB = BM / 32;                           // 32-row blocks of the tile
    float *piv = smem + BUF;                              // [BN]  the tile's first row
    float *red = piv + BN;                                // [RB][BN][2]
    int *last_sh = reinterpret_cast<int *>(red + RB * BN * 2);
    unsigned taken = 0;
    if constexpr (STATS) {
        const int c = lane & 31, h = lane >> 5;
        float bj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + c;
            bj[j] = (bias && col < N) ? bias[col] : 0.f;
            if (wm == 0 && h == 0) piv[wn * TN * 32 + j * 32 + c] = acc[0][j][0] + bj[j];      // row m0 (< M: the tile exists)
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float pv = piv[wn * TN * 32 + j * 32 + c];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int64_t rbase = m0 + wm * TM * 32 + i * 32 + 4 * h;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (rbase + (r & 3) + 8 * (r >> 2) < M) {
                        const float d = (acc[i][j][r] + bj[j]) - pv;
                        s1 += d;
                        s2 = __builtin_fmaf(d, d, s2);
                    }
                }
                s1 += __shfl_xor(s1, 32, 64);             // the two half-waves hold the other 16 rows of the column
                s2 += __shfl_xor(s2, 32, 64);
                if (h == 0) {
                    float *dst = red + (((wm * TM + i) * BN) + wn * TN * 32 + j * 32 + c) * 2;
                    dst[0] = s1;
                    dst[1] = s2;
                }
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < N) {
            float s1 = red[tid * 2], s2 = red[tid * 2 + 1];
#pragma unroll
            for (int b = 1; b < RB; ++b) { s1 += red[(b * BN + tid) * 2]; s2 += red[(b * BN + tid) * 2 + 1]; }
            float *rec = ex.stat_partial + ((size_t)bm * 3) * N + n0 + tid;
            handoff_put(rec, piv[tid]);
            handoff_put(rec + N, s1);
            handoff_put(rec + 2 * (size_t)N, s2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave's records have reached memory ...
        __syncthreads();                                   // ... before ONE thread takes the strip's ticket -- whose round trip
        if (tid == 0) taken = ticket_take(ex.tickets + bn);   // through the fabric then passes under the tile's stores below
    }
    float *stg = As + wave * 32 * LDK;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
            store_tile_32x32(acc[i][j], stg, lane, C, m0 + wm * TM * 32 + i * 32, M, n0 + wn * TN * 32 + j * 32, N, bias,
                             bn_scale, bn_shift, relu);
    LCREC_GMARK(3);
    if constexpr (STATS) {
        if (tid == 0) *last_sh = ticket_finish(ex.tickets + bn, taken, (unsigned)bm_blocks) ? 1 : 0;
        __syncthreads();
        if (*last_sh) {
            // the strip's last tile (uniform over the workgroup): merge the row tiles' (n_T, mean_T, M2_T) in tile order -- the same
            // bits whichever tile comes last.  The acquire in ticket_is_last + the barrier above make plain loads of the records
            // valid; all 256 threads fetch a chunk of them into LDS side by side (one memory round trip, not one per tile), then
            // one thread per column walks the chunk in order.  Two passes (mean, then M2 about it), as lcrec_bn_merge_stats.
            constexpr int CH = BUF / (3 * BN);            // row tiles per chunk (the first K-tile buffer: the staging patches are done)
            float *rec_l = smem;
            const bool mine = tid < BN && n0 + tid < N;
            const int col = n0 + tid;
            const float total = (float)M;
            float acc_mean = 0.f, mean = 0.f, m2 = 0.f;
            for (int pass = 0; pass < 2; ++pass) {
                for (int b0 = 0; b0 < bm_blocks; b0 += CH) {
                    const int nb = bm_blocks - b0 < CH ? bm_blocks - b0 : CH;
                    if (pass == 0 || bm_blocks > CH) {
                        __syncthreads();
                        for (int e = tid; e < nb * 3 * BN; e += 256) {
                            const int cidx = e % BN, bf = e / BN;          // bf = tile * 3 + field
                            rec_l[e] = n0 + cidx < N ? ex.stat_partial[((size_t)b0 * 3 + bf) * N + n0 + cidx] : 0.f;
                        }
                        __syncthreads();
                    }
                    if (mine) {
                        for (int b = 0; b < nb; ++b) {
                            const int64_t left = M - (int64_t)(b0 + b) * BM;
                            const float nt = (float)(left < BM ? left : BM);
                            const float pv = rec_l[(b * 3) * BN + tid], s1 = rec_l[(b * 3 + 1) * BN + tid], s2 = rec_l[(b * 3 + 2) * BN + tid];
                            const float dm = s1 / nt;
                            if (pass == 0) {
                                acc_mean += nt * (pv + dm);
                            } else {
                                float m2_t = s2 - s1 * dm;
                                m2_t = m2_t > 0.f ? m2_t : 0.f;
                                const float d = (pv + dm) - mean;
                                m2 += m2_t + nt * (d * d);
                            }
                        }
                    }
                }
                if (pass == 0) mean = acc_mean / total;
            }
            if (!mine) return;
            const float rstd = 1.0f / __builtin_sqrtf(m2 / total + ex.eps);
            const float sc = (ex.gamma ? ex.gamma[col] : 1.0f) * rstd;
            ex.mean_out[col] = mean;
            ex.rstd_out[col] = rstd;
            ex.scale_out[col] = sc;
            ex.shift_out[col] = __builtin_fmaf(-mean, sc, ex.beta ? ex.beta[col] : 0.0f);
            if (ex.running_mean) ex.running_mean[col] = (1.0f - ex.momentum) * ex.running_mean[col] + ex.momentum * mean;
            if (ex.running_var)
                ex.running_var[col] = (1.0f - ex.momentum) * ex.running_var[col] + ex.momentum * (m2 / (total > 1.f ? total - 1.f : 1.f));
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, bool FAST, bool TA = false, bool TB = false>
__global__ __launch_bounds__(256) void linear_fwd_kernel(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    const float *__restrict__ bn_scale, const float *__restrict__ bn_shift, float *__restrict__ C,
    int64_t M, int N, int K, int relu, int bn_blocks, int bm_blocks, int tune, int kt_per_split, int64_t split_stride)
{
    linear_tile_body<WAVES_M, WAVES_N, TM, TN, FAST, TA, TB>(A, W, bias, bn_scale, bn_shift, C, M, N, K, relu, bn_blocks, bm_blocks,
                                                             tune, kt_per_split, split_stride, blockIdx.x, blockIdx.y);
}

// The training step's forward product (lcrec_linear_bn_forward): register-buffered tiles only, extras by value.
template <int WAVES_M, int WAVES_N, int TM, int TN, int PRO, bool STATS>
__global__ __launch_bounds__(256) void linear_train_fwd_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                               const float *__restrict__ bias, float *__restrict__ C, int64_t M, int N,
                                                               int K, int bn_blocks, int bm_blocks, int tune, TileExtras ex)
{
    linear_tile_body<WAVES_M, WAVES_N, TM, TN, true, false, false, PRO, STATS>(A, W, bias, nullptr, nullptr, C, M, N, K, 0, bn_blocks,
                                                                             bm_blocks, tune, 1 << 30, (int64_t)0, blockIdx.x, 0u, ex);
}

// Several independent weight-gradient products dW_p = dY_p^T X_p in ONE launch (64 x 64 tiles, both operands k-major):
// the narrow layers of a training step are a handful of tiles each and leave most CUs idle when launched one by one;
// side by side their workgroups fill the chip.  Workgroup w belongs to the problem whose [wg_start, wg_start + wgs)
// range holds it; inside a problem, w - wg_start = split * tiles + tile.
constexpr int DW_GROUP_MAX = 16;
struct DwGroup {
    int count;
    const float *A[DW_GROUP_MAX], *B[DW_GROUP_MAX];
    float *C[DW_GROUP_MAX];                      // partial buffer when splits > 1, else the gradient itself
    int M[DW_GROUP_MAX], N[DW_GROUP_MAX], K[DW_GROUP_MAX];
    int bn_blocks[DW_GROUP_MAX], bm_blocks[DW_GROUP_MAX], tiles[DW_GROUP_MAX], kt_per_split[DW_GROUP_MAX];
    unsigned wg_start[DW_GROUP_MAX + 1];
    // (PRO form) the layer input X_p is given before ITS BatchNorm + ReLU: u = max(x * xs[col] + xh[col], lo); NULL = as it is
    const float *xs[DW_GROUP_MAX], *xh[DW_GROUP_MAX];
    float lo[DW_GROUP_MAX];
};

template <bool PRO>
__global__ __launch_bounds__(256) void linear_dw_grouped_kernel(DwGroup g)
{
    int p = 0;
#pragma unroll 1
    while (p + 1 < g.count && blockIdx.x >= g.wg_start[p + 1]) ++p;
    const unsigned local = blockIdx.x - g.wg_start[p];
    const unsigned tile = local % (unsigned)g.tiles[p], split = local / (unsigned)g.tiles[p];
    TileExtras ex = {};
    if constexpr (PRO) {
        ex.in_scale = g.xs[p];
        ex.in_shift = g.xh[p];
        ex.in_lo = g.lo[p];
    }
    linear_tile_body<2, 2, 1, 1, true, true, true, PRO ? 2 : 0>(g.A[p], g.B[p], nullptr, nullptr, nullptr, g.C[p], g.M[p], g.N[p], g.K[p], 0,
                                                              g.bn_blocks[p], g.bm_blocks[p], 1, g.kt_per_split[p],
                                                              (int64_t)g.M[p] * g.N[p], tile, split, ex);
}

struct DwReduce {
    int count;
    const float *partial[DW_GROUP_MAX];
    float *out[DW_GROUP_MAX];
    int splits[DW_GROUP_MAX];
    unsigned total4[DW_GROUP_MAX], wg_start[DW_GROUP_MAX + 1];
};

// gw_p = ((p0 + p1) + p2) + ... for every problem with more than one K-run, one launch
__global__ __launch_bounds__(256) void splitk_reduce_grouped_kernel(DwReduce r)
{
    int p = 0;
#pragma unroll 1
    while (p + 1 < r.count && blockIdx.x >= r.wg_start[p + 1]) ++p;
    const unsigned q = (blockIdx.x - r.wg_start[p]) * 256u + threadIdx.x;
    if (q >= r.total4[p]) return;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(r.partial[p]);
    f32x4 acc = src[q];
    for (int s2 = 1; s2 < r.splits[p]; ++s2) acc = acc + src[(size_t)s2 * r.total4[p] + q];
    reinterpret_cast<f32x4 *>(r.out[p])[q] = acc;
}

// ------------------------------------------------------------------------------------------
// Ping-pong kernel for the wide layers (K % 32 == 0): 512 threads = 8 waves = two groups of 4; a workgroup
// computes a 256 x 128 tile, group g owning rows [128g, 128g+128).  The two waves that share a SIMD belong
// to different groups and ALTERNATE roles every phase (one barrier per phase):
//
//   phase 2u   : group 0 issues its global loads for K-tile u+1, then runs K-tile u's 64 MFMAs;
//                group 1 writes the tile it loaded last phase (A1[u], upper half of W[u+1]) to LDS
//   phase 2u+1 : group 1 loads (A1[u+1], upper W[u+2]) and runs K-tile u's MFMAs;
//                group 0 writes (A0[u+1], lower half of W[u+1]) to LDS
//
// so each SIMD's MFMA pipe always has exactly one wave in its MFMA phase while its partner stages, instead of
// co-resident workgroups drifting into the same phase.  A0/A1 need one LDS buffer each (a group reads and
// writes its own tile in different phases); W is double-buffered because one group still reads W[u] while the
// other writes W[u+1].  Arithmetic is unchanged: one fma chain per output over k ascending.
//
// Which ISSUE PORT the non-MFMA work uses decides whether this runs at the pipe's rate.  In-kernel cycle stamps
// (tools/stamp_probe.py) showed that a wave with back-to-back fp32 MFMAs queued keeps the SIMD's VALU port: its
// partner's VALU instructions get through about once per 64-cycle MFMA, so a staging role made of 48 v_mov (the
// k de-interleave) + 12 ds_write_b128 + 64-bit address adds took 2 400 cycles instead of ~200 and the computing
// wave sat 500-700 cycles per phase at the barrier waiting for it (first version of this kernel: 125 TFLOP/s).
// LDS, vector-memory and scalar instructions issue on their own ports and are not held up.  So here:
//   * the de-interleave is done by the LDS unit: ds_write2_b32 places any two registers at any two dword
//     offsets, so (k0,k1) -> slots 0 and 4, (k2,k3) -> 1 and 5, ... : 24 LDS writes, zero VALU;
//   * global loads are buffer loads: a scalar resource descriptor per tile (base = the tile's first row,
//     extent = its valid rows), a per-thread byte offset that never changes, and the K-tile offset in an
//     SGPR advanced by the scalar unit -- no VALU address arithmetic, and rows past M or N read as zero
//     through the descriptor's range check instead of through compares and selects;
//   * the K loop is unrolled by two so both W buffers are compile-time LDS offsets;
//   * sched_barriers pin each fragment refill right behind the MFMAs that free its registers;
//   * after the mid-phase barrier the computing wave's remaining MFMAs run at raised priority (see there).
// ------------------------------------------------------------------------------------------
// Measured and rejected on MI355X (tools/gemm_probe.py, TFLOP/s on the 768->2048 / 2048->1024 layers; this form: 141 / 151):
//   K slice of 64 per phase (half the barriers, 139 KB of LDS)                       133 / 144
//   two workgroups per CU (register-capped to 128 VGPRs, 2 x 74 KB LDS)               137 / 140
//   fragment refills dealt out one ds_read per MFMA, or staggered between the waves   123-132 / 129-140
//   persistent workgroups with run-time decisions inside every phase (which descriptor, which epilogue piece):
//   bit-exact, but the K loop lost 5-8 %.  The cause was found later (a uniform boolean materialised through a VGPR
//   costs VALU instructions, which starve in the staging role) and the idea lives on as linear_fwd_pp3_kernel below,
//   whose steady-state loop is this kernel's loop instruction for instruction.
//   A control experiment -- THIS kernel's body in a loop over a static list of tiles, 256 workgroups -- runs at exactly
//   the speed of one workgroup per tile: neither the hardware's workgroup turn-around nor static assignment costs
//   anything measurable here.
//   Also without effect: prefetching two K-tiles ahead instead of one (second register set), unrolling the K loop by
//   four (the ~400-cycle barrier delay the stamps show once per loop iteration is an artefact of the stamp build).
//   Natural k order in LDS (6 ds_write_b128 in the staging role instead of 12 ds_write2_b32, but 2 ds_read2_b32 per
//   fragment in the computing role instead of 1 ds_read_b128)                         133 / 141
//   256 x 256 workgroup tile (each wave 64 x 128, 128 MFMAs per phase, 222 VGPRs)      142 / 150  (= the persistent form;
//   the epilogue is proportional to the output, so a bigger tile amortises only the prologue)
// What is left is per-tile: ~3.5 k cycles of prologue and ~10 k of epilogue (all 256 CUs store their 128 KB
// tiles at the same moment) against 8 225 cycles per K-tile in the loop (ideal 8 192).
__global__ __launch_bounds__(512) void linear_fwd_pp2_kernel(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    const float *__restrict__ bn_scale, const float *__restrict__ bn_shift, float *__restrict__ C,
    int64_t M, int N, int K, int relu, int bn_blocks, int bm_blocks, int xcd_order)
{
    constexpr int GM = 128, BN = 128, BKT = BK;
    constexpr int LDT = BKT + 4;                 // padded LDS row (floats): 16 consecutive rows hit 16 distinct 16-B slots
    constexpr int KG = BKT / 8;                  // 8-float groups per row = fragment groups per K slice
    constexpr int RPI = 256 / KG;                // tile rows one pass of the group's 256 threads covers
    constexpr int ITA = GM / RPI, ITW = 64 / RPI;
    extern __shared__ __attribute__((aligned(16))) float pp2_lds[];
    float *As0 = pp2_lds;                        // [2][GM * LDT]
    float *Ws0 = pp2_lds + 2 * GM * LDT;         // [2][BN * LDT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    const int gt = tid & 255;

    int64_t bm;
    int bn;
    if (xcd_order) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int panels = (bm_blocks - xcd + 7) >> 3;
        if (j >= panels * bn_blocks) return;
        bm = (int64_t)(j / bn_blocks) * 8 + xcd;
        bn = j % bn_blocks;
    } else {
        bm = blockIdx.x / bn_blocks;
        bn = blockIdx.x % bn_blocks;
    }
    const int64_t m0 = bm * (2 * GM) + grp * GM;
    const int n0 = bn * BN;
    const int64_t w_row0 = n0 + grp * 64;

    LCREC_MARK(0);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BKT;
    float *my_a = As0 + grp * GM * LDT;

    // global side: one buffer descriptor per operand tile (stride 0 = raw; extent = the tile's valid rows, so
    // rows past the matrix read as 0.0f and an all-out-of-range tile touches no memory), a fixed per-thread
    // byte offset (row gt/KG, k-group gt%KG), and the K-tile (and row-pass) byte offset in an SGPR
    const __amdgpu_buffer_rsrc_t a_rsrc = tile_rsrc(A, m0, M, GM, K);
    const __amdgpu_buffer_rsrc_t w_rsrc = tile_rsrc(W, w_row0, N, 64, K);
    const int t_g = ((gt / KG) * K + (gt % KG) * 8) * 4;
    const int pass_g = RPI * K * 4;
    // LDS side: this thread's 8-float slot in row gt/KG (+ RPI per pass) of a tile
    const uint32_t t_s = (uint32_t)(((gt / KG) * LDT + (gt % KG) * 8) * 4);
    uint32_t a_s[ITA], w_s[2][ITW];
#pragma unroll
    for (int it = 0; it < ITA; ++it) a_s[it] = lds_addr(my_a) + t_s + it * RPI * LDT * 4;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int it = 0; it < ITW; ++it) w_s[b][it] = lds_addr(Ws0 + b * BN * LDT + grp * 64 * LDT) + t_s + it * RPI * LDT * 4;

    f32x4 ra[ITA][2], rw[ITW][2];   // this group's activation tile (128 x BKT) and its half of the weight tile (64 x BKT)
    auto load_a = [&](int kt) {
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            const int so = kt * (BKT * 4) + it * pass_g;
            ra[it][0] = buffer_load_f32x4(a_rsrc, t_g, so);
            ra[it][1] = buffer_load_f32x4(a_rsrc, t_g + 16, so);
        }
    };
    auto load_w = [&](int kt) {
#pragma unroll
        for (int it = 0; it < ITW; ++it) {
            const int so = kt * (BKT * 4) + it * pass_g;
            rw[it][0] = buffer_load_f32x4(w_rsrc, t_g, so);
            rw[it][1] = buffer_load_f32x4(w_rsrc, t_g + 16, so);
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int it = 0; it < ITA; ++it) lds_store_deint8(a_s[it], ra[it][0], ra[it][1]);
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int it = 0; it < ITW; ++it) lds_store_deint8(w_s[buf][it], rw[it][0], rw[it][1]);
    };
    auto lds_drain = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };   // the asm stores are invisible to the compiler's counters

    // ---- prologue: W[0] (both halves) and A0[0] into LDS; group 1 leaves (A1[0], upper W[1]) in registers
    load_w(0);
    store_w(0);
    load_a(0);
    if (grp == 0) {
        store_a();
    } else if (nk > 1) {
        load_w(1);
    }
    lds_drain();
    __syncthreads();
    LCREC_MARK(1);

    const float *a_base = my_a + (wm * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int w_off = (wn * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;

    auto mfma_group = [&](const f32x4 (&af)[2], const f32x4 (&wf)[2]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], wf[j][q], acc[i][j], 0, 0, 0);
    };

    // one phase of K-tile u (PAR = u & 1 as a constant, half = which group computes)
    // one phase of K-tile u (PAR = u & 1 as a constant, half = which group computes)
    auto phase = [&](auto par_c, auto half_c, int u) {
        constexpr int PAR = decltype(par_c)::value, half = decltype(half_c)::value;
        if (grp == half) {
            LCREC_STAMP(0);
            const float *w_base = Ws0 + PAR * BN * LDT + w_off;
            // fragment registers: one 16-VGPR buffer per 8-wide k group; a buffer is refilled as soon as the MFMAs that
            // read it have been issued, so the refill is in flight under the 16 MFMAs that follow.  (Measured and
            // rejected: dealing a refill out one read per MFMA, and staggering it between the four waves -- both
            // 5-7 % slower than one burst.)
            f32x4 af[2][2], wf[2][2];
            auto frags = [&](int buf, int g) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[buf][i] = *reinterpret_cast<const f32x4 *>(a_base + i * 32 * LDT + g * 8);
                    wf[buf][i] = *reinterpret_cast<const f32x4 *>(w_base + i * 32 * LDT + g * 8);
                }
            };
            auto reads_then_mfmas = [&]() {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            };
            frags(0, 0);
            frags(1, 1);
            if (half == 0) {
                if (u + 1 < nk) { load_a(u + 1); load_w(u + 1); }      // A0[u+1], lower W[u+1]
            } else {
                if (u + 1 < nk) load_a(u + 1);                          // A1[u+1]
                if (u + 2 < nk) load_w(u + 2);                          // upper W[u+2]
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(af[0], wf[0]);                                   // k group 0
            __builtin_amdgcn_sched_barrier(0);
            frags(0, 2);
            mfma_group(af[1], wf[1]);                                   // k group 1
            reads_then_mfmas();
            __builtin_amdgcn_sched_barrier(0);
            frags(1, 3);
            mfma_group(af[0], wf[0]);                                   // k group 2
            reads_then_mfmas();
            __builtin_amdgcn_sched_barrier(0);                          // the barrier stays where it is written
            LCREC_STAMP(1);
            __syncthreads();                                            // after this wave's LAST LDS read of the tile
            LCREC_STAMP(2);
            // From here on this wave's MFMAs outrank its partner's: the partner has just become the computing wave
            // of the next phase and its first MFMAs are ready too, but every MFMA of theirs that overtakes one of
            // these delays this wave's staging role, and the partner would then wait for that staging at the next
            // barrier (stamps: 330 cycles per phase).
            __builtin_amdgcn_s_setprio(3);
            mfma_group(af[1], wf[1]);                                   // k group 3
            __builtin_amdgcn_s_setprio(0);
            LCREC_STAMP(3);
        } else {
            LCREC_STAMP(0);
            // half == 0: group 1 writes A1[u] and the upper half of W[u+1]; half == 1: group 0 writes A0[u+1] and
            // the lower half of W[u+1] (registers of a tile past the end hold stale data nobody reads)
            store_a();
            store_w(PAR ^ 1);
            lds_drain();
            LCREC_STAMP(1);
            __syncthreads();
            LCREC_STAMP(2);
        }
    };

    for (int u = 0; u < nk; u += 2) {
        phase(IntC<0>{}, IntC<0>{}, u);
        phase(IntC<0>{}, IntC<1>{}, u);
        if (u + 1 < nk) {
            phase(IntC<1>{}, IntC<0>{}, u + 1);
            phase(IntC<1>{}, IntC<1>{}, u + 1);
        }
    }

    LCREC_MARK(2);
    // epilogue: each wave transposes through 32 x 36 floats of its OWN group's activation buffer
    float *stg = my_a + w4 * 32 * LDK;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            store_tile_32x32(acc[i][j], stg, lane, C, m0 + wm * 64 + i * 32, M, n0 + wn * 64 + j * 32, N, bias, bn_scale,
                             bn_shift, relu);
    LCREC_MARK(3);
}

// ------------------------------------------------------------------------------------------
// Ping-pong kernel, persistent form (K % 64 == 0, K >= 384, N % 128 == 0): one workgroup per CU walks a list of
// output tiles and never drains its pipeline between them.
//   * The K-tile stream is flat across tiles: the last two K-tiles of a tile prefetch the first two of the next
//     one (other buffer descriptors), so only the workgroup's first tile has a prologue.
//   * A finished accumulator set moves to `prev` THROUGH the epilogue arithmetic (prev = relu(bn(acc + bias)) is the
//     copy) in the wave's first COMPUTE phase of the next tile, where its own VALU work issues in the shadow of its
//     own MFMAs; one 16 x 32 piece per STAGING phase of K-tiles 1..8 then goes through a wave-private LDS patch to
//     16-byte buffer stores (LDS and vector-memory ports only; rows past M are dropped by the descriptor's range
//     check).  The 128 KB per CU that linear_fwd_pp2_kernel stores in one burst at the end of every tile -- all 256
//     CUs at the same instant -- drains under eight K-tiles of the next tile.
//   * The K loop is split into HEAD (K-tiles 0..9, which carry the previous tile's epilogue), STEADY (exactly the
//     phase code of linear_fwd_pp2_kernel: no extra branch, no extra scalar or vector instruction) and TAIL (the
//     last two K-tiles, whose prefetch targets are static).  An earlier persistent form with run-time decisions
//     inside every phase lost 5-8 % in the K loop: a uniform boolean that LLVM materialises through a VGPR costs two
//     VALU instructions, and in the staging role those wait for gaps in the partner's MFMA stream.
//   * Measured with per-wave stamps: a wave in the staging role is issued roughly one instruction per MFMA of its
//     partner (~64 per phase, LDS / vector-memory / scalar alike), and in the steady state it reaches the barrier only
//     ~100 cycles before the computing waves.  So the epilogue is cut into eight pieces and each piece between the
//     roles: patch writes in the compute role, two reads + two stores in the staging role.
// Measured on MI355X (131 072 rows, TFLOP/s, this / linear_fwd_pp2_kernel): 768->2048 143 / 140, 2048->1024 151 / 150,
// 1024->512 146 / 144, 512->256 138 / 134; C3 end to end 16.49 M vs 16.30 M items/s.
// Same arithmetic as every other kernel here: one fma chain per output over k ascending, epilogue after it.
// ------------------------------------------------------------------------------------------
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <bool HAS_BN>
__global__ __launch_bounds__(512) void linear_fwd_pp3_kernel(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    const float *__restrict__ bn_scale, const float *__restrict__ bn_shift, float *__restrict__ C,
    int64_t M, int N, int K, int relu, int bn_blocks, int bm_blocks, int total_virtual, int xcd_order)
{
    constexpr int GM = 128, BN = 128, LDT = LDK;
    // parts of a tile's K loop: HEAD0 = K-tile 0, HEAD_S0 + p = K-tile 1 + p carrying epilogue piece p (8 pieces: the
    // two 16-row halves of the four 32 x 32 sub-tiles), STEADY, TAIL0 / TAIL1 = the last two K-tiles
    // (HEAD_S0 + 8 = K-tile 9: group 1, whose staging phase precedes its compute phase in a K-tile, sends its last piece)
    constexpr int STEADY = 0, HEAD0 = 1, HEAD_S0 = 2, HEAD_S8 = 10, TAIL0 = 11, TAIL1 = 12;
    extern __shared__ __attribute__((aligned(16))) float pp3_lds[];
    float *As0 = pp3_lds;                        // [2][GM * LDT]
    float *Ws0 = As0 + 2 * GM * LDT;             // [2][BN * LDT]
    float *Ep0 = Ws0 + 2 * BN * LDT;             // [8][32 * LDT] wave-private transpose patches

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    const int gt = tid & 255;
    const int nk = K / BK;
    float *my_a = As0 + grp * GM * LDT;
    float *patch = Ep0 + wave * 32 * LDT;

    // tile walker: (m0 = first row of this GROUP's 128-row half, n0 = first column, valid row counts) as plain scalars
    auto tile_of = [&](int t, int64_t &o_m0, int &o_n0, int &o_arows, int &o_wrows) __attribute__((always_inline)) -> bool {
        int64_t bm;
        int bn;
        if (xcd_order) {
            const int xcd = t & 7, j = t >> 3;
            const int panels = (bm_blocks - xcd + 7) >> 3;
            if (j >= panels * bn_blocks) return false;
            bm = (int64_t)(j / bn_blocks) * 8 + xcd;
            bn = j % bn_blocks;
        } else {
            if (t >= bm_blocks * bn_blocks) return false;
            bm = t / bn_blocks;
            bn = t % bn_blocks;
        }
        o_m0 = bm * (2 * GM) + grp * GM;
        o_n0 = bn * BN;
        const int64_t ar = M - o_m0, wr = (int64_t)N - (o_n0 + grp * 64);
        o_arows = __builtin_amdgcn_readfirstlane((int)(ar < 0 ? 0 : (ar > GM ? GM : ar)));
        o_wrows = __builtin_amdgcn_readfirstlane((int)(wr < 0 ? 0 : (wr > 64 ? 64 : wr)));
        return true;
    };
    auto next_tile = [&](int t, int64_t &o_m0, int &o_n0, int &o_arows, int &o_wrows) __attribute__((always_inline)) -> int {
        for (t += gridDim.x; t < total_virtual; t += gridDim.x)
            if (tile_of(t, o_m0, o_n0, o_arows, o_wrows)) return t;
        return total_virtual;
    };

    int64_t cur_m0 = 0, nxt_m0 = 0, prv_m0 = 0;
    int cur_n0 = 0, nxt_n0 = 0, prv_n0 = 0, cur_ar = 0, nxt_ar = 0, prv_ar = 0, cur_wr = 0, nxt_wr = 0;
    int t = blockIdx.x;
    if (!tile_of(t, cur_m0, cur_n0, cur_ar, cur_wr)) t = next_tile(t, cur_m0, cur_n0, cur_ar, cur_wr);
    if (t >= total_virtual) return;
    int has_next = 0, have_prev = 0;

    const int t_g = ((gt >> 2) * K + (gt & 3) * 8) * 4;
    const int pass_g = 64 * K * 4;
    const uint32_t t_s = (uint32_t)(((gt >> 2) * LDT + (gt & 3) * 8) * 4);
    const uint32_t a_s0 = lds_addr(my_a) + t_s, a_s1 = a_s0 + 64 * LDT * 4;
    const uint32_t w_s[2] = {lds_addr(Ws0 + grp * 64 * LDT) + t_s, lds_addr(Ws0 + BN * LDT + grp * 64 * LDT) + t_s};

    // operand descriptors of the current and the next tile, rebuilt once per tile
    __amdgpu_buffer_rsrc_t cur_a = tile_rsrc(A, cur_m0, M, GM, K), cur_w = tile_rsrc(W, cur_n0 + grp * 64, N, 64, K);
    __amdgpu_buffer_rsrc_t nxt_a = cur_a, nxt_w = cur_w;

    f32x4 ra[2][2], rw[2];
    auto load_a = [&](__amdgpu_buffer_rsrc_t r, int kt) __attribute__((always_inline)) {
        const int so = kt * (BK * 4);
        ra[0][0] = buffer_load_f32x4(r, t_g, so);
        ra[0][1] = buffer_load_f32x4(r, t_g + 16, so);
        ra[1][0] = buffer_load_f32x4(r, t_g, so + pass_g);
        ra[1][1] = buffer_load_f32x4(r, t_g + 16, so + pass_g);
    };
    auto load_w = [&](__amdgpu_buffer_rsrc_t r, int kt) __attribute__((always_inline)) {
        const int so = kt * (BK * 4);
        rw[0] = buffer_load_f32x4(r, t_g, so);
        rw[1] = buffer_load_f32x4(r, t_g + 16, so);
    };
    auto store_a = [&]() __attribute__((always_inline)) {
        lds_store_deint8(a_s0, ra[0][0], ra[0][1]);
        lds_store_deint8(a_s1, ra[1][0], ra[1][1]);
    };
    auto store_w = [&](int buf) __attribute__((always_inline)) { lds_store_deint8(w_s[buf], rw[0], rw[1]); };
    auto lds_drain = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    // Workgroup barrier for LDS traffic only.  __syncthreads() is a release/acquire fence over ALL memory: with the
    // previous tile's buffer stores in flight it waits for their completion (vmcnt(0), ~3 000 cycles per phase) although
    // nothing in this kernel ever reads what another wave stored to global memory.
    auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // per-column epilogue constants of a tile: this lane's two columns (j = 0, 1)
    struct EpiConst { float b[2], sc[2], sh[2]; };
    auto load_epi = [&](EpiConst &e, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            e.b[j] = bias ? bias[col] : 0.f;
            e.sc[j] = HAS_BN ? bn_scale[col] : 1.f;
            e.sh[j] = HAS_BN ? bn_shift[col] : 0.f;
        }
    };
    auto epi_math = [&](f32x16 (&dst)[2][2], const f32x16 (&src)[2][2], const EpiConst &e) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = src[i][j][r] + e.b[j];
                    if (HAS_BN) v = __builtin_fmaf(v, e.sc[j], e.sh[j]);
                    if (relu) v = (v > 0.f) ? v : 0.f;
                    dst[i][j][r] = v;
                }
    };
    // one 32 x 32 sub-tile of a post-processed accumulator set -> C, through this wave's LDS patch
    // The epilogue pieces run in the STAGING role, where every VALU instruction waits for a gap in the partner's MFMA
    // stream: all addresses are computed once, here, and pinned in registers; the patch writes are ds_write_b32 with
    // immediate offsets (written as asm so that they are not re-paired into ds_write2 with fresh base registers).
    uint32_t patch_w = lds_addr(patch) + (uint32_t)((4 * (lane >> 5) * LDT + (lane & 31)) * 4);
    uint32_t patch_r = lds_addr(patch) + (uint32_t)(((lane >> 3) * LDT + (lane & 7) * 4) * 4);
    int c_voff = ((lane >> 3) * N + (lane & 7) * 4) * 4;
    asm volatile("" : "+v"(patch_w), "+v"(patch_r), "+v"(c_voff));
    // One piece = 16 rows x 32 columns (half hf of sub-tile (i, j)).  A wave in the staging role is issued roughly ONE
    // instruction per MFMA of its partner -- LDS, vector-memory and scalar ones too, measured -- and is within ~100
    // cycles of being the barrier's last arrival even in the steady state.  So a piece is split between the roles: its 8
    // patch writes go out in the COMPUTE role (epi_patch: LDS writes issue freely in the shadow of the wave's own MFMAs),
    // and only 2 patch reads + 2 stores remain for the wave's next STAGING role (epi_send).
    auto epi_patch = [&](const f32x16 &v, auto hf_c) __attribute__((always_inline)) {
        constexpr int hf = decltype(hf_c)::value;
        const uint32_t pw = patch_w;          // (named here: an implicit capture inside a dependent asm operand is not seen)
#define LCREC_PATCH_W(R)                                                                                         \
        asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(pw), "v"(v[8 * hf + (R)]),                            \
                     "n"((((R) & 3) + 8 * ((8 * hf + (R)) >> 2)) * LDT * 4) : "memory")
        LCREC_PATCH_W(0); LCREC_PATCH_W(1); LCREC_PATCH_W(2); LCREC_PATCH_W(3);
        LCREC_PATCH_W(4); LCREC_PATCH_W(5); LCREC_PATCH_W(6); LCREC_PATCH_W(7);
#undef LCREC_PATCH_W
    };
    auto epi_send = [&](auto i_c, auto j_c, auto hf_c, int64_t t_m0, int t_n0, int rows) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value, j = decltype(j_c)::value, hf = decltype(hf_c)::value;
        const int extent = rows > 0 ? ((rows - 1) * N + BN) * 4 : 0;
        const __amdgpu_buffer_rsrc_t c_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(C + t_m0 * (int64_t)N + t_n0, 0, extent, 0x00020000);
        f32x4 q[2];
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     "ds_read_b128 %0, %2 offset:%3\n\t"
                     "ds_read_b128 %1, %2 offset:%4\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(q[0]), "=&v"(q[1])
                     : "v"(patch_r), "n"((2 * hf) * 8 * LDT * 4), "n"((2 * hf + 1) * 8 * LDT * 4)
                     : "memory");
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int soff = ((wm * 64 + i * 32 + 8 * (2 * hf + p)) * N + wn * 64 + j * 32) * 4;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, q[p]), c_rsrc, c_voff, soff, 0);
            // A 16-byte buffer store reads its data VGPRs a few cycles AFTER issue.  The compiler inserts the required
            // wait state before an instruction that overwrites them -- except when the store has an SGPR offset (LLVM
            // assumes that form is safe; on gfx950 it is not: the next VALU write, e.g. the v_cndmask of a branch
            // condition, showed up as the value 1 in the first element of the stored vector).  So wait here.
            asm volatile("s_nop 3" ::: "memory");
        }
    };
    auto epi_store = [&](const f32x16 &v, auto i_c, auto j_c, auto hf_c, int64_t t_m0, int t_n0, int rows) __attribute__((always_inline)) {
        epi_patch(v, hf_c);
        epi_send(i_c, j_c, hf_c, t_m0, t_n0, rows);
    };

    // ---- prologue of the workgroup's first tile: W[0] (both halves) and A0[0] into LDS; group 1 keeps (A1[0], upper W[1])
    load_w(cur_w, 0);
    store_w(0);
    load_a(cur_a, 0);
    if (grp == 0) {
        store_a();
    } else {
        load_w(cur_w, 1);
    }
    lds_drain();
    lds_barrier();

    const float *a_base = my_a + (wm * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int w_off = (wn * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;

    f32x16 acc[2][2], prev[2][2];
    EpiConst ec;
    // one phase of K-tile u (PAR = u & 1 and half = which group computes, as constants; MODE = which part of the tile)
    auto phase = [&](auto par_c, auto half_c, auto mode_c, int u) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value, half = decltype(half_c)::value, MODE = decltype(mode_c)::value;
        if (grp == half) {
            LCREC_STAMP(0);
            const float *w_base = Ws0 + PAR * BN * LDT + w_off;
            f32x4 af[2][2], wf[2][2];
            auto frags = [&](int buf, int g) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[buf][i] = *reinterpret_cast<const f32x4 *>(a_base + i * 32 * LDT + g * 8);
                    wf[buf][i] = *reinterpret_cast<const f32x4 *>(w_base + i * 32 * LDT + g * 8);
                }
            };
            auto mfma_group = [&](const f32x4 (&a2)[2], const f32x4 (&w2)[2]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[i][q], w2[j][q], acc[i][j], 0, 0, 0);
            };
            auto reads_then_mfmas = [&]() {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            };
            if constexpr (MODE == HEAD0) {
                // first K-tile of a tile: the previous tile's post-processing (with ITS constants), then this tile's
                // constants and a zeroed accumulator set
                if (have_prev) epi_math(prev, acc, ec);
                load_epi(ec, cur_n0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
            frags(0, 0);
            frags(1, 1);
            __builtin_amdgcn_sched_barrier(0);                          // fragment reads first: the MFMAs wait for them
            if constexpr (MODE == TAIL0) {                              // u = nk - 2
                if (half == 0) {
                    load_a(cur_a, u + 1);
                    load_w(cur_w, u + 1);
                } else {
                    load_a(cur_a, u + 1);
                    if (has_next) load_w(nxt_w, 0);
                }
            } else if constexpr (MODE == TAIL1) {                       // u = nk - 1: the next tile's first K-tiles
                if (has_next) {
                    load_a(nxt_a, 0);
                    load_w(nxt_w, half == 0 ? 0 : 1);
                }
            } else {
                load_a(cur_a, u + 1);                                   // half 0: A0[u+1], lower W[u+1]
                load_w(cur_w, half == 0 ? u + 1 : u + 2);               // half 1: A1[u+1], upper W[u+2]
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(af[0], wf[0]);                                   // k group 0
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE >= HEAD_S0 && MODE < HEAD_S8) {          // this K-tile's epilogue piece into the patch
                if (have_prev) {
                    constexpr int s = (MODE - HEAD_S0) >> 1, hf = (MODE - HEAD_S0) & 1;
                    epi_patch(prev[s >> 1][s & 1], IntC<hf>{});
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            frags(0, 2);
            mfma_group(af[1], wf[1]);                                   // k group 1
            reads_then_mfmas();
            __builtin_amdgcn_sched_barrier(0);
            frags(1, 3);
            mfma_group(af[0], wf[0]);                                   // k group 2
            reads_then_mfmas();
            __builtin_amdgcn_sched_barrier(0);
            LCREC_STAMP(1);
            lds_barrier();                                            // after this wave's LAST LDS read of the tile
            LCREC_STAMP(2);
            __builtin_amdgcn_s_setprio(3);
            mfma_group(af[1], wf[1]);                                   // k group 3
            __builtin_amdgcn_s_setprio(0);
            LCREC_STAMP(3);
        } else {
            LCREC_STAMP(0);
            store_a();
            store_w(PAR ^ 1);
            lds_drain();
            LCREC_STAMP(1);
            // K-tiles 1..4 of a tile: one sub-tile of the previous tile's output per staging phase
            // the piece this wave wrote to its patch in its last compute phase: this K-tile's for group 0 (half == 1
            // phases), the previous K-tile's for group 1 (whose staging phase comes first in a K-tile)
            constexpr int PIECE = MODE - HEAD_S0 - (half == 0 ? 1 : 0);
            if constexpr (MODE >= HEAD_S0 && MODE <= HEAD_S8 && PIECE >= 0 && PIECE < 8) {
                if (have_prev) {
                    constexpr int s = PIECE >> 1, hf = PIECE & 1;
                    epi_send(IntC<(s >> 1)>{}, IntC<(s & 1)>{}, IntC<hf>{}, prv_m0, prv_n0, prv_ar);
                }
            }
            LCREC_STAMP(3);
            lds_barrier();
            LCREC_STAMP(2);
        }
    };
    auto pair = [&](auto par_c, auto mode_c, int u) __attribute__((always_inline)) {
        phase(par_c, IntC<0>{}, mode_c, u);
        phase(par_c, IntC<1>{}, mode_c, u);
    };
    auto finish = [&]() __attribute__((always_inline)) {     // the workgroup's last tile: nothing left to hide behind
        epi_math(acc, acc, ec);
        epi_store(acc[0][0], IntC<0>{}, IntC<0>{}, IntC<0>{}, cur_m0, cur_n0, cur_ar);
        epi_store(acc[0][0], IntC<0>{}, IntC<0>{}, IntC<1>{}, cur_m0, cur_n0, cur_ar);
        epi_store(acc[0][1], IntC<0>{}, IntC<1>{}, IntC<0>{}, cur_m0, cur_n0, cur_ar);
        epi_store(acc[0][1], IntC<0>{}, IntC<1>{}, IntC<1>{}, cur_m0, cur_n0, cur_ar);
        epi_store(acc[1][0], IntC<1>{}, IntC<0>{}, IntC<0>{}, cur_m0, cur_n0, cur_ar);
        epi_store(acc[1][0], IntC<1>{}, IntC<0>{}, IntC<1>{}, cur_m0, cur_n0, cur_ar);
        epi_store(acc[1][1], IntC<1>{}, IntC<1>{}, IntC<0>{}, cur_m0, cur_n0, cur_ar);
        epi_store(acc[1][1], IntC<1>{}, IntC<1>{}, IntC<1>{}, cur_m0, cur_n0, cur_ar);
    };

    for (;;) {
        const int tn = next_tile(t, nxt_m0, nxt_n0, nxt_ar, nxt_wr);
        has_next = tn < total_virtual ? 1 : 0;
        pair(IntC<0>{}, IntC<HEAD0>{}, 0);
        pair(IntC<1>{}, IntC<HEAD_S0 + 0>{}, 1);
        pair(IntC<0>{}, IntC<HEAD_S0 + 1>{}, 2);
        pair(IntC<1>{}, IntC<HEAD_S0 + 2>{}, 3);
        pair(IntC<0>{}, IntC<HEAD_S0 + 3>{}, 4);
        pair(IntC<1>{}, IntC<HEAD_S0 + 4>{}, 5);
        pair(IntC<0>{}, IntC<HEAD_S0 + 5>{}, 6);
        pair(IntC<1>{}, IntC<HEAD_S0 + 6>{}, 7);
        pair(IntC<0>{}, IntC<HEAD_S0 + 7>{}, 8);
        pair(IntC<1>{}, IntC<HEAD_S8>{}, 9);
        for (int u = 10; u < nk - 2; u += 2) {                  // the loop of linear_fwd_pp2_kernel, instruction for instruction
            pair(IntC<0>{}, IntC<STEADY>{}, u);
            pair(IntC<1>{}, IntC<STEADY>{}, u + 1);
        }
        nxt_a = tile_rsrc(A, nxt_m0, M, GM, K);                // built here, not at the tile's start: 8 SGPRs less in the loop
        nxt_w = tile_rsrc(W, nxt_n0 + grp * 64, N, 64, K);
        pair(IntC<0>{}, IntC<TAIL0>{}, nk - 2);
        pair(IntC<1>{}, IntC<TAIL1>{}, nk - 1);
        if (!has_next) { finish(); break; }
        prv_m0 = cur_m0; prv_n0 = cur_n0; prv_ar = cur_ar;
        cur_m0 = nxt_m0; cur_n0 = nxt_n0; cur_ar = nxt_ar; cur_wr = nxt_wr; cur_a = nxt_a; cur_w = nxt_w;
        t = tn;
        have_prev = 1;
    }
}

// ------------------------------------------------------------------------------------------
// Small-tile kernel for launches that 64 x 64 tiles cannot fill the chip with (a training step's layers, the row tail of a
// Games-sized launch): 32 x 64 tiles on v_mfma_f32_16x16x4_f32.
//
// Why another instruction.  At batch 1024 a 1024-column layer is 4096 outputs per CU: with 32 x 32 x 2 tiles exactly one
// accumulator per SIMD, one wave per SIMD, and nothing to run under that wave's LDS round trips and barriers (the 64 x 64
// kernel's MFMA pipe is busy 0.69-0.79 of such a launch, profiles/r03_pmc_mfma.txt).  v_mfma_f32_16x16x4_f32 has the same peak
// (2048 flop / 32 cycles) on a quarter of the outputs, so the same work makes TWO waves per SIMD (two workgroups per CU), each
// with two independent 16 x 16 accumulators; and its dependent chain is 40 cycles per 4 k against 64 per 2 k, which is what
// bounds a launch of few tiles (a row tail's serial K loop).  It keeps the arithmetic contract: its four k-products are ONE fp32
// fma chain, k ascending, onto the accumulator -- measured bit for bit against fmaf on gfx950 (tools/mfma16_probe.hip) -- so a
// 16 x 16 tile fed k = 4s .. 4s+3 for s ascending reproduces oracle/lcrec_oracle.c like the big tiles do.
//
// A workgroup of four waves covers 32 x 64: wave (wm, wn) rows 16 wm .., columns 32 wn .. (two 16 x 16 blocks).  Same K-tile
// pipeline as the 64 x 64 kernel: two LDS buffers, operand fragments of K-tile kt+1 read into a second register set under
// K-tile kt's 16 MFMAs, K-tile kt+2 stored into the buffer kt was read from, a ring of register sets for the global prefetch,
// every LDS / memory instruction in an MFMA gap of its own, one barrier per K-tile.
// LDS image of a row-major operand tile: rows of 32 k padded to 36 floats, k at position (k % 4) * 8 + k / 4, so that lane
// (r = lane % 16, q = lane / 16) finds its operands of all eight MFMAs of the K-tile -- k = q, 4 + q, ..., 28 + q -- in 8
// consecutive floats: two ds_read_b128 (conflict-free: 36 = 4 mod 32).  A k-major W (the dX product reads W [out][in] as
// [K][N]) keeps a k-major image [32 k][72]: 16-byte stores as loaded, 4-byte fragment reads.
constexpr int S16_BM = 32, S16_BN = 64, S16_LDN = 72, S16_RING = 4;

template <bool TB>
__global__ __launch_bounds__(256) void linear_s16_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                         const float *__restrict__ bias, const float *__restrict__ bn_scale,
                                                         const float *__restrict__ bn_shift, float *__restrict__ C, int64_t M, int N,
                                                         int K, int relu, int bn_blocks)
{
    constexpr int A_FLOATS = S16_BM * LDK, W_FLOATS = TB ? BK * S16_LDN : S16_BN * LDK, BUF = A_FLOATS + W_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, q = lane >> 4;
    const int64_t bm = blockIdx.x / bn_blocks;
    const int bn = blockIdx.x % bn_blocks;
    const int64_t m0 = bm * S16_BM;
    const int n0 = bn * S16_BN;
    const int nk = K / BK;                                    // K % 32 == 0 (the launcher's condition)

    const __amdgpu_buffer_rsrc_t a_rsrc = tile_rsrc(A, m0, M, S16_BM, K);
    const __amdgpu_buffer_rsrc_t w_rsrc = TB ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W), 0, (int)((int64_t)K * N * 4), 0x00020000)
                                             : tile_rsrc(W, n0, N, S16_BN, K);
    struct Regs { f32x4 a, w[2]; };
    struct Frag { f32x4 a[2], b[2][2]; };                     // [k half][...]: MFMAs s = 0..3 use half 0, s = 4..7 half 1
    constexpr unsigned SOFF_OUT = 0x7fffff00u;
    const int arow = tid >> 3, akq = tid & 7;                 // A: 32 rows x 8 float4; row-major W: rows arow, arow + 32
    const int wkk = tid >> 4, wn4 = tid & 15;                 // k-major W: 32 k x 16 float4, k = wkk, wkk + 16
    const bool w_in = !TB || n0 + wn4 * 4 < N;
    const int a_vo = (arow * K + akq * 4) * 4;
    const int w_vo0 = TB ? (w_in ? (int)(((int64_t)wkk * N + n0 + wn4 * 4) * 4) : 0x7fffff00) : a_vo;
    const int w_vo1 = TB ? (w_in ? (int)(((int64_t)(wkk + 16) * N + n0 + wn4 * 4) * 4) : 0x7fffff00) : a_vo + 32 * K * 4;
    // one of the three 16-byte loads of a K-tile (piece 0: A, 1 / 2: W); a K-tile past the end is "loaded" from beyond the
    // descriptor's extent: zeros, no memory access, no branch for the waitcnt bookkeeping to merge
    auto load_piece = [&](Regs &g, int kt, bool live, auto piece_c) {
        constexpr int piece = decltype(piece_c)::value;
        const unsigned so = live ? (TB && piece > 0 ? (unsigned)((int64_t)kt * BK * N * 4) : (unsigned)(kt * (BK * 4))) : SOFF_OUT;
        if constexpr (piece == 0) g.a = buffer_load_f32x4(a_rsrc, a_vo, (int)so);
        else if constexpr (piece == 1) g.w[0] = buffer_load_f32x4(w_rsrc, w_vo0, (int)so);
        else g.w[1] = buffer_load_f32x4(w_rsrc, w_vo1, (int)so);
    };
    const uint32_t a_st = (uint32_t)((arow * LDK + akq) * 4);                       // k = 4 akq + i -> position i * 8 + akq
    const uint32_t wk_st = (uint32_t)((A_FLOATS + wkk * S16_LDN + wn4 * 4) * 4);
    // six LDS store instructions per K-tile (row-major: ds_write2_b32 pairs; k-major W: ds_write_b128): piece 0, 1 = A
    auto store_piece = [&](const Regs &g, int buf, auto piece_c) {
        constexpr int piece = decltype(piece_c)::value;
        const uint32_t base = lds_addr(smem) + (uint32_t)(buf * BUF * 4);
        if constexpr (piece < 2) {
            const uint32_t ad = base + a_st + piece * 64;
            asm volatile("ds_write2_b32 %0, %1, %2 offset1:8" ::"v"(ad), "v"(g.a[2 * piece]), "v"(g.a[2 * piece + 1]) : "memory");
        } else if constexpr (TB) {
            const uint32_t ad = base + wk_st;                 // (named here: clang does not capture a variable first used inside the nested constexpr-if)
            if constexpr (piece < 4) asm volatile("ds_write_b128 %0, %1" ::"v"(ad + (uint32_t)((piece - 2) * 16 * S16_LDN * 4)), "v"(g.w[piece - 2]) : "memory");
        } else {
            constexpr int j = (piece - 2) >> 1, h = (piece - 2) & 1;
            const uint32_t ad = base + (uint32_t)(A_FLOATS * 4) + a_st + (uint32_t)(j * 32 * LDK * 4) + h * 64;
            asm volatile("ds_write2_b32 %0, %1, %2 offset1:8" ::"v"(ad), "v"(g.w[j][2 * h]), "v"(g.w[j][2 * h + 1]) : "memory");
        }
    };
    // six (row-major W) fragment reads of 16 bytes per K-tile: slot 0, 1 = A halves; 2 .. 5 = W block cb = (slot - 2) / 2, half
    const float *a_fr = smem + (16 * wm + r16) * LDK + q * 8;
    const float *w_fr = smem + A_FLOATS + (32 * wn + r16) * LDK + q * 8;
    const float *wk_fr = smem + A_FLOATS + q * S16_LDN + 32 * wn + r16;
    auto frag_read = [&](Frag &f, int buf, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        if constexpr (slot < 2) {
            f.a[slot] = *reinterpret_cast<const f32x4 *>(a_fr + buf * BUF + slot * 4);
        } else {
            constexpr int cb = (slot - 2) >> 1, h = (slot - 2) & 1;
            if constexpr (TB) {
#pragma unroll
                for (int s = 0; s < 4; ++s) f.b[h][cb][s] = wk_fr[buf * BUF + (4 * s + 16 * h) * S16_LDN + 16 * cb];
            } else {
                f.b[h][cb] = *reinterpret_cast<const f32x4 *>(w_fr + buf * BUF + cb * 16 * LDK + h * 4);
            }
        }
    };

    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    Frag fa, fb;
    Regs ring[S16_RING];
    if (nk > 0) {
        static_for<S16_RING>([&](auto j_c) {
            constexpr int j = decltype(j_c)::value;
            static_for<3>([&](auto p_c) { load_piece(ring[j], j, j < nk, p_c); });
        });
        static_for<6>([&](auto p_c) { store_piece(ring[0], 0, p_c); });
        static_for<6>([&](auto p_c) { store_piece(ring[1], 1, p_c); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<3>([&](auto p_c) { load_piece(ring[0], S16_RING, S16_RING < nk, p_c); });
        __syncthreads();
        static_for<6>([&](auto s_c) { frag_read(fa, 0, s_c); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                      // iteration 0 stores K-tile 2 into buffer 0
    }
    // iteration kt: MFMAs of kt from `fc`; fragments of kt+1 from buffer (kt+1)&1 into `fn`; K-tile kt+2 (ring set (kt+2) % RING)
    // into buffer kt&1; K-tile kt+RING+1 loaded into set (kt+1) % RING
    auto k_tile = [&](int kt, const Frag &fc, Frag &fn, Regs &r_free, const Regs &r_next, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value;
        const bool live = kt + S16_RING + 1 < nk;
        static_for<16>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value, cb = s & 1, ks = s >> 1, h = ks >> 2, sq = ks & 3;
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fc.a[h][sq], fc.b[h][cb][sq], acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s < 6) frag_read(fn, cur ^ 1, IntC<s>{});
            if constexpr (s >= 6 && s < 9) load_piece(r_free, kt + S16_RING + 1, live, IntC<s - 6>{});
            if constexpr (s >= 9 && s < 15) store_piece(r_next, cur, IntC<s - 9>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += S16_RING) {
        static_for<S16_RING>([&](auto u_c) {
            constexpr int u = decltype(u_c)::value;
            if (u == 0 || kt + u < nk) {
                if constexpr (u % 2 == 0) k_tile(kt + u, fa, fb, ring[(u + 1) % S16_RING], ring[(u + 2) % S16_RING], IntC<0>{});
                else k_tile(kt + u, fb, fa, ring[(u + 1) % S16_RING], ring[(u + 2) % S16_RING], IntC<1>{});
            }
        });
    }

    // epilogue: lane holds rows 4 q + r (r = 0..3) of column r16 of each 16 x 16 block
    const bool has_bn = bn_scale != nullptr;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = n0 + 32 * wn + 16 * cb + r16;
        if (col >= N) continue;
        const float bj = bias ? bias[col] : 0.f;
        const float sc = has_bn ? bn_scale[col] : 1.f, sh = has_bn ? bn_shift[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = m0 + 16 * wm + 4 * q + r;
            if (row >= M) continue;
            float t = acc[cb][r] + bj;
            if (has_bn) t = __builtin_fmaf(t, sc, sh);
            if (relu) t = t > 0.f ? t : 0.f;
            C[row * (int64_t)N + col] = t;
        }
    }
}

static int launch_s16(const float *A, const float *W, const float *b, const float *sc, const float *sh, int relu, float *C, int64_t M,
                      int N, int K, bool kmajor_w, hipStream_t stream)
{
    const int64_t bm_blocks = (M + S16_BM - 1) / S16_BM;
    const int bn_blocks = (N + S16_BN - 1) / S16_BN;
    const int64_t grid = bm_blocks * bn_blocks;
    if (grid > 0x7fffffffLL) return fail(LCREC_EINVAL, "linear (32 x 64 tiles): grid too large");
    TraceScope trace(K_LINEAR_32x64, stream);
    if (kmajor_w) hipLaunchKernelGGL(linear_s16_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, A, W, b, sc, sh, C, M, N, K, relu, bn_blocks);
    else hipLaunchKernelGGL(linear_s16_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, A, W, b, sc, sh, C, M, N, K, relu, bn_blocks);
    return check_launch("linear_s16_kernel");
}

// Which launches take the 32 x 64 tiles: K a multiple of 32, and 64 x 64 tiles would leave the chip under-filled -- at most
// LCREC_GEMM_S16_TILES (default below) of them.  LCREC_GEMM_S16=0 never, =1 whenever the shape allows (tuning).
static bool use_s16_tiles(int64_t M, int N, int K)
{
    static const int mode = [] { const char *e = getenv("LCREC_GEMM_S16"); return e ? atoi(e) : -1; }();
    static const int limit = [] { const char *e = getenv("LCREC_GEMM_S16_TILES"); return e ? atoi(e) : 128; }();
    if (mode == 0 || K % BK != 0 || K < BK || N % 4 != 0 || (int64_t)K * 4 * (M + 64) >= (1ll << 31) || (int64_t)K * N * 4 >= (1ll << 31)) return false;
    if (mode == 1) return true;
    return ((M + 63) / 64) * ((N + 63) / 64) <= limit;
}

static int launch_pp2(dim3 grid, hipStream_t stream, const float *x, const float *W, const float *b, const float *sc,
                      const float *sh, float *y, int64_t n, int out_dim, int in_dim, int relu, int bn_blocks, int bm_blocks,
                      int xcd_order)
{
    constexpr size_t lds = (size_t)(2 * 128 + 2 * 128) * (BK + 4) * sizeof(float);     // 73 728 B
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(linear_fwd_pp2_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return fail(LCREC_EHIP, "linear_forward: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(attr));
    hipLaunchKernelGGL(linear_fwd_pp2_kernel, grid, dim3(512), lds, stream, x, W, b, sc, sh, y, n, out_dim, in_dim, relu,
                       bn_blocks, bm_blocks, xcd_order);
    return LCREC_OK;
}

static int launch_pp3(int total_virtual, hipStream_t stream, const float *x, const float *W, const float *b, const float *sc,
                      const float *sh, float *y, int64_t n, int out_dim, int in_dim, int relu, int bn_blocks, int bm_blocks,
                      int xcd_order)
{
    constexpr size_t lds = (size_t)(2 * 128 + 2 * 128 + 8 * 32) * LDK * sizeof(float);     // 110 592 B
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 256;
        return v > 8 ? v / 8 * 8 : 8;                // a multiple of 8 keeps a workgroup's tiles on its own XCD
    }();
    static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void *>(linear_fwd_pp3_kernel<false>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void *>(linear_fwd_pp3_kernel<true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr0 != hipSuccess || attr1 != hipSuccess)
        return fail(LCREC_EHIP, "linear_forward: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(attr0 != hipSuccess ? attr0 : attr1));
    const int grid = total_virtual < cus ? total_virtual : cus;
    if (sc)
        hipLaunchKernelGGL(linear_fwd_pp3_kernel<true>, dim3((unsigned)grid), dim3(512), lds, stream, x, W, b, sc, sh, y, n, out_dim,
                           in_dim, relu, bn_blocks, bm_blocks, total_virtual, xcd_order);
    else
        hipLaunchKernelGGL(linear_fwd_pp3_kernel<false>, dim3((unsigned)grid), dim3(512), lds, stream, x, W, b, sc, sh, y, n, out_dim,
                           in_dim, relu, bn_blocks, bm_blocks, total_virtual, xcd_order);
    return LCREC_OK;
}

static int launch_linear_pp(const float *x, int64_t n, int in_dim, const float *W, const float *b, const float *sc,
                            const float *sh, int relu, int out_dim, float *y, int xcd_order, hipStream_t stream)
{
    const int64_t bm_blocks = (n + 255) / 256;
    const int bn_blocks = (out_dim + 127) / 128;
    const int64_t grid = xcd_order ? ((bm_blocks + 7) / 8) * 8 * bn_blocks : bm_blocks * bn_blocks;
    if (grid > 0x7fffffffLL) return fail(LCREC_EINVAL, "linear_forward: grid too large (n=%lld)", (long long)n);
    TraceScope trace(K_LINEAR_PP, stream);
    // the persistent form where it applies (measured +1.2 % on C3, +0.5 .. +3.4 % per layer); LCREC_GEMM_PP3=0 turns it off
    static const int pp3 = [] { const char *e = getenv("LCREC_GEMM_PP3"); return e ? atoi(e) : 1; }();
    int rc;
    if (pp3 && in_dim % 64 == 0 && in_dim >= 12 * BK && out_dim % 128 == 0 && (int64_t)out_dim * 4 * 128 < (1ll << 31))
        rc = launch_pp3((int)grid, stream, x, W, b, sc, sh, y, n, out_dim, in_dim, relu, bn_blocks, (int)bm_blocks, xcd_order);
    else
        rc = launch_pp2(dim3((unsigned)grid), stream, x, W, b, sc, sh, y, n, out_dim, in_dim, relu, bn_blocks, (int)bm_blocks,
                        xcd_order);
    return rc ? rc : check_launch("linear_fwd_pp2_kernel");
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
static int launch_linear(const float *x, int64_t n, int in_dim, const float *W, const float *b,
                         const float *sc, const float *sh, int relu, int out_dim, float *y,
                         hipStream_t stream)
{
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    const int64_t bm_blocks = (n + BM - 1) / BM;
    const int bn_blocks = (out_dim + BN - 1) / BN;
    // LCREC_GEMM_TUNE=0 turns the XCD-aware tile order off (default on: same speed, 2.3x less fabric traffic by FETCH_SIZE)
    static const int tune = [] { const char *e = getenv("LCREC_GEMM_TUNE"); return (e ? atoi(e) : 1) & 1; }();
    const int64_t grid = (tune & 1) ? ((bm_blocks + 7) / 8) * 8 * bn_blocks : bm_blocks * bn_blocks;
    if (grid > 0x7fffffffLL) return fail(LCREC_EINVAL, "linear_forward: grid too large (n=%lld)", (long long)n);
    TraceScope trace(BM == 64 ? K_LINEAR_64x64 : BN == 128 ? K_LINEAR_128x128 : BN == 64 ? K_LINEAR_128x64 : K_LINEAR_128x32, stream);
    // K % 32 == 0 (every layer of the run.sh architecture): the instantiation whose staging path has no VALU
    static const int fast = [] { const char *e = getenv("LCREC_GEMM_FAST"); return e ? atoi(e) : 1; }();
    if (fast && in_dim % BK == 0 && (int64_t)in_dim * 4 * (BM + 64) < (1ll << 31))
        hipLaunchKernelGGL((linear_fwd_kernel<WAVES_M, WAVES_N, TM, TN, true>), dim3((unsigned)grid), dim3(256), 0,
                           stream, x, W, b, sc, sh, y, n, out_dim, in_dim, relu, bn_blocks, (int)bm_blocks, tune, 1 << 30,
                           (int64_t)0);
    else
        hipLaunchKernelGGL((linear_fwd_kernel<WAVES_M, WAVES_N, TM, TN, false>), dim3((unsigned)grid), dim3(256), 0,
                           stream, x, W, b, sc, sh, y, n, out_dim, in_dim, relu, bn_blocks, (int)bm_blocks, tune, 1 << 30,
                           (int64_t)0);
    return check_launch("linear_fwd_kernel");
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ partial, int splits, int64_t total4,
                                                          float *__restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= total4) return;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(partial);
    f32x4 acc = p[q];
    for (int s = 1; s < splits; ++s) acc = acc + p[(int64_t)s * total4 + q];       // run order: ((p0 + p1) + p2) + ...
    reinterpret_cast<f32x4 *>(out)[q] = acc;
}

// a batch-sized launch (fewer than 512 tiles of 128 x 128) whose 64 x 128 tiles still give every CU one
static bool wide_tiles_fill(int64_t M, int N) { return ((M + 63) / 64) * ((N + 127) / 128) >= 256; }

// C[M][N] = A * B with A given [M][K] (TA false) or [K][M] (TA true) and B given [K][N] (always k-major here):
// the two backward products of a Linear layer.  Same tiles and the same dispatch by (M, N) as the forward launches;
// `splits` > 1 cuts K into that many runs of K-tiles whose partial products go to `partial` [splits][M][N].
template <int WAVES_M, int WAVES_N, int TM, int TN, bool TA>
static int launch_kmajor(const float *A, const float *B, int64_t M, int N, int K, float *C, int splits, float *partial,
                         hipStream_t stream)
{
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    const int64_t bm_blocks = (M + BM - 1) / BM;
    const int bn_blocks = (N + BN - 1) / BN;
    const int64_t grid = ((bm_blocks + 7) / 8) * 8 * bn_blocks;
    if (grid > 0x7fffffffLL) return fail(LCREC_EINVAL, "linear_backward: grid too large");
    const int nk = (K + BK - 1) / BK;
    const int per = splits > 1 ? (nk + splits - 1) / splits : 1 << 30;
    TraceScope trace(BM == 64 ? K_LINEAR_64x64 : BN == 128 ? K_LINEAR_128x128 : BN == 64 ? K_LINEAR_128x64 : K_LINEAR_128x32, stream);
    hipLaunchKernelGGL((linear_fwd_kernel<WAVES_M, WAVES_N, TM, TN, true, TA, true>), dim3((unsigned)grid, (unsigned)(splits > 1 ? splits : 1)),
                       dim3(256), 0, stream, A, B, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                       splits > 1 ? partial : C, M, N, K, 0, bn_blocks, (int)bm_blocks, 1, per, splits > 1 ? M * (int64_t)N : (int64_t)0);
    if (splits > 1) {
        const int64_t total4 = M * (int64_t)N / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, partial, splits, total4, C);
    }
    return check_launch("linear_fwd_kernel (k-major operands)");
}

// which tile shape a (M, N) output gets: 0 = 64x64, 1 = 128x128, 2 = 128x64, 3 = 128x32, 4 = 64x128 (the forward rule)
static int kmajor_shape(int64_t M, int N)
{
    if (N > 64) return ((M + 127) / 128) * ((N + 127) / 128) >= 512 ? 1 : (wide_tiles_fill(M, N) ? 4 : 0);
    return N > 32 ? ((M + 127) / 128 < 256 ? 0 : 2) : 3;
}

template <bool TA>
static int gemm_kmajor(const float *A, const float *B, int64_t M, int N, int K, float *C, int splits, float *partial,
                       hipStream_t stream)
{
    if constexpr (!TA) {
        if (splits <= 1 && use_s16_tiles(M, N, K))          // dX of an under-filled launch: the 32 x 64 tiles
            return launch_s16(A, B, nullptr, nullptr, nullptr, 0, C, M, N, K, true, stream);
    }
    switch (kmajor_shape(M, N)) {
    case 0: return launch_kmajor<2, 2, 1, 1, TA>(A, B, M, N, K, C, splits, partial, stream);
    case 1: return launch_kmajor<2, 2, 2, 2, TA>(A, B, M, N, K, C, splits, partial, stream);
    case 2: return launch_kmajor<4, 1, 1, 2, TA>(A, B, M, N, K, C, splits, partial, stream);
    case 4: return launch_kmajor<2, 2, 1, 2, TA>(A, B, M, N, K, C, splits, partial, stream);
    default: return launch_kmajor<4, 1, 1, 1, TA>(A, B, M, N, K, C, splits, partial, stream);
    }
}

// dW = dY^T X contracts over the batch: a [out][in] output of a narrow layer is a handful of tiles, each a serial
// chain of n/32 K-tiles (66 us at n = 2048 whatever the layer).  So K is cut into S runs of K-tiles -- S chosen to put
// at least ~512 workgroups on the chip, at least 4 K-tiles per run, at most 16 runs -- and the S partial products are
// added in run order.  This is part of the arithmetic contract of lcrec_linear_backward (include/lcrec.h): gw is the
// ordered sum of S fma chains, S = lcrec_linear_backward_splits(n, in_dim, out_dim).
int linear_backward_splits(int64_t n, int in_dim, int out_dim)
{
    static const int cap = [] { const char *e = getenv("LCREC_GEMM_SPLITK"); return e ? atoi(e) : 16; }();
    const int shape = kmajor_shape(out_dim, in_dim);
    const int bm = shape == 0 ? 64 : 128, bn = shape == 0 ? 64 : shape == 1 ? 128 : shape == 2 ? 64 : 32;
    const int64_t tiles = ((out_dim + bm - 1) / bm) * (int64_t)((in_dim + bn - 1) / bn);
    const int64_t nk = (n + BK - 1) / BK;
    int64_t s = 512 / (tiles > 0 ? tiles : 1);
    if (s > nk / 4) s = nk / 4;
    if (s > cap) s = cap;
    if (s < 2) return 1;
    const int64_t per = (nk + s - 1) / s;          // K-tiles per run; the count returned has no empty run
    return (int)((nk + per - 1) / per);
}

size_t linear_backward_workspace(int64_t n, int in_dim, int out_dim)
{
    const int s = linear_backward_splits(n, in_dim, out_dim);
    return s > 1 ? (size_t)s * out_dim * in_dim * sizeof(float) : 0;
}

int linear_backward(const float *gy, const float *x, const float *W, int64_t n, int in_dim, int out_dim, float *gx, float *gw,
                    void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    if (n == 0) return LCREC_OK;
    if (!gy || (gx && !W) || (gw && !x)) return fail(LCREC_EINVAL, "linear_backward: NULL pointer");
    if (n < 0 || in_dim <= 0 || out_dim <= 0) return fail(LCREC_EINVAL, "linear_backward: bad shape");
    if (in_dim % 4 != 0 || out_dim % 4 != 0)
        return fail(LCREC_EUNSUPPORTED, "linear_backward: in_dim=%d / out_dim=%d must be multiples of 4", in_dim, out_dim);
    if (((uintptr_t)gy | (uintptr_t)x | (uintptr_t)W) & 15) return fail(LCREC_EINVAL, "linear_backward: operands must be 16-byte aligned");
    const int64_t widest = in_dim > out_dim ? in_dim : out_dim;
    if ((n + 64) * widest * 4 >= (1ll << 31) || (int64_t)out_dim * in_dim * 4 >= (1ll << 31))
        return fail(LCREC_EUNSUPPORTED, "linear_backward: operands of %lld x %lld floats exceed the 2 GiB a buffer descriptor spans "
                                        "(a training batch is expected here)", (long long)n, (long long)widest);
    if (gx) {
        // dX [n][in] = dY [n][out] * W [out][in]: A row-major (K = out), B = W as it is stored
        if (out_dim % BK != 0) return fail(LCREC_EUNSUPPORTED, "linear_backward: out_dim=%d is not a multiple of 32", out_dim);
        int rc = gemm_kmajor<false>(gy, W, n, in_dim, out_dim, gx, 1, nullptr, stream);
        if (rc) return rc;
    }
    if (gw) {
        // dW [out][in] = dY^T * X: A = dY [n][out] and B = X [n][in], both as they are stored (K = n, any length)
        const int splits = linear_backward_splits(n, in_dim, out_dim);
        const size_t need = linear_backward_workspace(n, in_dim, out_dim);
        if (splits > 1 && (!workspace || workspace_bytes < need))
            return fail(LCREC_EWORKSPACE, "linear_backward: workspace %zu B < required %zu B", workspace_bytes, need);
        int rc = gemm_kmajor<true>(gy, x, out_dim, in_dim, (int)n, gw, splits, reinterpret_cast<float *>(workspace), stream);
        if (rc) return rc;
    }
    return LCREC_OK;
}

// ---- grouped weight gradients (lcrec_linear_backward_weights): same S and run lengths per problem as linear_backward,
// hence the same bits; one launch for the products, one for the ordered sums of their K-runs
// K-runs of problem q: its `splits` field when set (normalised so that no run is empty), else the per-layer rule
static int dw_problem_splits(const lcrec_dw_problem &q)
{
    if (q.splits <= 0) return linear_backward_splits(q.n, q.in_dim, q.out_dim);
    const int64_t nk = (q.n + BK - 1) / BK;
    const int64_t s = q.splits < nk ? q.splits : (nk > 0 ? nk : 1);
    const int64_t per = (nk + s - 1) / s;
    return per > 0 ? (int)((nk + per - 1) / per) : 1;
}
static size_t dw_problem_workspace(const lcrec_dw_problem &q)
{
    const int s = dw_problem_splits(q);
    return s > 1 ? (size_t)s * q.out_dim * q.in_dim * sizeof(float) : 0;
}

size_t linear_backward_weights_workspace(const lcrec_dw_problem *pr, int count)
{
    size_t total = 0;
    for (int i = 0; i < count; ++i) total += align_up(dw_problem_workspace(pr[i]), 256);
    return total;
}

int linear_backward_weights(const lcrec_dw_problem *pr, int count, void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    if (count == 0) return LCREC_OK;
    if (!pr || count < 0 || count > DW_GROUP_MAX)
        return fail(LCREC_EINVAL, "linear_backward_weights: %d problems (1..%d supported)", count, DW_GROUP_MAX);
    if (workspace_bytes < linear_backward_weights_workspace(pr, count) || (!workspace && linear_backward_weights_workspace(pr, count)))
        return fail(LCREC_EWORKSPACE, "linear_backward_weights: workspace %zu B < required %zu B", workspace_bytes,
                    linear_backward_weights_workspace(pr, count));
    DwGroup g = {};
    DwReduce r = {};
    char *ws = reinterpret_cast<char *>(workspace);
    unsigned wg = 0, rwg = 0;
    bool any_pro = false;
    for (int i = 0; i < count; ++i) {
        const lcrec_dw_problem &q = pr[i];
        if (!q.gy || !q.x || !q.gw) return fail(LCREC_EINVAL, "linear_backward_weights: NULL pointer in problem %d", i);
        if (q.n <= 0 || q.in_dim <= 0 || q.out_dim <= 0 || q.in_dim % 4 || q.out_dim % 4)
            return fail(LCREC_EUNSUPPORTED, "linear_backward_weights: problem %d: n=%lld in=%d out=%d (positive, widths multiples of 4)", i,
                        (long long)q.n, q.in_dim, q.out_dim);
        if (((uintptr_t)q.gy | (uintptr_t)q.x | (uintptr_t)q.gw) & 15)
            return fail(LCREC_EINVAL, "linear_backward_weights: operands must be 16-byte aligned");
        const int64_t widest = q.in_dim > q.out_dim ? q.in_dim : q.out_dim;
        if ((q.n + 64) * widest * 4 >= (1ll << 31) || (int64_t)q.out_dim * q.in_dim * 4 >= (1ll << 31))
            return fail(LCREC_EUNSUPPORTED, "linear_backward_weights: problem %d exceeds the 2 GiB a buffer descriptor spans", i);
        const int splits = dw_problem_splits(q);
        const int nk = (int)((q.n + BK - 1) / BK);
        const int64_t bm_blocks = (q.out_dim + 63) / 64;
        const int bn_blocks = (q.in_dim + 63) / 64;
        const int64_t tiles = ((bm_blocks + 7) / 8) * 8 * bn_blocks;      // XCD-aware numbering has holes (see linear_tile_body)
        g.A[i] = q.gy; g.B[i] = q.x;
        if ((q.x_scale == nullptr) != (q.x_shift == nullptr)) return fail(LCREC_EINVAL, "linear_backward_weights: problem %d: x_scale and x_shift go together", i);
        if (((uintptr_t)q.x_scale | (uintptr_t)q.x_shift) & 15) return fail(LCREC_EINVAL, "linear_backward_weights: x_scale / x_shift must be 16-byte aligned");
        g.xs[i] = q.x_scale; g.xh[i] = q.x_shift;
        g.lo[i] = (q.x_scale && q.x_relu) ? 0.0f : -__builtin_inff();
        any_pro = any_pro || q.x_scale != nullptr;
        g.M[i] = q.out_dim; g.N[i] = q.in_dim; g.K[i] = (int)q.n;
        g.bn_blocks[i] = bn_blocks; g.bm_blocks[i] = (int)bm_blocks; g.tiles[i] = (int)tiles;
        g.kt_per_split[i] = splits > 1 ? (nk + splits - 1) / splits : 1 << 30;
        g.wg_start[i] = wg;
        wg += (unsigned)(tiles * (splits > 1 ? splits : 1));
        if (splits > 1) {
            float *partial = reinterpret_cast<float *>(ws);
            ws += align_up(dw_problem_workspace(q), 256);
            g.C[i] = partial;
            const int j = r.count++;
            r.partial[j] = partial; r.out[j] = q.gw; r.splits[j] = splits;
            r.total4[j] = (unsigned)((int64_t)q.out_dim * q.in_dim / 4);
            r.wg_start[j] = rwg;
            rwg += (r.total4[j] + 255) / 256;
        } else {
            g.C[i] = q.gw;
        }
    }
    g.count = count;
    g.wg_start[count] = wg;
    r.wg_start[r.count] = rwg;
    TraceScope trace(K_LINEAR_64x64, stream);
    if (any_pro) hipLaunchKernelGGL(linear_dw_grouped_kernel<true>, dim3(wg), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(linear_dw_grouped_kernel<false>, dim3(wg), dim3(256), 0, stream, g);
    if (r.count) hipLaunchKernelGGL(splitk_reduce_grouped_kernel, dim3(rwg), dim3(256), 0, stream, r);
    return check_launch("linear_dw_grouped_kernel");
}

// ---- lcrec_linear_bn_forward: the training step's Linear with its input's BatchNorm + ReLU folded into the operand staging
// and its output's batch statistics taken in the epilogue (TileExtras)
static int train_tile_shape(int64_t n, int out_dim)
{
    if (out_dim > 64) return wide_tiles_fill(n, out_dim) ? 4 : 0;       // 64 x 128 : 64 x 64
    return out_dim > 32 ? 0 : 3;                                        // 64 x 64 : 128 x 32
}

size_t linear_bn_forward_workspace(int64_t n, int out_dim)
{
    const int bm = train_tile_shape(n, out_dim) == 3 ? 128 : 64;
    return (size_t)((n + bm - 1) / bm) * 3 * (size_t)out_dim * sizeof(float);
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
static int launch_train_fwd(const float *x, int64_t n, int in_dim, const float *W, const float *b, int out_dim, float *t,
                            bool pro, bool stats, const TileExtras &ex, hipStream_t stream)
{
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    const int64_t bm_blocks = (n + BM - 1) / BM;
    const int bn_blocks = (out_dim + BN - 1) / BN;
    const int64_t grid = ((bm_blocks + 7) / 8) * 8 * bn_blocks;
    TraceScope trace(BM == 64 ? K_LINEAR_64x64 : K_LINEAR_128x32, stream);
#define LCREC_TRAIN_LAUNCH(P, S)                                                                                                  \
    hipLaunchKernelGGL((linear_train_fwd_kernel<WAVES_M, WAVES_N, TM, TN, P, S>), dim3((unsigned)grid), dim3(256), 0, stream, x, W, b, t, \
                       n, out_dim, in_dim, bn_blocks, (int)bm_blocks, 1, ex)
    if (pro && stats) LCREC_TRAIN_LAUNCH(1, true);
    else if (pro) LCREC_TRAIN_LAUNCH(1, false);
    else LCREC_TRAIN_LAUNCH(0, true);
#undef LCREC_TRAIN_LAUNCH
    return check_launch("linear_train_fwd_kernel");
}

int linear_bn_forward(const float *x, int64_t n, int in_dim, const float *in_scale, const float *in_shift, int in_relu,
                      const float *W, const float *b, int out_dim, float *t_out, int want_stats, const float *gamma,
                      const float *beta, float eps, float momentum, float *running_mean, float *running_var, float *mean_out,
                      float *rstd_out, float *scale_out, float *shift_out, void *workspace, size_t workspace_bytes,
                      unsigned *tickets, hipStream_t stream)
{
    if (!x || !W || !t_out) return fail(LCREC_EINVAL, "linear_bn_forward: NULL pointer");
    if (n < 1 || in_dim <= 0 || out_dim <= 0) return fail(LCREC_EINVAL, "linear_bn_forward: bad shape");
    if ((in_scale == nullptr) != (in_shift == nullptr)) return fail(LCREC_EINVAL, "linear_bn_forward: in_scale and in_shift go together");
    const bool pro = in_scale != nullptr, stats = want_stats != 0;
    if (!pro && !stats) return linear_forward(x, n, in_dim, W, b, nullptr, nullptr, 0, out_dim, t_out, stream);
    if (in_dim % BK != 0 || out_dim % 4 != 0)
        return fail(LCREC_EUNSUPPORTED, "linear_bn_forward: in_dim=%d must be a multiple of 32, out_dim=%d of 4", in_dim, out_dim);
    if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)in_scale | (uintptr_t)in_shift) & 15)
        return fail(LCREC_EINVAL, "linear_bn_forward: operands must be 16-byte aligned");
    if (((n + 127) / 128) * ((out_dim + 127) / 128) >= 512 || (int64_t)in_dim * 4 * (n + 128) >= (1ll << 31) || out_dim > 64 * 64)
        return fail(LCREC_EUNSUPPORTED, "linear_bn_forward: sized for training batches (n=%lld, out_dim=%d)", (long long)n, out_dim);
    if (stats) {
        if (n < 2) return fail(LCREC_EINVAL, "linear_bn_forward: training-mode BatchNorm needs more than 1 row (n=%lld)", (long long)n);
        if (!mean_out || !rstd_out || !scale_out || !shift_out || !tickets) return fail(LCREC_EINVAL, "linear_bn_forward: NULL statistics output or tickets");
        if (!workspace || workspace_bytes < linear_bn_forward_workspace(n, out_dim))
            return fail(LCREC_EWORKSPACE, "linear_bn_forward: workspace %zu B < required %zu B", workspace_bytes, linear_bn_forward_workspace(n, out_dim));
    }
    TileExtras ex = {};
    ex.in_scale = in_scale; ex.in_shift = in_shift; ex.in_lo = (pro && in_relu) ? 0.0f : -__builtin_inff();
    ex.stat_partial = reinterpret_cast<float *>(workspace); ex.tickets = tickets;
    ex.gamma = gamma; ex.beta = beta; ex.eps = eps; ex.momentum = momentum;
    ex.running_mean = running_mean; ex.running_var = running_var;
    ex.mean_out = mean_out; ex.rstd_out = rstd_out; ex.scale_out = scale_out; ex.shift_out = shift_out;
    switch (train_tile_shape(n, out_dim)) {
    case 4: return launch_train_fwd<2, 2, 1, 2>(x, n, in_dim, W, b, out_dim, t_out, pro, stats, ex, stream);
    case 3: return launch_train_fwd<4, 1, 1, 1>(x, n, in_dim, W, b, out_dim, t_out, pro, stats, ex, stream);
    default: return launch_train_fwd<2, 2, 1, 1>(x, n, in_dim, W, b, out_dim, t_out, pro, stats, ex, stream);
    }
}

int linear_forward(const float *x, int64_t n, int in_dim, const float *W, const float *b,
                   const float *bn_scale, const float *bn_shift, int relu, int out_dim, float *y,
                   hipStream_t stream)
{
    if (n == 0) return LCREC_OK;                       // empty batch: nothing to read or write
    if (!x || !W || !y) return fail(LCREC_EINVAL, "linear_forward: NULL pointer");
    if (n < 0 || in_dim <= 0 || out_dim <= 0) return fail(LCREC_EINVAL, "linear_forward: bad shape");
    if ((bn_scale == nullptr) != (bn_shift == nullptr))
        return fail(LCREC_EINVAL, "linear_forward: bn_scale and bn_shift must both be given or both NULL");
    if (in_dim % 8 != 0)
        return fail(LCREC_EUNSUPPORTED, "linear_forward: in_dim=%d is not a multiple of 8", in_dim);
    if (((uintptr_t)x | (uintptr_t)W) & 15)
        return fail(LCREC_EINVAL, "linear_forward: x and W must be 16-byte aligned");
    // Wide layers: the ping-pong kernel (one 512-thread workgroup per CU) wins as soon as its 256 x 128 tiles fill
    // the 256 CUs in whole rounds: at least one round, and either >= 8 rounds or <= 20 % of the last round empty
    // (measured with tools/pp_sweep.sh: 8 192 x 768->2048 = 512 tiles: 117 vs 89 TFLOP/s; 16 896 x 1024->512 = 264
    // tiles: 76 vs 83).  Otherwise the 128 x 128 / 64 x 64 kernels (several workgroups per CU, finer tail).
    // LCREC_GEMM_PP=0/1 forces one or the other (tuning only).
    static const int pp = [] { const char *e = getenv("LCREC_GEMM_PP"); return e ? atoi(e) : -1; }();
    static const int pp_tune = [] { const char *e = getenv("LCREC_GEMM_TUNE"); return (e ? atoi(e) : 1) & 1; }();
    const int64_t pp_tiles = ((n + 255) / 256) * ((out_dim + 127) / 128);
    const int64_t pp_rounds = (pp_tiles + 255) / 256;
    const bool pp_fits = pp_tiles >= 256 && (pp_rounds >= 8 || pp_tiles * 5 >= pp_rounds * 256 * 4);
    const bool pp_ok = in_dim % BK == 0 && (int64_t)in_dim * 4 * 192 < (1ll << 31);     // else the generic kernels
    // Mid-sized launches (Games: 16 859 rows = 65.9 row panels) leave the last round of 256 x 128 tiles mostly empty --
    // 1056 tiles are 4.1 rounds, 528 are 2.06 -- and a persistent workgroup's time is its LONGEST tile list.  Rows are
    // independent, so such a launch is cut at a row-panel boundary: the largest head whose tiles fill whole rounds (or
    // >= 90 % of the last one) goes to the ping-pong kernel, the tail (here 475 rows) to the generic kernels in a second
    // launch.  Same chains, same bits.  Launches of >= 8 rounds are left alone (their partial round is noise).
    const int64_t pp_last = pp_tiles % 256;
    if (out_dim > 64 && pp_ok && pp == -1 && pp_rounds < 8 && pp_last != 0 && pp_last * 10 < 256 * 9) {
        const int64_t ntile = (out_dim + 127) / 128;
        // (the tail stays below one round of tiles: look back at most 256 / ntile panels)
        for (int64_t p = n / 256; p >= 1 && p * ntile >= 256 && p >= n / 256 - 256 / ntile; --p) {
            const int64_t t = p * ntile, last = t % 256;
            if (last == 0 || last * 10 >= 256 * 9) {
                const int64_t head = p * 256;
                if (head >= n) break;                     // the whole launch is already whole rounds
                int rc = launch_linear_pp(x, head, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, pp_tune, stream);
                if (rc) return rc;
                return linear_forward(x + head * in_dim, n - head, in_dim, W, b, bn_scale, bn_shift, relu, out_dim,
                                      y + head * out_dim, stream);
            }
        }
    }
    const bool use_pp = out_dim > 64 && pp_ok && (pp == 1 || (pp == -1 && pp_fits));
    if (use_pp) return launch_linear_pp(x, n, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, pp_tune, stream);
    // launches that 64 x 64 tiles cannot fill the chip with: 32 x 64 tiles on the 16 x 16 x 4 MFMA (linear_s16_kernel)
    if (use_s16_tiles(n, out_dim, in_dim))
        return launch_s16(x, W, b, bn_scale, bn_shift, relu, y, n, out_dim, in_dim, false, stream);
    if (out_dim > 64) {
        // batch-sized problems (a training step has 1-2 k rows): 128 x 128 tiles would leave most CUs idle,
        // so launches with fewer than two tiles per CU use 64 x 64 tiles (4x the workgroups)
        const int64_t tiles128 = ((n + 127) / 128) * ((out_dim + 127) / 128);
        static const int small_tile = [] { const char *e = getenv("LCREC_GEMM_SMALL"); return e ? atoi(e) : 0; }();   // tuning only
        if (tiles128 < 512 && small_tile == 1) return launch_linear<2, 2, 1, 2>(x, n, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, stream);
        if (tiles128 < 512 && small_tile == 2) return launch_linear<2, 2, 2, 1>(x, n, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, stream);
        // ... and 64 x 128 tiles (two accumulators per wave: a K-tile's fixed costs are paid once per 32 MFMAs instead of 16)
        // when those still fill the chip: 768 -> 2048 at batch 2048 72 -> 62 us, 4096 -> 2048 at batch 1024 148 -> 144 us
        if (tiles128 < 512 && small_tile == 0 && wide_tiles_fill(n, out_dim))
            return launch_linear<2, 2, 1, 2>(x, n, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, stream);
        if (tiles128 < 512) return launch_linear<2, 2, 1, 1>(x, n, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, stream);
        return launch_linear<2, 2, 2, 2>(x, n, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, stream);
    }
    // 33 .. 64 columns: 128 x 64 tiles, or 64 x 64 while those would not give every CU one (a batch-sized launch: twice the
    // workgroups, and the register-buffered K-tile: 128 -> 64 at batch 1024 9.0 -> ~5 us)
    if (out_dim > 32 && (n + 127) / 128 < 256) return launch_linear<2, 2, 1, 1>(x, n, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, stream);
    if (out_dim > 32) return launch_linear<4, 1, 1, 2>(x, n, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, stream);
    return launch_linear<4, 1, 1, 1>(x, n, in_dim, W, b, bn_scale, bn_shift, relu, out_dim, y, stream);
}

}  // namespace lcrec

#ifdef LCREC_GEMM_STAMP
extern "C" __attribute__((visibility("default"))) int lcrec_debug_gemm_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcrec::g_stamps), sizeof(unsigned long long) * 8 * 64 * 4);
}
extern "C" __attribute__((visibility("default"))) int lcrec_debug_generic_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcrec::g_gen_stamps), sizeof(unsigned long long) * 4 * 64 * 6);
}
extern "C" __attribute__((visibility("default"))) int lcrec_debug_gemm_marks(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcrec::g_marks), sizeof(unsigned long long) * 8 * 4);
}
#endif
