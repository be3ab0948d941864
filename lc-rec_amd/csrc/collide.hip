// Collision detection over index tuples: which items share an identical L-tuple of codes.
//
// Replaces the Python string/dict passes of the reference:
//   index/trainer.py:139-150              collision_rate = (N - |set of "-".join(codes)|) / N
//   index/generate_indices.py:18-42       check_collision / get_indices_count / get_collision_item
// get_collision_item's contract is kept exactly: groups (tuples held by >= 2 items) are ordered by
// the first occurrence of their tuple in item order, item ids ascend inside a group.
//
// Method (all on device, O(n) memory, integer work -> HBM-bound sorts):
//   1. pack each tuple into a <=128-bit mixed-radix key (ceil(log2 K_l) bits per level);
//   2. stable LSD radix sort of (key, item id)  [rocPRIM device radix sort];
//   3. segment heads -> every item learns the id of its tuple's FIRST item (stable sort => minimum id);
//   4. stable sort of (first id, item id): groups now lie in first-occurrence order, ids ascending;
//   5. segment sizes, compaction of the members of groups with size >= 2, group offsets, counters.
#include "common.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace lcrec {

struct PackDesc {
    int L;
    int bits[LCREC_MAX_LEVELS];
    int K[LCREC_MAX_LEVELS];
};

__global__ void pack_keys_kernel(const int64_t *__restrict__ idx, int64_t n, PackDesc d, uint64_t *klo, uint64_t *khi,
                                 int64_t *ids)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned __int128 key = 0;
    for (int l = 0; l < d.L; ++l) {
        int64_t v = idx[i * d.L + l];
        v = v < 0 ? 0 : (v >= d.K[l] ? d.K[l] - 1 : v);
        key = (key << d.bits[l]) | (unsigned __int128)(uint64_t)v;
    }
    klo[i] = (uint64_t)key;
    khi[i] = (uint64_t)(key >> 64);
    ids[i] = i;
}

__global__ void gather_u64_kernel(const uint64_t *src, const int64_t *ids, int64_t n, uint64_t *dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[ids[i]];
}

// flag[i] = position i if it starts a new tuple in sorted order, else 0 (for a max-scan)
__global__ void tuple_heads_kernel(const uint64_t *klo, const uint64_t *khi, const int64_t *ids, int64_t n, int64_t *headpos)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool head = i == 0;
    if (!head) {
        const int64_t a = ids[i], b = ids[i - 1];
        head = klo[a] != klo[b] || khi[a] != khi[b];
    }
    headpos[i] = head ? i : 0;
}

__global__ void first_ids_kernel(const int64_t *ids, const int64_t *headpos, int64_t n, uint64_t *fid)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fid[i] = (uint64_t)ids[headpos[i]];
}

__global__ void seg_flags_kernel(const uint64_t *fid, int64_t n, int64_t *flag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || fid[i] != fid[i - 1]) ? 1 : 0;
}

// seg[i] = inclusive scan of flags = 1-based segment number; count sizes
__global__ void seg_sizes_kernel(const int64_t *seg, int64_t n, unsigned long long *sizes)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&sizes[seg[i] - 1], 1ULL);
}

__global__ void member_flags_kernel(const int64_t *seg, const int64_t *flag, const unsigned long long *sizes, int64_t n,
                                    int64_t *member, int64_t *ghead, unsigned long long *counters)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long sz = sizes[seg[i] - 1];
    const bool m = sz > 1;
    member[i] = m ? 1 : 0;
    ghead[i] = (m && flag[i]) ? 1 : 0;
    if (flag[i]) atomicMax(&counters[3], sz);
    if (i == n - 1) counters[0] = (unsigned long long)seg[i];   // number of distinct tuples
}

__global__ void emit_groups_kernel(const int64_t *ids2, const int64_t *member, const int64_t *ghead, const int64_t *mpos,
                                   const int64_t *gpos, int64_t n, int64_t *members_out, int64_t *offsets_out,
                                   unsigned long long *counters)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (member[i]) members_out[mpos[i]] = ids2[i];
    if (ghead[i]) offsets_out[gpos[i]] = mpos[i];
    if (i == n - 1) {
        const int64_t groups = gpos[i] + ghead[i], members = mpos[i] + member[i];
        offsets_out[groups] = members;
        counters[1] = (unsigned long long)groups;
        counters[2] = (unsigned long long)members;
    }
}

struct MaxOp {
    __device__ __host__ int64_t operator()(int64_t a, int64_t b) const { return a > b ? a : b; }
};

static size_t prim_temp_bytes(int64_t n)
{
    size_t a = 0, b = 0, c = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (uint64_t *)nullptr, (uint64_t *)nullptr, (int64_t *)nullptr,
                                    (int64_t *)nullptr, (size_t)n, 0, 64, (hipStream_t)0, false);
    (void)rocprim::inclusive_scan(nullptr, b, (int64_t *)nullptr, (int64_t *)nullptr, (size_t)n, MaxOp(), (hipStream_t)0, false);
    (void)rocprim::exclusive_scan(nullptr, c, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n,
                                  rocprim::plus<int64_t>(), (hipStream_t)0, false);
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    // the size query needs a device; without one (build host) fall back to a generous bound
    const size_t floor_bytes = (size_t)n * 32 + (1u << 20);
    if (m < 256) m = floor_bytes;
    return align_up(m, 256) + 256;
}

size_t collision_workspace(int64_t n, int L)
{
    (void)L;
    const size_t arr = align_up((size_t)(n > 0 ? n : 1) * 8, 256);
    return 10 * arr + prim_temp_bytes(n > 0 ? n : 1);
}

int collision_groups(const int64_t *idx, int64_t n, int L, const int *K, int64_t *members_out, int64_t *offsets_out,
                     int64_t *counters_out, void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    if ((n > 0 && !idx) || !K || !counters_out) return fail(LCREC_EINVAL, "collision_groups: NULL pointer");
    if (n < 0 || L < 1 || L > LCREC_MAX_LEVELS) return fail(LCREC_EINVAL, "collision_groups: bad n or L");
    if ((members_out == nullptr) != (offsets_out == nullptr))
        return fail(LCREC_EINVAL, "collision_groups: members_out and offsets_out go together");
    PackDesc d;
    d.L = L;
    int total_bits = 0;
    for (int l = 0; l < L; ++l) {
        if (K[l] < 1) return fail(LCREC_EINVAL, "collision_groups: K[%d]=%d", l, K[l]);
        int b = 0;
        while ((1LL << b) < K[l]) ++b;
        d.bits[l] = b;
        d.K[l] = K[l];
        total_bits += b;
    }
    if (total_bits > 128)
        return fail(LCREC_EUNSUPPORTED, "collision_groups: tuples need %d bits (> 128)", total_bits);
    hipError_t he = hipMemsetAsync(counters_out, 0, 4 * sizeof(int64_t), stream);
    if (he != hipSuccess) return fail(LCREC_EHIP, "collision_groups: %s", hipGetErrorString(he));
    if (n == 0) {
        if (offsets_out) he = hipMemsetAsync(offsets_out, 0, sizeof(int64_t), stream);
        return he == hipSuccess ? LCREC_OK : fail(LCREC_EHIP, "collision_groups: %s", hipGetErrorString(he));
    }
    const size_t need = collision_workspace(n, L);
    if (!workspace || workspace_bytes < need)
        return fail(LCREC_EWORKSPACE, "collision_groups: workspace %zu B < required %zu B", workspace_bytes, need);

    const size_t arr = align_up((size_t)n * 8, 256);
    char *ws = reinterpret_cast<char *>(workspace);
    auto take = [&]() { char *p = ws; ws += arr; return p; };
    uint64_t *klo = (uint64_t *)take(), *khi = (uint64_t *)take();
    uint64_t *ka = (uint64_t *)take(), *kb = (uint64_t *)take();
    int64_t *ia = (int64_t *)take(), *ib = (int64_t *)take();
    int64_t *s0 = (int64_t *)take(), *s1 = (int64_t *)take(), *s2 = (int64_t *)take(), *s3 = (int64_t *)take();
    void *temp = ws;
    size_t temp_bytes = prim_temp_bytes(n);
    unsigned long long *counters = reinterpret_cast<unsigned long long *>(counters_out);

    const unsigned grid = (unsigned)((n + 255) / 256);
    TraceScope trace(K_COLLISION, stream);
    hipLaunchKernelGGL(pack_keys_kernel, dim3(grid), dim3(256), 0, stream, idx, n, d, klo, khi, ia);

    // stable sort by the low word, then (if the tuple is wider than 64 bits) by the high word
    const int lo_bits = total_bits > 64 ? 64 : (total_bits > 0 ? total_bits : 1);
    size_t tb = temp_bytes;
    he = rocprim::radix_sort_pairs(temp, tb, klo, ka, ia, ib, (size_t)n, 0, lo_bits, stream, false);
    if (he != hipSuccess) return fail(LCREC_EHIP, "collision_groups: sort: %s", hipGetErrorString(he));
    int64_t *ids = ib, *spare = ia;
    if (total_bits > 64) {
        hipLaunchKernelGGL(gather_u64_kernel, dim3(grid), dim3(256), 0, stream, khi, ids, n, ka);
        tb = temp_bytes;
        he = rocprim::radix_sort_pairs(temp, tb, ka, kb, ids, spare, (size_t)n, 0, total_bits - 64, stream, false);
        if (he != hipSuccess) return fail(LCREC_EHIP, "collision_groups: sort: %s", hipGetErrorString(he));
        int64_t *t = ids; ids = spare; spare = t;
    }
    // first item id of every item's tuple
    hipLaunchKernelGGL(tuple_heads_kernel, dim3(grid), dim3(256), 0, stream, klo, khi, ids, n, s0);
    tb = temp_bytes;
    he = rocprim::inclusive_scan(temp, tb, s0, s1, (size_t)n, MaxOp(), stream, false);
    if (he != hipSuccess) return fail(LCREC_EHIP, "collision_groups: scan: %s", hipGetErrorString(he));
    hipLaunchKernelGGL(first_ids_kernel, dim3(grid), dim3(256), 0, stream, ids, s1, n, ka);
    // order groups by first occurrence (ids stay ascending inside a group: the sort is stable)
    int idbits = 1;
    while ((1LL << idbits) < n) ++idbits;
    tb = temp_bytes;
    he = rocprim::radix_sort_pairs(temp, tb, ka, kb, ids, spare, (size_t)n, 0, idbits, stream, false);
    if (he != hipSuccess) return fail(LCREC_EHIP, "collision_groups: sort: %s", hipGetErrorString(he));
    int64_t *ids2 = spare;
    uint64_t *fid2 = kb;
    // segments, sizes, members
    hipLaunchKernelGGL(seg_flags_kernel, dim3(grid), dim3(256), 0, stream, fid2, n, s0);                 // s0 = flag
    tb = temp_bytes;
    he = rocprim::inclusive_scan(temp, tb, s0, s1, (size_t)n, rocprim::plus<int64_t>(), stream, false);  // s1 = seg number
    if (he != hipSuccess) return fail(LCREC_EHIP, "collision_groups: scan: %s", hipGetErrorString(he));
    unsigned long long *sizes = reinterpret_cast<unsigned long long *>(ka);
    he = hipMemsetAsync(sizes, 0, (size_t)n * 8, stream);
    if (he != hipSuccess) return fail(LCREC_EHIP, "collision_groups: %s", hipGetErrorString(he));
    hipLaunchKernelGGL(seg_sizes_kernel, dim3(grid), dim3(256), 0, stream, s1, n, sizes);
    hipLaunchKernelGGL(member_flags_kernel, dim3(grid), dim3(256), 0, stream, s1, s0, sizes, n, s2, s3, counters);
    if (members_out) {
        int64_t *mpos = reinterpret_cast<int64_t *>(klo), *gpos = reinterpret_cast<int64_t *>(khi);
        tb = temp_bytes;
        he = rocprim::exclusive_scan(temp, tb, s2, mpos, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), stream, false);
        if (he == hipSuccess) {
            tb = temp_bytes;
            he = rocprim::exclusive_scan(temp, tb, s3, gpos, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), stream, false);
        }
        if (he != hipSuccess) return fail(LCREC_EHIP, "collision_groups: scan: %s", hipGetErrorString(he));
        hipLaunchKernelGGL(emit_groups_kernel, dim3(grid), dim3(256), 0, stream, ids2, s2, s3, mpos, gpos, n, members_out,
                           offsets_out, counters);
    }
    return check_launch("collision_groups kernels");
}

}  // namespace lcrec
