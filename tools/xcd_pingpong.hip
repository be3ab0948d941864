// What does a hand-over between two workgroups cost when both sit on the SAME XCD and talk through its L2 (loads and stores at
// "group" scope: sc0 -- miss the CU's L1, hit the XCD's L2) instead of at agent scope (sc1: performed at the memory side, because
// the eight L2s are not coherent with each other)?  Two single-wave workgroups bounce a counter; cycles per round trip by s_memtime.
//   hipcc --offload-arch=gfx950 -O2 tools/xcd_pingpong.hip -o tools/diag/xcd_pingpong && tools/diag/xcd_pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned long long u64;
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u; }   // HW_REG_XCC_ID[3:0]

// MODE = 4 * store bits + load bits; bits: 0 none, 1 sc0, 2 sc1, 3 sc0 sc1 (LLVM's gfx942 memory model: sc0 = workgroup, sc1 = agent,
// both = system scope).  MODE 16: the compiler's agent-scope atomics.
template <int MODE> __device__ __forceinline__ void put(u64 *p, u64 v)
{
    constexpr int S = MODE / 4;
    if (MODE == 16) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (S == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if (S == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (S == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int MODE> __device__ __forceinline__ u64 get(const u64 *p)
{
    constexpr int L = MODE % 4;
    if (MODE == 16) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 v;
    if (L == 0) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (L == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (L == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// slots[0]: initiator -> responder, slots[8]: responder -> initiator (separate 64-byte lines); out: {cycles, rounds done, xcc a, xcc b}
template <int MODE>
__global__ void pingpong(u64 *slots, int a, int b, int rounds, u64 *out, unsigned *xcc_of)
{
    if (threadIdx.x == 0) xcc_of[blockIdx.x] = xcc_id();
    if (threadIdx.x != 0 || ((int)blockIdx.x != a && (int)blockIdx.x != b)) return;
    const bool init = (int)blockIdx.x == a;
    u64 *mine = slots + (init ? 0 : 8), *theirs = slots + (init ? 8 : 0);
    u64 t0 = __builtin_readcyclecounter();
    int done = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (init) put<MODE>(mine, (u64)i);
        unsigned spins = 0;
        while (get<MODE>(theirs) != (u64)i) {
            if (++spins > 200000u) goto out_;       // never hang the GPU: report how far it got
        }
        if (!init) put<MODE>(mine, (u64)i);
        done = i;
    }
out_:
    u64 t1 = __builtin_readcyclecounter();
    if (init) { out[0] = t1 - t0; out[1] = (u64)done; }
    out[init ? 2 : 3] = xcc_id();
}

int main()
{
    const int G = 64, rounds = 2000;
    u64 *slots, *out;
    unsigned *xcc;
    hipMalloc(&slots, 4096); hipMalloc(&out, 64); hipMalloc(&xcc, G * 4);
    unsigned hx[G];
    u64 ho[4];
    // where do the workgroups of a launch land?
    hipMemset(slots, 0, 4096); hipMemset(out, 0, 64);
    hipLaunchKernelGGL(pingpong<0>, dim3(G), dim3(64), 0, 0, slots, -1, -1, 0, out, xcc);
    hipMemcpy(hx, xcc, G * 4, hipMemcpyDeviceToHost);
    printf("XCC_ID of workgroups 0..%d:", G - 1);
    for (int i = 0; i < G; ++i) printf(" %u", hx[i]);
    printf("\n");
    int same = -1, diff = -1;
    for (int i = 1; i < G && (same < 0 || diff < 0); ++i) {
        if (hx[i] == hx[0] && same < 0) same = i;
        if (hx[i] != hx[0] && diff < 0) diff = i;
    }
    printf("partner on the same XCD: workgroup %d; on another: %d\n", same, diff);
    const char *bits[] = {"none   ", "sc0    ", "sc1    ", "sc0 sc1"};
    for (int where = 0; where < 2; ++where) {
        const int partner = where ? diff : same;
        if (partner < 0) continue;
        for (int mode = 0; mode <= 16; ++mode) {
            hipMemset(slots, 0, 4096); hipMemset(out, 0, 64);
#define RUN(M) case M: hipLaunchKernelGGL(pingpong<M>, dim3(G), dim3(64), 0, 0, slots, 0, partner, rounds, out, xcc); break;
            switch (mode) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) }
            if (hipMemcpy(ho, out, 32, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
            printf("%s XCD, store %s load %s: %4llu of %d round trips, %5.0f cycles each (XCCs %llu / %llu)\n", where ? "other" : "same ",
                   mode == 16 ? "agent-scope atomic" : bits[mode / 4], mode == 16 ? "(compiler)" : bits[mode % 4], ho[1], rounds,
                   ho[1] ? (double)ho[0] / (double)ho[1] : 0.0, ho[2], ho[3]);
            fflush(stdout);
        }
    }
    return 0;
}
