// Does v_mfma_f32_16x16x4_f32 accumulate its four k-products as ONE sequential fp32 fma chain (k = 0, 1, 2, 3 onto C), the
// way v_mfma_f32_32x32x2_f32 does its two?  If so a 16 x 16 tile reproduces the library's arithmetic contract bit for bit with
// a dependent chain of 40 cycles per 4 k instead of 64 per 2 -- a 3.2x shorter serial chain for latency-bound (few-tile) GEMMs.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma16_probe.hip -o tools/diag/mfma16_probe && tools/diag/mfma16_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float *A, const float *B, float *D, int K)   // A [16][K], B [K][16], D [16][16]
{
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float a = A[(l % 16) * K + k0 + l / 16];
        const float b = B[(k0 + l / 16) * 16 + l % 16];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = acc[r];
}

int main()
{
    const int K = 512;
    float *hA = (float *)malloc(16 * K * 4), *hB = (float *)malloc(K * 16 * 4), hD[256], ref[4][256];
    srand(7);
    for (int i = 0; i < 16 * K; ++i) { hA[i] = (float)rand() / RAND_MAX * 2 - 1; hB[i] = (float)rand() / RAND_MAX * 2 - 1; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, 16 * K * 4); hipMalloc(&dB, K * 16 * 4); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, hA, 16 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, K * 16 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    if (hipMemcpy(hD, dD, 256 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    // candidate orders: 0 = one chain k ascending; 1 = pairs (k0,k1),(k2,k3) chained; 2 = tree ((p0+p1)+(p2+p3)) + C with exact products
    // rounded once; 3 = chain k descending inside each group of four
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float c0 = 0.f, c1 = 0.f, c3 = 0.f;
            double c2 = 0.0;
            for (int k0 = 0; k0 < K; k0 += 4) {
                for (int t = 0; t < 4; ++t) c0 = fmaf(hA[i * K + k0 + t], hB[(k0 + t) * 16 + j], c0);
                const float p01 = fmaf(hA[i * K + k0 + 1], hB[(k0 + 1) * 16 + j], hA[i * K + k0] * hB[k0 * 16 + j]);
                const float p23 = fmaf(hA[i * K + k0 + 3], hB[(k0 + 3) * 16 + j], hA[i * K + k0 + 2] * hB[(k0 + 2) * 16 + j]);
                c1 = (c1 + p01) + p23;
                double s = (double)c2;
                for (int t = 0; t < 4; ++t) s += (double)hA[i * K + k0 + t] * (double)hB[(k0 + t) * 16 + j];
                c2 = (double)(float)s;
                for (int t = 3; t >= 0; --t) c3 = fmaf(hA[i * K + k0 + t], hB[(k0 + t) * 16 + j], c3);
            }
            ref[0][i * 16 + j] = c0; ref[1][i * 16 + j] = c1; ref[2][i * 16 + j] = (float)c2; ref[3][i * 16 + j] = c3;
        }
    const char *names[4] = {"sequential fma chain, k ascending", "pairwise", "exact sum of four, rounded once", "chain, k descending in each four"};
    for (int c = 0; c < 4; ++c) {
        int same = 0;
        for (int q = 0; q < 256; ++q) same += memcmp(&hD[q], &ref[c][q], 4) == 0;
        printf("%-40s %3d / 256 outputs bit-identical\n", names[c], same);
    }
    return 0;
}
