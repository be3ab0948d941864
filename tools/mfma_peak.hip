// Calibration: fp32 MFMA issue rate on this device by waves per SIMD, with and without LDS operand reads.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(float *out, int iters, float a0, float b0)
{
    __shared__ __attribute__((aligned(16))) float lds[128 * 36 * 2];
    for (int i = threadIdx.x; i < 128 * 36 * 2; i += blockDim.x) lds[i] = a0 + i * 1e-6f;
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *ab = lds + ((wave & 1) * 64 + (lane & 31)) * 36 + (lane >> 5) * 4;
    const float *wb = lds + 128 * 36 + (((wave >> 1) & 1) * 64 + (lane & 31)) * 36 + (lane >> 5) * 4;
    float a = a0 + lane, b = b0 + lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 af[2], wf[2];
                af[0] = *(const f32x4 *)(ab + g * 8);
                af[1] = *(const f32x4 *)(ab + 32 * 36 + g * 8);
                wf[0] = *(const f32x4 *)(wb + g * 8);
                wf[1] = *(const f32x4 *)(wb + 32 * 36 + g * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][q], wf[0][q], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][q], wf[1][q], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][q], wf[0][q], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][q], wf[1][q], acc[3], 0, 0, 0);
                }
            }
            if (MODE == 2) __syncthreads();
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char *name, int threads, int blocks_per_cu)
{
    float *out;
    hipMalloc(&out, 4);
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * (threads / 64) * iters * 64.0 * 4096.0;
    printf("%-28s threads %4d  blocks/CU %d : %7.1f TFLOP/s\n", name, threads, blocks_per_cu, flop / ms / 1e9);
    hipFree(out);
}

int main()
{
    run<0>("regs only", 256, 1); run<0>("regs only", 512, 1); run<0>("regs only", 256, 3);
    run<1>("lds operands", 256, 1); run<1>("lds operands", 512, 1); run<1>("lds operands", 256, 2); run<1>("lds operands", 256, 3);
    run<2>("lds operands + barrier", 256, 1); run<2>("lds operands + barrier", 512, 1); run<2>("lds operands + barrier", 256, 3);
    return 0;
}
