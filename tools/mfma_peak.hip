#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 4; ++v) acc[i][v] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int v = 0; v < 4; ++v) s += acc[i][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        int grid = 256 * blocks_per_cu, iters = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)grid * 4 * iters * 16 * 32 * 32 * 2 * 2;
            printf("32x32x2 4acc %d blk/CU: %.3f ms %.1f TF\n", blocks_per_cu, ms, fl / ms / 1e9);
            hipEventRecord(e0); hipLaunchKernelGGL(k16<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            fl = (double)grid * 4 * iters * 32 * 16 * 16 * 4 * 2;
            printf("16x16x4 8acc %d blk/CU: %.3f ms %.1f TF\n", blocks_per_cu, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
