// Does a run of DEPENDENT fp32 matrix instructions (same accumulator, back to back) cost matrix-pipe time against the same instructions dealt
// round-robin over independent accumulators?  v_mfma_f32_16x16x4_f32 (8 passes) and v_mfma_f32_32x32x2_f32 (16 passes), 6 accumulators, runs of
// 1 (round-robin) / 2 / 4 / 8 on one accumulator before moving on; 1, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_chain.hip -o tools/bin/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int RUN>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) for (int v = 0; v < 4; ++v) acc[i][v] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8 / RUN; ++u)
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int r = 0; r < RUN; ++r) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 6; ++i) for (int v = 0; v < 4; ++v) s += acc[i][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int RUN>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8 / RUN; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < RUN; ++r) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kern, float* out, int wgs_per_cu, double flops_per_wave_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu, iters = 2000;
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%s %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", name, wgs_per_cu, ms, (double)grid * 4 * iters * flops_per_wave_iter / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 1 << 24);
    const double f16 = 48.0 * 16 * 16 * 4 * 2, f32 = 32.0 * 32 * 32 * 2 * 2;
    for (int w = 1; w <= 4; w *= 2) {
        run("16x16x4 run 1 (round-robin)", k16<1>, out, w, f16);
        run("16x16x4 run 2              ", k16<2>, out, w, f16);
        run("16x16x4 run 4              ", k16<4>, out, w, f16);
        run("16x16x4 run 8              ", k16<8>, out, w, f16);
        run("32x32x2 run 1 (round-robin)", k32<1>, out, w, f32);
        run("32x32x2 run 4              ", k32<4>, out, w, f32);
        run("32x32x2 run 8              ", k32<8>, out, w, f32);
    }
    return 0;
}
