// Cost of one Philox4x32-10 call per lane in its two spellings: v_mul_hi_u32 + v_mul_lo_u32 per product (what hipcc 7.2 emits for __umulhi + a
// 32-bit product) against one v_mad_u64_u32 per product (a 64-bit product).  Build: hipcc --offload-arch=gfx950 -O3 tools/philox_rate.hip -o tools/bin/philox_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <bool WIDE>
__device__ __forceinline__ void rounds(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        if (WIDE) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
            hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
        } else {
            hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0; hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
        }
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = blockIdx.x, a2 = seed, a3 = 1, b0 = a0 ^ 77u, b1 = a1, b2 = seed, b3 = 2;
    for (int i = 0; i < iters; ++i) {           // two independent calls per turn (as a 32 x 32 attention tile's lane draws)
        rounds<WIDE>(a0, a1, a2, a3, seed, i);
        rounds<WIDE>(b0, b1, b2, b3, seed, i);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3;
}

int main() {
    const int blocks = 256 * 8, iters = 2000;
    uint32_t* out;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    uint32_t h[2][4];
    for (int wide = 0; wide < 2; ++wide) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (wide) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, out, iters, 5u);
            else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, out, iters, 5u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: blocks * 4 waves * iters * 2 calls / (256 CUs * 4 SIMDs)
            const double calls_per_simd = (double)blocks * 4 * iters * 2 / (256.0 * 4);
            if (rep == 2) printf("%s: %.3f ms, %.1f ns per wave-call per SIMD (= %.0f cycles at 2.4 GHz)\n", wide ? "v_mad_u64_u32        " : "v_mul_hi + v_mul_lo  ",
                                 ms, ms * 1e6 / calls_per_simd, ms * 1e6 / calls_per_simd * 2.4);
        }
        hipMemcpy(h[wide], out, 16, hipMemcpyDeviceToHost);
    }
    printf("same values: %s\n", (h[0][0] == h[1][0] && h[0][3] == h[1][3]) ? "yes" : "NO");
    return 0;
}
