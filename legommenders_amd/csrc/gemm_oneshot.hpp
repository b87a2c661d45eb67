// The small, latency-bound row products of the path (user-side additive attention, the category column: a few thousand rows,
// K <= 1024): light_kernel below.  (Rounds 2-4 also had a "one-shot" 512-thread form with the whole A strip and B panel in LDS; on a
// side stream it could not START while a row-strip / Winograd product held the CU's LDS -- removed in round 6)
// Loaders and epilogue kinds are those of gemm_core.hpp / gemm_ops.hip.
#pragma once
#include "gemm_core.hpp"

namespace lego {

constexpr int ONE_BM = 64, ONE_BN = 64, ONE_THREADS = 512, ONE_MC_LD = ONE_BN + 16, ONE_KMAX = 256;

// ONE WAVE per 16 x 32 output tile, MFMA fragments read straight from global memory
// (the operands are L2-resident), no LDS, ~70 VGPRs.  A 512-thread workgroup with up to 135 KB of LDS
// on a side stream cannot START while a row-strip / Winograd product (122 KB of LDS, 2 x 212-255 of a SIMD's 512 registers) holds
// the CU -- a rocprofv3 timeline of the NAML step shows two 11 us launches taking 90 and 69 us of the side stream, which then ends
// after the main stream.  A single wave with a few registers fits beside anything.
template <bool B_MC, class ALoad, class BLoad, class Epi, int U = 2>
__global__ __launch_bounds__(64) void light_kernel(GemmDims dims, ALoad la, BLoad lb, Epi epi) {
    int M = dims.M;
    if (dims.m_dyn != nullptr) M = min(M, *dims.m_dyn);
    const int N = dims.N, K = dims.K;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 32;
    if (m0 >= M) return;
    const int lane = threadIdx.x, l16 = lane & 15, g4 = lane >> 4;
    epi.setup(M, N, 0);
    la.ext = M;
    la.K = K;
    lb.K = K;
    la.prepare(0);
    lb.prepare(0);
    la.tile(0);
    lb.tile(0);
    const typename ALoad::Row ra = la.row(m0 + l16);
    f32x4 acc[1][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
    // U: k groups of 16 whose loads are in flight together
    for (int k0 = 0; k0 < K; k0 += 16 * U) {
        f32x4 fa[U], fb[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 16 * u + 4 * g4;              // this lane's 4 reduction indices of the group
            fa[u] = la.load(ra, k);                          // (the loader clamps k to K - 4)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = n0 + 16 * b + l16;
                if constexpr (B_MC) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[u][b][j] = lb.p[(size_t)min(k + j, K - 1) * lb.ld + min(col, N - 1)];
                } else {
                    fb[u][b] = lb.load(lb.row(col), k);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                   // every load of the batch is issued before the first MFMA waits
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float in = k0 + 16 * u + 4 * g4 < K ? 1.f : 0.f;       // K % 4 == 0: a lane's four indices are in or out together
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[0][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][j] * in, fb[u][b][j], acc[0][b], 0, 0, 0);
        }
    }
    epi.template run16<1>(acc, m0, min(M, m0 + 16), n0, l16, g4);
}

}  // namespace lego
