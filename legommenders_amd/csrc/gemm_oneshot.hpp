// One-shot GEMM for the small, latency-bound row products of the path (user-side additive attention, the
// category column: a few thousand rows, K <= 256; longer reductions run as 256-wide chunks of the same scheme).  A pipelined k loop spends one global-memory latency per
// k tile there (8 tiles x ~1.5 us for almost no MFMA work); here a workgroup issues ALL loads of its 64 x K
// A strip and K x 64 B panel at once, waits once, and runs the whole reduction from LDS.
//   * 512 threads = 8 waves as 4 (rows) x 2 (columns), wave tile 16 x 32 on v_mfma_f32_16x16x4_f32
//   * A image [64][ldk] (k contiguous, ldk = ceil64(K) + 8: conflict-free ds_read_b128 for the 16-row x 4-k-group
//     lane mapping), B image the same (KC) or [K][80] (MC, the NN products)
// Loaders and epilogue kinds are those of gemm_core.hpp / gemm_ops.hip.
#pragma once
#include "gemm_core.hpp"

namespace lego {

constexpr int ONE_BM = 64, ONE_BN = 64, ONE_THREADS = 512, ONE_MC_LD = ONE_BN + 16, ONE_KMAX = 256;

__host__ __device__ inline int one_ldk(int K) { return ((K + 63) & ~63) + 8; }
template <bool B_MC>
inline size_t oneshot_lds_bytes(int K) {
    const int kp = (K + 15) & ~15;
    return (size_t)(ONE_BM * one_ldk(K) + (B_MC ? kp * ONE_MC_LD : ONE_BN * one_ldk(K))) * sizeof(float);
}

template <bool B_MC, class ALoad, class BLoad, class Epi>
__global__ __launch_bounds__(ONE_THREADS) void oneshot_kernel(GemmDims dims, ALoad la, BLoad lb, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int M = dims.M;
    if (dims.m_dyn != nullptr) M = min(M, *dims.m_dyn);
    const int N = dims.N, K = dims.K;
    const int m0 = blockIdx.x * ONE_BM, n0 = blockIdx.y * ONE_BN;
    if (m0 >= M) return;
    // K > ONE_KMAX (the NRMS user side's in-projection data gradient reduces over 3 D = 768): the same one-shot scheme per
    // 256-wide chunk of the reduction -- 3 round trips instead of the 24 dependent k tiles of the pipelined tile kernel (37.7 us
    // for 0.5 GFLOP on 42 workgroups)
    const int Kc_max = min(K, ONE_KMAX);
    const int ldk = one_ldk(Kc_max);
    float* const As = smem;
    float* const Bs = smem + ONE_BM * ldk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    epi.setup(M, N, 0);
    la.ext = M;
    la.K = K;
    lb.K = K;
    la.prepare(0);
    lb.prepare(0);
    la.tile(0);
    lb.tile(0);
    constexpr int QMAX = ONE_KMAX / 4;                  // float4 per row and chunk
    const int r = tid >> 3;
    const typename ALoad::Row ra = la.row(m0 + r);
    f32x4 acc[1][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
    const float* arow = As + (wm * 16 + l16) * ldk + 4 * g4;

    for (int kc = 0; kc < K; kc += ONE_KMAX) {
        const int Kc = min(ONE_KMAX, K - kc);
        const int kp = (Kc + 15) & ~15;                 // chunk length padded to the MFMA k group (zero filled)
        // ---- every load of the chunk in flight, then one wait
        {
            // A: 64 rows x kp / 4 quads over 512 threads: thread -> (row = tid / 8, quad = tid % 8 + 8 * j)
            f32x4 va[QMAX / 8];
            bool pa[QMAX / 8];
#pragma unroll
            for (int j = 0; j < QMAX / 8; ++j) {
                const int k = ((tid & 7) + 8 * j) * 4;
                pa[j] = k < kp && la.keep(ra, kc + k);
                va[j] = la.load(ra, min(kc + k, K - 4));
            }
            f32x4 vb[QMAX / 8];
            bool pb[QMAX / 8];
            if constexpr (B_MC) {
                // B: Kc rows x 16 quads (64 columns): thread -> (k = tid / 16 + 32 * j, quad = tid % 16)
#pragma unroll
                for (int j = 0; j < QMAX / 8; ++j) {
                    const int kk = (tid >> 4) + 32 * j;
                    vb[j] = lb.load(min(kc + kk, K - 1), n0 + (tid & 15) * 4, pb[j]);
                    pb[j] = pb[j] && kk < Kc;
                }
            } else {
                const typename BLoad::Row rb = lb.row(n0 + r);
#pragma unroll
                for (int j = 0; j < QMAX / 8; ++j) {
                    const int k = ((tid & 7) + 8 * j) * 4;
                    pb[j] = k < kp && lb.keep(rb, kc + k);
                    vb[j] = lb.load(rb, min(kc + k, K - 4));
                }
            }
            if (kc > 0) __syncthreads();                // the previous chunk's MFMA reads are done
#pragma unroll
            for (int j = 0; j < QMAX / 8; ++j) {
                const int k = ((tid & 7) + 8 * j) * 4;
                if (k < kp) *reinterpret_cast<f32x4*>(As + r * ldk + k) = zero_unless(pa[j], va[j]);
            }
            if constexpr (B_MC) {
#pragma unroll
                for (int j = 0; j < QMAX / 8; ++j) {
                    const int kk = (tid >> 4) + 32 * j;
                    if (kk < kp) *reinterpret_cast<f32x4*>(Bs + kk * ONE_MC_LD + (tid & 15) * 4) = zero_unless(pb[j], vb[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < QMAX / 8; ++j) {
                    const int k = ((tid & 7) + 8 * j) * 4;
                    if (k < kp) *reinterpret_cast<f32x4*>(Bs + r * ldk + k) = zero_unless(pb[j], vb[j]);
                }
            }
        }
        __syncthreads();

        for (int k0 = 0; k0 < kp; k0 += 16) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(arow + k0);
            f32x4 fb[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = wn * 32 + b * 16 + l16;
                if constexpr (B_MC) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[b][j] = Bs[(k0 + 4 * g4 + j) * ONE_MC_LD + col];
                } else {
                    fb[b] = *reinterpret_cast<const f32x4*>(Bs + col * ldk + k0 + 4 * g4);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[0][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[b][j], acc[0][b], 0, 0, 0);
        }
    }
    epi.template run16<1>(acc, m0 + wm * 16, min(M, m0 + ONE_BM), n0 + wn * 32, l16, g4);
}

// "Light" form of the same small products: ONE WAVE per 16 x 32 output tile, MFMA fragments read straight from global memory
// (the operands are L2-resident), no LDS, ~70 VGPRs.  The one-shot kernel above is a 512-thread workgroup with up to 135 KB of LDS:
// on a side stream it cannot START while a row-strip / Winograd product (122 KB of LDS, 2 x 212-255 of a SIMD's 512 registers) holds
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
