// Row-strip GEMM for the big ragged products of the item tower (rows x N; N > 256 runs as ceil(N / 256) column
// panels, partial fragments are guarded) on v_mfma_f32_16x16x4_f32.
//
// Why a second tiling: the token-row count of a batch (~26 k rows at the headline size) cut into
// 128 x 128 tiles gives ~410 blocks for 256 CUs -- 1.6 rounds, so a fifth of the machine idles in the
// second round.  Here the grid is ONE block per CU and every block owns a contiguous strip of
// ceil(M / #CU) rows (rounded to the 16-row MFMA granule, <= 128 rows per pass) across ALL N columns:
//   * 8 waves side by side over the columns (32 columns = two 16-wide fragments each), every wave
//     covers all NF row fragments of the strip -> accumulators NF x 2 x 4 floats,
//   * A strip [<=128][BK] and the whole B panel [256][BK] staged in LDS (KC images, +32 B pad, one
//     ds_read_b128 feeds four MFMAs through the k permutation k = 16q + 4g + j; MC image for the NN
//     products with a +4 float row pad so the four k groups of a wave land on distinct banks),
//   * register-staged double buffering with an explicit software pipeline (see strip_pass).
// Operand loaders and epilogue kinds are shared with gemm_core.hpp.
#pragma once
#include "gemm_core.hpp"

namespace lego {

constexpr int STRIP_BN = 256;
constexpr int STRIP_BM = 128;
constexpr int STRIP_MC_LD = STRIP_BN + 4;
constexpr int STRIP_THREADS = 512;
constexpr int STRIP_KC_LD = BK + 8;   // 40 floats: conflict-free ds_read_b128 for the 16-row x 4-k-group lane mapping (36 is 2-way)

template <bool B_MC>
constexpr size_t strip_lds_bytes() {
    return 2 * (size_t)(STRIP_BM * STRIP_KC_LD + (B_MC ? BK * STRIP_MC_LD : STRIP_BN * STRIP_KC_LD)) * sizeof(float);
}

// rows of strip partition shared by host (grid) and device
struct StripPlan { int s, sub, nf; };
__host__ __device__ inline StripPlan strip_plan(int M, int G) {
    int s = ((M + G - 1) / G + 15) & ~15;
    if (s < 16) s = 16;
    const int nsub = (s + STRIP_BM - 1) / STRIP_BM;
    const int sub = (((s + nsub - 1) / nsub) + 15) & ~15;
    return {s, sub, sub / 16};
}

template <int NF, bool B_MC, class ALoad, class BLoad, class Epi>
__device__ __forceinline__ void strip_pass(const ALoad& la0, const BLoad& lb0, Epi& epi, float* As0, float* Bs0,
                                           const typename BLoad::Row (&rb)[4], int m0, int m_end, int n0, int K) {
    constexpr int NT = STRIP_THREADS, BN = STRIP_BN;
    constexpr int A_FLOATS = STRIP_BM * STRIP_KC_LD;
    constexpr int B_FLOATS = B_MC ? BK * STRIP_MC_LD : BN * STRIP_KC_LD;
    constexpr int AN = (NF * 16 * 8 + NT - 1) / NT;        // float4 of the A strip per thread (1 or 2)
    ALoad la = la0;
    BLoad lb = lb0;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g4 = lane >> 4;

    typename ALoad::Row ra[AN];
#pragma unroll
    for (int j = 0; j < AN; ++j) ra[j] = la.row(m0 + (tid >> 3) + 64 * j);
    f32x4 sa[AN], sb[4];
    bool pa[AN], pb[4];
    const int k_last = (K - 1) / BK * BK;                  // fetches past the end re-read the last tile (never used)
    auto fetch = [&](int k0) {
        k0 = min(k0, k_last);
        la.tile(k0);
        lb.tile(k0);
#pragma unroll
        for (int j = 0; j < AN; ++j) { sa[j] = la.load(ra[j], k0 + (tid & 7) * 4); pa[j] = la.keep(ra[j], k0 + (tid & 7) * 4); }
        if constexpr (B_MC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sb[j] = lb.load(k0 + (tid >> 6) + 8 * j, n0 + (tid & 63) * 4, pb[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { sb[j] = lb.load(rb[j], k0 + (tid & 7) * 4); pb[j] = lb.keep(rb[j], k0 + (tid & 7) * 4); }
        }
    };
    auto commit = [&](float* A_, float* B_) {
#pragma unroll
        for (int j = 0; j < AN; ++j)      // the A image always has STRIP_BM rows: no guard
            *reinterpret_cast<f32x4*>(A_ + ((tid >> 3) + 64 * j) * STRIP_KC_LD + (tid & 7) * 4) = zero_unless(pa[j], sa[j]);
        if constexpr (B_MC) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(B_ + ((tid >> 6) + 8 * j) * STRIP_MC_LD + (tid & 63) * 4) = zero_unless(pb[j], sb[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(B_ + ((tid >> 3) + 64 * j) * STRIP_KC_LD + (tid & 7) * 4) = zero_unless(pb[j], sb[j]);
        }
    };

    f32x4 acc[NF][2];
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment registers of the two 16-wide k groups of a tile (F0: k 0..15, F1: k 16..31)
    f32x4 fa0[NF], fb0[2], fa1[NF], fb1[2];
    auto read_frags = [&](const float* A_, const float* B_, int q, f32x4 (&fa)[NF], f32x4 (&fb)[2]) {
#pragma unroll
        for (int a = 0; a < NF; ++a)
            fa[a] = *reinterpret_cast<const f32x4*>(A_ + (a * 16 + l16) * STRIP_KC_LD + 16 * q + 4 * g4);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = wave * 32 + b * 16 + l16;
            if constexpr (B_MC) {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[b][j] = B_[(16 * q + 4 * g4 + j) * STRIP_MC_LD + col];
            } else {
                fb[b] = *reinterpret_cast<const f32x4*>(B_ + col * STRIP_KC_LD + 16 * q + 4 * g4);
            }
        }
    };
    auto mfma_j = [&](const f32x4 (&fa)[NF], const f32x4 (&fb)[2], int j) {
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
    };

    // Software pipeline (one barrier per tile, placed where every wave has long since arrived):
    //   top of tile t: LDS buf[t&1] = tile t, F0 = its first k group (read at the end of tile t-1),
    //                  staging registers = tile t+1 (global loads issued during tile t-1)
    //   read F1(t) | MFMA F0 | commit tile t+1 -> buf[(t+1)&1] | fetch tile t+2 | MFMA F1 j=0..2 |
    //   barrier | read F0(t+1) | MFMA F1 j=3
    // buf[(t+1)&1] was last read as tile t-1, whose reads completed before the barrier of tile t-1.
    fetch(0);
    commit(As0, Bs0);
    fetch(BK);
    __syncthreads();
    read_frags(As0, Bs0, 0, fa0, fb0);
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const float* A_ = As0 + buf * A_FLOATS;
        const float* B_ = Bs0 + buf * B_FLOATS;
        float* An = As0 + (buf ^ 1) * A_FLOATS;
        float* Bn = Bs0 + (buf ^ 1) * B_FLOATS;
        read_frags(A_, B_, 1, fa1, fb1);
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma_j(fa0, fb0, j);
        // ONE scheduling pin per tile: left alone, the compiler lifts the zeroing selects of `commit` (and with them the
        // s_waitcnt vmcnt(0) on the staged global loads) to the TOP of the tile to free registers, which halves the
        // distance between a tile's loads and their first use and stalls every MFMA behind that wait
        __builtin_amdgcn_sched_barrier(0);
        commit(An, Bn);                     // unconditional: past the end this writes the unused buffer
        fetch(k0 + 2 * BK);
#pragma unroll
        for (int j = 0; j < 3; ++j) mfma_j(fa1, fb1, j);
        __syncthreads();
        read_frags(An, Bn, 0, fa0, fb0);
        mfma_j(fa1, fb1, 3);
        buf ^= 1;
    }
    __syncthreads();        // the next pass refills both buffers
    // lane holds column l16 x rows 4*g4 + {0..3} of each 16 x 16 fragment
    epi.template run16<NF>(acc, m0, m_end, n0 + wave * 32, l16, g4);
}

template <bool B_MC, class ALoad, class BLoad, class Epi>
__global__ __launch_bounds__(STRIP_THREADS) void strip_kernel(GemmDims dims, ALoad la, BLoad lb, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * STRIP_BM * STRIP_KC_LD;
    int M = dims.M;
    if (dims.m_dyn != nullptr) M = min(M, *dims.m_dyn);
    const int N = dims.N, K = dims.K;
    // N > 256: the grid is split into ceil(N / 256) column panels, each with its own set of row strips
    const int n_panels = (N + STRIP_BN - 1) / STRIP_BN;
    const int panel = blockIdx.x % n_panels;
    const int n0 = panel * STRIP_BN;
    const StripPlan sp = strip_plan(M, max((int)gridDim.x / n_panels, 1));
    const int strip0 = (blockIdx.x / n_panels) * sp.s;
    if (strip0 >= M) return;
    const int strip_end = min(M, strip0 + sp.s);
    epi.setup(M, N, 0);
    la.ext = M;
    la.K = K;
    lb.K = K;
    la.prepare(0);
    lb.prepare(0);
    typename BLoad::Row rb[4];
    if constexpr (!B_MC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[j] = lb.row(n0 + (threadIdx.x >> 3) + 64 * j);
    }
    LEGO_CLOCK_BEGIN
    for (int m0 = strip0; m0 < strip_end; m0 += sp.sub) {
        const int m_end = min(strip_end, m0 + sp.sub);
        const int nf = (m_end - m0 + 15) >> 4;              // block-uniform
        switch (nf) {
            case 1: case 2: strip_pass<2, B_MC>(la, lb, epi, As0, Bs0, rb, m0, m_end, n0, K); break;
            case 3: case 4: strip_pass<4, B_MC>(la, lb, epi, As0, Bs0, rb, m0, m_end, n0, K); break;
            case 5: case 6: strip_pass<6, B_MC>(la, lb, epi, As0, Bs0, rb, m0, m_end, n0, K); break;
            case 7: strip_pass<7, B_MC>(la, lb, epi, As0, Bs0, rb, m0, m_end, n0, K); break;
            default: strip_pass<8, B_MC>(la, lb, epi, As0, Bs0, rb, m0, m_end, n0, K); break;
        }
    }
    LEGO_CLOCK_END(1)
}

}  // namespace lego
