// Split-K TN products (weight gradients): C[M,N] (+)= sum_r A[r,m] * B[r,n], both operands M-contiguous in memory (the reduction
// index r is the memory row), on v_mfma_f32_32x32x2_f32.  Included by gemm_ops.hip after EpiArgs / gemm_wino.hpp.
//
// Why its own kernel (the generic tile kernel of gemm_core.hpp served these in round 1): a weight gradient has a SMALL output
// (256 x 256 .. 1024 x 256 floats) and a LONG ragged reduction (13 k pairs .. 27 k rows), so every workgroup owns one
// 128 x 128 output tile and a short k range (a few hundred rows = ~13 k tiles).  With so few tiles per workgroup the PMC pass
// showed the waves waiting on s_waitcnt for 41-55 % of their cycles (profiles/r01_pmc_mfma_lds.json): one k tile of MFMA
// work (0.85 us per wave) does not cover a global load issued at the top of that tile, least of all the Winograd pair
// operands, whose row address depends on a pair_info word that is itself a global load.  Here
//   * global loads run TWO k tiles ahead of their use (two staging register sets, loop unrolled by two),
//   * the pair_info words of the workgroup's whole k range are copied to LDS once, so an operand load never waits for another,
//   * one workgroup per CU (512 threads, 8 waves as 2 x 4, wave tile 64 x 32), 2 LDS stages of 32 KB.
// Partial tiles either go out as fp32 atomics (ATOMIC) or as plain stores into a slab [split][M][N] that one combine launch
// folds later (the conv's unpack kernel, lego_combine_slabs for the Linear weights).
#pragma once
#include "wino_common.hpp"

namespace lego {

constexpr int TN_BM = 128, TN_BN = 128, TN_THREADS = 512;     // the long-reduction configuration (2 x 4 waves of 64 x 32)
constexpr int TN_BM_S = 64, TN_BN_S = 64, TN_THREADS_S = 256;   // plain-row products: 2 x 2 waves of 32 x 32, four workgroups per CU
constexpr int TN_INFO_CAP = 2048;            // pair_info words cached in LDS
constexpr int TN_WINDOW = TN_INFO_CAP - BK;  // rows of a k range walked per refill of that cache (a multiple of BK)

struct TnDims {
    int M, N, K;                 // output M x N, static bound of the reduction length
    const int* k_dyn;            // device scalar overriding K
    int split_k, taps;           // gridDim.z = taps * split_k
    size_t slab_stride;          // SLAB: floats between the partial outputs of consecutive k splits (taps * tap_stride)
    int tiles_m, tiles_n;        // output tiles
    int deal;                    // 1: 1-D grid, the workgroups of one k split dealt to ONE XCD (split_k % 8 == 0)
    int min_chunk = 0;           // a k split covers at least this many reduction rows: when the dynamic K is far below the capacity
                                 // the split was sized for, the trailing splits are empty instead of every split getting a sliver
};

constexpr size_t tn_lds_bytes(int bm = TN_BM, int bn = TN_BN, bool info = true) {
    return (size_t)(2 * BK * (bm + bn) + (info ? TN_INFO_CAP : 0)) * sizeof(float);
}

template <class ALoad, class BLoad, bool SLAB, int WM_ = 2, int WN_ = 4, int TM = 2>
__global__ __launch_bounds__(WM_ * WN_ * 64) void tn_kernel(TnDims dims, ALoad la, BLoad lb, EpiArgs e) {
    constexpr int BM = WM_ * TM * 32, BN = WN_ * 32, NT = WM_ * WN_ * 64;
    static_assert(BM == BN, "square workgroup tiles (one staging pattern for both operands)");
    constexpr int STAGE = BK * BM;                       // floats per operand stage ([BK][BM])
    constexpr int PER = BM / 4, STEP = NT / PER;         // float4 per k row, k rows per pass of the workgroup (16)
    constexpr int NJ = BK / STEP;                        // 2 float4 per thread, operand and k tile
    constexpr bool A2 = IsDual<ALoad>::value, B2 = IsDual<BLoad>::value;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * STAGE;
    int* const s_info = reinterpret_cast<int*>(smem + 4 * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN_, wn = wave % WN_;          // WM_ x WN_ waves, wave tile (TM * 32) x 32

    int K = dims.K;
    if (dims.k_dyn != nullptr) K = min(K, *dims.k_dyn);
    const int M = dims.M, N = dims.N;
    // XCD-aware dealing: every workgroup of k split z reads the same rows of both operands (its tap / output tile only picks
    // columns and row combinations).  Dispatch is round-robin over the 8 XCDs, so in (x, y, z) grid order those workgroups land on
    // 8 different L2s.  Dealt, split z lives on XCD z % 8.  (Measured: FETCH_SIZE of the conv launch stayed at 332 MB either way
    // -- 430 MB are requested from L2, 16 workgroups x 2 KB per pair, against 54 MB of operands: with every split resident at once
    // an XCD's share of the operands, 6.7 MB, does not fit its 4 MB L2.  Kept: it costs nothing)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (dims.deal) {
        const int per = dims.tiles_m * dims.tiles_n * dims.taps;          // workgroups of one k split
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int zz = xcd + 8 * (j / per), inner = j % per;
        bx = inner % dims.tiles_m;
        by = (inner / dims.tiles_m) % dims.tiles_n;
        bz = (inner / (dims.tiles_m * dims.tiles_n)) * dims.split_k + zz;
    }
    const int m0 = bx * BM, n0 = by * BN;
    const int z = bz % dims.split_k, tap = bz / dims.split_k;
    int chunk = (K + dims.split_k - 1) / dims.split_k;
    chunk = max((chunk + BK - 1) / BK * BK, dims.min_chunk);
    const int kbeg = z * chunk, kend = min(K, kbeg + chunk);
    float* C = e.C + (size_t)tap * e.tap_stride + (SLAB ? (size_t)z * dims.slab_stride : 0);
    if (kbeg >= kend) {
        if constexpr (SLAB) {                            // an empty split still owns a slab tile: the combine reads every slab
            for (int i = tid; i < BM * (BN / 4); i += NT) {
                const int r = m0 + i / (BN / 4), c = n0 + (i % (BN / 4)) * 4;
                if (r < M && c < N) *reinterpret_cast<f32x4*>(C + (size_t)r * e.ldc + c) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        return;
    }
    la.prepare(tap); lb.prepare(tap);
    // The k range is walked in WINDOWS of at most TN_WINDOW rows: the pair_info words of a window are what the LDS cache holds.  One
    // window is the rule (the host sizes the split for the rows a batch usually has); a batch near the CAPACITY of the plan takes
    // several.  Round 2 first sized the split so that the capacity fitted one window: 27 splits instead of 16 for the headline shape --
    // 432 workgroups in two rounds on 256 CUs, no XCD dealing, 27 slabs, an L2 hit rate of 0.13 instead of 0.82.
    constexpr int WIN = (A2 || B2) ? TN_WINDOW : (1 << 30);      // plain operands need no cache: one window
    int wbeg = kbeg, wend = min(kend, kbeg + WIN);

    const int kr = tid / PER, c4 = (tid % PER) * 4;      // this thread's k row inside a pass and its 4 columns
    struct Regs { f32x4 a[NJ], b[NJ], a2[A2 ? NJ : 1], b2[B2 ? NJ : 1]; bool pa[NJ], pb[NJ], pa2[A2 ? NJ : 1], pb2[B2 ? NJ : 1]; };
    auto fetch = [&](Regs& r, int k0) {
        k0 = min(k0, wend - 1);                          // past the end: re-read the last rows (never committed to a used stage)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int kk = k0 + kr + STEP * j;
            if constexpr (A2) la.template load2<true>(kk, m0 + c4, r.a[j], r.pa[j], r.a2[j], r.pa2[j]); else r.a[j] = la.load(kk, m0 + c4, r.pa[j]);
            if constexpr (B2) lb.template load2<true>(kk, n0 + c4, r.b[j], r.pb[j], r.b2[j], r.pb2[j]); else r.b[j] = lb.load(kk, n0 + c4, r.pb[j]);
        }
    };
    auto commit = [&](const Regs& r, float* A_, float* B_) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 va, vb;
            if constexpr (A2) va = la.combine(r.a[j], r.pa[j], r.a2[j], r.pa2[j]); else va = zero_unless(r.pa[j], r.a[j]);
            if constexpr (B2) vb = lb.combine(r.b[j], r.pb[j], r.b2[j], r.pb2[j]); else vb = zero_unless(r.pb[j], r.b[j]);
            *reinterpret_cast<f32x4*>(A_ + (kr + STEP * j) * BM + c4) = va;
            *reinterpret_cast<f32x4*>(B_ + (kr + STEP * j) * BN + c4) = vb;
        }
    };

    f32x16 acc[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
    // fragments of k group q+1 are read while the MFMAs of group q run (two fragment register sets): read just in time, every
    // pair of MFMAs waited on its own ds_read (s_waitcnt lgkmcnt before each pair in the round-1 code)
    constexpr int NFR = 4 * (TM + 1);                    // fragment words of one k group: TM A fragments + one B fragment
    auto read_frags = [&](const float* A_, const float* B_, int q, float (&f)[NFR]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 8 * q + 4 * lh + j;            // lane half h takes k = 8q + 4h + j for MFMA j, A and B alike
#pragma unroll
            for (int a = 0; a < TM; ++a) f[4 * a + j] = A_[k * BM + (wm * TM + a) * 32 + li];
            f[4 * TM + j] = B_[k * BN + wn * 32 + li];
        }
    };
    auto mfma8 = [&](const float (&f)[NFR]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < TM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[4 * a + j], f[4 * TM + j], acc[a], 0, 0, 0);
    };
    auto tile_mfma = [&](const float* A_, const float* B_) {
        float f0[NFR], f1[NFR];
        read_frags(A_, B_, 0, f0);
        read_frags(A_, B_, 1, f1);
        mfma8(f0);
        read_frags(A_, B_, 2, f0);
        mfma8(f1);
        read_frags(A_, B_, 3, f1);
        mfma8(f0);
        mfma8(f1);
        // order for the scheduler: the ds_reads of a k group BEFORE the 8 MFMAs of the previous group, not one ds_read in
        // front of every MFMA (0x100 = DS read, 0x008 = MFMA; a ds_read2st64_b32 carries two of the twelve fragment words)
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + 1), 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + 1), 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * TM, 0);
    };

    // tile t lives in LDS stage t & 1; its global loads are issued at the top of iteration t - 2 and committed at the bottom of
    // iteration t - 1 (after that iteration's MFMAs), i.e. two tiles of matrix work cover every load
    Regs r0, r1;
    for (; wbeg < kend; wbeg = wend, wend = min(kend, wbeg + WIN)) {
        la.K = wend; lb.K = wend;                        // rows past the window read as zero; the prefetch past it is never used
        if constexpr (A2 || B2) {
            if (wbeg > kbeg) __syncthreads();            // the previous window's loads have left the cache
            const int* src = A2 ? la.info_src() : lb.info_src();
            for (int i = tid; i < wend - wbeg; i += NT) s_info[i] = src[wbeg + i];
            if constexpr (A2) la.cache(s_info, wbeg);
            if constexpr (B2) lb.cache(s_info, wbeg);
            __syncthreads();
        }
        fetch(r0, wbeg);
        commit(r0, As0, Bs0);
        fetch(r1, wbeg + BK);
        __syncthreads();
        const int nt = (wend - wbeg + BK - 1) / BK;
        for (int t = 0; t < nt; t += 2) {
            fetch(r0, wbeg + (t + 2) * BK);
            tile_mfma(As0, Bs0);
            commit(r1, As0 + STAGE, Bs0 + STAGE);        // tile t + 1
            __syncthreads();
            if (t + 1 >= nt) break;
            fetch(r1, wbeg + (t + 3) * BK);
            tile_mfma(As0 + STAGE, Bs0 + STAGE);
            commit(r0, As0, Bs0);                        // tile t + 2
            __syncthreads();
        }
    }

    // epilogue: lane holds column li x rows {(v & 3) + 8 * (v >> 2) + 4 * lh} of each 32 x 32 sub-tile
    const int col = n0 + wn * 32 + li;
    if (col < N) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m0 + (wm * TM + a) * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                if (row < M) {
                    float* dst = C + (size_t)row * e.ldc + col;
                    if constexpr (SLAB) *dst = acc[a][v]; else atomicAdd(dst, acc[a][v]);
                }
            }
    }
}

// out[i] += sum_s slabs[s * stride + i]  (i < n): the fold of a slab set, 16 B per lane
__global__ __launch_bounds__(256) void combine_slabs_kernel(const float* __restrict__ slabs, size_t stride, int S, float* __restrict__ out, int n4) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        f32x4 s = *reinterpret_cast<const f32x4*>(out + (size_t)i * 4);
        for (int k = 0; k < S; ++k) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)k * stride + (size_t)i * 4);
        *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = s;
    }
}

}  // namespace lego
