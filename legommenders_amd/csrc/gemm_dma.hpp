// Row-strip GEMM with LDS-DMA operand staging (global_load_lds_dwordx4) for the plain-row products of the path
// (rows x N, N > 256 as column panels): the same work split and MFMA micro-tile as gemm_strip.hpp -- one workgroup per CU,
// a strip of ceil(M / #CU) rows x 256 columns, 8 waves side by side over the columns, v_mfma_f32_16x16x4_f32 -- with the
// register staging of its k loop replaced:
//
//   * both operand tiles go global -> LDS by LDS-DMA: no staging VGPRs, no zeroing selects, no ds_write pass, nothing in the
//     k loop waits on a global load result.  Six glds per wave and k tile (A: 128 rows x 128 B = 2, B: 256 x 128 B = 4).
//   * THREE LDS stages (3 x 48 KB): tile t+2 is issued at the top of tile t, so two tiles are in flight across the one
//     barrier of a tile -- raw s_barrier behind a COUNTED s_waitcnt vmcnt (a __syncthreads() would drain the DMAs:
//     cdna_hip_programming.md section 5, "Pipelining across barriers").  gemm_strip.hpp could hold only ONE tile ahead in
//     registers (two staging sets spilled), which at the path's shapes is less than the latency of a row that comes from HBM.
//   * a DMA wave-instruction writes 1 KB of LDS linearly, so the KC images are unpadded [row][32 floats] and the bank
//     spread comes from an XOR swizzle applied on BOTH sides: lane (row r, slot s) of a glds fetches the row's 16-B chunk
//     s ^ (r & 7), a fragment read of chunk c of row r goes to slot c ^ (r & 7) (conflict-free for the 16-lane groups of
//     ds_read_b128: tools/lds_swizzle_check.py).  The MC image of an NN product's weight panel is [k][256 + 4] floats: one
//     glds per k row, rows may be padded because no instruction's bytes cross a row.
//   * K % 32 != 0 (the 300-wide GloVe rows): the DMA cannot zero, so the last tile's sources are clamped into the row and
//     the B fragments of the k chunks past K are zeroed after the LDS read (branch-free selects; the clamped A values they
//     meet are real elements of the same row).  Rows past the operand's extent are clamped, never zeroed: they only feed outputs the epilogue
//     does not store.
// Epilogue: gemm_ops.hip's EpiT kinds in their row-major form (run16_rows: the accumulators go through wave-private LDS tiles
// and every global access of the epilogue covers whole 128-B lines); run16 when that form's alignment conditions fail.
#pragma once
#include "gemm_strip.hpp"

namespace lego {

#ifdef LEGO_TUNING_HOOKS
#define LEGO_DMA_ABL_OF(d) ((d).abl)
#else
#define LEGO_DMA_ABL_OF(d) 0          // the product kernel has no ablation branches
#endif

#ifndef DMA_PIN
#define DMA_PIN 3
#endif
constexpr int DMA_BM = 112;
constexpr int DMA_STAGES = 3;
constexpr int DMA_ROW_BYTES = BK * 4;                                  // 128: one KC image row = 8 chunks of 16 B
constexpr int DMA_A_BYTES = STRIP_BM * DMA_ROW_BYTES;                  // 16 KB
constexpr int DMA_MC_LD = STRIP_BN + 4;                                // floats per k row of the MC image
template <bool B_MC> constexpr int dma_b_bytes() { return B_MC ? BK * DMA_MC_LD * 4 : STRIP_BN * DMA_ROW_BYTES; }
template <bool B_MC> constexpr int dma_stage_bytes() { return DMA_A_BYTES + dma_b_bytes<B_MC>(); }
template <bool B_MC> constexpr size_t dma_lds_bytes() { return (size_t)DMA_STAGES * dma_stage_bytes<B_MC>(); }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ void glds16(const char* src, char* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)lds_dst_uniform, 16, 0, 0);
}

// operands: A = plain K-contiguous rows (KcRows); B = KcRows (NT) or McRows (NN)
// NW waves side by side over the 256 columns, CF = 16 / NW column fragments each: 8 x 2 (two waves per SIMD, 256 VGPRs each) or
// 4 x 4 (ONE wave per SIMD with up to 512 registers: half the LDS fragment reads per MFMA, a 4-wave barrier, no partner wave to
// arbitrate the matrix pipe with -- and none to hide a stall behind)
template <int NF, bool B_MC, int NW, class BLoad, class Epi>
__device__ __forceinline__ void dma_pass(const KcRows& la, const BLoad& lb, Epi& epi, char* lds, int m0, int m_end, int n0, int K, int abl = 0) {
    // abl (tuning build, LEGO_DMA_ABL): 1 no DMA in the loop, 4 no wait / barrier in the loop, 8 no epilogue, 32 no MFMAs -- timing only
    constexpr int STAGE = dma_stage_bytes<B_MC>();
    constexpr int CF = 16 / NW, AI = 16 / NW, BI = 32 / NW;     // column fragments per wave; glds per wave and tile for A / B
    static_assert(NW == 4 || NW == 8, "8 x 2 or 4 x 4 waves x column fragments");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g4 = lane >> 4;
    const int T = (K + BK - 1) / BK;
    const int rem = K - (T - 1) * BK;                        // k extent of the last tile: 4 .. 32

    // ---- DMA sources.  KC images: lane = (row lane >> 3 of the instruction's 8 rows, slot lane & 7); chunk = slot ^ (row & 7)
    const int chunk = (lane & 7) ^ (lane >> 3);              // every instruction starts at a row that is a multiple of 8
    const char* pa[AI];
#pragma unroll
    for (int j = 0; j < AI; ++j)
        pa[j] = reinterpret_cast<const char*>(la.p + (size_t)min(m0 + 8 * (AI * wave + j) + (lane >> 3), la.ext - 1) * la.ld);
    const char* pb[BI];
    if constexpr (B_MC) {                                    // instruction j of the wave = k row BI * wave + j, lane = 4 columns
#pragma unroll
        for (int j = 0; j < BI; ++j) pb[j] = reinterpret_cast<const char*>(lb.p + min(n0 + 4 * lane, lb.ext - 4));
    } else {
#pragma unroll
        for (int j = 0; j < BI; ++j)
            pb[j] = reinterpret_cast<const char*>(lb.p + (size_t)min(n0 + 8 * (BI * wave + j) + (lane >> 3), lb.ext - 1) * lb.ld);
    }
    auto issue = [&](int t, int stage) {
        char* sA = lds + stage * STAGE;
        char* sB = sA + DMA_A_BYTES;
        const int kc = min(t * BK + 4 * chunk, K - 4) * 4;   // byte offset of this lane's chunk inside its row (clamped: tail tile)
#pragma unroll
        for (int j = 0; j < AI; ++j) glds16(pa[j] + kc, sA + 8 * (AI * wave + j) * DMA_ROW_BYTES);
        if constexpr (B_MC) {
#pragma unroll
            for (int j = 0; j < BI; ++j)
                glds16(pb[j] + (size_t)min(t * BK + BI * wave + j, K - 1) * lb.ld * 4, sB + (BI * wave + j) * (DMA_MC_LD * 4));
        } else {
#pragma unroll
            for (int j = 0; j < BI; ++j) glds16(pb[j] + kc, sB + 8 * (BI * wave + j) * DMA_ROW_BYTES);
        }
    };

    // ---- fragment reads: byte offsets inside a stage for k group q (chunk 4q + g4 of the lane's row, swizzled)
    int offA[2], offB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        offA[q] = l16 * DMA_ROW_BYTES + (((4 * q + g4) ^ (l16 & 7)) << 4);
        if constexpr (B_MC) offB[q] = DMA_A_BYTES + ((16 * q + 4 * g4) * DMA_MC_LD + wave * (16 * CF) + l16) * 4;
        else offB[q] = DMA_A_BYTES + (wave * (16 * CF) + l16) * DMA_ROW_BYTES + (((4 * q + g4) ^ (l16 & 7)) << 4);
    }
    auto read_frags = [&](int stage, int q, f32x4 (&fa)[NF], f32x4 (&fb)[CF]) {
        const char* sA = lds + stage * STAGE + offA[q];
        const char* sB = lds + stage * STAGE + offB[q];
#pragma unroll
        for (int a = 0; a < NF; ++a) fa[a] = *reinterpret_cast<const f32x4*>(sA + a * 16 * DMA_ROW_BYTES);
#pragma unroll
        for (int b = 0; b < CF; ++b) {
            if constexpr (B_MC) {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[b][j] = *reinterpret_cast<const float*>(sB + (j * DMA_MC_LD + b * 16) * 4);
            } else {
                fb[b] = *reinterpret_cast<const f32x4*>(sB + b * 16 * DMA_ROW_BYTES);
            }
        }
    };

    f32x4 acc[NF][CF];
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < CF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 fa0[NF], fb0[CF], fa1[NF], fb1[CF];
    auto mfma_j = [&](const f32x4 (&fa)[NF], const f32x4 (&fb)[CF], int j) {
#ifdef LEGO_TUNING_HOOKS
        if (abl & 32) {
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int b = 0; b < CF; ++b) asm volatile("" : "+v"(acc[a][b]) : "v"(fa[a][j]), "v"(fb[b][j]));
            return;
        }
#endif
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < CF; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
    };

    // ---- pipeline.  Top of tile t: F0(t) in registers, tile t+1 issued.
    //   issue t+2 -> stage (t+2) % 3 (= the stage of tile t-1: every wave's reads of it completed before the barrier of t-1) |
    //   read F1(t) | MFMA F0 | MFMA F1 j=0..2 | vmcnt: tile t+1 landed (t+2 stays in flight) | barrier | read F0(t+1) | MFMA F1 j=3
    // (a previous pass's epilogue stores may still be draining: vmcnt retires in issue order on gfx9-family parts -- LLVM's own
    // counted waits rely on it -- so older stores only make the counted waits below conservative, and they drain under this
    // pass's MFMAs)
    auto wait_one_in_flight = [&]() {                        // the tile issued last stays in flight: AI + BI glds of this wave
        if constexpr (NW == 8) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    };
    issue(0, 0);
    if (T > 1) {
        issue(1, 1);
        wait_one_in_flight();
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    read_frags(0, 0, fa0, fb0);
    int sc = 0, sn = 2;                                     // stage of tile t, stage of tile t + 2
    auto body = [&](int t) {
        if (t + 2 < T && !(abl & 1)) issue(t + 2, sn);
        read_frags(sc, 1, fa1, fb1);
        if (DMA_PIN & 1) __builtin_amdgcn_sched_barrier(0);   // F1's reads go out BEFORE the F0 MFMAs (left alone, hipcc sinks them
                                                               // behind 38 MFMAs and then waits lgkmcnt(0) right behind each group)
        {   // k chunks past K (last tile of a K % 32 != 0 product): the B side becomes an exact zero.  Branch-free and in every
            // tile (16 v_cndmask beside 112 MFMAs): a peeled tail copy of the body cost 48 VGPRs
            const int kmax = t + 1 < T ? BK : rem;
            const bool v0 = 4 * g4 < kmax, v1 = 16 + 4 * g4 < kmax;
#pragma unroll
            for (int b = 0; b < CF; ++b) { fb0[b] = zero_unless(v0, fb0[b]); fb1[b] = zero_unless(v1, fb1[b]); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma_j(fa0, fb0, j);
#pragma unroll
        for (int j = 0; j < 3; ++j) mfma_j(fa1, fb1, j);
        if (t + 1 < T) {
            if (abl & 5) { if (!(abl & 4)) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }
            else if (t + 2 < T) wait_one_in_flight();
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            const int s1 = sc == 2 ? 0 : sc + 1;
            read_frags(s1, 0, fa0, fb0);
            if (DMA_PIN & 2) __builtin_amdgcn_sched_barrier(0);   // ... and F0(t+1)'s before the last 14 MFMAs of tile t
        }
        mfma_j(fa1, fb1, 3);
        sn = sc;
        sc = sc == 2 ? 0 : sc + 1;
    };
    for (int t = 0; t < T; ++t) body(t);
    asm volatile("s_barrier" ::: "memory");                // every DMA has landed (vmcnt(0) above) and every wave has read its last
                                                            // fragments: the stages are free for the epilogue's row tiles
    if (abl & 8) {
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < CF; ++b) asm volatile("" :: "v"(acc[a][b]));
        asm volatile("s_barrier" ::: "memory");
        return;
    }
    const bool rows_form = epi.rows_form_ok();
    float* tile = reinterpret_cast<float*>(lds) + wave * (NF * 16 * 32);
    auto slab = [&](auto hh) {                               // one 32-column slab of the wave's columns
        constexpr int h = decltype(hh)::value;
        f32x4 part[NF][2];
#pragma unroll
        for (int a = 0; a < NF; ++a) { part[a][0] = acc[a][2 * h]; part[a][1] = acc[a][2 * h + 1]; }
        const int nb = n0 + wave * (16 * CF) + 32 * h;
        if (rows_form) epi.template run16_rows<NF>(part, m0, m_end, nb, l16, g4, tile);
        else epi.template run16<NF>(part, m0, m_end, nb, l16, g4);
    };
    slab(std::integral_constant<int, 0>{});
    if constexpr (CF == 4) slab(std::integral_constant<int, 1>{});
    if (rows_form) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // a next pass refills the stages
}

// Passes of at most DMA_BM = 112 rows (7 row fragments), not 128: with 8 fragments the kernel needs all 256 VGPRs that 512
// threads allow, two of its waves fill a SIMD's 512 registers, and the side streams' single-wave products (gemm_oneshot.hpp
// light_kernel, fold_ops.hip: 40-72 VGPRs, no LDS) can not START beside it -- the step was 25 us LONGER than with the
// register-staged strip kernel although every product that uses this kernel was shorter.
template <bool B_MC, int NW, class BLoad, class Epi>
__global__ __launch_bounds__(NW * 64) void dma_strip_kernel(GemmDims dims, KcRows la, BLoad lb, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    int M = dims.M;
    if (dims.m_dyn != nullptr) M = min(M, *dims.m_dyn);
    const int N = dims.N, K = dims.K;
    const int n_panels = (N + STRIP_BN - 1) / STRIP_BN;
    const int panel = blockIdx.x % n_panels;
    const int n0 = panel * STRIP_BN;
    StripPlan sp = strip_plan(M, max((int)gridDim.x / n_panels, 1));
    const int nsub = (sp.s + DMA_BM - 1) / DMA_BM;
    sp.sub = (((sp.s + nsub - 1) / nsub) + 15) & ~15;
    const int strip0 = (blockIdx.x / n_panels) * sp.s;
    if (strip0 >= M) return;
    const int strip_end = min(M, strip0 + sp.s);
    epi.setup(M, N, 0);
    la.ext = M;
    la.K = K;
    lb.K = K;
    la.prepare(0);
    lb.prepare(0);
    for (int m0 = strip0; m0 < strip_end; m0 += sp.sub) {
        const int m_end = min(strip_end, m0 + sp.sub);
        const int nf = (m_end - m0 + 15) >> 4;              // block-uniform
        switch (nf) {
            case 1: case 2: dma_pass<2, B_MC, NW>(la, lb, epi, lds, m0, m_end, n0, K, LEGO_DMA_ABL_OF(dims)); break;
            case 3: case 4: dma_pass<4, B_MC, NW>(la, lb, epi, lds, m0, m_end, n0, K, LEGO_DMA_ABL_OF(dims)); break;
            case 5: case 6: dma_pass<6, B_MC, NW>(la, lb, epi, lds, m0, m_end, n0, K, LEGO_DMA_ABL_OF(dims)); break;
            default: dma_pass<7, B_MC, NW>(la, lb, epi, lds, m0, m_end, n0, K, LEGO_DMA_ABL_OF(dims)); break;
        }
    }
}

}  // namespace lego
