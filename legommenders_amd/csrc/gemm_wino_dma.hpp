// Winograd F(2,3) conv over row pairs (gemm_wino.hpp) with LDS-DMA operand staging: the work split, the four transformed weight
// sets, the accumulators and the epilogue of wino_kernel<false>, on the three-stage glds pipeline of gemm_dma.hpp.
//
// What changes against the register-staged kernel:
//   * the A tile of a virtual k tile (set s, channels 32 kt ..) is no longer the two-row COMBINATION formed in staging registers
//     (two clamped global loads + two zeroing selects + an add per 16-B piece, 255 VGPRs with spills): the RAW input rows of the
//     pair strip go global -> LDS by DMA, and a fragment is formed where it is read -- two ds_read_b128 (rows r + ra, r + rb of
//     the lane's pair) and one packed add / subtract.  Rows outside the item read a zero row of the stage instead.
//   * raw rows are laid out in two PARITY PLANES (logical row i -> plane i & 1, index i >> 1): the pairs of a strip sit on every
//     second row, so a plain [row][32] image would put the 16 lanes of a fragment read on rows of one parity = half the banks;
//     inside a plane they are consecutive rows and the XOR swizzle of gemm_dma.hpp (on the index) is conflict-free again.
//   * both operands have three stages (A 32 KB + B 16 KB each = 144 KB), six glds per wave and tile, tile t+2 issued at the top
//     of tile t, counted vmcnt + raw s_barrier -- the loop of dma_pass with two reads per A fragment.
// MEASURED (tools/wino_check.py, 23 k rows, D = 256): exact, and SLOWER than the register-staged kernel -- 91.6 vs 78.9 us forward,
// 93.3 vs 81.3 us data gradient -- the doubled LDS reads, the packed adds and 25 spilled VGPRs (256 allocated) cost more than
// the staging selects they replace.  Kept behind LEGO_WINO_DMA=1 as the record of the experiment; off by default.
// Virtual tiles stay set-major (y0 / y1 / one temporary accumulator set, 84 VGPRs), so the raw rows of a channel slice are
// staged once per set: 4 x 29 KB per slice from L2, what the register-staged kernel also fetched (8 row loads per pair).
#pragma once
#include "gemm_wino.hpp"
#include "gemm_dma.hpp"

namespace lego {

#ifndef WD_PIN
#define WD_PIN 0
#endif
constexpr int WD_PLANE_ROWS = 128;                          // rows per parity plane of an A stage (120 filled + zero rows)
constexpr int WD_A_BYTES = 2 * WD_PLANE_ROWS * DMA_ROW_BYTES;   // 32 KB
constexpr int WD_B_BYTES = WINO_BN * DMA_ROW_BYTES;         // 16 KB
constexpr int WD_STAGE = WD_A_BYTES + WD_B_BYTES;           // 48 KB
constexpr int WD_ZERO_ROW = WD_PLANE_ROWS - 1;              // plane 0, index 127: never written by a DMA, cleared at kernel start
constexpr size_t wino_dma_lds_bytes() { return (size_t)DMA_STAGES * WD_STAGE; }

template <int NF>
__device__ __forceinline__ void wino_dma_pass(const WinoArgs& w, const EpiArgs& e, char* lds, int p0, int p_end, int P, int n0) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g4 = lane >> 4;
    const int C = w.C, N = w.N;
    const int KT = C / BK, T = 4 * KT;
    const size_t set_stride = (size_t)N * C;

    // ---- the strip's rows: pairs are generated item by item, so their first rows ascend; raw rows r_base .. r_base + 239
    const int info_first = w.pair_info[p0], info_last = w.pair_info[p_end - 1];
    const int r_first = info_first >> PI_ROW_SHIFT;
    // the last row any pair of the strip READS: its own second row, or row r + 2 of the same item (it belongs to the next strip)
    const int r_last = (info_last >> PI_ROW_SHIFT) + ((info_last & PI_RIGHT2) ? 2 : ((info_last & PI_HAS2) ? 1 : 0));
    const int r_base = max(r_first - 1, 0) & ~1;                                         // even: plane = (r - r_base) & 1

    // ---- DMA sources.  A: 30 instructions of 8 plane rows (15 per plane), four per wave (ids 30, 31 repeat id 29)
    const int chunk = (lane & 7) ^ (lane >> 3);
    const char* pa[4];
    int la_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int id = min(wave * 4 + j, 29);
        const int plane = id / 15, blk = id - plane * 15;
        const int idx = 8 * blk + (lane >> 3);
        const int r = min(r_base + 2 * idx + plane, r_last);                             // clamped into rows that exist
        pa[j] = reinterpret_cast<const char*>(w.x + (size_t)r * w.ldx) + 16 * chunk;
        la_off[j] = (plane * WD_PLANE_ROWS + 8 * blk) * DMA_ROW_BYTES;                   // wave-uniform
    }
    const char* pb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        pb[j] = reinterpret_cast<const char*>(w.u + (size_t)min(n0 + 16 * wave + 8 * j + (lane >> 3), N - 1) * C) + 16 * chunk;
    auto issue = [&](int t, int stage) {
        const int set = t / KT, kt = t - set * KT;
        const int ws = w.swap ? (set == 0 ? 3 : (set == 3 ? 0 : set)) : set;
        char* sA = lds + stage * WD_STAGE;
        char* sB = sA + WD_A_BYTES;
        const size_t kb = (size_t)kt * (BK * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(pa[j] + kb, sA + __builtin_amdgcn_readfirstlane(la_off[j]));
        const size_t ub = (size_t)ws * set_stride * 4 + kb;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(pb[j] + ub, sB + (16 * wave + 8 * j) * DMA_ROW_BYTES);
    };

    // ---- fragment addresses: for pair (a, l16) the stage-relative byte address of chunk g4 (k group 0) of its rows d0 .. d3 =
    // r - 1 .. r + 2; rows outside the item point at the zero row.  k group 1 = the same address with bit 6 flipped.
    int offd[NF][4];
#pragma unroll
    for (int a = 0; a < NF; ++a) {
        const int info = w.pair_info[min(p0 + a * 16 + l16, p_end - 1)];       // (lanes past the strip repeat its last pair)
        const int rel = (info >> PI_ROW_SHIFT) - r_base;
        const bool ok[4] = {(info & PI_LEFT) != 0, true, (info & PI_HAS2) != 0, (info & PI_RIGHT2) != 0};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int i = rel + d - 1;
            const int idx = ok[d] ? (i >> 1) : WD_ZERO_ROW;
            const int plane = ok[d] ? (i & 1) : 0;
            offd[a][d] = (plane * WD_PLANE_ROWS + idx) * DMA_ROW_BYTES + (((idx >> 2) & 1) << 6) + ((g4 ^ (idx & 3)) << 4);
        }
    }
    const int colw = wave * 16 + l16;
    int offB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) offB[q] = WD_A_BYTES + colw * DMA_ROW_BYTES + (((4 * q + g4) ^ (colw & 7)) << 4);

    // set 0: d0 - d2   set 1: d1 + d2   set 2: d2 - d1   set 3: d1 - d3
    auto read_frags = [&](int stage, int set, int q, f32x4 (&fa)[NF], f32x4& fb) {
        const char* sA = lds + stage * WD_STAGE;
        const int flip = q << 6;
        auto rd = [&](int off) { return *reinterpret_cast<const f32x4*>(sA + (off ^ flip)); };
        if (set == 0) {
#pragma unroll
            for (int a = 0; a < NF; ++a) fa[a] = rd(offd[a][0]) - rd(offd[a][2]);
        } else if (set == 1) {
#pragma unroll
            for (int a = 0; a < NF; ++a) fa[a] = rd(offd[a][1]) + rd(offd[a][2]);
        } else if (set == 2) {
#pragma unroll
            for (int a = 0; a < NF; ++a) fa[a] = rd(offd[a][2]) - rd(offd[a][1]);
        } else {
#pragma unroll
            for (int a = 0; a < NF; ++a) fa[a] = rd(offd[a][1]) - rd(offd[a][3]);
        }
        fb = *reinterpret_cast<const f32x4*>(sA + offB[q]);
    };

    f32x4 y0a[NF], y1a[NF], tma[NF];
#pragma unroll
    for (int a = 0; a < NF; ++a) {
        y0a[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        y1a[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        tma[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 fa0[NF], fb0, fa1[NF], fb1;

    issue(0, 0);
    issue(1, 1);                                              // T = 4 KT >= 4
    asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    read_frags(0, 0, 0, fa0, fb0);
    int sc = 0, sn = 2, t = 0;
    auto run_set = [&](int set, f32x4 (&ac)[NF]) {
        for (int kt = 0; kt < KT; ++kt, ++t) {
            if (t + 2 < T) issue(t + 2, sn);
            read_frags(sc, set, 1, fa1, fb1);
            if (WD_PIN & 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < NF; ++a) ac[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[a][j], fb0[j], ac[a], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int a = 0; a < NF; ++a) ac[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[a][j], fb1[j], ac[a], 0, 0, 0);
            if (t + 1 < T) {
                if (t + 2 < T) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                const int s1 = sc == 2 ? 0 : sc + 1;
                read_frags(s1, kt + 1 < KT ? set : set + 1, 0, fa0, fb0);
                if (WD_PIN & 2) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int a = 0; a < NF; ++a) ac[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[a][3], fb1[3], ac[a], 0, 0, 0);
            sn = sc;
            sc = sc == 2 ? 0 : sc + 1;
        }
    };
    run_set(0, y0a);                                         // M0
    run_set(1, y1a);                                         // M1
    run_set(2, tma);                                         // M2
#pragma unroll
    for (int a = 0; a < NF; ++a) {
        y0a[a] += y1a[a] + tma[a];
        y1a[a] -= tma[a];
        tma[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    run_set(3, tma);                                         // M3
#pragma unroll
    for (int a = 0; a < NF; ++a) y1a[a] -= tma[a];
    asm volatile("s_barrier" ::: "memory");                 // a next pass refills the stages

    // ---- epilogue of wino_pass: lane holds column colw x pairs 4*g4 + {0..3} of each fragment
    const int col = n0 + colw;
    const int cc = min(col, N - 1);
    const float bcol = (e.bias != nullptr && col < N) ? e.bias[col] : 0.f;
    float csum = 0.f;
    const float dinv = e.drop.p > 0.f ? 1.f / (1.f - e.drop.p) : 1.f;
    const bool dropping = e.drop.p > 0.f;
#pragma unroll
    for (int a = 0; a < NF; ++a) {
        const int pb_ = p0 + a * 16 + 4 * g4;
        if (pb_ >= p_end) continue;
        int inf[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) inf[v] = w.pair_info[min(pb_ + v, P - 1)];
        uint32_t k0[4], k1[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = inf[v] >> PI_ROW_SHIFT;
            k0[v] = dropping ? dropout_bits4(e.drop, r & ~3, cc, e.drop_cols) : 15u;
            k1[v] = (dropping && (r & 3) == 3) ? dropout_bits4(e.drop, r + 1, cc, e.drop_cols) : k0[v];
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (pb_ + v >= p_end || col >= N) continue;
            const int r = inf[v] >> PI_ROW_SHIFT;
            const bool has2 = (inf[v] & PI_HAS2) != 0;
            float y0 = y0a[a][v] + bcol;
            float y1 = y1a[a][v] + bcol;
            if (e.act == 1) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
            y0 *= (k0[v] >> (r & 3)) & 1u ? dinv : 0.f;
            y1 *= (k1[v] >> ((r + 1) & 3)) & 1u ? dinv : 0.f;
            float* dst = e.C + (size_t)r * e.ldc + col;
            dst[0] = y0;
            csum += y0;
            if (has2) { dst[e.ldc] = y1; csum += y1; }
        }
    }
    if (e.colsum != nullptr) {
        float sum = csum;
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (g4 == 0 && col < N) atomicAdd(e.colsum + col, sum);
    }
}

__global__ __launch_bounds__(STRIP_THREADS) void wino_dma_kernel(WinoArgs w, EpiArgs e) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const int P = w.P_dyn != nullptr ? min(w.P_cap, *w.P_dyn) : w.P_cap;
    if (P <= 0) return;
    const int halves = (w.N + WINO_BN - 1) / WINO_BN;
    const int G = max((int)gridDim.x / halves, 1);
    int strip, half;
    if (halves == 2) { half = (blockIdx.x >> 3) & 1; strip = (blockIdx.x & 7) + 8 * (blockIdx.x >> 4); }
    else { half = 0; strip = blockIdx.x; }
    if (strip >= G) return;
    int s = ((P + G - 1) / G + 15) & ~15;
    const int nsub = (s + WINO_BP - 1) / WINO_BP;
    const int sub = (((s + nsub - 1) / nsub) + 15) & ~15;
    const int strip0 = strip * s;
    if (strip0 >= P) return;
    const int strip_end = min(P, strip0 + s);
    const int n0 = half * WINO_BN;
    // the zero rows of every A stage (plane 0, indices 120 .. 127: no DMA writes there)
    for (int i = threadIdx.x; i < DMA_STAGES * 8 * (DMA_ROW_BYTES / 4); i += STRIP_THREADS) {
        const int st = i / (8 * (DMA_ROW_BYTES / 4)), rem = i - st * (8 * (DMA_ROW_BYTES / 4));
        reinterpret_cast<float*>(lds + st * WD_STAGE + (WD_PLANE_ROWS - 8) * DMA_ROW_BYTES)[rem] = 0.f;
    }
    __syncthreads();
    for (int p0 = strip0; p0 < strip_end; p0 += sub) {
        const int p_end = min(strip_end, p0 + sub);
        switch ((p_end - p0 + 15) >> 4) {                                   // block-uniform
            case 1: case 2: wino_dma_pass<2>(w, e, lds, p0, p_end, P, n0); break;
            case 3: case 4: wino_dma_pass<4>(w, e, lds, p0, p_end, P, n0); break;
            case 5: wino_dma_pass<5>(w, e, lds, p0, p_end, P, n0); break;
            case 6: wino_dma_pass<6>(w, e, lds, p0, p_end, P, n0); break;
            default: wino_dma_pass<7>(w, e, lds, p0, p_end, P, n0); break;
        }
    }
}

}  // namespace lego
