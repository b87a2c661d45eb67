// Weight gradients (TN products C[M,N] += sum_r A[r,m] B[r,n], both operands row-major with the reduction index as the memory
// row) WITHOUT any LDS staging (round 5).
//
// In a TN product the operand fragment a wave needs IS a contiguous piece of an operand row: for v_mfma_f32_16x16x4_f32 lane
// (l16, g4) supplies A[k = g4][m = l16], so a 16-byte load of A[r0 + g4][m0 + 4 l16 .. + 3] gives each lane the A operands of FOUR
// matrix instructions -- instruction i covers the output rows m0 + 4 l16 + i, an interleaved set of 16 rows, and since the epilogue
// knows the interleaving nothing is ever transposed.  One such load per operand and k step (4 reduction rows) feeds a 64 x 64 wave
// tile = 16 MFMAs; the 16 lanes of a k row read 256 contiguous bytes.  So:
//   * no LDS images, no staging stores, no barrier in the k loop: a wave is a self-contained stream of {2 buffer loads, 16 MFMAs},
//     its loads DEPTH steps ahead in a register ring; three to four such waves per SIMD hide each other's memory latency (the
//     tile kernel of gemm_tn.hpp: one k tile of global loads -> registers -> LDS -> fragment reads per barrier, matrix pipe 0.5 busy);
//   * rows past the end of a wave's k range are outside its buffer descriptor and read as zeros (hardware range check), so the
//     loop is branch-free and the ring needs no tail handling;
//   * the 8 waves of a workgroup own the SAME output tile and eight consecutive k sub-ranges; their partial tiles are summed in a
//     fixed order through LDS (lane-private slots, no transposition: every wave uses the same lane <-> element map) and leave as ONE
//     set of fp32 atomics (256 contiguous bytes per wave instruction) or slab stores per workgroup -- an eighth of the adds per output
//     element that one-wave-tile-per-workgroup kernels issue;
//   * the workgroups of one k split are dealt to one XCD, so the rows they share are fetched into ONE L2.
#pragma once
#include "gemm_epi.hpp"
#include "wino_common.hpp"      // PI_* bits of a pair_info word

namespace lego {

constexpr int TND_WAVES = 8, TND_THREADS = TND_WAVES * 64;
constexpr int TND_T = 64;                                   // wave tile: 64 x 64 outputs
constexpr int TND_LD = TND_T + 4;                           // floats per row of the final [64][64] image
constexpr size_t tnd_lds_bytes() { return (size_t)4 * 16 * 64 * sizeof(f32x4); }      // four partial tiles in lane-private slots: 64 KB

struct TndArgs {
    const float* a; int lda;                // A[r][m]: M columns
    const float* b; int ldb;                // B[r][n]: N columns
    float* c; int ldc;                      // C[m][n] += ...
    int M, N, K_cap; const int* k_dyn;      // reduction rows: min(K_cap, *k_dyn)
    const int* a_row_off; const int* b_row_off;   // optional device row offsets of the operands
    int tiles_m, tiles_n, split;            // output tiles, k splits (a multiple of 8 when dealt)
    int deal;
    size_t slab_stride;                     // != 0: split z stores its partial into c + z * slab_stride (plain stores, no atomics)
};

template <int DEPTH>
__global__ __launch_bounds__(TND_THREADS) void tnd_kernel(TndArgs t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g4 = lane >> 4;
    const int K = t.k_dyn != nullptr ? min(t.K_cap, *t.k_dyn) : t.K_cap;
    // XCD-aware dealing: dispatch is round-robin over the 8 XCDs, so blocks b and b + 8 share one; the tiles of k split z go to XCD z % 8
    const int tiles = t.tiles_m * t.tiles_n;
    int tile, z;
    if (t.deal) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; z = xcd + 8 * (j / tiles); tile = j % tiles; }
    else { tile = blockIdx.x % tiles; z = blockIdx.x / tiles; }
    const int m0 = (tile % t.tiles_m) * TND_T, n0 = (tile / t.tiles_m) * TND_T;
    // this wave's k sub-range: the live rows dealt evenly over split x 8 waves, in whole rounds of the load ring
    const int rpw = ((K + t.split * TND_WAVES - 1) / (t.split * TND_WAVES) + 4 * DEPTH - 1) / (4 * DEPTH) * (4 * DEPTH);
    if (z * TND_WAVES * rpw >= K && t.slab_stride == 0) return;             // a k split behind the live rows has nothing to add (block-uniform)
    const int wb = (z * TND_WAVES + wave) * rpw;
    const int we = min(K, wb + rpw);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (wb < we) {                                                           // wave-uniform
        const float* ap = t.a + (t.a_row_off != nullptr ? (size_t)(*t.a_row_off) * t.lda : 0);
        const float* bp = t.b + (t.b_row_off != nullptr ? (size_t)(*t.b_row_off) * t.ldb : 0);
        // extents end at this wave's last row: everything behind it reads as zero
        const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ap), 0, (int)((unsigned)we * (unsigned)t.lda * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bp), 0, (int)((unsigned)we * (unsigned)t.ldb * 4u), 0x00020000);
        // columns past M / N are clamped into the operand: they only feed outputs that are never stored
        unsigned va = ((unsigned)(wb + g4) * (unsigned)t.lda + (unsigned)min(m0 + 4 * l16, t.M - 4)) * 4u;
        unsigned vb = ((unsigned)(wb + g4) * (unsigned)t.ldb + (unsigned)min(n0 + 4 * l16, t.N - 4)) * 4u;
        const unsigned sa = (unsigned)t.lda * 16u, sb = (unsigned)t.ldb * 16u;   // 4 rows per step
        // register ring: step s lives in slot s % DEPTH; while step s is multiplied, the loads of step s + DEPTH - 1 go into the slot
        // step s - 1 has just left (DEPTH - 1 steps in flight per wave)
        f32x4 fa[DEPTH], fb[DEPTH];
        auto load = [&](int slot) {
            fa[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_, va, 0, 0));   // past the range: zeros
            fb[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_, vb, 0, 0));
            va += sa; vb += sb;
        };
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) load(d);
        const int rounds = (we - wb + 4 * DEPTH - 1) / (4 * DEPTH);
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                load((d + DEPTH - 1) % DEPTH);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[d][i], fb[d][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);       // the two loads of a step go out BEFORE its 16 MFMAs: hipcc
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);      // otherwise sinks all of a round's loads behind the round's MFMAs
            }
        }
    }

    // ---- sum of the 8 partial tiles, fixed order: (w, w + 4), then (w, w + 2), then (0, 1).  Slot s holds 16 f32x4 per lane.
    f32x4* const slots = reinterpret_cast<f32x4*>(smem);
    auto put = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) slots[(s * 16 + i * 4 + j) * 64 + lane] = acc[i][j];
    };
    auto add = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += slots[(s * 16 + i * 4 + j) * 64 + lane];
    };
#pragma unroll
    for (int h = 4; h >= 1; h >>= 1) {
        if (wave >= h && wave < 2 * h) put(wave - h);
        __syncthreads();
        if (wave < h) add(wave);
        __syncthreads();
    }
    // ---- wave 0 holds the tile: lane (l16, g4) has rows m0 + 4 (4 g4 + v) + i, columns n0 + 4 l16 + j.  Through a row-major LDS image
    // so that a wave instruction of the atomics / stores covers 256 contiguous bytes of one output row
    float* const img = smem;
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 x = {acc[i][0][v], acc[i][1][v], acc[i][2][v], acc[i][3][v]};
                *reinterpret_cast<f32x4*>(img + (4 * (4 * g4 + v) + i) * TND_LD + 4 * l16) = x;
            }
    }
    __syncthreads();
    float* const C = t.c + (t.slab_stride != 0 ? (size_t)z * t.slab_stride : 0);
    const int col = n0 + lane;
#pragma unroll
    for (int q = 0; q < TND_T / TND_WAVES; ++q) {
        const int rl = TND_WAVES * q + wave;
        const int row = m0 + rl;
        if (row < t.M && col < t.N) {
            const float v = img[rl * TND_LD + lane];
            if (t.slab_stride != 0) C[(size_t)row * t.ldc + col] = v;
            else atomicAdd(C + (size_t)row * t.ldc + col, v);
        }
    }
}


// ---------------------------------------------------------------- the Winograd conv's weight gradient in the same form
// dU_s[o][c] = sum over pairs of dM_s[o] * A_s[c] (gemm_wino.hpp), s = 0..3: dM_s from the two rows of gy, A_s from the four rows of h.
// A wave owns a 64 (o) x 32 (c) tile of ALL FOUR sets -- the raw rows are loaded once per pair and the four combinations are formed
// in registers (the tile kernel ran one set per workgroup: every pair row was fetched and combined 16 times) -- and a k step is 4 pairs:
// lane (l16, g4) loads gy[r, r+1][o0 + 4 l16 ..+3] (two 16-byte loads) and h[r-1 .. r+2][c0 + 2 l16 ..+1] (four 8-byte loads) of pair
// 4 step + g4, neighbours the plan says do not exist as out-of-range offsets (zeros).  32 MFMAs per step, 128 accumulator registers.
constexpr int TNDP_TM = 64, TNDP_TN = 32;
constexpr size_t tndp_lds_bytes() { return (size_t)4 * 16 * 64 * sizeof(f32x4); }      // the tree sum runs in two halves of 16 f32x4 per lane

struct TndpArgs {
    const float* gy; int ldg; const float* h; int ldh;
    const int* pair_info; int P_cap; const int* P_dyn;
    float* du;                              // [split][4][Dout][Din]
    int Dout, Din, split, deal;
    unsigned gy_bytes, h_bytes;             // extents of the two row spaces (rows < 2 * P_cap)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int DEPTH>
__global__ __launch_bounds__(TND_THREADS) void tndp_kernel(TndpArgs t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g4 = lane >> 4;
    const int P = t.P_dyn != nullptr ? min(t.P_cap, *t.P_dyn) : t.P_cap;
    const int tiles_m = t.Dout / TNDP_TM, tiles = tiles_m * (t.Din / TNDP_TN);
    int tile, z;
    if (t.deal) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; z = xcd + 8 * (j / tiles); tile = j % tiles; }
    else { tile = blockIdx.x % tiles; z = blockIdx.x / tiles; }
    const int o0 = (tile % tiles_m) * TNDP_TM, c0 = (tile / tiles_m) * TNDP_TN;
    const int ppw = ((P + t.split * TND_WAVES - 1) / (t.split * TND_WAVES) + 4 * DEPTH - 1) / (4 * DEPTH) * (4 * DEPTH);   // pairs per wave
    const int wb = (z * TND_WAVES + wave) * ppw;
    const int we = min(P, wb + ppw);

    f32x4 acc[4][4][2];                     // [set][o group][c group]
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[s][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (wb < we) {                                                           // wave-uniform
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t.gy), 0, (int)t.gy_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t.h), 0, (int)t.h_bytes, 0x00020000);
        constexpr unsigned kNone = 0x80000000u;                              // extents < 2^31 (launcher)
        const unsigned ldgb = (unsigned)t.ldg * 4u, ldhb = (unsigned)t.ldh * 4u;
        const unsigned cg = (unsigned)(o0 + 4 * l16) * 4u, ch = (unsigned)(c0 + 2 * l16) * 4u;
        struct Step { f32x4 y0, y1; f32x2 d0, d1, d2, d3; };
        Step ring[DEPTH];
        int p = wb + g4;                                                     // this lane's pair of the step being loaded
        int info = t.pair_info[min(p, t.P_cap - 1)];
        auto load = [&](Step& q) {
            const bool in = p < we;
            const unsigned r = (unsigned)(info >> PI_ROW_SHIFT);
            const unsigned g0 = in ? r * ldgb + cg : kNone;
            const unsigned g1 = (in && (info & PI_HAS2)) ? g0 + ldgb : kNone;
            const unsigned h1 = in ? r * ldhb + ch : kNone;
            const unsigned h0 = (in && (info & PI_LEFT)) ? h1 - ldhb : kNone;
            const unsigned h2 = (in && (info & PI_HAS2)) ? h1 + ldhb : kNone;
            const unsigned h3 = (in && (info & PI_RIGHT2)) ? h1 + 2u * ldhb : kNone;
            p += 4;
            info = t.pair_info[min(p, t.P_cap - 1)];                        // the next step's word: in flight while this step's rows are
            q.y0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, g0, 0, 0));
            q.y1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, g1, 0, 0));
            q.d0 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rh, h0, 0, 0));
            q.d1 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rh, h1, 0, 0));
            q.d2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rh, h2, 0, 0));
            q.d3 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rh, h3, 0, 0));
        };
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) load(ring[d]);
        const int rounds = (we - wb + 4 * DEPTH - 1) / (4 * DEPTH);
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                load(ring[(d + DEPTH - 1) % DEPTH]);
                __builtin_amdgcn_sched_barrier(0);       // the loads of step + DEPTH - 1 go out before this step's arithmetic (hipcc otherwise
                                                         // sinks them behind most of its MFMAs and waits for them a few hundred cycles later)
                const Step& q = ring[d];
                // dM = {dy0, dy0 + dy1, dy0 - dy1, dy1},  A = {d0 - d2, d1 + d2, d2 - d1, d3 - d1}: set 3 carries its sign on the A side
                const f32x4 m[4] = {q.y0, q.y0 + q.y1, q.y0 - q.y1, q.y1};
                const f32x2 a[4] = {q.d0 - q.d2, q.d1 + q.d2, q.d2 - q.d1, q.d3 - q.d1};
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[s][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[s][i], a[s][j], acc[s][i][j], 0, 0, 0);
            }
        }
    }

    // ---- tree sum of the 8 partial tiles (fixed order), two sets at a time: slot = 16 f32x4 per lane
    f32x4* const slots = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int hh = 4; hh >= 1; hh >>= 1) {
            if (wave >= hh && wave < 2 * hh) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) slots[((wave - hh) * 16 + s * 8 + i * 2 + j) * 64 + lane] = acc[2 * half + s][i][j];
            }
            __syncthreads();
            if (wave < hh) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[2 * half + s][i][j] += slots[(wave * 16 + s * 8 + i * 2 + j) * 64 + lane];
            }
            __syncthreads();
        }
    }
    // ---- wave 0 holds the four tiles: lane (l16, g4) has rows o0 + 4 (4 g4 + v) + i, columns c0 + 2 l16 + j of each set
    if (wave == 0) {
        float* const slab = t.du + (size_t)z * 4 * t.Dout * t.Din;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x2 x = {acc[s][i][0][v], acc[s][i][1][v]};
                    *reinterpret_cast<f32x2*>(slab + ((size_t)s * t.Dout + o0 + 4 * (4 * g4 + v) + i) * t.Din + c0 + 2 * l16) = x;
                }
    }
}

}  // namespace lego
