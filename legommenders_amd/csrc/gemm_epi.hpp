// Epilogue kinds of every dense product of liblego_hip.so (shared by gemm_ops.hip and wino2_ops.hip).
#pragma once
#include "gemm_strip.hpp"

namespace lego {

// Epilogue of every product.  The per-element memory traffic (live bits, previous C for accumulation,
// ReLU reference, atomics) is selected at COMPILE time so that each (sub-tile, row group) issues all of its
// loads back to back before the first use -- with run-time flags the loads sat behind branches and were
// serialised at L2 latency (measured: the accumulate + ReLU-backward product ran at half the plain rate).
// Cheap per-column things (bias, activation kind, dropout on/off, column sums) stay run-time.
struct EpiArgs {
    float* C; int ldc;
    const float* bias;        // [N] or null
    int act;                  // 0 none, 1 relu, 2 tanh; tile-kernel epilogue (`run`) only: 3 = exact GELU with the pre-activation kept in C2
                              // (BertIntermediate: z = x W1^T + b1 -> C2, gelu(z) -> C), 4 = the reference IS a GELU pre-activation:
                              // x * gelu'(ref) instead of the ReLU rule (data gradient of BertOutput.dense through the GELU)
    float* C2; int ldc2;      // act == 3: where the pre-activation goes
    const int* rowinfo;       // live-bit source (indexed by absolute row), kind ROWINFO
    Dropout drop;             // p == 0: off
    int drop_cols;            // column count of the dropout counter space
    const float* relu_ref; int ld_ref; float relu_scale;   // backward of ReLU(+dropout): ref>0 ? x*scale : 0
    float* colsum;            // += column sums of the stored values (bias gradients)
    size_t tap_stride;        // C offset per tap (TN conv weight gradient)
    const int* row_off_dyn;   // device row offset of C / rowinfo / relu_ref rows
    int M, N, row_off;
    int rows_form;            // gemm_dma.hpp: 1 = row-major epilogue through LDS (0: fragment-shaped stores)
};

template <bool ROWINFO, bool ACCUM, bool RELUREF, bool ATOMIC>
struct EpiT : EpiArgs {
    __device__ __forceinline__ void setup(int M_, int N_, int tap) {
        M = M_; N = N_;
        row_off = row_off_dyn != nullptr ? *row_off_dyn : 0;
        C += (size_t)tap * tap_stride;
    }
    template <int TM, int TN>
    __device__ __forceinline__ void run(f32x16 (&acc)[TM][TN], int m_base, int n_base, int li, int lh) {
        int col[TN], colc[TN];
        float bcol[TN], csum[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            col[b] = n_base + b * 32 + li;
            colc[b] = min(col[b], N - 1);
            bcol[b] = bias != nullptr ? bias[colc[b]] : 0.f;
            csum[b] = 0.f;
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r0 = m_base + a * 32 + 8 * g + 4 * lh;
                if (r0 >= M) continue;                       // wave-half uniform; rows below are clamped, stores guarded
                int ra[4];
                bool live[4];
                float old[TN][4], ref[TN][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {                // every load of this group is issued before the first use
                    ra[i] = min(r0 + i, M - 1) + row_off;
                    live[i] = ROWINFO ? (rowinfo[ra[i]] & RI_LIVE) != 0 : true;
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        old[b][i] = ACCUM ? C[(size_t)ra[i] * ldc + colc[b]] : 0.f;
                        ref[b][i] = RELUREF ? relu_ref[(size_t)ra[i] * ld_ref + colc[b]] : 1.f;
                    }
                }
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    float ds[4];
                    dropout_scale4(drop, r0 + row_off, col[b], drop_cols, ds);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float x = acc[a][b][4 * g + i] + bcol[b];
                        const bool in = r0 + i < M && col[b] < N;
                        if (act == 1) x = fmaxf(x, 0.f);
                        else if (act == 2) x = fast_tanh(x);
                        else if (act == 3) {
                            if (in) C2[(size_t)ra[i] * ldc2 + col[b]] = x;
                            x = gelu_exact(x);
                        }
                        if (ROWINFO && !live[i]) x = 0.f;
                        x *= ds[i];
                        if (ACCUM) x += old[b][i];
                        if (RELUREF) x = act == 4 ? x * gelu_exact_grad(ref[b][i]) : (ref[b][i] > 0.f ? x * relu_scale : 0.f);
                        if (in) {
                            float* dst = C + (size_t)ra[i] * ldc + col[b];
                            if (ATOMIC) atomicAdd(dst, x); else *dst = x;
                            csum[b] += x;
                        }
                    }
                }
            }
        if (colsum != nullptr) {
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const float s = csum[b] + __shfl_xor(csum[b], 32, 64);
                if (lh == 0 && col[b] < N) atomicAdd(colsum + col[b], s);
            }
        }
    }
    // 16 x 16 fragments of the row-strip kernel: lane holds column l16 x rows 4*g4 + {0..3} of fragment (a, b)
    template <int NF>
    __device__ __forceinline__ void run16(f32x4 (&acc)[NF][2], int m_base, int m_end, int n_base, int l16, int g4) {
        int col[2], colc[2];
        float bcol[2], csum[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            col[b] = n_base + b * 16 + l16;
            colc[b] = min(col[b], N - 1);
            bcol[b] = bias != nullptr ? bias[colc[b]] : 0.f;
            csum[b] = 0.f;
        }
        // keep bits of the whole tile first (one byte load each when the mask is precomputed)
        uint32_t kb[NF][2];
        const float dinv = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                kb[a][b] = dropout_bits4(drop, min(m_base + a * 16 + 4 * g4, max(m_end - 1, 0) & ~3) + row_off, colc[b], drop_cols);
#pragma unroll
        for (int a = 0; a < NF; ++a) {
            const int r0 = m_base + a * 16 + 4 * g4;
            if (r0 >= m_end) continue;
            int ra[4];
            bool live[4];
            float old[2][4], ref[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = min(r0 + i, m_end - 1) + row_off;
                live[i] = ROWINFO ? (rowinfo[ra[i]] & RI_LIVE) != 0 : true;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    old[b][i] = ACCUM ? C[(size_t)ra[i] * ldc + colc[b]] : 0.f;
                    ref[b][i] = RELUREF ? relu_ref[(size_t)ra[i] * ld_ref + colc[b]] : 1.f;
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float ds[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ds[i] = (kb[a][b] >> i) & 1u ? dinv : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = acc[a][b][i] + bcol[b];
                    if (act == 1) x = fmaxf(x, 0.f);
                    else if (act == 2) x = fast_tanh(x);
                    if (ROWINFO && !live[i]) x = 0.f;
                    x *= ds[i];
                    if (ACCUM) x += old[b][i];
                    if (RELUREF) x = ref[b][i] > 0.f ? x * relu_scale : 0.f;
                    if (r0 + i < m_end && col[b] < N) {
                        float* dst = C + (size_t)ra[i] * ldc + col[b];
                        if (ATOMIC) atomicAdd(dst, x); else *dst = x;
                        csum[b] += x;
                    }
                }
            }
        }
        if (colsum != nullptr) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float s = csum[b];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                if (g4 == 0 && col[b] < N) atomicAdd(colsum + col[b], s);
            }
        }
    }
    // The same epilogue in ROW-MAJOR form (gemm_dma.hpp): the wave's 32-column slab of accumulators goes through a wave-private
    // LDS tile [NF*16][32] (ds_write_b32 two-way = free, ds_read_b128 conflict-free: tools/lds_swizzle_check.py notes), comes back
    // as lane = (row lane >> 3 of a group of 8, 4 consecutive columns), and every global access of the epilogue -- the store,
    // the previous C of an accumulation, the ReLU reference, the keep bits -- is a 16-B (4-B for the bits) access that covers
    // whole 128-B lines: 8 rows x 128 B per wave-instruction instead of 4 rows x 64 B, a quarter of the store instructions.
    // Measured on the 26 368 x 256 x 256 product: the fragment-shaped epilogue cost 12 us of a 42 us launch.
    // Requires N % 4 == 0 and a precomputed keep-bit mask when dropout is on (else the caller takes run16).
    __device__ __forceinline__ bool rows_form_ok() const { return rows_form != 0 && (N & 3) == 0 && (ldc & 3) == 0 && (drop.p <= 0.f || drop.mask != nullptr) &&
                                                                  (!RELUREF || (ld_ref & 3) == 0) && (drop_cols & 3) == 0; }
    template <int NF>
    __device__ __forceinline__ void run16_rows(f32x4 (&acc)[NF][2], int m_base, int m_end, int n_base, int l16, int g4, float* tile) {
        constexpr int LD = 32;
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) tile[(a * 16 + 4 * g4 + i) * LD + b * 16 + l16] = acc[a][b][i];
        const int lane = g4 * 16 + l16;
        const int rsub = lane >> 3, col = n_base + 4 * (lane & 7);
        const bool col_ok = col < N;
        const int colc = min(col, N - 4);
        const f32x4 b4 = bias != nullptr ? *reinterpret_cast<const f32x4*>(bias + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float dinv = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
        const bool dropping = drop.p > 0.f;
        f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int g = 0; g < NF * 2; ++g) {
            const int r = m_base + 8 * g + rsub;
            const bool ok = r < m_end && col_ok;
            const int ra = min(r, m_end - 1) + row_off;
            f32x4 x = *reinterpret_cast<const f32x4*>(tile + (8 * g + rsub) * LD + 4 * (lane & 7));
            f32x4 old = f32x4{0.f, 0.f, 0.f, 0.f}, ref = f32x4{1.f, 1.f, 1.f, 1.f};
            if constexpr (ACCUM) old = *reinterpret_cast<const f32x4*>(C + (size_t)ra * ldc + colc);
            if constexpr (RELUREF) ref = *reinterpret_cast<const f32x4*>(relu_ref + (size_t)ra * ld_ref + colc);
            const bool live = ROWINFO ? (rowinfo[ra] & RI_LIVE) != 0 : true;
            uint32_t kw = 0x0f0f0f0fu;
            if (dropping) kw = *reinterpret_cast<const uint32_t*>(drop.mask + (uint64_t)(ra >> 2) * (uint64_t)drop_cols + (uint64_t)colc);
            kw >>= (ra & 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = x[i] + b4[i];
                if (act == 1) v = fmaxf(v, 0.f);
                else if (act == 2) v = fast_tanh(v);
                if (ROWINFO && !live) v = 0.f;
                v *= (kw >> (8 * i)) & 1u ? dinv : 0.f;
                if (ACCUM) v += old[i];
                if (RELUREF) v = ref[i] > 0.f ? v * relu_scale : 0.f;
                x[i] = v;
            }
            if (ok) {
                float* dst = C + (size_t)ra * ldc + col;
                // rows_form 2 / 3: write-through (sc1) / streaming (nt) stores -- the output leaves L2 while the kernel still
                // computes instead of waiting as dirty lines for the write-back at the kernel boundary
                if (rows_form == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(x) : "memory");
                else if (rows_form == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dst), "v"(x) : "memory");
                else *reinterpret_cast<f32x4*>(dst) = x;
                cs += x;
            }
        }
        if (colsum != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sum = cs[i];
                sum += __shfl_xor(sum, 8, 64);
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                if (rsub == 0 && col_ok) atomicAdd(colsum + col + i, sum);
            }
        }
    }
};

using Epi = EpiArgs;      // host-side view; the kernel is instantiated with one EpiT<...> kind

}  // namespace lego
