// Winograd F(2,3) form of the k=3 'same' conv over ragged token rows (cnn_operator.py:54-57), on the row-strip
// MFMA kernel of gemm_strip.hpp.  Included by gemm_ops.hip after EpiArgs.
//
// Two consecutive rows (r, r+1) of one item form a PAIR (plan_pairs_kernel).  With d0..d3 = rows r-1..r+2 (zero
// outside the item) and taps g0, g1, g2:
//     M0 = (d0 - d2) g0        M1 = (d1 + d2) (g0+g1+g2)/2       M2 = (d2 - d1) (g0-g1+g2)/2        M3 = (d1 - d3) g2
//     y[r] = M0 + M1 + M2      y[r+1] = M1 - M2 - M3
// i.e. four [pairs x C] x [C x N] products instead of one [rows x 3C] x [3C x N]: 4 C MACs per pair and output
// column instead of 6 C -- two thirds of the direct conv's MFMA work (the reference's result up to fp32 rounding;
// the parity bar of the path is 1e-3 on fp32 logits).  The same kernel gives the data gradient (correlation with
// the taps reversed: sets 0 and 3 swap their weights, the weight panel is read k-major).
//
// Layout per workgroup (one per CU): a strip of <= 64 pairs x all N <= 256 columns; 8 waves side by side over the
// columns; accumulators (y0, y1, one temporary set) x 4 row fragments x 2 column fragments x 4 floats = 96 VGPRs.  The reduction runs
// over 4 * C/32 virtual k tiles (set-major); the A tile of a set is the two-row combination, formed when the staged
// registers are written to LDS.
#pragma once
#include "gemm_strip.hpp"

namespace lego {

constexpr int WINO_BP = 64;                 // pairs per pass
constexpr int WINO_NF = WINO_BP / 16;

constexpr int PI_HAS2 = 1, PI_LEFT = 2, PI_RIGHT2 = 4, PI_ROW_SHIFT = 3;

struct WinoArgs {
    const float* x; int ldx;                // input rows: h (forward) or gy (backward-data)
    const float* u;                         // [4][Dout][Din] transformed weights (conv3_wino_pack_kernel)
    int C;                                  // reduction channels per set: Din forward, Dout backward-data
    int N;                                  // output channels
    const int* pair_info; int P_cap; const int* P_dyn;
    int swap;                               // backward-data: sets 0 and 3 swap weights
};

template <bool B_MC>
constexpr size_t wino_lds_bytes() {
    return 2 * (size_t)(WINO_BP * STRIP_KC_LD + (B_MC ? BK * STRIP_MC_LD : STRIP_BN * STRIP_KC_LD)) * sizeof(float);
}

template <bool B_MC>
__global__ __launch_bounds__(STRIP_THREADS) void wino_kernel(WinoArgs w, EpiArgs e) {
    constexpr int NF = WINO_NF, BN = STRIP_BN;
    constexpr int A_FLOATS = WINO_BP * STRIP_KC_LD;
    constexpr int B_FLOATS = B_MC ? BK * STRIP_MC_LD : BN * STRIP_KC_LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * A_FLOATS;

    const int P = w.P_dyn != nullptr ? min(w.P_cap, *w.P_dyn) : w.P_cap;
    if (P <= 0) return;
    const int C = w.C, N = w.N;
    // strip of pairs for this workgroup, cut into passes of <= 64 pairs
    int s = ((P + (int)gridDim.x - 1) / (int)gridDim.x + 15) & ~15;
    const int nsub = (s + WINO_BP - 1) / WINO_BP;
    const int sub = (((s + nsub - 1) / nsub) + 15) & ~15;
    const int strip0 = blockIdx.x * s;
    if (strip0 >= P) return;
    const int strip_end = min(P, strip0 + s);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g4 = lane >> 4;
    const int KT = C / BK, T = 4 * KT;
    const size_t set_stride = (size_t)(B_MC ? C * N : N * C);     // floats per transformed weight matrix

    // B staging addresses that do not depend on the pass
    const float* brow[4];
    if constexpr (!B_MC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) brow[j] = w.u + (size_t)min((tid >> 3) + 64 * j, N - 1) * C + (tid & 7) * 4;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) brow[j] = w.u + (size_t)((tid >> 6) + 8 * j) * N + min((tid & 63) * 4, N - 4);
    }

    for (int p0 = strip0; p0 < strip_end; p0 += sub) {
        const int p_end = min(strip_end, p0 + sub);
        // ---- this thread's pair of the A tile (pair row tid >> 3, k quad tid & 7)
        const int info = w.pair_info[min(p0 + (tid >> 3), P - 1)];
        const float* base = w.x + (size_t)(info >> PI_ROW_SHIFT) * w.ldx + (tid & 7) * 4;
        const bool ok0 = (info & PI_LEFT) != 0, ok2 = (info & PI_HAS2) != 0, ok3 = (info & PI_RIGHT2) != 0;

        f32x4 sa1, sa2, sb[4];
        bool pa1 = false, pa2 = false;
        float sgn = 0.f;
        auto fetch = [&](int t) {
            t = min(t, T - 1);
            const int set = t / KT;
            const int k0 = (t - set * KT) * BK;
            // set 0: d0 - d2   set 1: d1 + d2   set 2: d2 - d1   set 3: d1 - d3
            const int ra = set == 0 ? -1 : (set == 2 ? 1 : 0);
            const int rb = set == 2 ? 0 : (set == 3 ? 2 : 1);
            pa1 = set == 0 ? ok0 : (set == 2 ? ok2 : true);
            pa2 = set == 2 ? true : (set == 3 ? ok3 : ok2);
            sgn = set == 1 ? 1.f : -1.f;
            sa1 = *reinterpret_cast<const f32x4*>(base + (pa1 ? ra * w.ldx : 0) + k0);
            sa2 = *reinterpret_cast<const f32x4*>(base + (pa2 ? rb * w.ldx : 0) + k0);
            const int ws = w.swap ? (set == 0 ? 3 : (set == 3 ? 0 : set)) : set;
            const size_t uoff = (size_t)ws * set_stride;
            if constexpr (B_MC) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sb[j] = *reinterpret_cast<const f32x4*>(brow[j] + uoff + (size_t)k0 * N);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) sb[j] = *reinterpret_cast<const f32x4*>(brow[j] + uoff + k0);
            }
        };
        auto commit = [&](float* A_, float* B_) {
            const f32x4 a = zero_unless(pa1, sa1), b = zero_unless(pa2, sa2);
            *reinterpret_cast<f32x4*>(A_ + (tid >> 3) * STRIP_KC_LD + (tid & 7) * 4) = a + sgn * b;
            if constexpr (B_MC) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(B_ + ((tid >> 6) + 8 * j) * STRIP_MC_LD + (tid & 63) * 4) = sb[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(B_ + ((tid >> 3) + 64 * j) * STRIP_KC_LD + (tid & 7) * 4) = sb[j];
            }
        };

        // y0 = M0 + M1 + M2, y1 = M1 - M2 - M3: sets 0 and 1 accumulate straight into y0 / y1, sets 2 and 3 into a
        // temporary that is folded in when the set is done (96 accumulator VGPRs instead of 128)
        f32x4 y0a[NF][2], y1a[NF][2], tma[NF][2];
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                y0a[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                y1a[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                tma[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }

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

        // software pipeline of strip_pass (gemm_strip.hpp); the accumulator set is a compile-time index
        fetch(0);
        commit(As0, Bs0);
        fetch(1);
        __syncthreads();
        read_frags(As0, Bs0, 0, fa0, fb0);
        int buf = 0, t = 0;
        auto run_set = [&](f32x4 (&ac)[NF][2]) {
            for (int kt = 0; kt < KT; ++kt, ++t) {
                const float* A_ = As0 + buf * A_FLOATS;
                const float* B_ = Bs0 + buf * B_FLOATS;
                float* An = As0 + (buf ^ 1) * A_FLOATS;
                float* Bn = Bs0 + (buf ^ 1) * B_FLOATS;
                read_frags(A_, B_, 1, fa1, fb1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < NF; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            ac[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[a][j], fb0[b][j], ac[a][b], 0, 0, 0);
                commit(An, Bn);                 // tile t+1 (past the end: the unused buffer)
                fetch(t + 2);
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int a = 0; a < NF; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            ac[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[a][j], fb1[b][j], ac[a][b], 0, 0, 0);
                __syncthreads();
                read_frags(An, Bn, 0, fa0, fb0);
#pragma unroll
                for (int a = 0; a < NF; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        ac[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[a][3], fb1[b][3], ac[a][b], 0, 0, 0);
                buf ^= 1;
            }
        };
        run_set(y0a);                                    // M0
        run_set(y1a);                                    // M1
        run_set(tma);                                    // M2
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                y0a[a][b] += y1a[a][b] + tma[a][b];
                y1a[a][b] -= tma[a][b];
                tma[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        run_set(tma);                                    // M3
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) y1a[a][b] -= tma[a][b];
        __syncthreads();

        // ---- epilogue: lane holds column l16 x pairs 4*g4 + {0..3} of each fragment
        int col[2];
        float bcol[2], csum[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            col[b] = wave * 32 + b * 16 + l16;
            bcol[b] = (e.bias != nullptr && col[b] < N) ? e.bias[col[b]] : 0.f;
            csum[b] = 0.f;
        }
        const float dinv = e.drop.p > 0.f ? 1.f / (1.f - e.drop.p) : 1.f;
        const bool dropping = e.drop.p > 0.f;
#pragma unroll
        for (int a = 0; a < NF; ++a) {
            const int pb = p0 + a * 16 + 4 * g4;
            if (pb >= p_end) continue;
            int inf[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) inf[v] = w.pair_info[min(pb + v, P - 1)];
            // keep bits of the fragment's 8 rows x 2 columns first: rows r and r + 1 of a pair sit in the same 4-row
            // group unless r % 4 == 3 (then the second group is fetched too)
            uint32_t k0[4][2], k1[4][2];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = inf[v] >> PI_ROW_SHIFT;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int cc = min(col[b], N - 1);
                    k0[v][b] = dropping ? dropout_bits4(e.drop, r & ~3, cc, e.drop_cols) : 15u;
                    k1[v][b] = (dropping && (r & 3) == 3) ? dropout_bits4(e.drop, r + 1, cc, e.drop_cols) : k0[v][b];
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (pb + v >= p_end) continue;
                const int r = inf[v] >> PI_ROW_SHIFT;
                const bool has2 = (inf[v] & PI_HAS2) != 0;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (col[b] >= N) continue;
                    float y0 = y0a[a][b][v] + bcol[b];
                    float y1 = y1a[a][b][v] + bcol[b];
                    if (e.act == 1) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
                    y0 *= (k0[v][b] >> (r & 3)) & 1u ? dinv : 0.f;
                    y1 *= (k1[v][b] >> ((r + 1) & 3)) & 1u ? dinv : 0.f;
                    float* dst = e.C + (size_t)r * e.ldc + col[b];
                    dst[0] = y0;
                    csum[b] += y0;
                    if (has2) { dst[e.ldc] = y1; csum[b] += y1; }
                }
            }
        }
        if (e.colsum != nullptr) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float sum = csum[b];
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                if (g4 == 0 && col[b] < N) atomicAdd(e.colsum + col[b], sum);
            }
        }
    }
}

// ---------------------------------------------------------------- weight gradient: MC loaders over pair rows
// dU_s[o][c] += sum_pairs dM_s[o] * A_s[c] with dM = {dy0, dy0+dy1, dy0-dy1, -dy1} and A_s the forward's row
// combinations: four TN products (set = blockIdx.z / split) on the split-K kernel of gemm_core.hpp.
struct McPair {
    static constexpr bool kDual = true;
    const float* p; int ld; int ext; int K; const int* pair_info; int side;     // side 0: A_s from h, 1: dM_s from gy
    int set;
    const int* s_info = nullptr; int s_base = 0, s_n = 0;     // pair_info words [s_base, s_base + s_n) cached in LDS (gemm_tn.hpp)
    struct Row {};
    __device__ __forceinline__ void prepare(int tap) { set = tap; }
    __device__ __forceinline__ void tile(int) {}
    __device__ __forceinline__ const int* info_src() const { return pair_info; }
    __device__ __forceinline__ void cache(const int* s, int base, int n) { s_info = s; s_base = base; s_n = n; }
    __device__ __forceinline__ int info_at(int kk) const {
        const int kc = min(kk, K - 1);
        const int i = kc - s_base;
        return (s_info != nullptr && i < s_n) ? s_info[i] : pair_info[kc];
    }
    // value = c1 * (keep1 ? v1 : 0) + c2 * (keep2 ? v2 : 0)
    __device__ __forceinline__ void load2(int kk, int c, f32x4& v1, bool& k1, f32x4& v2, bool& k2) const {
        const int info = info_at(kk);
        const bool in = kk < K;
        const bool ok0 = (info & PI_LEFT) != 0, ok2 = (info & PI_HAS2) != 0, ok3 = (info & PI_RIGHT2) != 0;
        const float* base = p + (size_t)(info >> PI_ROW_SHIFT) * ld + min(c, ext - 4);
        int ra, rb;
        if (side == 0) {
            ra = set == 0 ? -1 : (set == 2 ? 1 : 0);
            rb = set == 2 ? 0 : (set == 3 ? 2 : 1);
            k1 = in && (set == 0 ? ok0 : (set == 2 ? ok2 : true));
            k2 = in && (set == 2 ? true : (set == 3 ? ok3 : ok2));
        } else {
            ra = 0; rb = 1;
            k1 = in && set != 3;
            k2 = in && ok2 && set != 0;
        }
        v1 = *reinterpret_cast<const f32x4*>(base + (k1 ? ra * ld : 0));
        v2 = *reinterpret_cast<const f32x4*>(base + (k2 ? rb * ld : 0));
    }
    __device__ __forceinline__ f32x4 combine(const f32x4& v1, bool k1, const f32x4& v2, bool k2) const {
        const float c2 = set == 1 ? 1.f : -1.f;
        return zero_unless(k1, v1) + c2 * zero_unless(k2, v2);
    }
};

}  // namespace lego
