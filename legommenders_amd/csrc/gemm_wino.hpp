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
// Layout per workgroup (one per CU): a strip of <= 128 pairs x 128 columns (N <= 256 = two column halves); 8 waves side by side
// over the columns, 16 each; accumulators (y0, y1, one temporary set) x NF <= 8 row fragments x 4 floats <= 96 VGPRs.  The
// reduction runs over 4 * C/32 virtual k tiles (set-major); the A tile of a set is the two-row combination, formed when the
// staged registers are written to LDS.
#pragma once
#include "gemm_strip.hpp"

namespace lego {

constexpr int WINO_BN = 128;                // output columns per workgroup: N <= 256 runs as two column halves
constexpr int WINO_BP = 112;                // pairs per pass: <= 7 row fragments (8 would spill: 3 x 8 accumulator + 2 x 8 operand fragments)
constexpr int WINO_MC_LD = WINO_BN + 4;

constexpr int PI_HAS2 = 1, PI_LEFT = 2, PI_RIGHT2 = 4, PI_ROW_SHIFT = 3;

struct WinoArgs {
    const float* x; int ldx;                // input rows: h (forward) or gy (backward-data)
    const float* u;                         // [4][Dout][Din] transformed weights (conv3_wino_pack_kernel)
    int C;                                  // reduction channels per set: Din forward, Dout backward-data
    int N;                                  // output channels
    const int* pair_info; int P_cap; const int* P_dyn;
    int swap;                               // backward-data: sets 0 and 3 swap weights
    unsigned x_bytes = 0;                   // gemm_wino2.hpp: extent of x for its buffer descriptor (set by launch_wino2)
};

template <bool B_MC>
constexpr size_t wino_lds_bytes() {
    return 2 * (size_t)(WINO_BP * STRIP_KC_LD + (B_MC ? BK * WINO_MC_LD : WINO_BN * STRIP_KC_LD)) * sizeof(float);
}

// Work split (round 2): the grid is (#CU / 2) pair strips x 2 column halves, 8 waves of 16 columns each, every wave over
// all NF row fragments of the strip.  Round 1 gave every workgroup all 256 columns and a strip of ceil(P / #CU) pairs: at the
// headline batch that is 51 pairs, rounded up to the 16-row MFMA granule = 64 -- a fifth of the matrix work went to padding
// rows.  With half as many, twice as long strips the same batch needs 7 fragments of 16 (102 -> 112 pairs): 7 MFMAs per k step
// and wave instead of 8, and ceil(2y) <= 2 ceil(y) makes this split never worse.  The two halves of a strip sit on the same
// XCD, so the second reader of the input rows hits that XCD's L2.
template <bool B_MC, int NF>
__device__ __forceinline__ void wino_pass(const WinoArgs& w, const EpiArgs& e, float* As0, float* Bs0, int p0, int p_end, int P, int n0) {
    constexpr int BN = WINO_BN;
    constexpr int A_FLOATS = WINO_BP * STRIP_KC_LD;
    constexpr int B_FLOATS = B_MC ? BK * WINO_MC_LD : BN * STRIP_KC_LD;
    constexpr int AN = (NF * 16 + 63) / 64;              // pair rows of the A tile per thread (1 or 2)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g4 = lane >> 4;
    const int C = w.C, N = w.N;
    const int KT = C / BK, T = 4 * KT;
    const size_t set_stride = (size_t)(B_MC ? C * N : N * C);     // floats per transformed weight matrix

    // B staging addresses (2 float4 per thread and k tile)
    const float* brow[2];
    if constexpr (!B_MC) {
#pragma unroll
        for (int j = 0; j < 2; ++j) brow[j] = w.u + (size_t)min(n0 + (tid >> 3) + 64 * j, N - 1) * C + (tid & 7) * 4;
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) brow[j] = w.u + (size_t)((tid >> 5) + 16 * j) * N + min(n0 + (tid & 31) * 4, N - 4);
    }
    // ---- this thread's pairs of the A tile (pair row (tid >> 3) + 64 j, k quad tid & 7)
    const float* base[AN];
    bool ok0[AN], ok2[AN], ok3[AN];
#pragma unroll
    for (int j = 0; j < AN; ++j) {
        const int info = w.pair_info[min(p0 + (tid >> 3) + 64 * j, P - 1)];
        base[j] = w.x + (size_t)(info >> PI_ROW_SHIFT) * w.ldx + (tid & 7) * 4;
        ok0[j] = (info & PI_LEFT) != 0; ok2[j] = (info & PI_HAS2) != 0; ok3[j] = (info & PI_RIGHT2) != 0;
    }

    f32x4 sa1[AN], sa2[AN], sb[2];
    bool pa1[AN], pa2[AN];
    float sgn = 0.f;
    auto fetch = [&](int t) {
        t = min(t, T - 1);
        const int set = t / KT;
        const int k0 = (t - set * KT) * BK;
        // set 0: d0 - d2   set 1: d1 + d2   set 2: d2 - d1   set 3: d1 - d3
        const int ra = set == 0 ? -1 : (set == 2 ? 1 : 0);
        const int rb = set == 2 ? 0 : (set == 3 ? 2 : 1);
        sgn = set == 1 ? 1.f : -1.f;
#pragma unroll
        for (int j = 0; j < AN; ++j) {
            pa1[j] = set == 0 ? ok0[j] : (set == 2 ? ok2[j] : true);
            pa2[j] = set == 2 ? true : (set == 3 ? ok3[j] : ok2[j]);
            sa1[j] = *reinterpret_cast<const f32x4*>(base[j] + (pa1[j] ? ra * w.ldx : 0) + k0);
            sa2[j] = *reinterpret_cast<const f32x4*>(base[j] + (pa2[j] ? rb * w.ldx : 0) + k0);
        }
        const int ws = w.swap ? (set == 0 ? 3 : (set == 3 ? 0 : set)) : set;
        const size_t uoff = (size_t)ws * set_stride;
        if constexpr (B_MC) {
#pragma unroll
            for (int j = 0; j < 2; ++j) sb[j] = *reinterpret_cast<const f32x4*>(brow[j] + uoff + (size_t)k0 * N);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) sb[j] = *reinterpret_cast<const f32x4*>(brow[j] + uoff + k0);
        }
    };
    auto commit = [&](float* A_, float* B_) {
#pragma unroll
        for (int j = 0; j < AN; ++j) {
            const f32x4 a = zero_unless(pa1[j], sa1[j]), b = zero_unless(pa2[j], sa2[j]);
            if ((tid >> 3) + 64 * j < NF * 16)           // the A image holds NF * 16 <= 112 pair rows
                *reinterpret_cast<f32x4*>(A_ + ((tid >> 3) + 64 * j) * STRIP_KC_LD + (tid & 7) * 4) = a + sgn * b;
        }
        if constexpr (B_MC) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(B_ + ((tid >> 5) + 16 * j) * WINO_MC_LD + (tid & 31) * 4) = sb[j];
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(B_ + ((tid >> 3) + 64 * j) * STRIP_KC_LD + (tid & 7) * 4) = sb[j];
        }
    };

    // y0 = M0 + M1 + M2, y1 = M1 - M2 - M3: sets 0 and 1 accumulate straight into y0 / y1, sets 2 and 3 into a
    // temporary that is folded in when the set is done
    f32x4 y0a[NF], y1a[NF], tma[NF];
#pragma unroll
    for (int a = 0; a < NF; ++a) {
        y0a[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        y1a[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        tma[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    f32x4 fa0[NF], fb0, fa1[NF], fb1;
    const int colw = wave * 16 + l16;                    // this lane's column inside the workgroup's 128
    auto read_frags = [&](const float* A_, const float* B_, int q, f32x4 (&fa)[NF], f32x4& fb) {
#pragma unroll
        for (int a = 0; a < NF; ++a)
            fa[a] = *reinterpret_cast<const f32x4*>(A_ + (a * 16 + l16) * STRIP_KC_LD + 16 * q + 4 * g4);
        if constexpr (B_MC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = B_[(16 * q + 4 * g4 + j) * WINO_MC_LD + colw];
        } else {
            fb = *reinterpret_cast<const f32x4*>(B_ + colw * STRIP_KC_LD + 16 * q + 4 * g4);
        }
    };

    // software pipeline of strip_pass (gemm_strip.hpp); the accumulator set is a compile-time index
    fetch(0);
    commit(As0, Bs0);
    fetch(1);
    __syncthreads();
    read_frags(As0, Bs0, 0, fa0, fb0);
    int buf = 0, t = 0;
    auto run_set = [&](f32x4 (&ac)[NF]) {
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
                    ac[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[a][j], fb0[j], ac[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // keep `commit` (and its wait on the staged loads) behind the F0 MFMAs: gemm_strip.hpp
            commit(An, Bn);                 // tile t+1 (past the end: the unused buffer)
            fetch(t + 2);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int a = 0; a < NF; ++a)
                    ac[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[a][j], fb1[j], ac[a], 0, 0, 0);
            __syncthreads();
            read_frags(An, Bn, 0, fa0, fb0);
#pragma unroll
            for (int a = 0; a < NF; ++a)
                ac[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[a][3], fb1[3], ac[a], 0, 0, 0);
            buf ^= 1;
        }
    };
    run_set(y0a);                                    // M0
    run_set(y1a);                                    // M1
    run_set(tma);                                    // M2
#pragma unroll
    for (int a = 0; a < NF; ++a) {
        y0a[a] += y1a[a] + tma[a];
        y1a[a] -= tma[a];
        tma[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    run_set(tma);                                    // M3
#pragma unroll
    for (int a = 0; a < NF; ++a) y1a[a] -= tma[a];
    __syncthreads();        // the next pass refills both buffers

    // ---- epilogue: lane holds column colw x pairs 4*g4 + {0..3} of each fragment
    const int col = n0 + colw;
    const int cc = min(col, N - 1);
    const float bcol = (e.bias != nullptr && col < N) ? e.bias[col] : 0.f;
    float csum = 0.f;
    const float dinv = e.drop.p > 0.f ? 1.f / (1.f - e.drop.p) : 1.f;
    const bool dropping = e.drop.p > 0.f;
#pragma unroll
    for (int a = 0; a < NF; ++a) {
        const int pb = p0 + a * 16 + 4 * g4;
        if (pb >= p_end) continue;
        int inf[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) inf[v] = w.pair_info[min(pb + v, P - 1)];
        // keep bits of the fragment's 8 rows first: rows r and r + 1 of a pair sit in the same 4-row
        // group unless r % 4 == 3 (then the second group is fetched too)
        uint32_t k0[4], k1[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = inf[v] >> PI_ROW_SHIFT;
            k0[v] = dropping ? dropout_bits4(e.drop, r & ~3, cc, e.drop_cols) : 15u;
            k1[v] = (dropping && (r & 3) == 3) ? dropout_bits4(e.drop, r + 1, cc, e.drop_cols) : k0[v];
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (pb + v >= p_end || col >= N) continue;
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

template <bool B_MC>
__global__ __launch_bounds__(STRIP_THREADS) void wino_kernel(WinoArgs w, EpiArgs e) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * WINO_BP * STRIP_KC_LD;
    const int P = w.P_dyn != nullptr ? min(w.P_cap, *w.P_dyn) : w.P_cap;
    if (P <= 0) return;
    // (strip, column half) of this workgroup: blockIdx and blockIdx + 8 share a strip and an XCD
    const int halves = (w.N + WINO_BN - 1) / WINO_BN;                       // 1 or 2
    const int G = max((int)gridDim.x / halves, 1);                          // pair strips
    // the two column halves of a strip run on the same XCD (blockIdx and blockIdx + 8: dispatch is round-robin over 8 XCDs), so
    // the second reader of the strip's input rows hits that XCD's L2.  (Aligning the strips of consecutive kernels to XCDs, so
    // that a consumer would find its producer's rows in its own L2, measured no change: 93.3 vs 93.1 us.)
    int strip, half;
    if (halves == 2) { half = (blockIdx.x >> 3) & 1; strip = (blockIdx.x & 7) + 8 * (blockIdx.x >> 4); }
    else { half = 0; strip = blockIdx.x; }
    if (strip >= G) return;
    // strip of pairs for this workgroup, cut into passes of <= 112 pairs
    int s = ((P + G - 1) / G + 15) & ~15;
    const int nsub = (s + WINO_BP - 1) / WINO_BP;
    const int sub = (((s + nsub - 1) / nsub) + 15) & ~15;
    const int strip0 = strip * s;
    if (strip0 >= P) return;
    const int strip_end = min(P, strip0 + s);
    const int n0 = half * WINO_BN;
    LEGO_CLOCK_BEGIN
    for (int p0 = strip0; p0 < strip_end; p0 += sub) {
        const int p_end = min(strip_end, p0 + sub);
        switch ((p_end - p0 + 15) >> 4) {                                   // block-uniform
            case 1: case 2: wino_pass<B_MC, 2>(w, e, As0, Bs0, p0, p_end, P, n0); break;
            case 3: case 4: wino_pass<B_MC, 4>(w, e, As0, Bs0, p0, p_end, P, n0); break;
            case 5: wino_pass<B_MC, 5>(w, e, As0, Bs0, p0, p_end, P, n0); break;
            case 6: wino_pass<B_MC, 6>(w, e, As0, Bs0, p0, p_end, P, n0); break;
            default: wino_pass<B_MC, 7>(w, e, As0, Bs0, p0, p_end, P, n0); break;
        }
    }
    LEGO_CLOCK_END(0)
}

// ---------------------------------------------------------------- weight gradient: MC loaders over pair rows
// dU_s[o][c] += sum_pairs dM_s[o] * A_s[c] with dM = {dy0, dy0+dy1, dy0-dy1, -dy1} and A_s the forward's row
// combinations: four TN products (set = blockIdx.z / split) on the split-K kernel of gemm_core.hpp.
struct McPair {
    static constexpr bool kDual = true;
    const float* p; int ld; int ext; int K; const int* pair_info; int side;     // side 0: A_s from h, 1: dM_s from gy
    int set;
    const int* s_info = nullptr; int s_base = 0;              // tn_kernel: the pair_info words of the workgroup's k range, in LDS
    struct Row {};
    __device__ __forceinline__ void prepare(int tap) { set = tap; }
    __device__ __forceinline__ void tile(int) {}
    __device__ __forceinline__ const int* info_src() const { return pair_info; }
    __device__ __forceinline__ void cache(const int* s, int base) { s_info = s; s_base = base; }
    // kCached is a compile-time property of the CALLER (tn_kernel caches, the generic tile kernel does not): a run-time choice
    // here put a branch and a full s_waitcnt in front of every operand load
    template <bool kCached>
    __device__ __forceinline__ int info_at(int kk) const {
        const int kc = min(kk, K - 1);
        if constexpr (kCached) return s_info[kc - s_base];
        else return pair_info[kc];
    }
    // value = c1 * (keep1 ? v1 : 0) + c2 * (keep2 ? v2 : 0)
    template <bool kCached = false>
    __device__ __forceinline__ void load2(int kk, int c, f32x4& v1, bool& k1, f32x4& v2, bool& k2) const {
        const int info = info_at<kCached>(kk);
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
