// Winograd F(2,3) form of the k=3 'same' conv over ragged token rows (cnn_operator.py:54-57): what the kernels of
// gemm_wino2.hpp (forward / data gradient), gemm_tnd.hpp (weight gradient, long reductions) and gemm_tn.hpp (weight
// gradient, short reductions) share -- the pair plan's bits, the launch arguments, the pair-row operand loader.
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
