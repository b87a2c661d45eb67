// Grouped ranking metrics of the evaluation path (MetricPool.calculate, utils/metrics.py:313-369) on the device:
// GAUC, MRR, MRR0, LRAP and NDCG@k / HitRatio@k / Recall@k of every `group_col` group in one launch.
//
// Rows arrive sorted by group (stable), one wave owns one group.  Every metric of a group is a function of, per row i,
//   gt_i = #{j : s_j > s_i},  eq_i = #{j : s_j == s_i},  eqb_i = #{j < i : s_j == s_i}   (stable descending rank = gt + eqb)
// and the same counts restricted to positives / negatives, so one O(n_g^2) counting pass (lanes over i, all lanes
// read the same s_j: one broadcast load per step) feeds all of them.  Groups are tens to a few hundred rows (one user's
// impressions), 50-90 k groups per split: integer work, no MFMA; f64 for the few sums so that the per-group values
// round to the same fp32 the reference averages (utils/metrics.py:367).
#include "../../include/lego_hip.h"
#include <limits.h>
#include "common.hpp"

namespace lego {

struct MetricKs { int n; int k[LEGO_METRIC_MAX_K]; };

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// sum_{p = lo}^{hi - 1} 1 / log2(p + 2): the discounts a tie group occupying ranks [lo, hi) shares (sklearn _tie_averaged_dcg)
__device__ __forceinline__ double discount_sum(int lo, int hi) {
    double d = 0.0;
    for (int p = lo; p < hi; ++p) d += 1.0 / log2((double)p + 2.0);
    return d;
}

__global__ __launch_bounds__(256) void grouped_metrics_kernel(const float* __restrict__ s, const int32_t* __restrict__ lab,
                                                              const int32_t* __restrict__ off, int G, MetricKs ks,
                                                              float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;                                   // wave-uniform
    const int a = off[g], n = off[g + 1] - a;
    s += a;
    lab += a;
    int cnt = 0;
    for (int i = lane; i < n; i += 64) cnt += lab[i] == 1;
    const int n_pos = wave_sum_i32(cnt), n_neg = n - n_pos;

    double u_stat = 0.0, rr = 0.0, lrap = 0.0, dcg[LEGO_METRIC_MAX_K];
    int first = INT_MAX, hits[LEGO_METRIC_MAX_K];
#pragma unroll
    for (int q = 0; q < LEGO_METRIC_MAX_K; ++q) { dcg[q] = 0.0; hits[q] = 0; }
    for (int i = lane; i < n; i += 64) {
        if (lab[i] != 1) continue;                        // every metric here sums over the positives only
        const float si = s[i];
        int gt = 0, eq = 0, eqb = 0, neg_lt = 0, neg_eq = 0, pos_ge = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = s[j];
            const int pj = lab[j] == 1;
            const int is_gt = sj > si, is_eq = sj == si;
            gt += is_gt;
            eq += is_eq;
            eqb += is_eq & (j < i);
            neg_lt += (sj < si) & !pj;
            neg_eq += is_eq & !pj;
            pos_ge += (is_gt | is_eq) & pj;
        }
        const int rank = gt + eqb;                         // 0-based position after a stable descending sort
        u_stat += (double)neg_lt + 0.5 * (double)neg_eq;   // Mann-Whitney U == trapezoid ROC area (roc_auc_score)
        rr += 1.0 / (double)(rank + 1);
        first = min(first, rank);
        lrap += (double)pos_ge / (double)(gt + eq);
#pragma unroll
        for (int q = 0; q < LEGO_METRIC_MAX_K; ++q) {
            if (q < ks.n) {
                hits[q] += rank < ks.k[q];
                dcg[q] += discount_sum(gt, min(gt + eq, ks.k[q])) / (double)eq;
            }
        }
    }
    u_stat = wave_sum_f64(u_stat);
    rr = wave_sum_f64(rr);
    lrap = wave_sum_f64(lrap);
    first = wave_min_i32(first);
#pragma unroll
    for (int q = 0; q < LEGO_METRIC_MAX_K; ++q) {
        if (q < ks.n) { dcg[q] = wave_sum_f64(dcg[q]); hits[q] = wave_sum_i32(hits[q]); }
    }
    if (lane != 0) return;
    const float nan = __builtin_nanf("");
    const size_t ld = (size_t)G;
    out[0 * ld + g] = (n_pos > 0 && n_neg > 0) ? (float)(u_stat / ((double)n_pos * (double)n_neg)) : nan;
    out[1 * ld + g] = n_pos > 0 ? (float)(rr / (double)n_pos) : nan;
    out[2 * ld + g] = n_pos > 0 ? (float)(1.0 / (double)(first + 1)) : 0.f;
    out[3 * ld + g] = n_pos > 0 ? (float)(lrap / (double)n_pos) : 1.f;
    for (int q = 0; q < ks.n; ++q) {
        const double ideal = discount_sum(0, min(n_pos, ks.k[q]));
        out[(4 + 3 * q) * ld + g] = ideal > 0.0 ? (float)(dcg[q] / ideal) : 0.f;
        out[(5 + 3 * q) * ld + g] = hits[q] > 0 ? 1.f : 0.f;
        out[(6 + 3 * q) * ld + g] = n_pos > 0 ? (float)((double)hits[q] / (double)n_pos) : nan;
    }
}

}  // namespace lego

using namespace lego;

extern "C" int lego_grouped_metrics(const float* scores, const int32_t* labels, const int32_t* group_off, int n_groups,
                                    const int32_t* ks, int n_k, float* out, void* stream) {
    LEGO_REQUIRE(n_k >= 0 && n_k <= LEGO_METRIC_MAX_K, "lego_grouped_metrics: n_k=%d exceeds LEGO_METRIC_MAX_K", n_k);
    if (n_groups <= 0) return 0;
    MetricKs mk;
    mk.n = n_k;
    for (int q = 0; q < LEGO_METRIC_MAX_K; ++q) mk.k[q] = q < n_k ? ks[q] : 0;
    for (int q = 0; q < n_k; ++q) LEGO_REQUIRE(ks[q] > 0, "lego_grouped_metrics: k[%d]=%d must be positive", q, ks[q]);
    hipLaunchKernelGGL(grouped_metrics_kernel, dim3((n_groups + 3) / 4), dim3(256), 0, (hipStream_t)stream, scores, labels,
                       group_off, n_groups, mk, out);
    return check_launch("lego_grouped_metrics");
}
