// Multi-head self-attention core of NRMS's AttentionOperator over ragged segments (L <= 64 rows).
// One wave per (segment, head): lane = query row; K/V/Q of the head are staged in LDS, the
// L x L score tile lives in LDS, softmax row reductions are per-lane loops (the row is lane-local),
// column reductions of the backward pass (dK, dV) are re-mapped to lane = key row.
// ~1.5 % of NRMS's flops: VALU kernel; the QKV / output projections around it run on the MFMA core.
#include "../../include/lego_hip.h"
#include "common.hpp"

namespace lego {

constexpr int kMaxL = 64;

// One wave per (segment, head).  SPLIT (L <= 32): lane = (query row i, half of the head dim) so 2L of the 64 lanes work
// and every q.k product is two 16-wide partial dots joined by one wave shuffle; otherwise lane = query row.
// The saved probability carries the dropout decision in its sign bit (p >= 0: kept, stored -p: dropped), so the
// backward pass needs no Philox.  Dot products run on 4 independent accumulators (no 32-long dependent FMA chain).
template <int W>
__device__ __forceinline__ float dotw(const float (&a)[W], const float* __restrict__ b) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int c = 0; c < W; c += 4) {
        s0 += a[c] * b[c]; s1 += a[c + 1] * b[c + 1]; s2 += a[c + 2] * b[c + 2]; s3 += a[c + 3] * b[c + 3];
    }
    return (s0 + s1) + (s2 + s3);
}

template <int HD, bool SPLIT>
__global__ __launch_bounds__(64) void mhsa_fwd_kernel(const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off,
                                                      int n_cap, const int* __restrict__ n_dyn, int D, int heads,
                                                      float* __restrict__ out, int ldo, float* __restrict__ probs, int Lmax,
                                                      Dropout drop, int drop_cols) {
    constexpr int W = SPLIT ? HD / 2 : HD;
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int seg = blockIdx.x, h = blockIdx.y;
    if (seg >= n) return;
    const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;
    if (L <= 0) return;
    if (SPLIT != (L <= 32)) return;                    // the other instantiation handles this segment
    const int lane = threadIdx.x;
    const int i = SPLIT ? (lane & 31) : lane, half = SPLIT ? (lane >> 5) : 0;
    const bool act = i < L;
    const int ic = act ? i : 0;
    const float scale = rsqrtf((float)HD);
    // K and V rows of this (segment, head) are staged in LDS with coalesced 16-B loads; the inner loops then read
    // one row per step at a wave-uniform address (broadcast) instead of waiting on a global load per key
    constexpr int LTS = SPLIT ? 32 : kMaxL;
    __shared__ __attribute__((aligned(16))) float Ks[LTS][HD];
    __shared__ __attribute__((aligned(16))) float Vs[LTS][HD];
    {
        const float* kg = qkv + (size_t)beg * ldq + D + h * HD;
        for (int e = lane; e < L * (HD / 4); e += 64) {
            const int j = e / (HD / 4), c4 = e - j * (HD / 4);
            *reinterpret_cast<f32x4*>(&Ks[j][4 * c4]) = *reinterpret_cast<const f32x4*>(kg + (size_t)j * ldq + 4 * c4);
            *reinterpret_cast<f32x4*>(&Vs[j][4 * c4]) = *reinterpret_cast<const f32x4*>(kg + D + (size_t)j * ldq + 4 * c4);
        }
    }
    float q[W];
    const float* qrow = qkv + (size_t)(beg + ic) * ldq + h * HD + half * W;
#pragma unroll
    for (int c = 0; c < W; ++c) q[c] = qrow[c] * scale;
    // the L x L score / probability tile of this (segment, head) lives in LDS as [key j][query i] (lane = i: conflict
    // free); it is saved to `probs` in the same layout at (beg*heads + h*L)*Lmax + j*L + i, so every global access of
    // the tile is coalesced (the first version kept one row per lane, 1 KB apart: 64 transactions per access)
    constexpr int LT = SPLIT ? 32 : kMaxL;
    __shared__ float Pl[LT][LT + 1];
    float* ptile = probs + ((size_t)beg * heads + (size_t)h * L) * Lmax;
    __syncthreads();
    float mx = -INFINITY;                              // pass 1: scores and the row max
    for (int j = 0; j < L; ++j) {
        float s = dotw<W>(q, &Ks[j][half * W]);
        if (SPLIT) s += __shfl_xor(s, 32, 64);
        mx = fmaxf(mx, s);
        if (half == 0) Pl[j][i < LT ? i : 0] = s;
    }
    __syncthreads();
    float se = 0.f;
    for (int j = 0; j < L; ++j) se += __expf(Pl[j][ic] - mx);
    const float inv = 1.f / se;
    float o[W];
#pragma unroll
    for (int c = 0; c < W; ++c) o[c] = 0.f;
    const int dcol = (beg + ic) * heads + h;
    for (int j0 = 0; j0 < L; j0 += 4) {
        float ds[4];
        dropout_scale4(drop, j0, dcol, drop_cols, ds);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            if (j < L) {
                const float* vr = &Vs[j][half * W];
                const float p = __expf(Pl[j][ic] - mx) * inv;
                const float pd = p * ds[u];
#pragma unroll
                for (int c = 0; c < W; ++c) o[c] += pd * vr[c];
                if (act && half == 0) ptile[(size_t)j * L + i] = ds[u] > 0.f ? p : -p;   // sign bit = dropped
            }
        }
    }
    if (act) {
        float* orow = out + (size_t)(beg + i) * ldo + h * HD + half * W;
#pragma unroll
        for (int c = 0; c < W; ++c) orow[c] = o[c];
    }
}

template <int HD, bool SPLIT>
__global__ __launch_bounds__(64) void mhsa_bwd_kernel(const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off,
                                                      int n_cap, const int* __restrict__ n_dyn, int D, int heads,
                                                      const float* __restrict__ gout, int ldgo, const float* __restrict__ probs,
                                                      int Lmax, float keep_scale, float* __restrict__ gqkv, int ldgq) {
    constexpr int W = SPLIT ? HD / 2 : HD;
    __shared__ float dotS[kMaxL];              // sum_j dP[i,j] P[i,j] of every query row
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int seg = blockIdx.x, h = blockIdx.y;
    if (seg >= n) return;
    const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;
    if (L <= 0) return;
    if (SPLIT != (L <= 32)) return;
    const int lane = threadIdx.x;
    const int i = SPLIT ? (lane & 31) : lane, half = SPLIT ? (lane >> 5) : 0;
    const bool act = i < L;
    const int ic = act ? i : 0;
    const float scale = rsqrtf((float)HD);
    const float* qbase = qkv + (size_t)beg * ldq + h * HD + half * W;
    const float* gbase = gout + (size_t)beg * ldgo + h * HD + half * W;
    constexpr int LT = SPLIT ? 32 : kMaxL;
    __shared__ float Pl[LT][LT + 1];           // signed probabilities [key j][query i], staged with coalesced loads
    __shared__ float dPl[LT][LT + 1];          // dP[i][j] = dOut_i . V_j, same layout: formed once, used by both passes
    // two row panels in LDS: (K, V) for pass 1, then (Q, dOut) for pass 2 -- rows are read at wave-uniform addresses
    __shared__ __attribute__((aligned(16))) float Ra[LT][HD];
    __shared__ __attribute__((aligned(16))) float Rb[LT][HD];
    auto stage = [&](const float* a, int lda, const float* b, int ldb) {
        for (int e = lane; e < L * (HD / 4); e += 64) {
            const int j = e / (HD / 4), c4 = e - j * (HD / 4);
            *reinterpret_cast<f32x4*>(&Ra[j][4 * c4]) = *reinterpret_cast<const f32x4*>(a + (size_t)j * lda + 4 * c4);
            *reinterpret_cast<f32x4*>(&Rb[j][4 * c4]) = *reinterpret_cast<const f32x4*>(b + (size_t)j * ldb + 4 * c4);
        }
    };
    const float* ptile = probs + ((size_t)beg * heads + (size_t)h * L) * Lmax;
    for (int e = lane; e < L * L; e += 64) Pl[e / L][e % L] = ptile[e];
    stage(qkv + (size_t)beg * ldq + D + h * HD, ldq, qkv + (size_t)beg * ldq + 2 * D + h * HD, ldq);      // K, V
    __syncthreads();
    {                                          // ---- pass 1, lane = (query row i, half): dot_i and dQ
        float g[W];
#pragma unroll
        for (int c = 0; c < W; ++c) g[c] = gbase[(size_t)ic * ldgo + c];
        float dot = 0.f;
        for (int j = 0; j < L; ++j) {
            float dpd = dotw<W>(g, &Rb[j][half * W]);
            if (SPLIT) dpd += __shfl_xor(dpd, 32, 64);
            const float ps = Pl[j][ic];                                // sign bit = dropped
            dot += (ps > 0.f ? dpd * keep_scale * ps : 0.f);
            if (half == 0) dPl[j][i] = dpd;                            // rows i >= L are never read back
        }
        if (act && half == 0) dotS[i] = dot;
        float dq[W];
#pragma unroll
        for (int c = 0; c < W; ++c) dq[c] = 0.f;
        for (int j = 0; j < L; ++j) {
            const float* kr = &Ra[j][half * W];
            const float dpd = dPl[j][i];                               // own write of the loop above (both halves: lane i)
            const float ps = Pl[j][ic];
            const float dS = fabsf(ps) * ((ps > 0.f ? dpd * keep_scale : 0.f) - dot);
#pragma unroll
            for (int c = 0; c < W; ++c) dq[c] += dS * kr[c];
        }
        if (act) {
            float* gq = gqkv + (size_t)(beg + i) * ldgq + h * HD + half * W;
#pragma unroll
            for (int c = 0; c < W; ++c) gq[c] = dq[c] * scale;
        }
    }
    __syncthreads();
    stage(qkv + (size_t)beg * ldq + h * HD, ldq, gout + (size_t)beg * ldgo + h * HD, ldgo);                   // Q, dOut
    __syncthreads();
    {                                          // ---- pass 2, lane = (key row j, half): dK[j], dV[j]; probs read by column
        float dk[W], dv[W];
#pragma unroll
        for (int c = 0; c < W; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        for (int r = 0; r < L; ++r) {
            const float* qr = &Ra[r][half * W];
            const float* gr = &Rb[r][half * W];
            const float ps = Pl[ic][r];
            const float dpd = dPl[ic][r];                              // dOut_r . V_j from pass 1 (other lanes' writes: barriers above)
            const float keep = ps > 0.f ? keep_scale : 0.f;
            const float p = fabsf(ps);
            const float dS = p * (dpd * keep - dotS[r]);
            const float pd = p * keep;
#pragma unroll
            for (int c = 0; c < W; ++c) {
                dk[c] += dS * qr[c];
                dv[c] += pd * gr[c];
            }
        }
        if (act) {
            float* gk = gqkv + (size_t)(beg + i) * ldgq + D + h * HD + half * W;
            float* gv = gqkv + (size_t)(beg + i) * ldgq + 2 * D + h * HD + half * W;
#pragma unroll
            for (int c = 0; c < W; ++c) { gk[c] = dk[c] * scale; gv[c] = dv[c]; }
        }
    }
}

static Dropout to_drop(const lego_dropout* d) {
    Dropout r = make_dropout(d);
    r.mask = nullptr;                 // the attention-probability site always draws in-kernel
    return r;
}

}  // namespace lego

using namespace lego;

extern "C" int lego_mhsa_core_fwd(const float* qkv, int ldq, const int32_t* seg_off, int n_cap, const int32_t* n_dyn,
                                  int D, int heads, float* out, int ldo, float* probs, int Lmax,
                                  const lego_dropout* drop, int rows_cap, void* stream) {
    LEGO_REQUIRE(heads > 0 && D % heads == 0, "lego_mhsa_core_fwd: D=%d not divisible by heads=%d", D, heads);
    LEGO_REQUIRE(Lmax <= kMaxL, "lego_mhsa_core_fwd: Lmax=%d exceeds %d", Lmax, kMaxL);
    if (n_cap <= 0) return 0;
    const int hd = D / heads;
    const Dropout dr = to_drop(drop);
    const int dc = rows_cap * heads;
    dim3 grid(n_cap, heads), block(64);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(HD) do { hipLaunchKernelGGL((mhsa_fwd_kernel<HD, true>), grid, block, 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, out, ldo, probs, Lmax, dr, dc); \
                        if (Lmax > 32) hipLaunchKernelGGL((mhsa_fwd_kernel<HD, false>), grid, block, 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, out, ldo, probs, Lmax, dr, dc); } while (0)
    switch (hd) {
        case 8: LAUNCH(8); break;
        case 16: LAUNCH(16); break;
        case 32: LAUNCH(32); break;
        case 64: LAUNCH(64); break;
        default: return set_error("lego_mhsa_core_fwd: head dim %d not in {8,16,32,64}", hd);
    }
#undef LAUNCH
    return check_launch("lego_mhsa_core_fwd");
}

extern "C" int lego_mhsa_core_bwd(const float* qkv, int ldq, const int32_t* seg_off, int n_cap, const int32_t* n_dyn,
                                  int D, int heads, const float* gout, int ldgo, const float* probs, int Lmax,
                                  const lego_dropout* drop, int rows_cap, float* gqkv, int ldgq, void* stream) {
    LEGO_REQUIRE(heads > 0 && D % heads == 0, "lego_mhsa_core_bwd: D=%d not divisible by heads=%d", D, heads);
    LEGO_REQUIRE(Lmax <= kMaxL, "lego_mhsa_core_bwd: Lmax=%d exceeds %d", Lmax, kMaxL);
    if (n_cap <= 0) return 0;
    const int hd = D / heads;
    const Dropout dr = to_drop(drop);
    const float ks = dr.p > 0.f ? 1.f / (1.f - dr.p) : 1.f;     // the keep / drop decision itself is the sign of the saved probability
    (void)rows_cap;
    dim3 grid(n_cap, heads), block(64);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(HD) do { hipLaunchKernelGGL((mhsa_bwd_kernel<HD, true>), grid, block, 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, gout, ldgo, probs, Lmax, ks, gqkv, ldgq); \
                        if (Lmax > 32) hipLaunchKernelGGL((mhsa_bwd_kernel<HD, false>), grid, block, 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, gout, ldgo, probs, Lmax, ks, gqkv, ldgq); } while (0)
    switch (hd) {
        case 8: LAUNCH(8); break;
        case 16: LAUNCH(16); break;
        case 32: LAUNCH(32); break;
        case 64: LAUNCH(64); break;
        default: return set_error("lego_mhsa_core_bwd: head dim %d not in {8,16,32,64}", hd);
    }
#undef LAUNCH
    return check_launch("lego_mhsa_core_bwd");
}
