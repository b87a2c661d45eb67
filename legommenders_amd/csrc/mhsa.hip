// Multi-head self-attention core of NRMS's AttentionOperator over ragged segments (L <= 64 rows).
// One wave per (segment, head): lane = query row; K/V/Q of the head are staged in LDS, the
// L x L score tile lives in LDS, softmax row reductions are per-lane loops (the row is lane-local),
// column reductions of the backward pass (dK, dV) are re-mapped to lane = key row.
// ~1.5 % of NRMS's flops: VALU kernel; the QKV / output projections around it run on the MFMA core.
#include "../../include/lego_hip.h"
#include "common.hpp"

namespace lego {

constexpr int kMaxL = 64;

template <int HD>
__global__ __launch_bounds__(64) void mhsa_fwd_kernel(const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off,
                                                      int n_cap, const int* __restrict__ n_dyn, int D, int heads,
                                                      float* __restrict__ out, int ldo, float* __restrict__ probs, int Lmax,
                                                      Dropout drop, int drop_cols) {
    __shared__ float Ks[kMaxL][HD + 1];
    __shared__ float Vs[kMaxL][HD + 1];
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int seg = blockIdx.x, h = blockIdx.y;
    if (seg >= n) return;
    const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;
    if (L <= 0) return;
    const int lane = threadIdx.x;
    const float scale = rsqrtf((float)HD);
    // stage K and V of this head (coalesced over the hd columns)
    for (int e = lane; e < L * HD; e += 64) {
        const int r = e / HD, c = e - r * HD;
        const float* row = qkv + (size_t)(beg + r) * ldq + h * HD + c;
        Ks[r][c] = row[D];
        Vs[r][c] = row[2 * D];
    }
    __syncthreads();
    if (lane >= L) return;
    float q[HD];
    const float* qrow = qkv + (size_t)(beg + lane) * ldq + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) q[c] = qrow[c] * scale;
    // pass 1: row max
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s += q[c] * Ks[j][c];
        mx = fmaxf(mx, s);
    }
    // pass 2: exp / sum, keep unnormalised probabilities in the probs buffer row
    float* prow = probs + ((size_t)(beg + lane) * heads + h) * Lmax;
    float se = 0.f;
    for (int j = 0; j < L; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s += q[c] * Ks[j][c];
        const float e = expf(s - mx);
        se += e;
        prow[j] = e;
    }
    const float inv = 1.f / se;
    float o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = 0.f;
    const int dcol = (beg + lane) * heads + h;
    for (int j0 = 0; j0 < L; j0 += 4) {
        float ds[4];
        dropout_scale4(drop, j0, dcol, drop_cols, ds);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            if (j < L) {
                const float p = prow[j] * inv;
                prow[j] = p;                       // saved softmax output (pre-dropout) for the backward pass
                const float pd = p * ds[u];
#pragma unroll
                for (int c = 0; c < HD; ++c) o[c] += pd * Vs[j][c];
            }
        }
    }
    float* orow = out + (size_t)(beg + lane) * ldo + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) orow[c] = o[c];
}

template <int HD>
__global__ __launch_bounds__(64) void mhsa_bwd_kernel(const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off,
                                                      int n_cap, const int* __restrict__ n_dyn, int D, int heads,
                                                      const float* __restrict__ gout, int ldgo, const float* __restrict__ probs,
                                                      int Lmax, Dropout drop, int drop_cols, float* __restrict__ gqkv, int ldgq) {
    __shared__ float Qs[kMaxL][HD + 1];
    __shared__ float Ks[kMaxL][HD + 1];
    __shared__ float Vs[kMaxL][HD + 1];
    __shared__ float Gs[kMaxL][HD + 1];
    __shared__ float Ps[kMaxL][kMaxL + 1];     // dropped probabilities Pd[i][j]
    __shared__ float Ss[kMaxL][kMaxL + 1];     // dS[i][j]
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int seg = blockIdx.x, h = blockIdx.y;
    if (seg >= n) return;
    const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;
    if (L <= 0) return;
    const int lane = threadIdx.x;
    const float scale = rsqrtf((float)HD);
    for (int e = lane; e < L * HD; e += 64) {
        const int r = e / HD, c = e - r * HD;
        const float* row = qkv + (size_t)(beg + r) * ldq + h * HD + c;
        Qs[r][c] = row[0];
        Ks[r][c] = row[D];
        Vs[r][c] = row[2 * D];
        Gs[r][c] = gout[(size_t)(beg + r) * ldgo + h * HD + c];
    }
    __syncthreads();
    if (lane < L) {                                   // lane = query row i
        const float* prow = probs + ((size_t)(beg + lane) * heads + h) * Lmax;
        const int dcol = (beg + lane) * heads + h;
        float g[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) g[c] = Gs[lane][c];
        float dot = 0.f;                              // sum_j dP[i,j] P[i,j]
        for (int j0 = 0; j0 < L; j0 += 4) {
            float ds[4];
            dropout_scale4(drop, j0, dcol, drop_cols, ds);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                if (j < L) {
                    float dpd = 0.f;
#pragma unroll
                    for (int c = 0; c < HD; ++c) dpd += g[c] * Vs[j][c];
                    const float p = prow[j];
                    const float dp = dpd * ds[u];
                    Ps[lane][j] = p * ds[u];
                    Ss[lane][j] = dp;                 // dP for now
                    dot += dp * p;
                }
            }
        }
        float dq[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] = 0.f;
        for (int j = 0; j < L; ++j) {
            const float dS = prow[j] * (Ss[lane][j] - dot);
            Ss[lane][j] = dS;
#pragma unroll
            for (int c = 0; c < HD; ++c) dq[c] += dS * Ks[j][c];
        }
        float* gq = gqkv + (size_t)(beg + lane) * ldgq + h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) gq[c] = dq[c] * scale;
    }
    __syncthreads();
    if (lane < L) {                                   // lane = key row j
        float dk[HD], dv[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        for (int i = 0; i < L; ++i) {
            const float dS = Ss[i][lane], pd = Ps[i][lane];
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                dk[c] += dS * Qs[i][c];
                dv[c] += pd * Gs[i][c];
            }
        }
        float* gk = gqkv + (size_t)(beg + lane) * ldgq + D + h * HD;
        float* gv = gqkv + (size_t)(beg + lane) * ldgq + 2 * D + h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) { gk[c] = dk[c] * scale; gv[c] = dv[c]; }
    }
}

static Dropout to_drop(const lego_dropout* d) {
    if (d != nullptr && d->p > 0.f) return Dropout{d->p, (uint32_t)d->seed, (uint32_t)(d->seed >> 32), d->site};
    return Dropout{0.f, 0u, 0u, 0u};
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
#define LAUNCH(HD) hipLaunchKernelGGL(mhsa_fwd_kernel<HD>, grid, block, 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, out, ldo, probs, Lmax, dr, dc)
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
    const int dc = rows_cap * heads;
    dim3 grid(n_cap, heads), block(64);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(HD) hipLaunchKernelGGL(mhsa_bwd_kernel<HD>, grid, block, 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, gout, ldgo, probs, Lmax, dr, dc, gqkv, ldgq)
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
