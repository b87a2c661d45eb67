// Multi-head self-attention core of NRMS's AttentionOperator over ragged segments (L <= 64 rows).
// One wave per (segment, head): lane = query row; K/V/Q of the head are staged in LDS, the
// L x L score tile lives in LDS, softmax row reductions are per-lane loops (the row is lane-local),
// column reductions of the backward pass (dK, dV) are re-mapped to lane = key row.
// ~1.5 % of NRMS's flops: VALU kernel; the QKV / output projections around it run on the MFMA core.
#include "../../include/lego_hip.h"
#include "common.hpp"

namespace lego {

constexpr int kMaxL = 64;

// K / V (forward, pass 1) and Q / dO (pass 2) rows are read with WAVE-UNIFORM addresses straight from global
// memory: hipcc turns them into scalar loads and the inner products into v_fma with an SGPR operand, so there is
// no LDS staging and no LDS-issue bottleneck (the first version, one broadcast ds_read per FMA, was LDS-bound).
template <int HD>
__global__ __launch_bounds__(64) void mhsa_fwd_kernel(const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off,
                                                      int n_cap, const int* __restrict__ n_dyn, int D, int heads,
                                                      float* __restrict__ out, int ldo, float* __restrict__ probs, int Lmax,
                                                      Dropout drop, int drop_cols) {
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int seg = blockIdx.x, h = blockIdx.y;
    if (seg >= n) return;
    const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;
    const int lane = threadIdx.x;
    if (L <= 0 || lane >= L) return;
    const float scale = rsqrtf((float)HD);
    const float* kbase = qkv + (size_t)beg * ldq + D + h * HD;         // wave-uniform
    const float* vbase = kbase + D;
    float q[HD];
    const float* qrow = qkv + (size_t)(beg + lane) * ldq + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) q[c] = qrow[c] * scale;
    float mx = -INFINITY;                              // pass 1: row max
    for (int j = 0; j < L; ++j) {
        const float* kr = kbase + (size_t)j * ldq;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s += q[c] * kr[c];
        mx = fmaxf(mx, s);
    }
    float* prow = probs + ((size_t)(beg + lane) * heads + h) * Lmax;
    float se = 0.f;                                    // pass 2: exp / sum, unnormalised probabilities parked in probs
    for (int j = 0; j < L; ++j) {
        const float* kr = kbase + (size_t)j * ldq;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s += q[c] * kr[c];
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
                const float* vr = vbase + (size_t)j * ldq;
                const float p = prow[j] * inv;
                prow[j] = p;                           // saved softmax output (pre-dropout) for the backward pass
                const float pd = p * ds[u];
#pragma unroll
                for (int c = 0; c < HD; ++c) o[c] += pd * vr[c];
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
    __shared__ float dotS[kMaxL];              // sum_j dP[i,j] P[i,j] of every query row
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int seg = blockIdx.x, h = blockIdx.y;
    if (seg >= n) return;
    const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;
    if (L <= 0) return;
    const int lane = threadIdx.x;
    const float scale = rsqrtf((float)HD);
    const float* qbase = qkv + (size_t)beg * ldq + h * HD;             // wave-uniform bases
    const float* kbase = qbase + D;
    const float* vbase = qbase + 2 * D;
    const float* gbase = gout + (size_t)beg * ldgo + h * HD;
    if (lane < L) {                            // ---- pass 1, lane = query row i: dot_i and dQ
        const float* prow = probs + ((size_t)(beg + lane) * heads + h) * Lmax;
        const int dcol = (beg + lane) * heads + h;
        float g[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) g[c] = gbase[(size_t)lane * ldgo + c];
        float dot = 0.f;
        for (int j0 = 0; j0 < L; j0 += 4) {
            float ds[4];
            dropout_scale4(drop, j0, dcol, drop_cols, ds);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                if (j < L) {
                    const float* vr = vbase + (size_t)j * ldq;
                    float dpd = 0.f;
#pragma unroll
                    for (int c = 0; c < HD; ++c) dpd += g[c] * vr[c];
                    dot += dpd * ds[u] * prow[j];
                }
            }
        }
        dotS[lane] = dot;
        float dq[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] = 0.f;
        for (int j0 = 0; j0 < L; j0 += 4) {
            float ds[4];
            dropout_scale4(drop, j0, dcol, drop_cols, ds);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                if (j < L) {
                    const float* vr = vbase + (size_t)j * ldq;
                    const float* kr = kbase + (size_t)j * ldq;
                    float dpd = 0.f;
#pragma unroll
                    for (int c = 0; c < HD; ++c) dpd += g[c] * vr[c];
                    const float dS = prow[j] * (dpd * ds[u] - dot);
#pragma unroll
                    for (int c = 0; c < HD; ++c) dq[c] += dS * kr[c];
                }
            }
        }
        float* gq = gqkv + (size_t)(beg + lane) * ldgq + h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) gq[c] = dq[c] * scale;
    }
    __syncthreads();
    if (lane < L) {                            // ---- pass 2, lane = key row j: dK[j], dV[j] (dS recomputed, probs read by column)
        float own[HD], dk[HD], dv[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) { own[c] = vbase[(size_t)lane * ldq + c]; dk[c] = 0.f; dv[c] = 0.f; }
        for (int i = 0; i < L; ++i) {
            const float* qr = qbase + (size_t)i * ldq;
            const float* gr = gbase + (size_t)i * ldgo;
            const float p = probs[((size_t)(beg + i) * heads + h) * Lmax + lane];
            const float dsc = dropout_scale1(drop, lane, (beg + i) * heads + h, drop_cols);
            float dpd = 0.f;
#pragma unroll
            for (int c = 0; c < HD; ++c) dpd += gr[c] * own[c];
            const float dS = p * (dpd * dsc - dotS[i]);
            const float pd = p * dsc;
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                dk[c] += dS * qr[c];
                dv[c] += pd * gr[c];
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
