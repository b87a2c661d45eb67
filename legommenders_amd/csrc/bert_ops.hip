// Row-wise kernels of the native BERT block (config 5: the BERT news encoder of config/model/bert-naml.yaml, reference call sites
// model/operators/once_operator.py:156-193, bert_operator.py:10-52; the arithmetic itself is the third-party `transformers`
// package's BertEmbeddings / BertSelfOutput / BertIntermediate / BertOutput; the test suite holds a CPU restatement):
//
//   lego_dropout_add_layernorm_fwd / _bwd   out = drop_post(LayerNorm(drop_pre(y) + resid) * gamma + beta)
//       BertSelfOutput / BertOutput: dense output -> Dropout -> + residual -> LayerNorm   (drop_pre)
//       BertEmbeddings: (inputs_embeds + position + token-type rows) -> LayerNorm -> Dropout  (drop_post)
//   lego_gelu_fwd / _bwd                     exact (erf) GELU of BertIntermediate and its derivative
//
// The products around them are the path's MFMA kernels (lego_linear_*), the attention core is lego_mhsa_core_* (head dim 64).
// All HBM-bound streaming kernels: one WAVE owns a group of 8 consecutive rows (the dropout keep bits of a group come from one
// Philox call per column, common.hpp dropout_draw8) and holds a row in registers -- 16-byte loads, two wave reductions per row
// (mean, centred variance: the two-pass form, as aten's CPU kernel), nothing staged through LDS in the forward pass.
#include "../../include/lego_hip.h"
#include "common.hpp"

namespace lego {

constexpr int LN_MAX_NJ = 4;          // float4 groups per lane: widths up to 1024

// keep bits of a lane's 4 * NJ columns for the 8 rows of group g8 (bit f of word [j][i] = row 8 * g8 + f kept)
template <int NJ>
__device__ __forceinline__ void ln_draw(const Dropout& d, int g8, int lane, int W, uint32_t (&bits)[NJ][4]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * (lane + 64 * j) + i;
            bits[j][i] = (d.p > 0.f && c < W) ? dropout_draw8(d, g8, c, W) : 0xFFu;
        }
}

template <int NJ>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ resid, int ldr,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                     Dropout dpre, Dropout dpost, float* __restrict__ out, int ldo,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int W) {
    const int lane = threadIdx.x & 63;
    const int waves = gridDim.x * (blockDim.x >> 6);
    const float inv_pre = dpre.p > 0.f ? 1.f / (1.f - dpre.p) : 1.f, inv_post = dpost.p > 0.f ? 1.f / (1.f - dpost.p) : 1.f;
    f32x4 gm[NJ], bt[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = 4 * (lane + 64 * j);
        gm[j] = c < W ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        bt[j] = c < W ? *reinterpret_cast<const f32x4*>(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float invW = 1.f / (float)W;
    for (int g8 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g8 * 8 < rows; g8 += waves) {
        uint32_t kpre[NJ][4], kpost[NJ][4];
        ln_draw<NJ>(dpre, g8, lane, W, kpre);
        ln_draw<NJ>(dpost, g8, lane, W, kpost);
#pragma unroll 1
        for (int f = 0; f < 8; ++f) {
            const int r = 8 * g8 + f;
            if (r >= rows) break;
            f32x4 v[NJ];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = 4 * (lane + 64 * j);
                v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c < W) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(y + (size_t)r * ldy + c);
                    const f32x4 b = resid != nullptr ? *reinterpret_cast<const f32x4*>(resid + (size_t)r * ldr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[j][i] = a[i] * (((kpre[j][i] >> f) & 1u) ? inv_pre : 0.f) + b[i];
                    s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
                }
            }
            const float mu = wave_sum(s) * invW;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (4 * (lane + 64 * j) < W)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float d = v[j][i] - mu; q += d * d; }
            const float rs = rsqrtf(wave_sum(q) * invW + eps);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = 4 * (lane + 64 * j);
                if (c < W) {
                    f32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        o[i] = ((v[j][i] - mu) * rs * gm[j][i] + bt[j][i]) * (((kpost[j][i] >> f) & 1u) ? inv_post : 0.f);
                    *reinterpret_cast<f32x4*>(out + (size_t)r * ldo + c) = o;
                }
            }
            if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
        }
    }
}

// backward: v = drop_pre(y) + resid is formed again from the saved y / resid rows and the redrawn keep bits; with xh = (v - mu) rs,
//   do = dout * post,  dxh = do * gamma,  dv = rs (dxh - mean(dxh) - xh mean(dxh xh)),  dresid = dv,  dy = dv * pre,
//   dgamma += sum_rows do xh,  dbeta += sum_rows do,  dybias += sum_rows dy   (per-wave register partials -> LDS fold over the workgroup's 4 waves -> one
//   atomicAdd per column and workgroup)
template <int NJ>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dout, int lddo, const float* __restrict__ y, int ldy,
                                                     const float* __restrict__ resid, int ldr, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, Dropout dpre,
                                                     Dropout dpost, float* __restrict__ dy, int lddy, float* __restrict__ dresid, int lddr,
                                                     float* dgamma, float* dbeta, float* dybias, int rows, int W) {
    __shared__ float red[4][3][LN_MAX_NJ * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int waves = gridDim.x * (blockDim.x >> 6);
    const float inv_pre = dpre.p > 0.f ? 1.f / (1.f - dpre.p) : 1.f, inv_post = dpost.p > 0.f ? 1.f / (1.f - dpost.p) : 1.f;
    f32x4 gm[NJ], ag[NJ], ab[NJ], ay[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = 4 * (lane + 64 * j);
        gm[j] = c < W ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        ag[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ay[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float invW = 1.f / (float)W;
    for (int g8 = blockIdx.x * (blockDim.x >> 6) + wave; g8 * 8 < rows; g8 += waves) {
        uint32_t kpre[NJ][4], kpost[NJ][4];
        ln_draw<NJ>(dpre, g8, lane, W, kpre);
        ln_draw<NJ>(dpost, g8, lane, W, kpost);
#pragma unroll 1
        for (int f = 0; f < 8; ++f) {
            const int r = 8 * g8 + f;
            if (r >= rows) break;
            const float mu = mean[r], rs = rstd[r];
            f32x4 xh[NJ], dx[NJ];
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = 4 * (lane + 64 * j);
                xh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                dx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c < W) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(y + (size_t)r * ldy + c);
                    const f32x4 b = resid != nullptr ? *reinterpret_cast<const f32x4*>(resid + (size_t)r * ldr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                    const f32x4 go = *reinterpret_cast<const f32x4*>(dout + (size_t)r * lddo + c);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = a[i] * (((kpre[j][i] >> f) & 1u) ? inv_pre : 0.f) + b[i];
                        const float g = go[i] * (((kpost[j][i] >> f) & 1u) ? inv_post : 0.f);
                        xh[j][i] = (v - mu) * rs;
                        ag[j][i] += g * xh[j][i];
                        ab[j][i] += g;
                        dx[j][i] = g * gm[j][i];
                        m1 += dx[j][i];
                        m2 += dx[j][i] * xh[j][i];
                    }
                }
            }
            m1 = wave_sum(m1) * invW;
            m2 = wave_sum(m2) * invW;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = 4 * (lane + 64 * j);
                if (c < W) {
                    f32x4 dv, dyv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        dv[i] = rs * (dx[j][i] - m1 - xh[j][i] * m2);
                        dyv[i] = dv[i] * (((kpre[j][i] >> f) & 1u) ? inv_pre : 0.f);
                        ay[j][i] += dyv[i];
                    }
                    if (dresid != nullptr) *reinterpret_cast<f32x4*>(dresid + (size_t)r * lddr + c) = dv;
                    if (dy != nullptr) *reinterpret_cast<f32x4*>(dy + (size_t)r * lddy + c) = dyv;
                }
            }
        }
    }
    // fold the four waves' partial column sums, one atomic per column and workgroup
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[wave][0][4 * (lane + 64 * j) + i] = ag[j][i];
            red[wave][1][4 * (lane + 64 * j) + i] = ab[j][i];
            red[wave][2][4 * (lane + 64 * j) + i] = ay[j][i];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        const float sg = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
        const float sb = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
        const float sy = red[0][2][c] + red[1][2][c] + red[2][2][c] + red[3][2][c];
        if (dgamma != nullptr) atomicAdd(dgamma + c, sg);
        if (dbeta != nullptr) atomicAdd(dbeta + c, sb);
        if (dybias != nullptr) atomicAdd(dybias + c, sy);       // column sums of dy: the bias gradient of the dense layer that produced y
    }
}

// exact GELU (torch.nn.functional.gelu, approximate='none'; BertConfig.hidden_act = "gelu"): g = z Phi(z), g' = Phi(z) + z phi(z)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ z, float* __restrict__ g, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(z)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = gelu_exact(v[k]);
        reinterpret_cast<f32x4*>(g)[i] = o;
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* dg, const float* __restrict__ z, float* dz, int64_t n4)   /* dz may alias dg */ {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(z)[i], go = reinterpret_cast<const f32x4*>(dg)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = go[k] * gelu_exact_grad(v[k]);
        reinterpret_cast<f32x4*>(dz)[i] = o;
    }
}

static int ln_grid(int rows) {
    const int groups = (rows + 7) / 8, wgs = (groups + 3) / 4;
    return wgs < 1024 ? (wgs < 1 ? 1 : wgs) : 1024;
}

}  // namespace lego

using namespace lego;

#define ST ((hipStream_t)stream)

extern "C" int lego_dropout_add_layernorm_fwd(const float* y, int ldy, const float* resid, int ldr, const float* gamma, const float* beta,
                                              float eps, const lego_dropout* drop_pre, const lego_dropout* drop_post, float* out, int ldo,
                                              float* mean, float* rstd, int rows, int width, void* stream) {
    LEGO_REQUIRE(width > 0 && (width & 3) == 0 && width <= 256 * LN_MAX_NJ && (ldy & 3) == 0 && (ldo & 3) == 0 && (resid == nullptr || (ldr & 3) == 0),
                 "lego_dropout_add_layernorm_fwd: width=%d (multiple of 4, <= %d), ldy=%d, ldr=%d, ldo=%d must be multiples of 4", width,
                 256 * LN_MAX_NJ, ldy, ldr, ldo);
    if (rows <= 0) return 0;
    const Dropout a = make_dropout(drop_pre), b = make_dropout(drop_post);
    const int nj = (width + 255) / 256;
#define GO(NJ) hipLaunchKernelGGL((ln_fwd_kernel<NJ>), dim3(ln_grid(rows)), dim3(256), 0, ST, y, ldy, resid, ldr, gamma, beta, eps, a, b, out, ldo, mean, rstd, rows, width)
    switch (nj) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; default: GO(4); break; }
#undef GO
    return check_launch("lego_dropout_add_layernorm_fwd");
}

extern "C" int lego_dropout_add_layernorm_bwd(const float* dout, int lddo, const float* y, int ldy, const float* resid, int ldr,
                                              const float* gamma, const float* mean, const float* rstd, const lego_dropout* drop_pre,
                                              const lego_dropout* drop_post, float* dy, int lddy, float* dresid, int lddr, float* dgamma,
                                              float* dbeta, float* dybias, int rows, int width, void* stream) {
    LEGO_REQUIRE(width > 0 && (width & 3) == 0 && width <= 256 * LN_MAX_NJ && (ldy & 3) == 0 && (lddo & 3) == 0 && (resid == nullptr || (ldr & 3) == 0) &&
                 (dy == nullptr || (lddy & 3) == 0) && (dresid == nullptr || (lddr & 3) == 0),
                 "lego_dropout_add_layernorm_bwd: width=%d (multiple of 4, <= %d) and every leading dimension must be a multiple of 4", width,
                 256 * LN_MAX_NJ);
    if (rows <= 0) return 0;
    const Dropout a = make_dropout(drop_pre), b = make_dropout(drop_post);
    const int nj = (width + 255) / 256;
    const int grid = ln_grid(rows) < 512 ? ln_grid(rows) : 512;          // fewer, longer-lived workgroups: fewer column atomics
#define GO(NJ) hipLaunchKernelGGL((ln_bwd_kernel<NJ>), dim3(grid), dim3(256), 0, ST, dout, lddo, y, ldy, resid, ldr, gamma, mean, rstd, a, b, dy, lddy, dresid, lddr, dgamma, dbeta, dybias, rows, width)
    switch (nj) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; default: GO(4); break; }
#undef GO
    return check_launch("lego_dropout_add_layernorm_bwd");
}

extern "C" int lego_gelu_fwd(const float* z, float* g, int64_t n, void* stream) {
    LEGO_REQUIRE((n & 3) == 0, "lego_gelu_fwd: n=%lld must be a multiple of 4", (long long)n);
    if (n <= 0) return 0;
    const int64_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid), dim3(256), 0, ST, z, g, n4);
    return check_launch("lego_gelu_fwd");
}

extern "C" int lego_gelu_bwd(const float* dg, const float* z, float* dz, int64_t n, void* stream) {
    LEGO_REQUIRE((n & 3) == 0, "lego_gelu_bwd: n=%lld must be a multiple of 4", (long long)n);
    if (n <= 0) return 0;
    const int64_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid), dim3(256), 0, ST, dg, z, dz, n4);
    return check_launch("lego_gelu_bwd");
}
