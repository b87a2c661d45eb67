// Parameter-space products of the folded AttentionOperator block (engine.py, NrmsEngine fold_linear): everything here is D x D /
// A x D work on WEIGHTS -- no sequence row is touched -- so launch latency, not throughput, is what it costs.  The first version
// issued these products one launch each through the generic GEMM entry points (13 launches + 3 multi-tensor torch ops per operator
// and backward pass, all on the side stream) and the NRMS step lost 100 us to them (1.31 -> 1.21 ms with the launches removed).
// Here each dependency level is ONE launch: a grid of 16 x 16 output tiles (one wave each) over all products of the level; the vector
// terms are 1 x N products.
// Reference: AttentionOperator.forward, model/operators/attention_operator.py:49-56 (the out-projection inside
// nn.MultiheadAttention, then self.linear, then AdditiveAttention, model/common/attention.py:31-38) and its autograd backward.
#include "../../include/lego_hip.h"
#include "common.hpp"

namespace lego {

constexpr int FT = 16;          // output tile edge: one wave, one v_mfma_f32_16x16x4_f32 accumulator

// C[M,N] (+)= sum_k A(m,k) B(k,n) + u[m] v[n] + addv[n]; element strides express NN / NT / TN on row-major operands (one of the two
// strides of an operand is 1).  The vector terms of a level are products with M = 1 (K = 0: a plain add), so that they get a tile's
// wave per 16 outputs instead of one thread per output walking K on its own (40 us for two 256 x 256 vector terms)
struct FoldGemm {
    const float* A; int a_sm, a_sk;
    const float* B; int b_sk, b_sn;
    float* C; int ldc;
    const float* u; const float* v;   // nullable rank-1 term
    const float* addv;                 // nullable: + addv[n] on every row
    int M, N, K;
    int overwrite;                     // 1: C = ..., 0: C += ...
};
constexpr int kMaxGemm = 4;
struct FoldLevel {
    FoldGemm g[kMaxGemm];
    int n_gemm;
    int tiles[kMaxGemm + 1];           // prefix sums of the products' tile counts
};

// One WAVE per 16 x 16 output tile, v_mfma_f32_16x16x4_f32 with both operands read straight from global memory (they are a few
// hundred KB and L2-resident), no LDS, ~40 VGPRs: a launch has to START while a row-strip product or the attention core holds
// most of every SIMD's registers and LDS -- the first version (256-thread workgroups, 33 KB of LDS, 192 then 80 VGPRs) sat behind the
// 130 us strip product it was meant to run beside.  Out-of-range elements: clamped address, multiplied by 0 (a conditional load
// becomes one branch + full wait per element; weights are finite).
__device__ __forceinline__ void fold_tile(const FoldGemm& p, int tile) {
    const int tn = (p.N + FT - 1) / FT;
    const int m0 = (tile / tn) * FT, n0 = (tile % tn) * FT;
    const int lane = threadIdx.x, l16 = lane & 15, g4 = lane >> 4;
    const int am = min(m0 + l16, p.M - 1), bn = min(n0 + l16, p.N - 1);
    const float fa = m0 + l16 < p.M ? 1.f : 0.f, fb = n0 + l16 < p.N ? 1.f : 0.f;
    const float* pa = p.A + (size_t)am * p.a_sm;
    const float* pb = p.B + (size_t)bn * p.b_sn;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 16;                                   // k steps (of 4) whose 2 x 16 loads are in flight together
    for (int k0 = 0; k0 < p.K; k0 += 4 * U) {
        float ra[U], rb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(k0 + 4 * u + g4, p.K - 1);
            ra[u] = pa[(size_t)k * p.a_sk];
            rb[u] = pb[(size_t)k * p.b_sk];
        }
        __builtin_amdgcn_sched_barrier(0);                  // every load of the batch is issued before the first product waits
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float in = k0 + 4 * u + g4 < p.K ? 1.f : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u] * (fa * in), rb[u] * fb, acc, 0, 0, 0);
        }
    }
    const int n = n0 + l16;                                  // lane holds column l16, rows 4 * g4 + i
    if (n >= p.N) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * g4 + i;
        if (m >= p.M) continue;
        float x = acc[i];
        if (p.u != nullptr) x += p.u[m] * p.v[n];
        if (p.addv != nullptr) x += p.addv[n];
        float* c = p.C + (size_t)m * p.ldc + n;
        *c = p.overwrite ? x : *c + x;
    }
}

__global__ __launch_bounds__(64) void fold_level_kernel(FoldLevel L) {
    const int b = blockIdx.x;
    int i = 0;
    while (i + 1 < L.n_gemm && b >= L.tiles[i + 1]) ++i;
    fold_tile(L.g[i], b - L.tiles[i]);
}

static int launch_level(FoldLevel& L, hipStream_t st, const char* what) {
    L.tiles[0] = 0;
    for (int i = 0; i < L.n_gemm; ++i)
        L.tiles[i + 1] = L.tiles[i] + ((L.g[i].M + FT - 1) / FT) * ((L.g[i].N + FT - 1) / FT);
    hipLaunchKernelGGL(fold_level_kernel, dim3(L.tiles[L.n_gemm]), dim3(64), 0, st, L);
    return check_launch(what);
}

// y[N] (+)= x[K] . W(k,n) + add   as a 1 x N product
static FoldGemm vec_term(const float* x, const float* W, int w_sk, int w_sn, const float* add, float* y, int N, int K, int overwrite) {
    return FoldGemm{x, 0, 1, W, w_sk, w_sn, y, N, nullptr, nullptr, add, 1, N, x != nullptr ? K : 0, overwrite};
}

}  // namespace lego

using namespace lego;

extern "C" int lego_attn_fold_prepare(const float* Wo, const float* bo, const float* Wl, const float* bl, const float* W1,
                                      const float* b1, float* Wc, float* bc, float* W2, float* b2, int D, int A, void* stream) {
    LEGO_REQUIRE(D > 0 && A >= 0, "lego_attn_fold_prepare: D=%d A=%d", D, A);
    hipStream_t st = (hipStream_t)stream;
    FoldLevel L{};
    L.n_gemm = 2;
    L.g[0] = FoldGemm{Wl, D, 1, Wo, D, 1, Wc, D, nullptr, nullptr, nullptr, D, D, D, 1};      // Wc = Wl Wo
    L.g[1] = vec_term(bo, Wl, 1, D, bl, bc, D, D, 1);                                            // bc = Wl bo + bl
    if (launch_level(L, st, "lego_attn_fold_prepare") != 0) return 1;
    if (W2 == nullptr || A == 0) return 0;
    FoldLevel M{};
    M.n_gemm = 2;
    M.g[0] = FoldGemm{W1, D, 1, Wc, D, 1, W2, D, nullptr, nullptr, nullptr, A, D, D, 1};      // W2 = W1 Wc
    M.g[1] = vec_term(bc, W1, 1, D, b1, b2, A, D, 1);                                            // b2 = W1 bc + b1
    return launch_level(M, st, "lego_attn_fold_prepare");
}

extern "C" int lego_attn_fold_grads(const float* Wo, const float* bo, const float* Wl, const float* W1, const float* Wc,
                                    const float* bc, const float* Tp, const float* sp, float* T, float* s,
                                    float* gWo, float* gbo, float* gWl, float* gbl, float* gW1, float* gb1,
                                    int D, int A, void* stream) {
    LEGO_REQUIRE(D > 0 && A >= 0, "lego_attn_fold_grads: D=%d A=%d", D, A);
    hipStream_t st = (hipStream_t)stream;
    if (W1 != nullptr && A > 0) {
        // level 1 (fold level 2 only): from Tp = dpre^T o and sp = colsum(dpre) to the additive hidden layer's gradients, and the
        // rest of dL/dWc, dL/dbc on top of what T / s already hold (d_out^T pooled, colsum(d_out))
        FoldLevel L{};
        L.n_gemm = 4;
        L.g[0] = FoldGemm{Tp, D, 1, Wc, 1, D, gW1, D, sp, bc, nullptr, A, D, D, 0};           // gW1 += Tp Wc^T + sp (x) bc
        L.g[1] = FoldGemm{W1, 1, D, Tp, D, 1, T, D, nullptr, nullptr, nullptr, D, D, A, 0};   // T   += W1^T Tp
        L.g[2] = vec_term(sp, W1, D, 1, nullptr, s, D, A, 0);                                   // s   += sp W1
        L.g[3] = vec_term(nullptr, nullptr, 0, 1, sp, gb1, A, 0, 0);                            // gb1 += sp
        if (launch_level(L, st, "lego_attn_fold_grads") != 0) return 1;
    }
    // level 2: T = dL/dWc, s = dL/dbc -> the two affine layers att = o Wo^T + bo, lin = att Wl^T + bl
    FoldLevel M{};
    M.n_gemm = 4;
    M.g[0] = FoldGemm{T, D, 1, Wo, 1, D, gWl, D, s, bo, nullptr, D, D, D, 0};                  // gWl += T Wo^T + s (x) bo
    M.g[1] = FoldGemm{Wl, 1, D, T, D, 1, gWo, D, nullptr, nullptr, nullptr, D, D, D, 0};      // gWo += Wl^T T
    M.g[2] = vec_term(s, Wl, D, 1, nullptr, gbo, D, D, 0);                                      // gbo += s Wl
    M.g[3] = vec_term(nullptr, nullptr, 0, 1, s, gbl, D, 0, 0);                                 // gbl += s
    return launch_level(M, st, "lego_attn_fold_grads");
}

// ---------------------------------------------------------------------------------------------------------------------------
// NRMS user "head" of a training step in one launch (fold level 2): the user vector from the pooled attention output, the dot
// predictor, CrossEntropy(label 0) and their backward down to the pooled vector --
//   u = Wc p + bc;  s_c = u . item_c;  loss += CE(s)[0] / B;  g = (softmax(s) - e_0) gscale;
//   d_user = sum_c g_c item_c;  d_item_c = g_c u;  d_pooled = Wc^T d_user
// (attention_operator.py:52-56 folded, dot_predictor.py:7-10, legommender.py:254,263 and autograd) -- five dependent launches of
// ~9 us each on the step's critical path before.  One workgroup per impression, waves over the rows of Wc with 16-byte coalesced
// loads; Wc is read twice per workgroup from L2.
namespace lego {

constexpr int kHeadMaxD = 1024, kHeadMaxC = 64;

constexpr int HR = 8;          // rows of Wc whose 16-byte loads a wave has in flight together (16 spill at 1024 threads)
constexpr int HW = 16;         // waves per workgroup: at D = 256 every wave has two batches of HR rows per pass over Wc

__global__ __launch_bounds__(64 * HW) void nrms_user_head_kernel(
    const float* __restrict__ pooled, int ldp, const float* __restrict__ Wc, const float* __restrict__ bc,
    const float* __restrict__ items, int ldi, int B, int C, int D, float gscale,
    float* __restrict__ user, int ldu, float* __restrict__ scores, float* loss,
    float* __restrict__ d_user, int lddu, float* __restrict__ d_items, int lddi, float* __restrict__ d_pooled, int lddp,
    const int* __restrict__ seg_off) {
    __shared__ __attribute__((aligned(16))) float p[kHeadMaxD];
    __shared__ __attribute__((aligned(16))) float red[HW][256];
    __shared__ float u[kHeadMaxD], du[kHeadMaxD], s[kHeadMaxC], g[kHeadMaxC];
    constexpr int NT = 64 * HW;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a user WITHOUT clicked items: the un-folded operator pools over no rows and yields the zero vector (the reference's
    // nn.MultiheadAttention yields NaN there: no value to match).  The folded form would give u = bc -- a third answer -- so the
    // empty segment is made the same constant zero: u = 0, and d_user = 0 keeps it out of the bias gradient (colsum of d_user)
    const bool empty = seg_off != nullptr && seg_off[b + 1] <= seg_off[b];
    for (int k = tid; k < D; k += NT) p[k] = pooled[(size_t)b * ldp + k];
    __syncthreads();
    // u[n] = Wc[n, :] . p + bc[n]: one wave per row, lanes over k; HR rows' loads in flight per wave (one row at a time is one
    // memory round trip per row)
    for (int n0 = wave * HR; n0 < D; n0 += HW * HR) {
        float acc[HR];
#pragma unroll
        for (int i = 0; i < HR; ++i) acc[i] = 0.f;
        for (int k = 4 * lane; k < D; k += 256) {
            f32x4 w[HR];
#pragma unroll
            for (int i = 0; i < HR; ++i) w[i] = *reinterpret_cast<const f32x4*>(Wc + (size_t)min(n0 + i, D - 1) * D + k);
            const f32x4 pv = *reinterpret_cast<const f32x4*>(p + k);
#pragma unroll
            for (int i = 0; i < HR; ++i) acc[i] += w[i].x * pv.x + w[i].y * pv.y + w[i].z * pv.z + w[i].w * pv.w;
        }
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const float v = wave_sum(acc[i]);
            const int n = n0 + i;
            if (lane == 0 && n < D) { const float x = empty ? 0.f : v + bc[n]; u[n] = x; user[(size_t)b * ldu + n] = x; }
        }
    }
    __syncthreads();
    for (int c = wave; c < C; c += HW) {
        const float* it = items + (size_t)(b * C + c) * ldi;
        float acc = 0.f;
        for (int k = lane; k < D; k += 64) acc += u[k] * it[k];
        acc = wave_sum(acc);
        if (lane == 0) { s[c] = acc; scores[b * C + c] = acc; }
    }
    __syncthreads();
    if (tid == 0) {
        float mx = -INFINITY, se = 0.f;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, s[c]);
        for (int c = 0; c < C; ++c) se += expf(s[c] - mx);
        if (loss != nullptr) atomicAdd(loss, (logf(se) + mx - s[0]) / (float)B);
        for (int c = 0; c < C; ++c) g[c] = (expf(s[c] - mx) / se - (c == 0 ? 1.f : 0.f)) * gscale;
    }
    __syncthreads();
    for (int n = tid; n < D; n += NT) {
        float acc = 0.f;
        const float un = u[n];
        for (int c = 0; c < C; ++c) {
            acc += g[c] * items[(size_t)(b * C + c) * ldi + n];
            d_items[(size_t)(b * C + c) * lddi + n] = g[c] * un;
        }
        if (empty) acc = 0.f;
        du[n] = acc;
        d_user[(size_t)b * lddu + n] = acc;
    }
    __syncthreads();
    // d_pooled[k] = sum_n du[n] Wc[n][k]: lanes over k (coalesced rows of Wc, 4 columns each), the waves split n and fold through LDS
    for (int k0 = 0; k0 < D; k0 += 256) {
        const int k = k0 + 4 * lane;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (k < D) {
            for (int n0 = wave * HR; n0 < D; n0 += HW * HR) {
                f32x4 w[HR];
#pragma unroll
                for (int i = 0; i < HR; ++i) w[i] = *reinterpret_cast<const f32x4*>(Wc + (size_t)min(n0 + i, D - 1) * D + k);
#pragma unroll
                for (int i = 0; i < HR; ++i) acc += (n0 + i < D ? du[n0 + i] : 0.f) * w[i];
            }
        }
        if (k0 > 0) __syncthreads();
        *reinterpret_cast<f32x4*>(&red[wave][4 * lane]) = acc;
        __syncthreads();
        if (wave == 0 && k < D) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < HW; ++w) t += *reinterpret_cast<const f32x4*>(&red[w][4 * lane]);
            *reinterpret_cast<f32x4*>(d_pooled + (size_t)b * lddp + k) = t;
        }
    }
}

}  // namespace lego

extern "C" int lego_nrms_user_head_train(const float* pooled, int ldp, const float* Wc, const float* bc, const float* items, int ldi,
                                         int B, int C, int D, float gscale, float* user, int ldu, float* scores, float* loss,
                                         float* d_user, int lddu, float* d_items, int lddi, float* d_pooled, int lddp,
                                         const int32_t* seg_off, void* stream) {
    LEGO_REQUIRE(D > 0 && (D & 3) == 0 && D <= kHeadMaxD && C > 0 && C <= kHeadMaxC && (lddp & 3) == 0,
                 "lego_nrms_user_head_train: D=%d (multiple of 4, <= %d), C=%d (<= %d), lddp=%d", D, kHeadMaxD, C, kHeadMaxC, lddp);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(nrms_user_head_kernel, dim3(B), dim3(64 * HW), 0, (hipStream_t)stream, pooled, ldp, Wc, bc, items, ldi, B, C, D, gscale,
                       user, ldu, scores, loss, d_user, lddu, d_items, lddi, d_pooled, lddp, seg_off);
    return check_launch("lego_nrms_user_head_train");
}
