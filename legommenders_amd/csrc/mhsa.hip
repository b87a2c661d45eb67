// Multi-head self-attention core of NRMS's AttentionOperator over ragged segments (L <= 64 rows), on the matrix cores:
// softmax(Q K^T / sqrt(hd)) V and its backward per (segment, head) as v_mfma_f32_32x32x2_f32 tiles (exact f32).
// Reference: nn.MultiheadAttention inside model/operators/attention_operator.py:46-50.
//
// One WAVE per (segment, head), four heads per workgroup, no LDS operand staging and no barrier in the forward pass:
//   * S^T = K Q^T: the A operand is the key row held by lane li (its hd floats sit in registers, loaded with 16-B loads),
//     the B operand the query row of lane li; the accumulator then holds, for query i = lane, 16 of the 32 keys of a tile
//     (the other 16 sit in lane i + 32), so the softmax max / sum are in-lane reductions plus ONE cross-half shuffle;
//   * O = P V takes the probabilities straight from those accumulator registers as its A operand (the k order of an MFMA
//     reduction is free: step s pairs the keys that registers s of the two lane halves hold), V rows come as coalesced
//     128-B loads;
//   * the backward pass builds dP in both orientations (lane = query for dQ and the row dots, lane = key for dK / dV) with
//     the same two tricks; the row dots travel between the two through 256 B of LDS.
// Segments of 33..64 rows run as 2 x 2 tiles (second instantiation, launched only when Lmax > 32).
// The saved probability carries the dropout decision in its sign bit (p >= 0: kept, stored -p: dropped), so the backward
// pass needs no random numbers.  Round 1 ran this on the vector ALU, one 64-thread block per (segment, head): 91 + 268 us
// per NRMS step for 0.7 + 1.7 GFLOP (lanes two-thirds used, a Philox call per four keys).
#include "../../include/lego_hip.h"
#include "common.hpp"

namespace lego {

constexpr int kMaxL = 64;
constexpr int kHeadsPerBlock = 4;

__device__ __forceinline__ int acc_row(int v, int lh) { return (v & 3) + 8 * (v >> 2) + 4 * lh; }   // row of accumulator register v

// hd floats of row `row` (16-B loads; both lane halves hold the same row)
template <int HD>
__device__ __forceinline__ void load_row(const float* __restrict__ base, int ld, int row, float (&r)[HD]) {
    const float* p = base + (size_t)row * ld;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
        r[c] = v[0]; r[c + 1] = v[1]; r[c + 2] = v[2]; r[c + 3] = v[3];
    }
}

// acc += A B^T over the head dim: A row / B row of lane li in registers, MFMA step s takes columns 2s + lh
template <int HD>
__device__ __forceinline__ void rows_mfma(const float (&a)[HD], const float (&b)[HD], int lh, f32x16& acc) {
#pragma unroll
    for (int s = 0; s < HD / 2; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lh ? a[2 * s + 1] : a[2 * s], lh ? b[2 * s + 1] : b[2 * s], acc, 0, 0, 0);
}

// out[i][c] (+)= sum over the 32 keys of tile jt: coef(register s of this lane) * rows[jt * 32 + acc_row(s, lh)][c], c = lane column
template <int HD>
__device__ __forceinline__ void regs_mfma(const f32x16& coef, const float* __restrict__ rows, int ld, int jt, int L, int li, int lh,
                                          f32x16 (&out)[(HD + 31) / 32]) {
    constexpr int CT = (HD + 31) / 32;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = ct * 32 + li;
        float bv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int j = min(jt * 32 + acc_row(s, lh), L - 1);
            bv[s] = c < HD ? rows[(size_t)j * ld + c] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) out[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(coef[s], bv[s], out[ct], 0, 0, 0);
    }
}

template <int HD, int JT>
__global__ __launch_bounds__(64 * kHeadsPerBlock) void mhsa_fwd_kernel(
    const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off, int n_cap, const int* __restrict__ n_dyn, int D,
    int heads, float* __restrict__ out, int ldo, float* __restrict__ probs, int Lmax, Dropout drop) {
    constexpr int CT = (HD + 31) / 32;
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int seg = blockIdx.x, h = blockIdx.y * kHeadsPerBlock + (threadIdx.x >> 6);
    if (seg >= n || h >= heads) return;
    const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;
    if (L <= 0 || (JT == 1) != (L <= 32)) return;       // the other instantiation handles this segment
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const float scale = rsqrtf((float)HD);
    const float* qb = qkv + (size_t)beg * ldq + h * HD;
    const float* kb = qb + D;
    const float* vb = qb + 2 * D;
    float* ptile = probs + ((size_t)beg * heads + (size_t)h * L) * Lmax;

    float kreg[JT][HD];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) load_row<HD>(kb, ldq, min(jt * 32 + li, L - 1), kreg[jt]);
    const bool dropping = drop.p > 0.f;
    const float dinv = dropping ? 1.f / (1.f - drop.p) : 1.f;
    const uint32_t thr16 = (uint32_t)(drop.p * 65536.0f);

    for (int it = 0; it < JT; ++it) {
        if (it * 32 >= L) break;
        const int i = it * 32 + li;
        float qreg[HD];
        load_row<HD>(qb, ldq, min(i, L - 1), qreg);
#pragma unroll
        for (int c = 0; c < HD; ++c) qreg[c] *= scale;
        f32x16 acc[JT];
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[jt][v] = 0.f;
            rows_mfma<HD>(kreg[jt], qreg, lh, acc[jt]);                // S^T[j][i], j = jt*32 + acc_row(v, lh), i = it*32 + li
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                if (jt * 32 + acc_row(v, lh) >= L) acc[jt][v] = -INFINITY;
                mx = fmaxf(mx, acc[jt][v]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float se = 0.f;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[jt][v] = __expf(acc[jt][v] - mx); se += acc[jt][v]; }
        se += __shfl_xor(se, 32, 64);
        const float inv = 1.f / se;
        f32x16 o[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[ct][v] = 0.f;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            // keep bits of this lane's 16 (query, key) pairs: two Philox calls, one 16-bit field per decision
            uint32_t keep = 0xFFFFu;
            if (dropping) {
                keep = 0u;
                const uint32_t ctr = (uint32_t)((beg + min(i, L - 1)) * heads + h);
#pragma unroll
                for (int call = 0; call < 2; ++call) {
                    const Philox4 r = philox4x32_10(ctr, (uint32_t)((jt * 2 + lh) * 2 + call), drop.site, 0x6d687361u, drop.seed_lo, drop.seed_hi);
                    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int f = 0; f < 8; ++f)
                        keep |= (((w[f >> 1] >> (16 * (f & 1))) & 0xFFFFu) >= thr16 ? 1u : 0u) << (call * 8 + f);
                }
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float p = acc[jt][v] * inv;
                const bool kept = (keep >> v) & 1u;
                const int j = jt * 32 + acc_row(v, lh);
                if (i < L && j < L) ptile[(size_t)j * L + i] = kept ? p : -p;      // sign bit = dropped
                acc[jt][v] = kept ? p * dinv : 0.f;
            }
            regs_mfma<HD>(acc[jt], vb, ldq, jt, L, li, lh, o);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 32 + li;
            if (c >= HD) continue;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = it * 32 + acc_row(v, lh);
                if (r < L) out[(size_t)(beg + r) * ldo + h * HD + c] = o[ct][v];
            }
        }
    }
}

template <int HD, int JT>
__global__ __launch_bounds__(64 * kHeadsPerBlock) void mhsa_bwd_kernel(
    const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off, int n_cap, const int* __restrict__ n_dyn, int D,
    int heads, const float* __restrict__ gout, int ldgo, const float* __restrict__ probs, int Lmax, float keep_scale,
    float* __restrict__ gqkv, int ldgq) {
    constexpr int CT = (HD + 31) / 32;
    __shared__ float dots_all[kHeadsPerBlock][kMaxL];     // sum_j dP[i,j] Pd[i,j] of every query row, per wave
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int wave = threadIdx.x >> 6;
    const int seg = blockIdx.x, h = blockIdx.y * kHeadsPerBlock + wave;
    if (seg >= n || h >= heads) return;
    const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;
    if (L <= 0 || (JT == 1) != (L <= 32)) return;
    float* dots = dots_all[wave];
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const float scale = rsqrtf((float)HD);
    const float* qb = qkv + (size_t)beg * ldq + h * HD;
    const float* kb = qb + D;
    const float* vb = qb + 2 * D;
    const float* gb = gout + (size_t)beg * ldgo + h * HD;
    const float* ptile = probs + ((size_t)beg * heads + (size_t)h * L) * Lmax;
    float* gq = gqkv + (size_t)beg * ldgq + h * HD;

    // ---- orientation 1, lane = query i: dP^T tiles -> row dots, dS -> dQ
    {
        float vreg[JT][HD];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) load_row<HD>(vb, ldq, min(jt * 32 + li, L - 1), vreg[jt]);
        for (int it = 0; it < JT; ++it) {
            if (it * 32 >= L) break;
            const int i = it * 32 + li;
            float greg[HD];
            load_row<HD>(gb, ldgo, min(i, L - 1), greg);
            f32x16 dp[JT], ps[JT];
            float dot = 0.f;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int j = jt * 32 + acc_row(v, lh);
                    ps[jt][v] = (i < L && j < L) ? ptile[(size_t)j * L + i] : 0.f;     // signed: sign bit = dropped
                    dp[jt][v] = 0.f;
                }
                rows_mfma<HD>(vreg[jt], greg, lh, dp[jt]);             // dP^T[j][i] = V_j . dOut_i
#pragma unroll
                for (int v = 0; v < 16; ++v) dot += ps[jt][v] > 0.f ? dp[jt][v] * keep_scale * ps[jt][v] : 0.f;
            }
            dot += __shfl_xor(dot, 32, 64);
            if (lh == 0 && i < L) dots[i] = dot;
            f32x16 dq[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int v = 0; v < 16; ++v) dq[ct][v] = 0.f;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    dp[jt][v] = fabsf(ps[jt][v]) * ((ps[jt][v] > 0.f ? dp[jt][v] * keep_scale : 0.f) - dot);   // dS
                regs_mfma<HD>(dp[jt], kb, ldq, jt, L, li, lh, dq);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = ct * 32 + li;
                if (c >= HD) continue;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int r = it * 32 + acc_row(v, lh);
                    if (r < L) gq[(size_t)r * ldgq + c] = dq[ct][v] * scale;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();  // dots[] is written and read by this wave only: LDS executes a wave's accesses in order
    // ---- orientation 2, lane = key j: dP tiles -> dK, dV
    for (int jt = 0; jt < JT; ++jt) {
        if (jt * 32 >= L) break;
        const int j = jt * 32 + li;
        float vreg[HD];
        load_row<HD>(vb, ldq, min(j, L - 1), vreg);
        f32x16 dk[CT], dv[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) { dk[ct][v] = 0.f; dv[ct][v] = 0.f; }
        for (int it = 0; it < JT; ++it) {
            if (it * 32 >= L) break;
            float greg[HD];
            load_row<HD>(gb, ldgo, min(it * 32 + li, L - 1), greg);
            f32x16 dp, pd;
#pragma unroll
            for (int v = 0; v < 16; ++v) dp[v] = 0.f;
            rows_mfma<HD>(greg, vreg, lh, dp);                         // dP[i][j] = dOut_i . V_j, i = it*32 + acc_row(v, lh), j = lane
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = it * 32 + acc_row(v, lh);
                const float s = (i < L && j < L) ? ptile[(size_t)j * L + i] : 0.f;
                const float keep = s > 0.f ? keep_scale : 0.f;
                const float p = fabsf(s);
                dp[v] = p * (dp[v] * keep - dots[min(i, L - 1)]);     // dS[i][j]
                pd[v] = p * keep;
            }
            regs_mfma<HD>(dp, qb, ldq, it, L, li, lh, dk);             // dK[j][c] += dS[i][j] Q[i][c]
            regs_mfma<HD>(pd, gb, ldgo, it, L, li, lh, dv);            // dV[j][c] += Pd[i][j] dOut[i][c]
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 32 + li;
            if (c >= HD) continue;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = jt * 32 + acc_row(v, lh);
                if (r < L) {
                    gq[(size_t)r * ldgq + D + c] = dk[ct][v] * scale;
                    gq[(size_t)r * ldgq + 2 * D + c] = dv[ct][v];
                }
            }
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
    LEGO_REQUIRE((ldq & 3) == 0 && (D & 3) == 0, "lego_mhsa_core_fwd: ldq=%d and D=%d must be multiples of 4", ldq, D);
    if (n_cap <= 0) return 0;
    (void)rows_cap;
    const int hd = D / heads;
    const Dropout dr = to_drop(drop);
    dim3 grid(n_cap, (heads + kHeadsPerBlock - 1) / kHeadsPerBlock), block(64 * kHeadsPerBlock);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(HD) do { hipLaunchKernelGGL((mhsa_fwd_kernel<HD, 1>), grid, block, 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, out, ldo, probs, Lmax, dr); \
                        if (Lmax > 32) hipLaunchKernelGGL((mhsa_fwd_kernel<HD, 2>), grid, block, 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, out, ldo, probs, Lmax, dr); } while (0)
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
    LEGO_REQUIRE((ldq & 3) == 0 && (ldgo & 3) == 0 && (D & 3) == 0, "lego_mhsa_core_bwd: ldq=%d, ldgo=%d and D=%d must be multiples of 4", ldq, ldgo, D);
    if (n_cap <= 0) return 0;
    const int hd = D / heads;
    const Dropout dr = to_drop(drop);
    const float ks = dr.p > 0.f ? 1.f / (1.f - dr.p) : 1.f;     // the keep / drop decision itself is the sign of the saved probability
    (void)rows_cap;
    dim3 grid(n_cap, (heads + kHeadsPerBlock - 1) / kHeadsPerBlock), block(64 * kHeadsPerBlock);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(HD) do { hipLaunchKernelGGL((mhsa_bwd_kernel<HD, 1>), grid, block, 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, gout, ldgo, probs, Lmax, ks, gqkv, ldgq); \
                        if (Lmax > 32) hipLaunchKernelGGL((mhsa_bwd_kernel<HD, 2>), grid, block, 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, gout, ldgo, probs, Lmax, ks, gqkv, ldgq); } while (0)
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
