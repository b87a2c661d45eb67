// Multi-head self-attention core of NRMS's AttentionOperator over ragged segments (L <= 64 rows), on the matrix cores:
// softmax(Q K^T / sqrt(hd)) V and its backward per (segment, head) as v_mfma_f32_32x32x2_f32 tiles (exact f32).
// Reference: nn.MultiheadAttention inside model/operators/attention_operator.py:46-50.
//
// One WAVE per (segment, head), HPB heads per workgroup.  A wave first copies the Q / K / V (/ dOut / P) tiles of its head
// into its own LDS region with every global load in flight at once -- ONE memory round trip per wave; the first MFMA
// version fetched operands where it used them and paid six to eight dependent round trips in the backward pass (233 us per
// NRMS step) -- then:
//   * S^T = K Q^T: A operand = key row of lane li, B operand = query row of lane li, both read from LDS per MFMA step; the
//     accumulator then holds, for query i = lane, 16 of the 32 keys of a tile (the other 16 sit in lane i + 32), so the
//     softmax max / sum are in-lane reductions plus ONE cross-half shuffle;
//   * O = P V takes the probabilities straight from those accumulator registers as its A operand (the k order of an MFMA
//     reduction is free: step s pairs the keys that registers s of the two lane halves hold);
//   * the backward pass builds dP in both orientations (lane = query for dQ and the row dots, lane = key for dK / dV) with
//     the same two tricks; the row dots travel between the two through LDS.
// Segments of 33..64 rows run as 2 x 2 tiles (second instantiation, launched only when Lmax > 32).
// The saved probability carries the dropout decision in its sign bit (p >= 0: kept, stored -p: dropped), so the backward
// pass needs no random numbers.  Round 1 ran this on the vector ALU, one 64-thread block per (segment, head): 91 + 268 us
// per NRMS step for 0.7 + 1.7 GFLOP (lanes two-thirds used, a Philox call per four keys).
#include "../../include/lego_hip.h"
#include "common.hpp"

namespace lego {

constexpr int kMaxL = 64;

__device__ __forceinline__ int acc_row(int v, int lh) { return (v & 3) + 8 * (v >> 2) + 4 * lh; }   // row of accumulator register v

template <int HD> struct Tile { static constexpr int LD = HD + 4; };      // 16-B aligned rows, +4 floats of padding

// copy rows [0, L) x HD floats of a [*, ld] matrix into an LDS tile [LT][HD + 4]; rows L..LT-1 are zero-filled (they only feed
// accumulator entries that are masked or never stored, but they must be finite)
template <int HD, int LT>
__device__ __forceinline__ void stage_issue(const float* __restrict__ g, int ld, int L, int lane, f32x4 (&r)[LT * HD / 256]) {
#pragma unroll
    for (int t = 0; t < LT * HD / 256; ++t) {
        const int e = lane + 64 * t, row = e / (HD / 4), c4 = e % (HD / 4);
        r[t] = row < L ? *reinterpret_cast<const f32x4*>(g + (size_t)row * ld + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
template <int HD, int LT>
__device__ __forceinline__ void stage_commit(float* __restrict__ tile, int lane, const f32x4 (&r)[LT * HD / 256], float scale) {
#pragma unroll
    for (int t = 0; t < LT * HD / 256; ++t) {
        const int e = lane + 64 * t, row = e / (HD / 4), c4 = e % (HD / 4);
        *reinterpret_cast<f32x4*>(tile + row * Tile<HD>::LD + 4 * c4) = r[t] * scale;
    }
}

// acc += A B^T over the head dim: A row / B row of lane li (tiles in LDS), MFMA step s takes columns 2s + lh
template <int HD>
__device__ __forceinline__ void rows_mfma(const float* __restrict__ a_tile, const float* __restrict__ b_tile, int a_row, int b_row, int lh,
                                          f32x16& acc) {
    const float* a = a_tile + a_row * Tile<HD>::LD + lh;
    const float* b = b_tile + b_row * Tile<HD>::LD + lh;
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s], b[2 * s], acc, 0, 0, 0);
}

// out[m][c] += sum over the 32 rows of tile kt: coef(register s of this lane) * rows[kt * 32 + acc_row(s, lh)][c], c = lane column
template <int HD>
__device__ __forceinline__ void regs_mfma(const f32x16& coef, const float* __restrict__ rows, int kt, int li, int lh,
                                          f32x16 (&out)[(HD + 31) / 32]) {
    constexpr int CT = (HD + 31) / 32;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = min(ct * 32 + li, HD - 1);          // lanes past the head dim compute a duplicate column that is never stored
#pragma unroll
        for (int s = 0; s < 16; ++s)
            out[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(coef[s], rows[(kt * 32 + acc_row(s, lh)) * Tile<HD>::LD + c], out[ct], 0, 0, 0);
    }
}

// colsum[c] += scale * sum over the 32 rows of an accumulator tile (rows past the segment hold exact zeros): the bias gradient of
// the in-projection, folded into the kernel that produces d(qkv) instead of a separate pass over [rows, 3D]
template <int HD>
__device__ __forceinline__ void col_add(const f32x16 (&t)[(HD + 31) / 32], float scale, float* dst, int li, int lh, bool live) {
#pragma unroll
    for (int ct = 0; ct < (HD + 31) / 32; ++ct) {
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) s += t[ct][v];
        s += __shfl_xor(s, 32, 64);
        const int c = ct * 32 + li;
        if (live && lh == 0 && c < HD) atomicAdd(dst + c, s * scale);
    }
}

template <int HD, int JT, int HPB>
__global__ __launch_bounds__(64 * HPB) void mhsa_fwd_kernel(
    const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off, int n_cap, const int* __restrict__ n_dyn, int D,
    int heads, float* __restrict__ out, int ldo, float* __restrict__ probs, int Lmax, Dropout drop) {
    constexpr int CT = (HD + 31) / 32, LT = 32 * JT, TS = LT * Tile<HD>::LD, NR = LT * HD / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int wave = threadIdx.x >> 6;
    const int h = blockIdx.y * HPB + wave;
    // short segments: one workgroup each.  Long ones (33..64 rows) are rare -- 4 % of the news items, a fifth of the users -- so
    // their instantiation runs on a SMALL grid whose workgroups scan the segment list for them (a full-size grid of mostly
    // dead workgroups cost 22-39 us per launch in dispatch alone)
    for (int seg = blockIdx.x; seg < n; seg += gridDim.x) {
    const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;
    if (L <= 0 || (JT == 1) != (L <= 32)) continue;     // the other instantiation handles this segment (block-uniform)
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int hc = min(h, heads - 1);                    // heads % HPB != 0: the spare waves redo the last head and store nothing
    const bool live = h < heads;
    float* Qs = smem + wave * 3 * TS;
    float* Ks = Qs + TS;
    float* Vs = Ks + TS;
    const float* qb = qkv + (size_t)beg * ldq + hc * HD;
    {
        f32x4 rq[NR], rk[NR], rv[NR];
        stage_issue<HD, LT>(qb, ldq, L, lane, rq);
        stage_issue<HD, LT>(qb + D, ldq, L, lane, rk);
        stage_issue<HD, LT>(qb + 2 * D, ldq, L, lane, rv);
        stage_commit<HD, LT>(Qs, lane, rq, rsqrtf((float)HD));
        stage_commit<HD, LT>(Ks, lane, rk, 1.f);
        stage_commit<HD, LT>(Vs, lane, rv, 1.f);
    }
    __syncthreads();
    float* ptile = probs + ((size_t)beg * heads + (size_t)hc * L) * Lmax;
    const bool dropping = drop.p > 0.f;
    const float dinv = dropping ? 1.f / (1.f - drop.p) : 1.f;
    const uint32_t thr16 = (uint32_t)(drop.p * 65536.0f);

    for (int it = 0; it < JT; ++it) {
        if (it * 32 >= L) break;
        const int i = it * 32 + li;
        f32x16 acc[JT];
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[jt][v] = 0.f;
            rows_mfma<HD>(Ks, Qs, jt * 32 + li, i, lh, acc[jt]);       // S^T[j][i], j = jt*32 + acc_row(v, lh), i = it*32 + li
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
                const uint32_t ctr = (uint32_t)((beg + min(i, L - 1)) * heads + hc);
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
                if (live && i < L && j < L) ptile[(size_t)j * L + i] = kept ? p : -p;      // sign bit = dropped
                acc[jt][v] = kept ? p * dinv : 0.f;
            }
            regs_mfma<HD>(acc[jt], Vs, jt, li, lh, o);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 32 + li;
            if (c >= HD || !live) continue;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = it * 32 + acc_row(v, lh);
                if (r < L) out[(size_t)(beg + r) * ldo + hc * HD + c] = o[ct][v];
            }
        }
    }
    __syncthreads();                                     // the tiles are restaged for the next segment
    }
}

template <int HD, int JT, int HPB>
__global__ __launch_bounds__(64 * HPB) void mhsa_bwd_kernel(
    const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off, int n_cap, const int* __restrict__ n_dyn, int D,
    int heads, const float* __restrict__ gout, int ldgo, const float* __restrict__ probs, int Lmax, float keep_scale,
    float* __restrict__ gqkv, int ldgq, float* colsum) {
    constexpr int CT = (HD + 31) / 32, LT = 32 * JT, TS = LT * Tile<HD>::LD, NR = LT * HD / 256, PLD = LT + 1;
    constexpr int WS = 4 * TS + LT * PLD + LT;               // floats per wave: Q K V dOut tiles, P tile, row dots
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int wave = threadIdx.x >> 6;
    const int h = blockIdx.y * HPB + wave;
    for (int seg = blockIdx.x; seg < n; seg += gridDim.x) {
    const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;
    if (L <= 0 || (JT == 1) != (L <= 32)) continue;
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int hc = min(h, heads - 1);
    const bool live = h < heads;
    float* Qs = smem + wave * WS;
    float* Ks = Qs + TS;
    float* Vs = Ks + TS;
    float* Gs = Vs + TS;
    float* Ps = Gs + TS;                                     // signed probabilities [key j][query i]
    float* dots = Ps + LT * PLD;                             // sum_j dP[i,j] Pd[i,j] of every query row
    const float scale = rsqrtf((float)HD);
    const float* qb = qkv + (size_t)beg * ldq + hc * HD;
    const float* ptile = probs + ((size_t)beg * heads + (size_t)hc * L) * Lmax;
    {
        f32x4 rq[NR], rk[NR], rv[NR], rg[NR];
        constexpr int NP = LT * LT / 64;
        float rp[NP];
        stage_issue<HD, LT>(qb, ldq, L, lane, rq);
        stage_issue<HD, LT>(qb + D, ldq, L, lane, rk);
        stage_issue<HD, LT>(qb + 2 * D, ldq, L, lane, rv);
        stage_issue<HD, LT>(gout + (size_t)beg * ldgo + hc * HD, ldgo, L, lane, rg);
#pragma unroll
        for (int t = 0; t < NP; ++t) { const int e = lane + 64 * t; rp[t] = e < L * L ? ptile[e] : 0.f; }
        stage_commit<HD, LT>(Qs, lane, rq, 1.f);
        stage_commit<HD, LT>(Ks, lane, rk, 1.f);
        stage_commit<HD, LT>(Vs, lane, rv, 1.f);
        stage_commit<HD, LT>(Gs, lane, rg, 1.f);
        for (int e = lane; e < LT * PLD; e += 64) Ps[e] = 0.f;             // entries outside the L x L tile read as "probability 0"
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NP; ++t) { const int e = lane + 64 * t; if (e < L * L) Ps[(e / L) * PLD + (e % L)] = rp[t]; }
    }
    __syncthreads();
    float* gq = gqkv + (size_t)beg * ldgq + hc * HD;

    // ---- orientation 1, lane = query i: dP^T tiles -> row dots, dS -> dQ
    for (int it = 0; it < JT; ++it) {
        if (it * 32 >= L) break;
        const int i = it * 32 + li;
        f32x16 dp[JT], ps[JT];
        float dot = 0.f;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                ps[jt][v] = Ps[(jt * 32 + acc_row(v, lh)) * PLD + i];     // signed: sign bit = dropped
                dp[jt][v] = 0.f;
            }
            rows_mfma<HD>(Vs, Gs, jt * 32 + li, i, lh, dp[jt]);            // dP^T[j][i] = V_j . dOut_i
#pragma unroll
            for (int v = 0; v < 16; ++v) dot += ps[jt][v] > 0.f ? dp[jt][v] * keep_scale * ps[jt][v] : 0.f;
        }
        dot += __shfl_xor(dot, 32, 64);
        if (lh == 0) dots[i] = dot;
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
            regs_mfma<HD>(dp[jt], Ks, jt, li, lh, dq);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 32 + li;
            if (c >= HD || !live) continue;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = it * 32 + acc_row(v, lh);
                if (r < L) gq[(size_t)r * ldgq + c] = dq[ct][v] * scale;
            }
        }
        if (colsum != nullptr) col_add<HD>(dq, scale, colsum + hc * HD, li, lh, live);         // in_proj_bias gradient, Q third
    }
    __syncthreads();
    // ---- orientation 2, lane = key j: dP tiles -> dK, dV
    for (int jt = 0; jt < JT; ++jt) {
        if (jt * 32 >= L) break;
        const int j = jt * 32 + li;
        f32x16 dk[CT], dv[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) { dk[ct][v] = 0.f; dv[ct][v] = 0.f; }
        for (int it = 0; it < JT; ++it) {
            if (it * 32 >= L) break;
            f32x16 dp, pd;
#pragma unroll
            for (int v = 0; v < 16; ++v) dp[v] = 0.f;
            rows_mfma<HD>(Gs, Vs, it * 32 + li, j, lh, dp);                // dP[i][j] = dOut_i . V_j, i = it*32 + acc_row(v, lh), j = lane
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = it * 32 + acc_row(v, lh);
                const float s = Ps[j * PLD + i];
                const float keep = s > 0.f ? keep_scale : 0.f;
                const float p = fabsf(s);
                dp[v] = p * (dp[v] * keep - dots[i]);                      // dS[i][j]
                pd[v] = p * keep;
            }
            regs_mfma<HD>(dp, Qs, it, li, lh, dk);                         // dK[j][c] += dS[i][j] Q[i][c]
            regs_mfma<HD>(pd, Gs, it, li, lh, dv);                         // dV[j][c] += Pd[i][j] dOut[i][c]
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 32 + li;
            if (c >= HD || !live) continue;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = jt * 32 + acc_row(v, lh);
                if (r < L) {
                    gq[(size_t)r * ldgq + D + c] = dk[ct][v] * scale;
                    gq[(size_t)r * ldgq + 2 * D + c] = dv[ct][v];
                }
            }
        }
        if (colsum != nullptr) {                                               // ... K and V thirds
            col_add<HD>(dk, scale, colsum + D + hc * HD, li, lh, live);
            col_add<HD>(dv, 1.f, colsum + 2 * D + hc * HD, li, lh, live);
        }
    }
    __syncthreads();
    }
}

template <int HD, int JT, int HPB>
constexpr size_t mhsa_lds(bool bwd) {
    constexpr int LT = 32 * JT, TS = LT * (HD + 4);
    return (size_t)HPB * (bwd ? 4 * TS + LT * (LT + 1) + LT : 3 * TS) * sizeof(float);
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
    hipStream_t st = (hipStream_t)stream;
    // short segments: 4 heads per workgroup (3 tiles of 4.6 KB per wave at hd = 32); 33..64 rows: 2 heads per workgroup
#define LAUNCH(HD) do { \
        { auto k = mhsa_fwd_kernel<HD, 1, 4>; constexpr size_t lds = mhsa_lds<HD, 1, 4>(false); \
          { static bool once = false; if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; } } \
          hipLaunchKernelGGL(k, dim3(n_cap, (heads + 3) / 4), dim3(256), lds, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, out, ldo, probs, Lmax, dr); } \
        if (Lmax > 32) { auto k = mhsa_fwd_kernel<HD, 2, 2>; constexpr size_t lds = mhsa_lds<HD, 2, 2>(false); \
          { static bool once = false; if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; } } \
          hipLaunchKernelGGL(k, dim3(n_cap < 256 ? n_cap : 256, (heads + 1) / 2), dim3(128), lds, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, out, ldo, probs, Lmax, dr); } } while (0)
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
                                  const lego_dropout* drop, int rows_cap, float* gqkv, int ldgq, float* colsum, void* stream) {
    LEGO_REQUIRE(heads > 0 && D % heads == 0, "lego_mhsa_core_bwd: D=%d not divisible by heads=%d", D, heads);
    LEGO_REQUIRE(Lmax <= kMaxL, "lego_mhsa_core_bwd: Lmax=%d exceeds %d", Lmax, kMaxL);
    LEGO_REQUIRE((ldq & 3) == 0 && (ldgo & 3) == 0 && (D & 3) == 0, "lego_mhsa_core_bwd: ldq=%d, ldgo=%d and D=%d must be multiples of 4", ldq, ldgo, D);
    if (n_cap <= 0) return 0;
    const int hd = D / heads;
    const Dropout dr = to_drop(drop);
    const float ks = dr.p > 0.f ? 1.f / (1.f - dr.p) : 1.f;     // the keep / drop decision itself is the sign of the saved probability
    (void)rows_cap;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(HD) do { \
        { auto k = mhsa_bwd_kernel<HD, 1, 2>; constexpr size_t lds = mhsa_lds<HD, 1, 2>(true); \
          { static bool once = false; if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; } } \
          hipLaunchKernelGGL(k, dim3(n_cap, (heads + 1) / 2), dim3(128), lds, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, gout, ldgo, probs, Lmax, ks, gqkv, ldgq, colsum); } \
        if (Lmax > 32) { auto k = mhsa_bwd_kernel<HD, 2, 1>; constexpr size_t lds = mhsa_lds<HD, 2, 1>(true); \
          { static bool once = false; if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; } } \
          hipLaunchKernelGGL(k, dim3(n_cap < 256 ? n_cap : 256, heads), dim3(64), lds, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, gout, ldgo, probs, Lmax, ks, gqkv, ldgq, colsum); } } while (0)
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
