// Multi-head self-attention core of NRMS's AttentionOperator over ragged segments (L <= 64 rows), on the matrix cores:
// softmax(Q K^T / sqrt(hd)) V and its backward per (segment, head) as v_mfma_f32_32x32x2_f32 tiles (exact f32).
// Reference: nn.MultiheadAttention inside model/operators/attention_operator.py:46-50.
//
// One WAVE (= one 64-thread workgroup) per (segment, head); a persistent grid strides over the (segment, head) list.  Every
// MFMA operand comes STRAIGHT from global memory into registers, in the two shapes the 32x32x2 instruction wants:
//   * "row per lane" (S^T = K Q^T, dP = dOut V^T): lane (li, lh) holds half a head-dim row -- the reduction order of an MFMA
//     chain is free, so step s of lane half lh takes column lh * hd/2 + s and the lane's hd/2 values are CONTIGUOUS (hd/8
//     16-byte loads);
//   * "column per lane" (O = P V, dQ = dS K, dK = dS^T Q, dV = Pd^T dOut): lane li holds column li of the 16 rows that its
//     accumulator registers index -- 16 dword loads, each a coalesced 128-byte row segment per lane half.
// The accumulator of S^T holds, for query i = lane, 16 of the 32 keys of a tile (the other 16 sit in lane i + 32), so the
// softmax max / sum are in-lane reductions plus ONE cross-half shuffle, and O = P V takes the probabilities straight from
// those registers as its A operand.  The backward pass builds dP in both orientations (lane = query for dQ and the row dots,
// lane = key for dK / dV); only the probability tile (read coalesced, needed transposed) and the 32 row dots go through LDS
// -- 4.3 KB per wave, so occupancy is set by registers (3-4 waves per SIMD), not by the 23 KB per wave of staged Q / K / V /
// dOut tiles the previous version held (6 waves per CU: 134 + 69 us per NRMS step for the item side; this one: see DESIGN).
// Segments of 33..64 rows run as 2 x 2 tiles in a second instantiation on a small grid (launched only when Lmax > 32).
// Round 6: segments of <= 16 rows (a third of the news items: titles of 5..13 tokens + [SEP] category [SEP]) take a 16 x 16 tile on
// v_mfma_f32_16x16x4_f32 inside the same launch (wave-uniform branch per pair): a quarter of the matrix-pipe cycles and a quarter of the
// per-lane softmax / mask / store work of the 32 x 32 tile, which is 75 % padding for them -- the kernel is issue-bound.
// The layout idea is the same at both sizes: `Tile<T>` below holds what differs (lane split, accumulator registers, the instruction).
// The dropout keep bit of (query row, head, key j) is the same at every tile size (keep_bits), so which tile a segment takes is invisible.
// Round 4: two ways to hand the softmax to the backward pass.  `probs`: the forward saves the (sign-tagged) probabilities, the backward
// reads them -- the engine's default, measured faster (backward 63 against 80 us).  `lse`: the forward keeps one float
// per (row, head) -- the log-sum-exp of the row's scaled scores -- and the backward recomputes S^T = K Q^T (hd/2 MFMAs per tile, operands
// the wave reads anyway, in their other register shape) and p = exp(s - lse), and redraws the dropout keep bits from the same Philox
// counters: no [rows, heads, L] round trip (32 MB written + 32 MB read per NRMS step).  LEGO_MHSA_RECOMPUTE=1 selects it in the engine;
// the plug-in route (kernels.mhsa_fwd / ops.mhsa) always takes it -- its saved tensors then have static shapes and are 30x smaller.
#include "../../include/lego_hip.h"
#include "common.hpp"

namespace lego {

constexpr int kMaxL = 64;

// What differs between the two tile sizes.  T = 32: v_mfma_f32_32x32x2_f32, lane = (li = lane % 32, lh = lane / 32 in 0..1), 16 accumulator
// registers, register v of lane (li, lh) = C[row (v & 3) + 8 (v >> 2) + 4 lh][column li].  T = 16: v_mfma_f32_16x16x4_f32, lane = (li = lane % 16,
// lh = lane / 16 in 0..3), 4 registers, register v = C[row v + 4 lh][column li].  In both, the A / B operand of lane (li, lh) is element
// [li][k = lh] / [k = lh][li] of a K = KL step, so a "row per lane" operand is the lane's HD / KL contiguous values of row li, and an accumulator
// register doubles as the A operand of the next product with its row index as that product's reduction index.
template <int T> struct Tile;
template <> struct Tile<32> {
    static constexpr int KL = 2, NV = 16, SH = 5;
    typedef f32x16 acc_t;
    static __device__ __forceinline__ int row0(int v) { return (v & 3) + 8 * (v >> 2); }
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float sum(float x) { return x + __shfl_xor(x, 32, 64); }
    static __device__ __forceinline__ float max(float x) { return fmaxf(x, __shfl_xor(x, 32, 64)); }
};
template <> struct Tile<16> {
    static constexpr int KL = 4, NV = 4, SH = 4;
    typedef f32x4 acc_t;
    static __device__ __forceinline__ int row0(int v) { return v; }
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float sum(float x) { x += __shfl_xor(x, 16, 64); return x + __shfl_xor(x, 32, 64); }
    static __device__ __forceinline__ float max(float x) { x = fmaxf(x, __shfl_xor(x, 16, 64)); return fmaxf(x, __shfl_xor(x, 32, 64)); }
};
template <int T> __device__ __forceinline__ int acc_row(int v, int lh) { return Tile<T>::row0(v) + 4 * lh; }   // row of accumulator register v

// Round 4: every operand access of a (segment, head) goes through a BUFFER descriptor with the hardware range check (guide T8 /
// T20) instead of per-lane clamps (loads) and per-store branches.  A `View` is one matrix seen from row 0 of the segment at the
// head's first column, valid up to the end of the matrix's row L - 1; an access to a row past the segment is out of range: loads
// return 0 (every product such a row enters meets an exact zero or lands in an entry that is masked or never stored), stores are
// dropped.  The wave-uniform part of an address (which of the rows an accumulator register indexes, which third of a qkv row)
// goes into the descriptor's base and record count -- scalar ALU work -- so the loads / stores of a "column per lane" tile share
// ONE per-lane byte offset.  (Before: 16 clamped 32-bit offsets per matrix pitch and a branch per store kept ~70 VGPRs alive from
// the top of the kernel to its last store; the backward with the score recomputation did not fit three waves per SIMD.)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct View { const float* p; int bytes; int ld; };
__device__ __forceinline__ View make_view(const float* p, int ld, int L, int cols_left) { return {p, ((L - 1) * ld + cols_left) * 4, ld}; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t view_rsrc(const View& v, int off_floats) {          // off_floats: wave-uniform
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(v.p + off_floats), 0, max(v.bytes - off_floats * 4, 0), 0x00020000);
}
// a per-lane byte offset the compiler may neither hoist nor re-derive: the row offsets of a "column per lane" tile are walked as ONE running
// register (off += step) instead of NV registers that live from the top of the pair to its last store
__device__ __forceinline__ int pin(int x) { asm volatile("" : "+v"(x)); return x; }
// rows from the previous register's row to register v's, key / row tile jt (the walk starts at row 0 of tile 0)
template <int T> __device__ __forceinline__ constexpr int row_step(int jt, int v) {
    return v > 0 ? Tile<T>::row0(v) - Tile<T>::row0(v - 1) : (jt > 0 ? T - Tile<T>::row0(Tile<T>::NV - 1) : 0);
}
// "row per lane" operand: columns [col0 + lh * HD/KL, col0 + (lh + 1) * HD/KL) of row `row` (zeros past the segment)
template <int HD, int T>
__device__ __forceinline__ void load_row(const View& v, int col0, int row, int lh, float (&r)[HD / Tile<T>::KL]) {
    constexpr int HH = HD / Tile<T>::KL;
    static_assert(HH % 4 == 0, "a lane's share of a head-dim row is loaded in 16-byte pieces");
    const __amdgpu_buffer_rsrc_t rs = view_rsrc(v, col0);
    const int off = (row * v.ld + lh * HH) * 4;
#pragma unroll
    for (int t = 0; t < HH / 4; ++t) {
        // (bit_cast of the WHOLE vector: hipcc 7.2 lowers __builtin_bit_cast(float, v.y) on a vector-element lvalue to a read of element 0)
        const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16 * t, 0, 0));
        r[4 * t] = f.x; r[4 * t + 1] = f.y; r[4 * t + 2] = f.z; r[4 * t + 3] = f.w;
    }
}
// "column per lane" operand: column col0 + min(ct * T + li, HD - 1) of rows tile * T + acc_row(s, lh), s = 0..NV-1
template <int HD, int T>
__device__ __forceinline__ void load_cols(const View& v, int col0, int tile, int li, int lh, float (&r)[(HD + T - 1) / T][Tile<T>::NV]) {
    const __amdgpu_buffer_rsrc_t rs = view_rsrc(v, col0);
    const int ldb = v.ld * 4;
#pragma unroll
    for (int ct = 0; ct < (HD + T - 1) / T; ++ct) {
        const int c = min(ct * T + li, HD - 1);                  // lanes past the head dim compute a duplicate column that is never stored
        int off = ((tile * T + 4 * lh) * v.ld + c) * 4;
#pragma unroll
        for (int s = 0; s < Tile<T>::NV; ++s) {
            off = pin(off + row_step<T>(0, s) * ldb);
            r[ct][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
        }
    }
}
// acc += A B^T over the head dim, both operands "row per lane"
template <int HD, int T>
__device__ __forceinline__ void rows_mfma(const float (&a)[HD / Tile<T>::KL], const float (&b)[HD / Tile<T>::KL], typename Tile<T>::acc_t& acc) {
#pragma unroll
    for (int s = 0; s < HD / Tile<T>::KL; ++s) acc = Tile<T>::mfma(a[s], b[s], acc);
}
// out[m][c] += sum over the T rows of a tile: coef(register s of this lane) * rows[acc_row(s, lh)][c]
template <int HD, int T>
__device__ __forceinline__ void regs_mfma(const typename Tile<T>::acc_t& coef, const float (&cols)[(HD + T - 1) / T][Tile<T>::NV],
                                          typename Tile<T>::acc_t (&out)[(HD + T - 1) / T]) {
#pragma unroll
    for (int ct = 0; ct < (HD + T - 1) / T; ++ct)
#pragma unroll
        for (int s = 0; s < Tile<T>::NV; ++s) out[ct] = Tile<T>::mfma(coef[s], cols[ct][s], out[ct]);
}
template <int T, int CT>
__device__ __forceinline__ void zero(typename Tile<T>::acc_t (&t)[CT]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int v = 0; v < Tile<T>::NV; ++v) t[ct][v] = 0.f;
}
// out[tile * T + acc_row(v, lh)][col0 + ct * T + li] = t * scale; rows past the segment and columns past the head are dropped by the
// range check (their offset is out of range)
template <int HD, int T>
__device__ __forceinline__ void store_cols(const typename Tile<T>::acc_t (&t)[(HD + T - 1) / T], float scale, const View& o, int col0, int tile, int li, int lh) {
    const __amdgpu_buffer_rsrc_t rs = view_rsrc(o, col0);
    const int ldb = o.ld * 4;
#pragma unroll
    for (int ct = 0; ct < (HD + T - 1) / T; ++ct) {
        const int c = ct * T + li;
        int off = c < HD ? ((tile * T + 4 * lh) * o.ld + c) * 4 : 0x70000000;            // (+ 31 rows of < 2^20 bytes stays out of range and positive)
#pragma unroll
        for (int v = 0; v < Tile<T>::NV; ++v) {
            off = pin(off + row_step<T>(0, v) * ldb);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t[ct][v] * scale), rs, off, 0, 0);
        }
    }
}

// colsum[c] += scale * sum over the T rows of an accumulator tile (rows past the segment hold exact zeros): the bias gradient of
// the in-projection, folded into the kernel that produces d(qkv) instead of a separate pass over [rows, 3D]
template <int HD, int T>
__device__ __forceinline__ void col_add(const typename Tile<T>::acc_t (&t)[(HD + T - 1) / T], float scale, float* dst, int li, int lh) {
#pragma unroll
    for (int ct = 0; ct < (HD + T - 1) / T; ++ct) {
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < Tile<T>::NV; ++v) s += t[ct][v];
        s = Tile<T>::sum(s);
        const int c = ct * T + li;
        if (lh == 0 && c < HD) atomicAdd(dst + c, s * scale);
    }
}

// Segments of more than 32 rows, listed once per plan (lego_mhsa_long_segments): with the list the long-segment launch gives every
// (segment, head) pair its own workgroup.  Without it a workgroup finds its pairs by a strided ballot over ALL pairs and walks the
// two or three it happens to own one after the other -- the launch then lasts as long as the unluckiest workgroup's chain
// (21 us forward / 55 us backward for the 4 % of news items whose sequence has 33 rows).
__global__ __launch_bounds__(1024) void mhsa_long_segments_kernel(const int* __restrict__ seg_off, int n_cap, const int* __restrict__ n_dyn,
                                                                  int* __restrict__ list, int* __restrict__ count) {
    __shared__ int fill;
    if (threadIdx.x == 0) fill = 0;
    __syncthreads();
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    for (int s = threadIdx.x; s < n; s += blockDim.x)
        if (seg_off[s + 1] - seg_off[s] > 32) list[atomicAdd(&fill, 1)] = s;
    __syncthreads();
    if (threadIdx.x == 0) *count = fill;
}

// The (segment, head) pairs a workgroup visits.  JT == 1 (segments of <= 32 rows, the bulk): one wave per workgroup, a grid stride
// over all pairs.  JT == 2 (33..64 rows: 4 % of the news items, a fifth of the users): TWO waves per workgroup, wave t owns row
// tile t of the pair.  Lane l of every wave looks at pair blockIdx.x + l * gridDim.x -- one round trip for up to 64 candidates, a
// stride over all pairs would pay a dependent seg_off load per miss -- and the workgroup walks the long ones of its ballot; the
// interleaved assignment spreads the rare long pairs evenly over the grid.
#define LEGO_MHSA_WALK(...)                                                                                     \
    if constexpr (JT == 1) {                                                                                      \
        for (int w = blockIdx.x; w < n * heads; w += gridDim.x) {                                                 \
            const int seg = w / heads, h = w - seg * heads;                                                       \
            const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;                                             \
            if (L <= 0 || L > 32) continue;                                                                       \
            __VA_ARGS__;                                                                                          \
        }                                                                                                         \
    } else if (long_list != nullptr) {                                                                            \
        const int cnt_ = min(*long_count, n) * heads;        /* the long segments were listed ahead of time */    \
        for (int w = blockIdx.x; w < cnt_; w += gridDim.x) {                                                      \
            const int seg = long_list[w / heads], h = w - (w / heads) * heads;                                    \
            const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;                                             \
            __VA_ARGS__;                                                                                          \
        }                                                                                                         \
    } else {                                                                                                      \
        const int l_ = threadIdx.x & 63;                                                                          \
        for (int base = blockIdx.x; base < n * heads; base += 64 * gridDim.x) {                                   \
            const int w_ = base + l_ * gridDim.x;                                                                 \
            bool lng_ = false;                                                                                    \
            if (w_ < n * heads) { const int s_ = w_ / heads; lng_ = seg_off[s_ + 1] - seg_off[s_] > min_len; }   \
            unsigned long long todo = __ballot(lng_);                                                             \
            while (todo != 0ull) {                                                                                \
                const int w = base + (__ffsll((long long)todo) - 1) * gridDim.x;                                  \
                todo &= todo - 1ull;                                                                              \
                const int seg = w / heads, h = w - seg * heads;                                                   \
                const int beg = seg_off[seg], L = seg_off[seg + 1] - beg;                                         \
                __VA_ARGS__;                                                                                      \
            }                                                                                                     \
        }                                                                                                         \
    }

// keep bits of a lane's NV (query, key) pairs of key tile jt, one 16-bit Philox field per decision.  The bit of (query row, head, key j) is
// a function of those three alone: a 32-row tile's lane (i, lh) draws two Philox calls (index (2 jt + lh) 2 + call) for its 16 keys
// j = 32 jt + (v & 3) + 8 (v >> 2) + 4 lh, field v & 7 of call v >> 3; a 16-row tile's lane (i, lh in 0..3) holds keys j = v + 4 lh < 16, which in that
// numbering are lane half lh & 1, register v + 4 (lh >> 1) < 8: ONE call (index 2 (lh & 1)), fields v + 4 (lh >> 1).  Forward and backward call it
// with the same (row, head, tile, lane part), so the backward pass redraws the forward's mask -- whatever tile size either took.
template <int T>
__device__ __forceinline__ uint32_t keep_bits(const Dropout& drop, uint32_t thr16, uint32_t ctr, int jt, int lh) {
    uint32_t keep = 0u;
    if constexpr (T == 32) {
#pragma unroll
        for (int call = 0; call < 2; ++call) {
            const Philox4 r = philox4x32_10(ctr, (uint32_t)((jt * 2 + lh) * 2 + call), drop.site, 0x6d687361u, drop.seed_lo, drop.seed_hi);
            const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int f = 0; f < 8; ++f)
                keep |= (((w[f >> 1] >> (16 * (f & 1))) & 0xFFFFu) >= thr16 ? 1u : 0u) << (call * 8 + f);
        }
    } else {
        const Philox4 r = philox4x32_10(ctr, (uint32_t)((jt * 2 + (lh & 1)) * 2), drop.site, 0x6d687361u, drop.seed_lo, drop.seed_hi);
        const uint32_t lo = (lh >> 1) ? r.z : r.x, hi = (lh >> 1) ? r.w : r.y;          // fields 4 (lh >> 1) .. 4 (lh >> 1) + 3
        keep = ((lo & 0xFFFFu) >= thr16 ? 1u : 0u) | ((lo >> 16) >= thr16 ? 2u : 0u) | ((hi & 0xFFFFu) >= thr16 ? 4u : 0u) | ((hi >> 16) >= thr16 ? 8u : 0u);
    }
    return keep;
}

template <int HD, int JT, int T>
__device__ __forceinline__ void mhsa_fwd_tile(const float* __restrict__ qkv, int ldq, int D, int heads, float* __restrict__ out, int ldo,
                                              float* __restrict__ lse, float* __restrict__ probs, int Lmax, const Dropout& drop, int h,
                                              int beg, int L, int t0, int t1) {
    using TL = Tile<T>;
    using acc_t = typename TL::acc_t;
    constexpr int CT = (HD + T - 1) / T, HH = HD / TL::KL, NV = TL::NV;
    const int lane = threadIdx.x & 63, li = lane & (T - 1), lh = lane >> TL::SH;
    const bool dropping = drop.p > 0.f;
    const float dinv = dropping ? 1.f / (1.f - drop.p) : 1.f;
    const uint32_t thr16 = (uint32_t)(drop.p * 65536.0f);
    const float qscale = rsqrtf((float)HD);
    {
        const View qv = make_view(qkv + (size_t)beg * ldq + h * HD, ldq, L, 3 * D - h * HD);
        const View ov = make_view(out + (size_t)beg * ldo + h * HD, ldo, L, D - h * HD);
        const bool save_p = probs != nullptr;                    // the (segment, head) tile [key j][query i], L x L floats
        const View pv = make_view(save_p ? probs + ((size_t)beg * heads + (size_t)h * L) * Lmax : qkv, L, save_p ? L : 0, save_p ? L : 0);
        const __amdgpu_buffer_rsrc_t prs = view_rsrc(pv, 0);
        float kr[JT][HH], vc[JT][CT][NV];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            load_row<HD, T>(qv, D, jt * T + li, lh, kr[jt]);
            load_cols<HD, T>(qv, 2 * D, jt, li, lh, vc[jt]);
        }
#pragma unroll 1
        for (int it = t0; it < t1; ++it) {
            if (it * T >= L) break;
            const int i = it * T + li;
            int poff = i < L ? (4 * lh * L + i) * 4 : 0x70000000;
            float qr[HH];
            load_row<HD, T>(qv, 0, i, lh, qr);
            __builtin_amdgcn_sched_barrier(0);      // all operand loads in flight before the first MFMA
#pragma unroll
            for (int s = 0; s < HH; ++s) qr[s] *= qscale;
            acc_t acc[JT];
            zero<T, JT>(acc);
            float mx = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                rows_mfma<HD, T>(kr[jt], qr, acc[jt]);                     // S^T[j][i], j = jt*T + acc_row(v, lh), i = it*T + li
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (jt * T + acc_row<T>(v, lh) >= L) acc[jt][v] = -INFINITY;
                    mx = fmaxf(mx, acc[jt][v]);
                }
            }
            mx = TL::max(mx);
            float se = 0.f;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int v = 0; v < NV; ++v) { acc[jt][v] = __expf(acc[jt][v] - mx); se += acc[jt][v]; }
            se = TL::sum(se);
            const float inv = 1.f / se;
            if (lse != nullptr && lh == 0 && i < L) lse[(size_t)(beg + i) * heads + h] = mx + __logf(se);   // all a recomputing backward needs
            acc_t o[CT];
            zero<T, CT>(o);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                uint32_t keep = 0xFFFFu;
                if (dropping) keep = keep_bits<T>(drop, thr16, (uint32_t)((beg + min(i, L - 1)) * heads + h), jt, lh);
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[jt][v] *= inv;                  // the probabilities
                if (save_p) {                                    // ONE wave-uniform branch around the NV stores (the pinned offset is a volatile
#pragma unroll                                                   // statement: inside the register loop it made the branch per register)
                    for (int v = 0; v < NV; ++v) {               // sign bit = dropped; keys / queries past the segment are out of range
                        poff = pin(poff + row_step<T>(jt, v) * L * 4);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ((keep >> v) & 1u) ? acc[jt][v] : -acc[jt][v]), prs, poff, 0, 0);
                    }
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[jt][v] = ((keep >> v) & 1u) ? acc[jt][v] * dinv : 0.f;
                regs_mfma<HD, T>(acc[jt], vc[jt], o);
            }
            store_cols<HD, T>(o, 1.f, ov, 0, it, li, lh);
        }
    }
}

// one (segment, head) pair: segments of <= 16 rows take the 16 x 16 tile (head dims whose quarter rows are whole 16-byte pieces)
template <int HD, int JT>
__device__ __forceinline__ void mhsa_fwd_pair(const float* __restrict__ qkv, int ldq, int D, int heads, float* __restrict__ out, int ldo,
                                              float* __restrict__ lse, float* __restrict__ probs, int Lmax, const Dropout& drop, int h,
                                              int beg, int L, int t0, int t1) {
    if constexpr (JT == 1 && HD >= 16) {
        if (L <= 16) {                                           // wave-uniform
            mhsa_fwd_tile<HD, 1, 16>(qkv, ldq, D, heads, out, ldo, lse, probs, Lmax, drop, h, beg, L, t0, t1);
            return;
        }
    }
    mhsa_fwd_tile<HD, JT, 32>(qkv, ldq, D, heads, out, ldo, lse, probs, Lmax, drop, h, beg, L, t0, t1);
}

template <int HD, int JT>
__global__ __launch_bounds__(64 * JT) __attribute__((amdgpu_waves_per_eu(JT == 1 && HD <= 32 ? 4 : 2))) void mhsa_fwd_kernel(
    const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off, int n_cap, const int* __restrict__ n_dyn, int D,
    int heads, float* __restrict__ out, int ldo, float* __restrict__ lse, float* __restrict__ probs, int Lmax, Dropout drop,
    const int* __restrict__ long_list, const int* __restrict__ long_count, int min_len) {
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int t0 = JT == 1 ? 0 : (int)(threadIdx.x >> 6);           // JT == 2: wave t owns query tile t
    LEGO_MHSA_WALK(mhsa_fwd_pair<HD, JT>(qkv, ldq, D, heads, out, ldo, lse, probs, Lmax, drop, h, beg, L, t0, t0 + 1))
}

// RC (recompute): the probabilities come from S^T = K Q^T and the saved log-sum-exp rows, the keep bits from the forward's Philox
// counters (no [rows, heads, L] tensor exists).  !RC: the forward pass saved the signed probabilities (`probs`), read here through a
// descriptor of the (segment, head) tile.  Same-box measurement (tools/mhsa_bulk_probe.py): see DESIGN.md section 4.
template <int HD, int JT, bool RC, int T>
__device__ __forceinline__ void mhsa_bwd_tile(const float* __restrict__ qkv, int ldq, int D, int heads, const float* __restrict__ gout, int ldgo,
                                              const float* __restrict__ lse, const float* __restrict__ probs, int Lmax, const Dropout& drop,
                                              float keep_scale, float* __restrict__ gqkv, int ldgq,
                                              float* colsum, float* __restrict__ Pd, float* __restrict__ Ds, int h, int beg, int L, int t0, int t1) {
    using TL = Tile<T>;
    using acc_t = typename TL::acc_t;
    constexpr int CT = (HD + T - 1) / T, HH = HD / TL::KL, NV = TL::NV, LT = T * JT, PLD = LT + 1;
    const int lane = threadIdx.x & 63, li = lane & (T - 1), lh = lane >> TL::SH;
    const float scale = rsqrtf((float)HD);
    const bool dropping = drop.p > 0.f;
    const uint32_t thr16 = (uint32_t)(drop.p * 65536.0f);
    const View qv = make_view(qkv + (size_t)beg * ldq + h * HD, ldq, L, 3 * D - h * HD);
    const View gv = make_view(gout + (size_t)beg * ldgo + h * HD, ldgo, L, D - h * HD);
    const View dv_ = make_view(gqkv + (size_t)beg * ldgq + h * HD, ldgq, L, 3 * D - h * HD);
    __syncthreads();                                         // the previous pair's readers of Pd / Ds are done
    // ---- orientation 1, lane = query i: dP^T tiles -> row dots -> dS^T (kept for orientation 2 in LDS) -> dQ
    {
        float kr[RC ? JT : 1][HH], vr[JT][HH], kc[JT][CT][NV];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            if constexpr (RC) load_row<HD, T>(qv, D, jt * T + li, lh, kr[jt]);
            load_row<HD, T>(qv, 2 * D, jt * T + li, lh, vr[jt]);
            load_cols<HD, T>(qv, D, jt, li, lh, kc[jt]);
        }
#pragma unroll 1
        for (int it = t0; it < t1; ++it) {
            if (it * T >= L) break;
            const int i = it * T + li;
            float qr[RC ? HH : 1], gr[HH];
            acc_t dp[JT], ps[JT];
            float lse_i = 0.f;
            if constexpr (RC) {
                load_row<HD, T>(qv, 0, i, lh, qr);
                lse_i = lse[(size_t)(beg + min(i, L - 1)) * heads + h];
            } else {
                // the saved tile [key j][query i] (L x L floats) as a view of L rows of L columns: keys past the segment are out of
                // range, and so is every lane whose query is (offset past the tile)
                const View pv = make_view(probs + ((size_t)beg * heads + (size_t)h * L) * Lmax, L, L, L);
                const __amdgpu_buffer_rsrc_t prs = view_rsrc(pv, 0);
                int off = i < L ? (4 * lh * L + i) * 4 : 0x70000000;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        off = pin(off + row_step<T>(jt, v) * L * 4);
                        ps[jt][v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, off, 0, 0));
                    }
            }
            load_row<HD, T>(gv, 0, i, lh, gr);
            __builtin_amdgcn_sched_barrier(0);      // every load of this phase is in flight before the first MFMA (see orientation 2)
            float dot = 0.f;
            if constexpr (RC) {
                // the forward's probabilities again: S^T = K Q^T with the forward's operand values and order (bit-identical scores),
                // p = exp(s - lse); signed as the forward's debug output is: p > 0 kept, p < 0 dropped, 0 outside the L x L tile
#pragma unroll
                for (int s_ = 0; s_ < HH; ++s_) qr[s_] *= scale;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) ps[jt][v] = 0.f;
                    rows_mfma<HD, T>(kr[jt], qr, ps[jt]);
                    uint32_t keep = 0xFFFFu;
                    if (dropping) keep = keep_bits<T>(drop, thr16, (uint32_t)((beg + min(i, L - 1)) * heads + h), jt, lh);
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int j = jt * T + acc_row<T>(v, lh);
                        const float p = (i < L && j < L) ? __expf(ps[jt][v] - lse_i) : 0.f;
                        ps[jt][v] = ((keep >> v) & 1u) ? p : -p;
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
                for (int v = 0; v < NV; ++v) dp[jt][v] = 0.f;
                rows_mfma<HD, T>(vr[jt], gr, dp[jt]);                      // dP^T[j][i] = V_j . dOut_i
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    dp[jt][v] = ps[jt][v] > 0.f ? dp[jt][v] * keep_scale : 0.f;            // gradient of the un-dropped probability
                    dot += dp[jt][v] * ps[jt][v];
                }
            }
            dot = TL::sum(dot);
            acc_t dq[CT];
            zero<T, CT>(dq);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int j = jt * T + acc_row<T>(v, lh);
                    const float p = fabsf(ps[jt][v]);
                    dp[jt][v] = p * (dp[jt][v] - dot);                     // dS^T[j][i]
                    Ds[j * PLD + i] = dp[jt][v];
                    Pd[j * PLD + i] = ps[jt][v] > 0.f ? p * keep_scale : 0.f;              // dropped-and-rescaled probability
                }
                regs_mfma<HD, T>(dp[jt], kc[jt], dq);
            }
            store_cols<HD, T>(dq, scale, dv_, 0, it, li, lh);
            if (colsum != nullptr) col_add<HD, T>(dq, scale, colsum + h * HD, li, lh);          // in_proj_bias gradient, Q third
        }
    }
    __syncthreads();
    // ---- orientation 2, lane = key j: dK = dS^T Q, dV = Pd^T dOut
#pragma unroll 1
    for (int jt = t0; jt < t1; ++jt) {
        if (jt * T >= L) break;
        const int j = jt * T + li;
        acc_t dk[CT], dv[CT];
        zero<T, CT>(dk);
        zero<T, CT>(dv);
#pragma unroll
        for (int it = 0; it < JT; ++it) {
            if (it * T >= L) break;
            float qc[CT][NV], gc[CT][NV];
            load_cols<HD, T>(qv, 0, it, li, lh, qc);
            load_cols<HD, T>(gv, 0, it, li, lh, gc);
            // without this fence the scheduler of the 2-tile instantiation pairs every load with the MFMA that consumes it -- 48
            // dependent round trips, 69 us per launch for 4 % of the pairs
            __builtin_amdgcn_sched_barrier(0);
            acc_t ds, pd;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = it * T + acc_row<T>(v, lh);
                ds[v] = Ds[j * PLD + i];
                pd[v] = Pd[j * PLD + i];
            }
            regs_mfma<HD, T>(ds, qc, dk);                                  // dK[j][c] += dS[i][j] Q[i][c]
            regs_mfma<HD, T>(pd, gc, dv);                                  // dV[j][c] += Pd[i][j] dOut[i][c]
        }
        store_cols<HD, T>(dk, scale, dv_, D, jt, li, lh);
        store_cols<HD, T>(dv, 1.f, dv_, 2 * D, jt, li, lh);
        if (colsum != nullptr) {                                               // ... K and V thirds
            col_add<HD, T>(dk, scale, colsum + D + h * HD, li, lh);
            col_add<HD, T>(dv, 1.f, colsum + 2 * D + h * HD, li, lh);
        }
    }
}

template <int HD, int JT, bool RC>
__device__ __forceinline__ void mhsa_bwd_pair(const float* __restrict__ qkv, int ldq, int D, int heads, const float* __restrict__ gout, int ldgo,
                                              const float* __restrict__ lse, const float* __restrict__ probs, int Lmax, const Dropout& drop,
                                              float keep_scale, float* __restrict__ gqkv, int ldgq,
                                              float* colsum, float* __restrict__ Pd, float* __restrict__ Ds, int h, int beg, int L, int t0, int t1) {
    if constexpr (JT == 1 && HD >= 16) {
        if (L <= 16) {                                           // wave-uniform
            mhsa_bwd_tile<HD, 1, RC, 16>(qkv, ldq, D, heads, gout, ldgo, lse, probs, Lmax, drop, keep_scale, gqkv, ldgq, colsum, Pd, Ds, h, beg, L, t0, t1);
            return;
        }
    }
    mhsa_bwd_tile<HD, JT, RC, 32>(qkv, ldq, D, heads, gout, ldgo, lse, probs, Lmax, drop, keep_scale, gqkv, ldgq, colsum, Pd, Ds, h, beg, L, t0, t1);
}

template <int HD, int JT, int OCC, bool RC>
__global__ __launch_bounds__(64 * JT) __attribute__((amdgpu_waves_per_eu(OCC))) void mhsa_bwd_kernel(
    const float* __restrict__ qkv, int ldq, const int* __restrict__ seg_off, int n_cap, const int* __restrict__ n_dyn, int D,
    int heads, const float* __restrict__ gout, int ldgo, const float* __restrict__ lse, const float* __restrict__ probs, int Lmax,
    Dropout drop, float keep_scale,
    float* __restrict__ gqkv, int ldgq, float* colsum, const int* __restrict__ long_list, const int* __restrict__ long_count, int min_len) {
    constexpr int LT = 32 * JT;
    __shared__ float Pd[LT * (LT + 1)];                       // dropped-and-rescaled probabilities [key j][query i]
    __shared__ float Ds[LT * (LT + 1)];                       // dS^T [key j][query i]
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int t0 = JT == 1 ? 0 : (int)(threadIdx.x >> 6);           // JT == 2: wave t owns query tile t, then key tile t
    LEGO_MHSA_WALK(mhsa_bwd_pair<HD, JT, RC>(qkv, ldq, D, heads, gout, ldgo, lse, probs, Lmax, drop, keep_scale, gqkv, ldgq, colsum, Pd, Ds, h, beg, L, t0, t0 + 1))
}
#undef LEGO_MHSA_WALK

// persistent grid of the short-segment kernels: 16 single-wave workgroups per CU (4 per SIMD), capped by the number of pairs
static int short_grid(int n_cap, int heads) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        cus = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256;
    }
    const long long pairs = (long long)n_cap * heads, cap = 16LL * cus;
    return (int)(pairs < cap ? pairs : cap);
}
static int long_grid(int n_cap, int heads) { const long long pairs = (long long)n_cap * heads; return (int)(pairs < 2048 ? pairs : 2048); }

static Dropout to_drop(const lego_dropout* d) {
    Dropout r = make_dropout(d);
    r.mask = nullptr;                 // the attention-probability site always draws in-kernel
    return r;
}

}  // namespace lego

using namespace lego;

extern "C" int lego_mhsa_long_segments(const int32_t* seg_off, int n_cap, const int32_t* n_dyn, int32_t* list, int32_t* count, void* stream) {
    if (n_cap <= 0) return hipMemsetAsync(count, 0, sizeof(int32_t), (hipStream_t)stream) == hipSuccess ? 0 : set_error("lego_mhsa_long_segments: memset failed");
    hipLaunchKernelGGL(mhsa_long_segments_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, seg_off, n_cap, n_dyn, list, count);
    return check_launch("lego_mhsa_long_segments");
}

extern "C" int lego_mhsa_core_fwd(const float* qkv, int ldq, const int32_t* seg_off, int n_cap, const int32_t* n_dyn,
                                  int D, int heads, float* out, int ldo, float* lse, float* probs, int Lmax,
                                  const lego_dropout* drop, int rows_cap, int part, const int32_t* long_list,
                                  const int32_t* long_count, void* stream) {
    LEGO_REQUIRE(lse != nullptr || probs != nullptr, "lego_mhsa_core_fwd: lse (recomputing backward) or probs (saved probabilities) is required");
    LEGO_REQUIRE(heads > 0 && D % heads == 0, "lego_mhsa_core_fwd: D=%d not divisible by heads=%d", D, heads);
    LEGO_REQUIRE(Lmax <= kMaxL, "lego_mhsa_core_fwd: Lmax=%d exceeds %d", Lmax, kMaxL);
    LEGO_REQUIRE((ldq & 3) == 0 && (D & 3) == 0, "lego_mhsa_core_fwd: ldq=%d and D=%d must be multiples of 4", ldq, D);
    LEGO_REQUIRE((long_list == nullptr) == (long_count == nullptr), "lego_mhsa_core_fwd: long_list and long_count go together");
    if (n_cap <= 0) return 0;
    (void)rows_cap;
    const int hd = D / heads;
    const Dropout dr = to_drop(drop);
    hipStream_t st = (hipStream_t)stream;
    const bool all_long = part == LEGO_MHSA_ALL_LONG && Lmax > 32;   // every segment through the two-wave instantiation: ONE launch
#define LAUNCH(HD) do { \
        if (part != LEGO_MHSA_LONG && !all_long) \
            hipLaunchKernelGGL((mhsa_fwd_kernel<HD, 1>), dim3(short_grid(n_cap, heads)), dim3(64), 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, out, ldo, lse, probs, Lmax, dr, nullptr, nullptr, 32); \
        if (Lmax > 32 && part != LEGO_MHSA_SHORT) \
            hipLaunchKernelGGL((mhsa_fwd_kernel<HD, 2>), dim3(long_grid(n_cap, heads)), dim3(128), 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, out, ldo, lse, probs, Lmax, dr, all_long ? nullptr : long_list, all_long ? nullptr : long_count, all_long ? 0 : 32); \
    } while (0)
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
                                  int D, int heads, const float* gout, int ldgo, const float* lse, const float* probs, int Lmax,
                                  const lego_dropout* drop, int rows_cap, float* gqkv, int ldgq, float* colsum, int part,
                                  const int32_t* long_list, const int32_t* long_count, void* stream) {
    LEGO_REQUIRE(heads > 0 && D % heads == 0, "lego_mhsa_core_bwd: D=%d not divisible by heads=%d", D, heads);
    LEGO_REQUIRE(Lmax <= kMaxL, "lego_mhsa_core_bwd: Lmax=%d exceeds %d", Lmax, kMaxL);
    LEGO_REQUIRE((ldq & 3) == 0 && (ldgo & 3) == 0 && (D & 3) == 0, "lego_mhsa_core_bwd: ldq=%d, ldgo=%d and D=%d must be multiples of 4", ldq, ldgo, D);
    if (n_cap <= 0) return 0;
    const int hd = D / heads;
    const Dropout dr = to_drop(drop);
    const float ks = dr.p > 0.f ? 1.f / (1.f - dr.p) : 1.f;     // the keep / drop decisions are redrawn in the kernel (same Philox counters)
    (void)rows_cap;
    hipStream_t st = (hipStream_t)stream;
    const bool all_long = part == LEGO_MHSA_ALL_LONG && Lmax > 32;
    LEGO_REQUIRE(lse != nullptr || probs != nullptr, "lego_mhsa_core_bwd: needs the forward's lse rows (recompute) or its saved probabilities");
    const bool shrt = part != LEGO_MHSA_LONG && !all_long, lng = Lmax > 32 && part != LEGO_MHSA_SHORT;
    const int* ll = all_long ? nullptr : long_list;
    const int* lc = all_long ? nullptr : long_count;
    const int ml = all_long ? 0 : 32;
    const bool rc = probs == nullptr;                      // saved probabilities win when both are given
    const dim3 gs(short_grid(n_cap, heads)), gl(long_grid(n_cap, heads));
#define BWD(HD, JT, OCC, RC, GRID, ...) \
    hipLaunchKernelGGL((mhsa_bwd_kernel<HD, JT, OCC, RC>), GRID, dim3(64 * JT), 0, st, qkv, ldq, seg_off, n_cap, n_dyn, D, heads, gout, ldgo, lse, probs, Lmax, dr, ks, gqkv, ldgq, colsum, __VA_ARGS__)
#define BOTH(HD, OS_RC, OS_SV, OL_RC, OL_SV) do { \
        if (shrt) { if (rc) BWD(HD, 1, OS_RC, true, gs, nullptr, nullptr, 32); else BWD(HD, 1, OS_SV, false, gs, nullptr, nullptr, 32); } \
        if (lng) { if (rc) BWD(HD, 2, OL_RC, true, gl, ll, lc, ml); else BWD(HD, 2, OL_SV, false, gl, ll, lc, ml); } \
    } while (0)
    switch (hd) {                                          // waves per SIMD: what the register counts allow (tools/regs.sh)
        case 8: BOTH(8, 3, 4, 1, 2); break;
        case 16: BOTH(16, 3, 4, 1, 2); break;
        case 32: BOTH(32, 3, 4, 1, 2); break;
        case 64: BOTH(64, 2, 2, 1, 1); break;
        default: return set_error("lego_mhsa_core_bwd: head dim %d not in {8,16,32,64}", hd);
    }
#undef BOTH
#undef BWD
    return check_launch("lego_mhsa_core_bwd");
}
