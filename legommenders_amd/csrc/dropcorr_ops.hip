// NRMS with the GloVe projection: the attention in-projection once per DISTINCT key, exactly, with a sparse Dropout correction
// (round 5; DESIGN.md section 10.6 of round 4 had the algebra, this is the build).
//
// embedding_hub.py:95-96 puts Dropout between the projection and nn.MultiheadAttention's in-projection
// (attention_operator.py:49-55), so a token row is  E_r = s (m_r . h_k)  with  h_k  the projection of its key k (a function of the
// token id alone), m_r the row's keep bits, s = 1 / (1 - p), and the rows of one token differ.  But the difference is SPARSE:
//     W E_r = s (W h_k - W ((1 - m_r) . h_k)) = s (Q_k - sum over the ~p D dropped coordinates c of h_k[c] W[:, c])
// The first term is one row of a product over the ~4.5 k distinct keys of a batch instead of its ~31 k sequence rows, the second
// ~26 multiply-adds of a 768-vector per row on the vector ALU against a slice of W^T held in LDS -- a twentieth of the dense
// product's flops.  The data gradient splits the same way:
//     dh_k = s sum_r m_r . (W^T g_r) = s (W^T (sum_r g_r) - sum_r (1 - m_r) . (W^T g_r))
// = a product over the per-key sums of d(qkv) minus ~26 dot products of 768 per row.  ([SEP] / category positions are keys of
// their own, carry no Dropout and take neither correction.)  The weight gradient's correction is a 10 %-dense sparse product that
// would not beat the dense one: dW stays d(qkv)^T E over the rows (side stream).
//
//   lego_qkv_expand_dropcorr   q|k|v rows from the per-key product:  out_r = tok_r ? s (Q_k - corr_r) + b : Q_k + b
//   lego_dropcorr_bwd          dEu[k][c] -= g_r . W[:, c]  for every dropped coordinate c of every token row r of key k
//   lego_scale_mask_rows       x_r = live_r ? scale x_r : 0   (the per-key gradient rows that go on into the projection)
#include <stdlib.h>
#include "../../include/lego_hip.h"
#include "common.hpp"

namespace lego {

constexpr int DC_CB = 128;                 // columns of W^T (= outputs of the in-projection) per workgroup slice
constexpr int DC_THREADS = 1024;           // 16 waves: four per SIMD hide the LDS round trip of every correction term
constexpr int DC_WAVES = DC_THREADS / 64;
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// stage W^T[0:D][cb:cb+128] (row stride ldw) into LDS with row stride `lds_ld` floats
__device__ __forceinline__ void stage_wt(const float* __restrict__ wt, int ldw, int D, int cb, int ncols, float* lds, int lds_ld) {
    for (int i = threadIdx.x; i < D * (DC_CB / 4); i += DC_THREADS) {
        const int c = i / (DC_CB / 4), j = (i % (DC_CB / 4)) * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (cb + j < ncols) v = *reinterpret_cast<const f32x4*>(wt + (size_t)c * ldw + cb + j);
        *reinterpret_cast<f32x4*>(lds + c * lds_ld + j) = v;
    }
}

// keep bits of row r, coordinates 4 lane .. 4 lane + 3 (lane < D / 4), as bits 0 / 8 / 16 / 24 of the returned word; 0x01010101 = all kept
__device__ __forceinline__ uint32_t keep_word(const uint8_t* __restrict__ mask, int r, int D, int lane) {
    if (4 * lane >= D) return 0x01010101u;
    return (*reinterpret_cast<const uint32_t*>(mask + (uint64_t)(r >> 2) * (uint64_t)D + (uint64_t)(4 * lane)) >> (r & 3)) & 0x01010101u;
}

// out[r][cb + 2 lane ..] for the rows of a strip; lane = two consecutive output columns.
//
// Arithmetic: per row the wave compacts its dropped coordinates into a list (LDS offset of the W^T row, h value) -- one entry per lane
// when read back -- and walks it four entries at a time: the entries come out of the list registers as scalars (v_readlane with a
// loop-counter index), so a turn is four independent ds_read_b64 of W^T rows and four packed multiply-adds with a scalar operand.
// Memory: a row's inputs are two dependent trips (its key, then q / h / keep bits at that key) and the LDS slice leaves room for four
// waves per SIMD, so the trips are taken in bulk: a wave owns a CONTIGUOUS run of rows, reads 64 keys / flags with one vector load,
// and fetches the inputs of four rows at a time, one group of four ahead of the arithmetic, with no branch around the loads (indices
// past the run are clamped) so the compiler's counted waits stay exact.  (Round 5's first forms: one coordinate per turn off the ballot
// mask, every step waiting on the one before, 93 us at the bench batch; then the four-entry turns with the loads still row by row:
// 96 us -- the waves were sitting on the two trips per row, not on the arithmetic.  With both: 88 us, of which 21 the plain
// expansion, 18 keep words + list, 49 the correction's arithmetic = vector-ALU issue: 2 v_readlane + v_add + v_pk_fma per coordinate
// and 128 columns.  Two forms that move work off the vector ALU, built and measured slower (tools/dropcorr_time.py): the address on
// the scalar side with ds_read_addtid_b32 (LDS address = M0 + offset + 4 lane; one v_readlane + one v_pk_fma per coordinate): 140 us --
// a write of M0 waits for the LDS reads in flight that used the old value, so the reads of a turn run one after the other; the
// coordinates straight off four ballot masks with s_ff1 / s_and chains instead of the list: 121 us -- the dependent scalar chain and the
// per-mask padding of the four-coordinate turns cost more than the list they save.)
constexpr int DC_LIST = 256;               // list entries per wave (= the widest row: D <= 256)
struct Rows4 {
    f32x2_t q[4];
    f32x4 h[4];
    uint32_t kw[4];
};
template <bool DROP>
__global__ __launch_bounds__(DC_THREADS) void qkv_expand_dropcorr_kernel(
    const float* __restrict__ qkvu, int ldq, const float* __restrict__ eu, int lde, const float* __restrict__ wt, int ldw,
    const float* __restrict__ bias, const int* __restrict__ inv, const int* __restrict__ rowinfo, const uint8_t* __restrict__ mask, float scale,
    int rows_cap, const int* __restrict__ rows_dyn, int D, int N, float* __restrict__ out, int ldo, int strips) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const int nblk = (N + DC_CB - 1) / DC_CB;
    const int blk = blockIdx.x % nblk, strip = blockIdx.x / nblk;
    const int cb = blk * DC_CB;
    const int per = (rows + strips - 1) / strips;
    const int r0 = strip * per, r1 = min(rows, r0 + per);
    if (r0 >= r1) return;
    if (DROP) stage_wt(wt, ldw, D, cb, N, smem, DC_CB);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    f32x2_t* const list = reinterpret_cast<f32x2_t*>(smem + D * DC_CB) + wave * DC_LIST;       // (LDS byte offset of W^T[c], h[c]) pairs
    const char* const wl = reinterpret_cast<const char*>(smem) + 8 * lane;                      // this lane's two columns of row 0
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int col = cb + 2 * lane;
    const bool col_ok = col < N;
    const int colc = min(col, N - 2);
    const f32x2_t b2 = bias != nullptr ? *reinterpret_cast<const f32x2_t*>(bias + colc) : f32x2_t{0.f, 0.f};
    const bool has = 4 * lane < D;
    const int hoff = has ? 4 * lane : 0;
    const int cpw = (r1 - r0 + DC_WAVES - 1) / DC_WAVES;
    const int ws = r0 + wave * cpw, we = min(r1, ws + cpw);
    for (int gb = ws; gb < we; gb += 64) {
        const int gi = min(gb + lane, we - 1);
        const int kv = inv[gi], riv = rowinfo[gi];
        const int gn = min(64, we - gb);                         // rows of this group
        auto fetch = [&](Rows4& R, int b) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(b + u, gn - 1);
                const int k = __builtin_amdgcn_readlane(kv, i);
                R.q[u] = *reinterpret_cast<const f32x2_t*>(qkvu + (size_t)k * ldq + colc);
                if (DROP) {
                    const int r = gb + i;
                    R.h[u] = *reinterpret_cast<const f32x4*>(eu + (size_t)k * lde + hoff);
                    R.kw[u] = *reinterpret_cast<const uint32_t*>(mask + (uint64_t)(r >> 2) * (uint64_t)D + (uint64_t)hoff);
                }
            }
        };
        Rows4 cur, nxt;
        fetch(cur, 0);
        for (int b = 0; b < gn; b += 4) {
            fetch(nxt, b + 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = b + u;
                if (i < gn) {                                    // wave-uniform
                    const int r = gb + i;
                    const int ri = __builtin_amdgcn_readlane(riv, i);
                    f32x2_t q = cur.q[u];
                    if (DROP && (ri & RI_LIVE) != 0) {           // wave-uniform
                        const float h0 = cur.h[u][0], h1 = cur.h[u][1], h2 = cur.h[u][2], h3 = cur.h[u][3];
                        const uint32_t kw = has ? (cur.kw[u] >> (r & 3)) : 0xFFFFFFFFu;     // bits 0 / 8 / 16 / 24: keep coordinate 4 lane + j
                        // the list: coordinate 4 lane + j of a lane that dropped it, in (j, lane) order
                        int n = 0;
                        {
                            const bool d = (kw & 0x00000001u) == 0u;
                            const unsigned long long m = __ballot(d);
                            if (d) list[n + __popcll(m & lt)] = f32x2_t{__int_as_float((4 * lane + 0) * DC_CB * 4), h0};
                            n += __popcll(m);
                        }
                        {
                            const bool d = (kw & 0x00000100u) == 0u;
                            const unsigned long long m = __ballot(d);
                            if (d) list[n + __popcll(m & lt)] = f32x2_t{__int_as_float((4 * lane + 1) * DC_CB * 4), h1};
                            n += __popcll(m);
                        }
                        {
                            const bool d = (kw & 0x00010000u) == 0u;
                            const unsigned long long m = __ballot(d);
                            if (d) list[n + __popcll(m & lt)] = f32x2_t{__int_as_float((4 * lane + 2) * DC_CB * 4), h2};
                            n += __popcll(m);
                        }
                        {
                            const bool d = (kw & 0x01000000u) == 0u;
                            const unsigned long long m = __ballot(d);
                            if (d) list[n + __popcll(m & lt)] = f32x2_t{__int_as_float((4 * lane + 3) * DC_CB * 4), h3};
                            n += __popcll(m);
                        }
                        n = __builtin_amdgcn_readfirstlane(n);
                        // (the list is written and read by this wave only: LDS operations of one wave complete in order)
                        f32x2_t a0 = f32x2_t{0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
                        for (int base = 0; base < n; base += 64) {
                            f32x2_t e = f32x2_t{0.f, 0.f};       // entry base + lane; past the end: row 0 of W^T times 0
                            if (base + lane < n) e = list[base + lane];
                            // (__builtin_bit_cast applied straight to an ELEMENT of an ext_vector reads element 0 whatever the index --
                            // hipcc 7.2, seen in the IR -- so the elements go through scalars of their own first)
                            const float e0 = e[0], e1 = e[1];
                            const int ev = __float_as_int(e0), hv = __float_as_int(e1);
                            const int cnt = min(64, n - base);
                            for (int t = 0; t < cnt; t += 4) {
                                const int o0 = __builtin_amdgcn_readlane(ev, t), o1 = __builtin_amdgcn_readlane(ev, t + 1);
                                const int o2 = __builtin_amdgcn_readlane(ev, t + 2), o3 = __builtin_amdgcn_readlane(ev, t + 3);
                                const float g0 = __int_as_float(__builtin_amdgcn_readlane(hv, t));
                                const float g1 = __int_as_float(__builtin_amdgcn_readlane(hv, t + 1));
                                const float g2 = __int_as_float(__builtin_amdgcn_readlane(hv, t + 2));
                                const float g3 = __int_as_float(__builtin_amdgcn_readlane(hv, t + 3));
                                const f32x2_t w0 = *reinterpret_cast<const f32x2_t*>(wl + o0), w1 = *reinterpret_cast<const f32x2_t*>(wl + o1);
                                const f32x2_t w2 = *reinterpret_cast<const f32x2_t*>(wl + o2), w3 = *reinterpret_cast<const f32x2_t*>(wl + o3);
                                a0 += g0 * w0;
                                a1 += g1 * w1;
                                a2 += g2 * w2;
                                a3 += g3 * w3;
                            }
                        }
                        q = scale * (q - ((a0 + a1) + (a2 + a3)));
                    }
                    q += b2;
                    if (col_ok) *reinterpret_cast<f32x2_t*>(out + (size_t)r * ldo + col) = q;
                }
            }
            cur = nxt;
        }
    }
}

// dEu[inv[r]][c] -= sum_j g[r][cb + j] W^T[c][cb + j]  (j over this workgroup's 128-column slice) for the dropped coordinates c of the
// token rows of a strip.  Lane = ONE dropped coordinate of the row (compacted list, up to 64 per turn); the row's slice of g sits
// in LDS (broadcast reads), the W^T slice with a padded row stride (lanes on different rows c: the 16-lane groups of a
// ds_read_b128 then spread over the banks)
constexpr int DC_WLD = DC_CB + 4;
__global__ __launch_bounds__(DC_THREADS) void dropcorr_bwd_kernel(
    const float* __restrict__ g, int ldg, const float* __restrict__ wt, int ldw, const int* __restrict__ inv, const int* __restrict__ rowinfo,
    const uint8_t* __restrict__ mask, int rows_cap, const int* __restrict__ rows_dyn, int D, int N, float* __restrict__ deu, int ldd, int strips) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wl = smem;                                      // [D][DC_WLD]
    float* const gl = smem + D * DC_WLD;                         // [waves][DC_CB]
    int* const cl = reinterpret_cast<int*>(gl + DC_WAVES * DC_CB);   // [waves][256] compacted coordinates
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const int nblk = (N + DC_CB - 1) / DC_CB;
    const int blk = blockIdx.x % nblk, strip = blockIdx.x / nblk;
    const int cb = blk * DC_CB;
    const int per = (rows + strips - 1) / strips;
    const int r0 = strip * per, r1 = min(rows, r0 + per);
    if (r0 >= r1) return;
    stage_wt(wt, ldw, D, cb, N, wl, DC_WLD);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* const gw = gl + wave * DC_CB;
    int* const cw = cl + wave * 256;
    for (int r = r0 + wave; r < r1; r += DC_WAVES) {
        if ((rowinfo[r] & RI_LIVE) == 0) continue;               // wave-uniform: [SEP] / category rows carry no Dropout
        const int k = inv[r];
        const uint32_t kw = keep_word(mask, r, D, lane);
        // compacted list of the dropped coordinates (order: coordinate j of lane l -> all j = 0 first)
        int n = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool d = ((kw >> (8 * j)) & 1u) == 0u;
            const unsigned long long m = __ballot(d);
            if (d) cw[n + __popcll(m & ((1ull << lane) - 1ull))] = 4 * lane + j;
            n += __popcll(m);
        }
        {   // this wave's slice of the row's gradient: 128 floats = 2 per lane
            const int col = cb + 2 * lane;
            f32x2_t g2 = f32x2_t{0.f, 0.f};
            if (col < N) g2 = *reinterpret_cast<const f32x2_t*>(g + (size_t)r * ldg + col);
            *reinterpret_cast<f32x2_t*>(gw + 2 * lane) = g2;
        }
        // (the list and the gradient slice are written and read by this wave only: LDS operations of one wave complete in order)
        for (int base = 0; base < n; base += 64) {
            const bool on = base + lane < n;
            const int c = on ? cw[base + lane] : 0;
            const float* wr = wl + c * DC_WLD;
            float dot = 0.f;
#pragma unroll 8
            for (int j = 0; j < DC_CB; j += 4) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + j);
                const f32x4 gv = *reinterpret_cast<const f32x4*>(gw + j);
                dot += wv[0] * gv[0] + wv[1] * gv[1] + wv[2] * gv[2] + wv[3] * gv[3];
            }
            if (on) atomicAdd(deu + (size_t)k * ldd + c, -dot);
        }
    }
}

__global__ void scale_mask_rows_kernel(float* x, int ld, int rows_cap, const int* __restrict__ rows_dyn, int width, const int* __restrict__ rowinfo,
                                       float scale) {
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const long long total = (long long)rows * (width / 4);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / (width / 4)), c = (int)(e % (width / 4)) * 4;
        f32x4* p = reinterpret_cast<f32x4*>(x + (size_t)r * ld + c);
        const bool live = (rowinfo[r] & RI_LIVE) != 0;
        *p = live ? *p * scale : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

static int dc_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace lego

using namespace lego;

extern "C" int lego_qkv_expand_dropcorr(const float* qkvu, int ldq, const float* eu, int lde, const float* wt, int ldw, const float* bias,
                                        const int32_t* inv, const int32_t* rowinfo, const lego_dropout* drop, int rows_cap,
                                        const int32_t* rows_dyn, int D, int N, float* out, int ldo, void* stream) {
    LEGO_REQUIRE((D & 3) == 0 && D <= 256 && (N & 1) == 0 && (ldq & 1) == 0 && (ldo & 1) == 0 && (lde & 3) == 0 && (ldw & 3) == 0,
                 "lego_qkv_expand_dropcorr: D=%d (multiple of 4, <= 256), N=%d and the row strides must be even / multiples of 4", D, N);
    LEGO_REQUIRE(drop == nullptr || drop->p <= 0.f || drop->mask != nullptr, "lego_qkv_expand_dropcorr: a dropout site needs its precomputed keep bits (lego_dropout_mask)");
    LEGO_REQUIRE(inv != nullptr && rowinfo != nullptr, "lego_qkv_expand_dropcorr: inv and rowinfo are required");
    if (rows_cap <= 0) return 0;
    const bool dropping = drop != nullptr && drop->p > 0.f;
    const int nblk = (N + DC_CB - 1) / DC_CB;
    int strips = dc_cus() / nblk;
    if (strips < 1) strips = 1;
    const size_t lds = dropping ? (size_t)D * DC_CB * sizeof(float) + (size_t)DC_WAVES * DC_LIST * 8 : 0;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_expand_dropcorr_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  256 * DC_CB * 4 + DC_WAVES * DC_LIST * 8);
        attr_done = true;
    }
    if (dropping)
        hipLaunchKernelGGL(qkv_expand_dropcorr_kernel<true>, dim3(nblk * strips), dim3(DC_THREADS), lds, (hipStream_t)stream, qkvu, ldq, eu, lde, wt, ldw,
                           bias, inv, rowinfo, drop->mask, 1.f / (1.f - drop->p), rows_cap, rows_dyn, D, N, out, ldo, strips);
    else
        hipLaunchKernelGGL(qkv_expand_dropcorr_kernel<false>, dim3(nblk * strips), dim3(DC_THREADS), 0, (hipStream_t)stream, qkvu, ldq, eu, lde, wt, ldw,
                           bias, inv, rowinfo, (const uint8_t*)nullptr, 1.f, rows_cap, rows_dyn, D, N, out, ldo, strips);
    return check_launch("lego_qkv_expand_dropcorr");
}

extern "C" int lego_dropcorr_bwd(const float* g, int ldg, const float* wt, int ldw, const int32_t* inv, const int32_t* rowinfo,
                                 const lego_dropout* drop, int rows_cap, const int32_t* rows_dyn, int D, int N, float* deu, int ldd, void* stream) {
    LEGO_REQUIRE((D & 3) == 0 && D <= 256 && (N & 1) == 0 && (ldg & 1) == 0 && (ldw & 3) == 0,
                 "lego_dropcorr_bwd: D=%d (multiple of 4, <= 256), N=%d and the row strides must be even / multiples of 4", D, N);
    LEGO_REQUIRE(inv != nullptr && rowinfo != nullptr, "lego_dropcorr_bwd: inv and rowinfo are required");
    if (rows_cap <= 0 || drop == nullptr || drop->p <= 0.f) return 0;          // nothing was dropped: no correction
    LEGO_REQUIRE(drop->mask != nullptr, "lego_dropcorr_bwd: the dropout site needs its precomputed keep bits (lego_dropout_mask)");
    const int nblk = (N + DC_CB - 1) / DC_CB;
    int strips = dc_cus() / nblk;
    if (strips < 1) strips = 1;
    const size_t lds = ((size_t)D * DC_WLD + (size_t)DC_WAVES * DC_CB) * sizeof(float) + (size_t)DC_WAVES * 256 * sizeof(int);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dropcorr_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(((size_t)256 * DC_WLD + (size_t)DC_WAVES * DC_CB) * sizeof(float) + (size_t)DC_WAVES * 256 * sizeof(int)));
        attr_done = true;
    }
    hipLaunchKernelGGL(dropcorr_bwd_kernel, dim3(nblk * strips), dim3(DC_THREADS), lds, (hipStream_t)stream, g, ldg, wt, ldw, inv, rowinfo, drop->mask,
                       rows_cap, rows_dyn, D, N, deu, ldd, strips);
    return check_launch("lego_dropcorr_bwd");
}

extern "C" int lego_scale_mask_rows(float* x, int ld, int rows_cap, const int32_t* rows_dyn, int width, const int32_t* rowinfo, float scale,
                                    void* stream) {
    LEGO_REQUIRE((width & 3) == 0 && (ld & 3) == 0 && rowinfo != nullptr, "lego_scale_mask_rows: width=%d ld=%d must be multiples of 4, rowinfo is required", width, ld);
    if (rows_cap <= 0) return 0;
    const long long tot = (long long)rows_cap * (width / 4);
    hipLaunchKernelGGL(scale_mask_rows_kernel, dim3((int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream, x, ld,
                       rows_cap, rows_dyn, width, rowinfo, scale);
    return check_launch("lego_scale_mask_rows");
}
