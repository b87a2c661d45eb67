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
// (Round 5 also built the data gradient's per-key form -- per-key sums of d(qkv), a product over the keys and a correction kernel whose
// atomics queued on the Zipf head's rows: 620 us against 130 for the row form; removed in round 6, DESIGN.md section 11.4 has the record.)
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

