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
#include <map>
#include <mutex>
#include <utility>
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

// no Dropout (evaluation, p = 0): out[r][cb + 2 lane ..] = Q_key(r) + b for the rows of a strip; lane = two consecutive output columns.  A wave
// owns a contiguous run of rows, reads 64 keys with one vector load and fetches the q pieces of four rows at a time, one group ahead
__global__ __launch_bounds__(DC_THREADS) void qkv_expand_plain_kernel(const float* __restrict__ qkvu, int ldq, const float* __restrict__ bias,
                                                                      const int* __restrict__ inv, int rows_cap, const int* __restrict__ rows_dyn, int N,
                                                                      float* __restrict__ out, int ldo, int strips) {
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const int nblk = (N + DC_CB - 1) / DC_CB;
    const int blk = blockIdx.x % nblk, strip = blockIdx.x / nblk;
    const int cb = blk * DC_CB;
    const int per = (rows + strips - 1) / strips;
    const int r0 = strip * per, r1 = min(rows, r0 + per);
    if (r0 >= r1) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = cb + 2 * lane;
    const bool col_ok = col < N;
    const int colc = min(col, N - 2);
    const f32x2_t b2 = bias != nullptr ? *reinterpret_cast<const f32x2_t*>(bias + colc) : f32x2_t{0.f, 0.f};
    const int cpw = (r1 - r0 + DC_WAVES - 1) / DC_WAVES;
    const int ws = r0 + wave * cpw, we = min(r1, ws + cpw);
    for (int gb = ws; gb < we; gb += 64) {
        const int kv = inv[min(gb + lane, we - 1)];
        const int gn = min(64, we - gb);
        f32x2_t cur[4], nxt[4];
        auto fetch = [&](f32x2_t (&q)[4], int b) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                q[u] = *reinterpret_cast<const f32x2_t*>(qkvu + (size_t)__builtin_amdgcn_readlane(kv, min(b + u, gn - 1)) * ldq + colc);
        };
        fetch(cur, 0);
        for (int b = 0; b < gn; b += 4) {
            fetch(nxt, b + 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (b + u < gn && col_ok) *reinterpret_cast<f32x2_t*>(out + (size_t)(gb + b + u) * ldo + col) = cur[u] + b2;
                cur[u] = nxt[u];
            }
        }
    }
}

// ---------------------------------------------------------------- the Dropout correction: dropped coordinates as (W^T row offset, h value) PAIRS
// Round 5's kernel built a row's list of dropped coordinates inside the expansion -- once per 128-column slice, six times per row -- kept it in
// registers and took offset and multiplier out of them with two v_readlane per coordinate: four vector instructions per coordinate and slice,
// and the vector ALU bounded it (88 us at the bench batch: 21 the plain expansion, 18 keep words + list, 49 the correction).  Round 6:
//   * dropcorr_pairs_kernel writes every row's list ONCE (one wave per row; the same (coordinate % 4, lane) order, so the sums keep their order), padded
//     with (offset 0, multiplier 0) pairs to a multiple of DCP_CHUNK: a padded turn adds 0 x W^T[0];
//   * qkv_expand_pairs_kernel reads it through the SCALAR unit -- s_load_dwordx16 = 8 pairs in 16 scalar registers -- so offset and multiplier
//     are scalar operands of the address add and of the packed multiply-add: TWO vector instructions + one ds_read_b64 per coordinate.
// 86 -> 76 us for both launches (tools/dropcorr_time.py), NRMS step 0.955 -> 0.950 ms.  Not the 2 x the instruction count promises: scalar loads and
// LDS reads share one counter (lgkmcnt) and scalar loads return out of order, so the wait in front of the first multiply-add of a turn also waits
// for the list -- every turn of 32 pairs pays an L2 round trip that only the other three waves of the SIMD cover.  (Round 6 also tried the lists read
// by wave-uniform VECTOR loads: 85 us, the 1 KB a uniform 16-byte load returns through the 64 B/clk path costs what the v_readlane pairs did.)
constexpr int DCP_CHUNK = 8;                 // pairs per scalar load
constexpr int DCP_WAVES = 4;

__global__ __launch_bounds__(DCP_WAVES * 64) void dropcorr_pairs_kernel(const float* __restrict__ eu, int lde, const int* __restrict__ inv,
                                                                        const int* __restrict__ rowinfo, const uint8_t* __restrict__ mask, int rows_cap,
                                                                        const int* __restrict__ rows_dyn, int D, f32x2_t* __restrict__ pairs, int stride,
                                                                        int* __restrict__ cnt) {
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool has = 4 * lane < D;
    const int hoff = has ? 4 * lane : 0;
    for (int r = blockIdx.x * DCP_WAVES + wave; r < rows; r += gridDim.x * DCP_WAVES) {
        const int k = inv[r], ri = rowinfo[r];
        if ((ri & RI_LIVE) == 0) {                               // wave-uniform: [SEP] / category positions carry no Dropout
            if (lane == 0) cnt[r] = 0;
            continue;
        }
        const f32x4 h = *reinterpret_cast<const f32x4*>(eu + (size_t)k * lde + hoff);
        const uint32_t kw = has ? (*reinterpret_cast<const uint32_t*>(mask + (uint64_t)(r >> 2) * (uint64_t)D + (uint64_t)hoff) >> (r & 3)) : 0xFFFFFFFFu;
        f32x2_t* const list = pairs + (size_t)r * stride;
        int n = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                            // (j, lane) order: the order the in-kernel list had
            const bool d = ((kw >> (8 * j)) & 1u) == 0u;
            const unsigned long long m = __ballot(d);
            if (d) list[n + __popcll(m & lt)] = f32x2_t{__int_as_float((4 * lane + j) * DC_CB * 4), h[j]};
            n += __popcll(m);
        }
        n = __builtin_amdgcn_readfirstlane(n);
        const int padded = (n + DCP_CHUNK - 1) / DCP_CHUNK * DCP_CHUNK;
        if (n + lane < padded) list[n + lane] = f32x2_t{0.f, 0.f};
        if (lane == 0) cnt[r] = n;
    }
}

__global__ __launch_bounds__(DC_THREADS) void qkv_expand_pairs_kernel(
    const float* __restrict__ qkvu, int ldq, const float* __restrict__ wt, int ldw, const float* __restrict__ bias, const int* __restrict__ inv,
    const uint4* __restrict__ pairs, int stride16, const int* __restrict__ cnt, float scale, int rows_cap, const int* __restrict__ rows_dyn, int D, int N,
    float* __restrict__ out, int ldo, int strips) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const int nblk = (N + DC_CB - 1) / DC_CB;
    const int blk = blockIdx.x % nblk, strip = blockIdx.x / nblk;
    const int cb = blk * DC_CB;
    const int per = (rows + strips - 1) / strips;
    const int r0 = strip * per, r1 = min(rows, r0 + per);
    if (r0 >= r1) return;
    stage_wt(wt, ldw, D, cb, N, smem, DC_CB);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const char* const wl = reinterpret_cast<const char*>(smem) + 8 * lane;                      // this lane's two columns of row 0
    const int col = cb + 2 * lane;
    const bool col_ok = col < N;
    const int colc = min(col, N - 2);
    const f32x2_t b2 = bias != nullptr ? *reinterpret_cast<const f32x2_t*>(bias + colc) : f32x2_t{0.f, 0.f};
    const int cpw = (r1 - r0 + DC_WAVES - 1) / DC_WAVES;
    const int ws = r0 + wave * cpw, we = min(r1, ws + cpw);
    if (ws >= we) return;
    // a row's inputs one row ahead: its key (scalar), then q at that key (vector) and the list's length (scalar)
    int k_nxt = inv[ws];
    f32x2_t q_nxt = *reinterpret_cast<const f32x2_t*>(qkvu + (size_t)k_nxt * ldq + colc);
    int n_nxt = cnt[ws];
    for (int r = ws; r < we; ++r) {
        f32x2_t q = q_nxt;
        const int n = n_nxt;
        if (r + 1 < we) {
            k_nxt = inv[r + 1];
            q_nxt = *reinterpret_cast<const f32x2_t*>(qkvu + (size_t)k_nxt * ldq + colc);
            n_nxt = cnt[r + 1];
        }
        if (n > 0) {                                             // wave-uniform
            const uint4* pp = pairs + (size_t)r * stride16;
            f32x2_t a0 = f32x2_t{0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
            for (int c0 = 0; c0 < n; c0 += DCP_CHUNK, pp += DCP_CHUNK / 2) {
                const uint4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];                  // 8 (offset, multiplier) pairs: s_load_dwordx16
                const f32x2_t w0 = *reinterpret_cast<const f32x2_t*>(wl + p0.x), w1 = *reinterpret_cast<const f32x2_t*>(wl + p0.z);
                const f32x2_t w2 = *reinterpret_cast<const f32x2_t*>(wl + p1.x), w3 = *reinterpret_cast<const f32x2_t*>(wl + p1.z);
                const f32x2_t w4 = *reinterpret_cast<const f32x2_t*>(wl + p2.x), w5 = *reinterpret_cast<const f32x2_t*>(wl + p2.z);
                const f32x2_t w6 = *reinterpret_cast<const f32x2_t*>(wl + p3.x), w7 = *reinterpret_cast<const f32x2_t*>(wl + p3.z);
                a0 += __uint_as_float(p0.y) * w0;
                a1 += __uint_as_float(p0.w) * w1;
                a2 += __uint_as_float(p1.y) * w2;
                a3 += __uint_as_float(p1.w) * w3;
                a0 += __uint_as_float(p2.y) * w4;
                a1 += __uint_as_float(p2.w) * w5;
                a2 += __uint_as_float(p3.y) * w6;
                a3 += __uint_as_float(p3.w) * w7;
            }
            q = scale * (q - ((a0 + a1) + (a2 + a3)));
        }
        q += b2;
        if (col_ok) *reinterpret_cast<f32x2_t*>(out + (size_t)r * ldo + col) = q;
    }
}

// the pair lists of a launch: [rows_cap][stride] pairs + [rows_cap] lengths, one buffer per (device, stream) -- launches of one stream run one after
// the other -- grown on demand
static bool dc_pair_buffer(hipStream_t st, int rows_cap, int stride, f32x2_t** pairs, int** cnt) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<void*, size_t>> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const size_t need = (size_t)rows_cap * stride * sizeof(f32x2_t) + (size_t)rows_cap * sizeof(int) + 256;
    std::lock_guard<std::mutex> lock(mu);
    auto& b = bufs[{dev, st}];
    if (b.second < need) {
        if (b.first != nullptr) { (void)hipStreamSynchronize(st); (void)hipFree(b.first); b.first = nullptr; b.second = 0; }
        void* p = nullptr;
        if (hipMalloc(&p, need) != hipSuccess) { (void)hipGetLastError(); return false; }
        b.first = p; b.second = need;
    }
    *pairs = reinterpret_cast<f32x2_t*>(b.first);
    *cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(b.first) + (size_t)rows_cap * stride * sizeof(f32x2_t));
    return true;
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
    hipStream_t st = (hipStream_t)stream;
    if (!dropping) {
        hipLaunchKernelGGL(qkv_expand_plain_kernel, dim3(nblk * strips), dim3(DC_THREADS), 0, st, qkvu, ldq, bias, inv, rows_cap, rows_dyn, N, out, ldo, strips);
        return check_launch("lego_qkv_expand_dropcorr");
    }
    LEGO_REQUIRE(eu != nullptr && wt != nullptr, "lego_qkv_expand_dropcorr: a dropout site needs the per-key embeddings and W^T");
    const int stride = (D + DCP_CHUNK - 1) / DCP_CHUNK * DCP_CHUNK;          // pairs per row of the list buffer
    f32x2_t* pairs = nullptr;
    int* cnt = nullptr;
    if (!dc_pair_buffer(st, rows_cap, stride, &pairs, &cnt))
        return set_error("lego_qkv_expand_dropcorr: no memory for the pair lists (%d rows x %d pairs)", rows_cap, stride);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_expand_pairs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * DC_CB * 4);
        attr_done = true;
    }
    const int pg = min((rows_cap + DCP_WAVES - 1) / DCP_WAVES, 8 * dc_cus());
    hipLaunchKernelGGL(dropcorr_pairs_kernel, dim3(pg), dim3(DCP_WAVES * 64), 0, st, eu, lde, inv, rowinfo, drop->mask, rows_cap, rows_dyn, D, pairs, stride, cnt);
    hipLaunchKernelGGL(qkv_expand_pairs_kernel, dim3(nblk * strips), dim3(DC_THREADS), (size_t)D * DC_CB * sizeof(float), st, qkvu, ldq, wt, ldw, bias, inv,
                       reinterpret_cast<const uint4*>(pairs), stride / 2, cnt, 1.f / (1.f - drop->p), rows_cap, rows_dyn, D, N, out, ldo, strips);
    return check_launch("lego_qkv_expand_dropcorr");
}

