// HBM-bound kernels of the path: ragged batch plan, embedding row gather / scatter-add,
// additive-attention pooling (wave-shuffle reductions, one wave per segment), dot + cross-entropy,
// Adam, device-side negative sampling.  Everything here is integer/byte or streaming fp32 work:
// coalesced 16-B accesses, no MFMA.
#include "../../include/lego_hip.h"
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"

namespace lego {

constexpr float kEps = 1.1920928955078125e-07f;   // torch.finfo(float32).eps (model/common/attention.py:36)

// ------------------------------------------------------------------ block scan (1024 threads)
__device__ __forceinline__ int block_incl_scan_1024(int v, int* wsum) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    if (w == 0) {
        int s = lane < 16 ? wsum[lane] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int t = __shfl_up(s, o, 64);
            if (lane >= o) s += t;
        }
        if (lane < 16) wsum[lane] = s;
    }
    __syncthreads();
    if (w > 0) v += wsum[w - 1];
    __syncthreads();
    return v;
}

// One workgroup: scan the history lengths, enumerate the live item instances
// (B*C candidates, then each user's hist_len clicked items), scan their title lengths.
__global__ __launch_bounds__(1024) void plan_scan_kernel(
    const int* __restrict__ cand, const int* __restrict__ hist, const int* __restrict__ hist_len,
    int B, int C, int S, const int* __restrict__ title_len,
    int* counters, int* inst_item, int* seg_off, int* hist_off) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int b = base + tid;
        int len = 0;
        if (b < B) len = min(max(hist_len[b], 0), S);
        const int inc = block_incl_scan_1024(len, wsum);
        const int carry = carry_s;
        if (b < B) hist_off[b] = carry + inc - len;
        __syncthreads();
        if (tid == 1023) carry_s = carry + inc;
        __syncthreads();
    }
    const int n_hist = carry_s;
    if (tid == 0) hist_off[B] = n_hist;
    __syncthreads();
    const int BC = B * C;
    const int NI = BC + n_hist;
    const int NI_cap = B * (C + S);
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < NI_cap; base += 1024) {
        const int i = base + tid;
        int len = 0;
        if (i < NI) {
            int item;
            if (i < BC) {
                item = cand[i];
            } else {
                const int j = i - BC;
                int lo = 0, hi = B - 1;          // largest b with hist_off[b] <= j
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (hist_off[mid] <= j) lo = mid; else hi = mid - 1;
                }
                item = hist[lo * S + (j - hist_off[lo])];
            }
            inst_item[i] = item;
            len = title_len[item];
        }
        const int inc = block_incl_scan_1024(len, wsum);
        const int carry = carry_s;
        if (i < NI_cap) seg_off[i] = carry + inc - len;
        __syncthreads();
        if (tid == 1023) carry_s = carry + inc;
        __syncthreads();
    }
    if (tid == 0) {
        const int R = carry_s;
        seg_off[NI_cap] = R;
        counters[0] = R; counters[1] = NI; counters[2] = R + NI; counters[3] = n_hist;
        counters[4] = BC; counters[5] = 0; counters[6] = 0; counters[7] = 0;
    }
}

__global__ void plan_rows_kernel(const int* __restrict__ counters, const int* __restrict__ inst_item,
                                 const int* __restrict__ seg_off, const int* __restrict__ title_tok,
                                 const int* __restrict__ title_len, int T, int NI_cap,
                                 int* rowinfo, int* row_tok) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid >> 5, p0 = gid & 31;
    if (i >= NI_cap || i >= counters[1]) return;
    const int item = inst_item[i];
    const int len = title_len[item];
    const int r0 = seg_off[i];
    for (int pos = p0; pos < len; pos += 32) {
        rowinfo[r0 + pos] = (pos > 0 ? RI_LEFT : 0) | (pos < len - 1 ? RI_RIGHT : 0) | RI_LIVE | (i << RI_INST_SHIFT);
        row_tok[r0 + pos] = title_tok[(size_t)item * T + pos];
    }
}

// Row pairs of the Winograd F(2,3) conv (gemm_wino.hpp): pair q of an item instance covers its rows 2q, 2q+1.
// pair_info[p] = (first row << 3) | has_second | has_left << 1 | has_right2 << 2; counters_out[0] = #pairs.
__global__ __launch_bounds__(1024) void plan_pairs_kernel(const int* __restrict__ seg_off, int n_cap,
                                                          const int* __restrict__ n_dyn, int* __restrict__ pair_info,
                                                          int* counters_out) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x;
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        int beg = 0, len = 0;
        if (i < n) { beg = seg_off[i]; len = seg_off[i + 1] - beg; }
        const int np = (len + 1) >> 1;
        const int inc = block_incl_scan_1024(np, wsum);
        const int carry = carry_s;
        int p = carry + inc - np;
        for (int q = 0; q < np; ++q, ++p)
            pair_info[p] = ((beg + 2 * q) << 3) | (2 * q + 1 < len ? 1 : 0) | (q > 0 ? 2 : 0) | (2 * q + 2 < len ? 4 : 0);
        __syncthreads();
        if (tid == 1023) carry_s = carry + inc;
        __syncthreads();
    }
    if (tid == 0) counters_out[0] = carry_s;
}

// Winograd F(2,3) weight transform: u[0] = g0, u[1] = (g0+g1+g2)/2, u[2] = (g0-g1+g2)/2, u[3] = g2 with g_t = w[:, :, t]
// ut (optional): the same four matrices transposed, ut[s][c][o] = u[s][o][c] -- the data gradient then reads its weight
// panel K-contiguous (one ds_read_b128 per fragment instead of four ds_read_b32 from a [k][n] image)
__global__ void conv3_wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, float* __restrict__ ut, int Dout, int Din) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;      // over [o][c]
    const int per = Dout * Din;
    if (e >= per) return;
    const float g0 = w[3 * e], g1 = w[3 * e + 1], g2 = w[3 * e + 2];
    const float v[4] = {g0, 0.5f * (g0 + g1 + g2), 0.5f * (g0 - g1 + g2), g2};
#pragma unroll
    for (int s = 0; s < 4; ++s) u[(size_t)s * per + e] = v[s];
    if (ut != nullptr) {
        const int o = e / Din, c = e - o * Din;
#pragma unroll
        for (int s = 0; s < 4; ++s) ut[(size_t)s * per + (size_t)c * Dout + o] = v[s];
    }
}
// ... and its transpose for the gradient: dw[:, :, 0] += du0 + (du1+du2)/2, [1] += (du1-du2)/2, [2] += (du1+du2)/2 + du3
// du holds n_slabs partial [4][Dout][Din] results (one per k split of the weight-gradient launch); a single slab is an
// atomics accumulator and is handed back clean
// Workgroup = 64 entry quads x 4 slab lanes: lane y sums slabs y, y + 4, ... (all of a lane's loads are independent: with 16
// slabs that is 16 x 16 B in flight per thread, where the one-thread-per-quad form walked the slabs with 4 loads in flight and only
// 64 workgroups: 18 us serial, 30 us at the end of the step's critical path); the lanes fold through LDS and lane 0 updates dw with
// plain 16-byte read-modify-writes (the 12 floats of an entry quad are contiguous and nobody else writes them).
__global__ __launch_bounds__(256) void conv3_wino_unpack_add_kernel(float* __restrict__ du, int n_slabs, float* __restrict__ dw, int Dout, int Din) {
    __shared__ f32x4 part[4][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int per = Dout * Din;
    const int e4 = (blockIdx.x * 64 + tx) * 4;       // four consecutive (o, c) entries per thread: 16-B slab loads
    const bool in = e4 < per;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (in) {
#pragma unroll 4
        for (int s = ty; s < n_slabs; s += 4) {
            const float* p = du + (size_t)s * 4 * per + e4;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += *reinterpret_cast<const f32x4*>(p + (size_t)t * per);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) part[ty][t][tx] = acc[t];
    __syncthreads();
    if (ty != 0 || !in) return;
    f32x4 a, b, c, d;
    a = (part[0][0][tx] + part[1][0][tx]) + (part[2][0][tx] + part[3][0][tx]);
    b = (part[0][1][tx] + part[1][1][tx]) + (part[2][1][tx] + part[3][1][tx]);
    c = (part[0][2][tx] + part[1][2][tx]) + (part[2][2][tx] + part[3][2][tx]);
    d = (part[0][3][tx] + part[1][3][tx]) + (part[2][3][tx] + part[3][3][tx]);
    float* w = dw + 3 * (size_t)e4;                  // 12 contiguous floats: entries e4 .. e4 + 3, three taps each
    f32x4 w0 = *reinterpret_cast<f32x4*>(w), w1 = *reinterpret_cast<f32x4*>(w + 4), w2 = *reinterpret_cast<f32x4*>(w + 8);
    float o[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[3 * i] = a[i] + 0.5f * (b[i] + c[i]);
        o[3 * i + 1] = 0.5f * (b[i] - c[i]);
        o[3 * i + 2] = 0.5f * (b[i] + c[i]) + d[i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { w0[j] += o[j]; w1[j] += o[4 + j]; w2[j] += o[8 + j]; }
    *reinterpret_cast<f32x4*>(w) = w0; *reinterpret_cast<f32x4*>(w + 4) = w1; *reinterpret_cast<f32x4*>(w + 8) = w2;
    if (n_slabs == 1) {                  // a single slab is an atomics accumulator: handed back clean
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(du + (size_t)t * per + e4) = z;
    }
}

__global__ void plan_dense_kernel(const int* __restrict__ mask, int n, int L, int* counters, int* seg_off, int* rowinfo) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) {
        counters[0] = n * L; counters[1] = n; counters[2] = n * L + n; counters[3] = 0;
        counters[4] = 0; counters[5] = 0; counters[6] = 0; counters[7] = 0;
    }
    if (r <= n) seg_off[r] = r * L;
    if (r >= n * L) return;
    const int i = r / L, pos = r - i * L;
    const int live = mask == nullptr ? 1 : (mask[r] != 0);
    rowinfo[r] = (pos > 0 ? RI_LEFT : 0) | (pos < L - 1 ? RI_RIGHT : 0) | (live ? RI_LIVE : 0) | (i << RI_INST_SHIFT);
}

// ------------------------------------------------------------------ gather / scatter
// out[r, :] = table[idx[r], :]; the rows are swept as one flat float4 stream so both the
// 1200-B (E0 = 300) source rows and the destination are read/written in whole 16-B pieces.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int ld_table, int w4,
                                                          const int* __restrict__ idx, int rows_cap,
                                                          const int* __restrict__ rows_dyn, float* __restrict__ out, int ld_out,
                                                          int accumulate) {
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const long long total = (long long)rows * w4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    constexpr int U = 4;                                // 4 independent 16-B row pieces in flight per lane
    for (long long e0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += U * stride) {
        f32x4 v[U];
        f32x4* dst[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long e = e0 + u * stride;
            live[u] = false;
            dst[u] = nullptr;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e < total) {
                const int r = (int)(e / w4);
                const int c = (int)(e - (long long)r * w4) * 4;
                const int i = idx[r];
                dst[u] = reinterpret_cast<f32x4*>(out + (size_t)r * ld_out + c);
                if (i >= 0) { v[u] = *reinterpret_cast<const f32x4*>(table + (size_t)i * ld_table + c); live[u] = true; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (dst[u] == nullptr) continue;
            if (accumulate) { if (live[u]) *dst[u] += v[u]; }
            else *dst[u] = v[u];
        }
    }
}

// The same gather as ONE WAVE PER DESTINATION ROW, four rows in flight per wave (MI355X_MICROARCH.md, "Indexed rows": random
// 1 152-B rows gathered into registers read at 5.5-5.8 TB/s in this shape): a row is w4 <= 128 pieces of 16 B, lane l moves
// pieces l and l + 64; the row index is wave-uniform (scalar load), there is no per-element division, and every row's pieces
// are contiguous in one instruction.  Rows are dealt to waves round-robin so that neighbouring waves write neighbouring rows.
template <int U>
__device__ __forceinline__ void gather_rows_wave_body(const float* __restrict__ table, int ld_table, int w4, const int* __restrict__ idx, int rows,
                                                      float* __restrict__ out, int ld_out, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    const bool two = lane + 64 < w4, one = lane < w4;
    for (int r0 = wave; r0 < rows; r0 += U * n_waves) {
        f32x4 a[U], b[U];
        int src[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u * n_waves;
            src[u] = r < rows ? idx[r] : -1;                    // wave-uniform
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = b[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (src[u] >= 0) {
                const f32x4* row = reinterpret_cast<const f32x4*>(table + (size_t)src[u] * ld_table);
                if (one) a[u] = row[lane];
                if (two) b[u] = row[lane + 64];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u * n_waves;
            if (r >= rows) continue;
            f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)r * ld_out);
            if (accumulate) {
                if (src[u] >= 0) { if (one) dst[lane] += a[u]; if (two) dst[lane + 64] += b[u]; }
            } else {
                if (one) dst[lane] = a[u];
                if (two) dst[lane + 64] = b[u];
            }
        }
    }
}

// Access policy (round 6): plain loads and stores.  Round 5 streamed (nt) launches of >= 64 MB because that measured best behind a 640 MB
// WRITE -- 256 MB of dirty lines in the Infinity Cache, which plain stores must evict first (3.5 TB/s against 4.3 streamed).  Behind a 640 MB
// READ (cold but clean caches) plain accesses move 5.1-5.3 TB/s against 4.3-4.4, and inside the dense training step -- the case that counts --
// the launch runs at 4.9 TB/s plain against 4.1-4.3 streamed (tools/gather_sweep.py, profiles/r06_gather.txt): the 127 MB it writes are read
// again by the projection right behind it and the Infinity Cache can hold them.
template <int U>
__global__ __launch_bounds__(256) void gather_rows_wave_kernel(const float* __restrict__ table, int ld_table, int w4,
                                                               const int* __restrict__ idx, int rows_cap,
                                                               const int* __restrict__ rows_dyn, float* __restrict__ out, int ld_out,
                                                               int accumulate) {
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    gather_rows_wave_body<U>(table, ld_table, w4, idx, rows, out, ld_out, accumulate);
}


// ------------------------------------------------------------------ unique tokens of a batch (projection de-duplication)
// The GloVe projection of a token row, Linear(glove[tok]), depends on the token id alone (the table is frozen, dropout comes
// after it), and a batch repeats tokens heavily (Zipf: ~26 k token rows hold ~4.6 k distinct ids on the bench world).  The
// engine therefore projects every DISTINCT token once and expands the result to the rows (lego_expand_rows), and forms the
// projection's weight gradient from per-token sums of dH (lego_segment_sum_rows): exact up to fp32 summation order.
// These kernels build, on the prefetch stream, from the plan's row_tok[R]:
//   uniq[U]   distinct token ids, ascending        inv[R]   row -> its index in uniq
//   perm[R]   the rows grouped by inv (a counting sort; order inside a group follows an atomic cursor)
//   *n_uniq   U
// through a stamp table over the vocabulary (stamp[v] == epoch <=> token v occurs in this batch; never cleared) and two scans.
constexpr int kUqBlock = 1024;

__global__ void uq_mark_kernel(const int* __restrict__ row_tok, int R_cap, const int* __restrict__ R_dyn, uint32_t* stamp, uint32_t epoch, int V) {
    const int R = R_dyn != nullptr ? min(R_cap, *R_dyn) : R_cap;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        const int t = row_tok[r];
        if (t >= 0 && t < V) stamp[t] = epoch;             // racing writers store the same value
    }
}

__global__ __launch_bounds__(kUqBlock) void uq_count_kernel(const uint32_t* __restrict__ stamp, int V, uint32_t epoch, int* bsum) {
    __shared__ int wsum[16];
    const int v = blockIdx.x * kUqBlock + threadIdx.x;
    const int f = (v < V && stamp[v] == epoch) ? 1 : 0;
    const int incl = block_incl_scan_1024(f, wsum);
    if (threadIdx.x == kUqBlock - 1) bsum[blockIdx.x] = incl;
}

// single workgroup: exclusive scan of n <= cap entries in place, total -> *total (n read from n_dyn when given)
__global__ __launch_bounds__(kUqBlock) void uq_scan_kernel(int* a, int n_cap, const int* __restrict__ n_dyn, int* total, int* copy) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += kUqBlock) {
        const int i = base + threadIdx.x;
        const int x = i < n ? a[i] : 0;
        const int incl = block_incl_scan_1024(x, wsum);
        const int c = carry;
        if (i < n) {
            a[i] = c + incl - x;
            if (copy != nullptr) copy[i] = c + incl - x;
        }
        __syncthreads();
        if (threadIdx.x == kUqBlock - 1) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total != nullptr) *total = carry;
}

__global__ __launch_bounds__(kUqBlock) void uq_assign_kernel(const uint32_t* __restrict__ stamp, int V, uint32_t epoch, const int* __restrict__ boff,
                                                             int* uniq, int* rank, int* cnt) {
    __shared__ int wsum[16];
    const int v = blockIdx.x * kUqBlock + threadIdx.x;
    const int f = (v < V && stamp[v] == epoch) ? 1 : 0;
    const int incl = block_incl_scan_1024(f, wsum);
    if (f) {
        const int u = boff[blockIdx.x] + incl - 1;
        uniq[u] = v;
        rank[v] = u;
        cnt[u] = 0;
    }
}

// (sort_keys, optional: inv with rows past R set to INT_MAX -- the key array of lego_sort_rows, which sends them to the end)
__global__ void uq_inverse_kernel(const int* __restrict__ row_tok, int R_cap, const int* __restrict__ R_dyn, const int* __restrict__ rank,
                                  int V, int* inv, int* cnt, int* sort_keys) {
    const int R = R_dyn != nullptr ? min(R_cap, *R_dyn) : R_cap;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R_cap; r += gridDim.x * blockDim.x) {
        if (r >= R) { if (sort_keys != nullptr) sort_keys[r] = 0x7fffffff; continue; }
        const int t = row_tok[r];
        const int u = (t >= 0 && t < V) ? rank[t] : 0;
        inv[r] = u;
        if (sort_keys != nullptr) sort_keys[r] = u;
        if (cnt != nullptr) atomicAdd(&cnt[u], 1);
    }
}

__global__ void uq_fill_kernel(const int* __restrict__ inv, int R_cap, const int* __restrict__ R_dyn, int* cursor, int* perm) {
    const int R = R_dyn != nullptr ? min(R_cap, *R_dyn) : R_cap;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x)
        perm[atomicAdd(&cursor[inv[r]], 1)] = r;
}

// out[r, :] = keep(r, :) / (1 - p) * src[inv[r], :] (rows r < R): Dropout(Linear(.)) of embedding_hub.py:95-96 applied while
// the per-token projections are expanded to the batch's token rows.  One wave per row, lane = 4 columns (width <= 256), four
// rows in flight; the keep bits are the site's precomputed mask (byte [(r / 4) * width + col], bit r % 4) or Philox draws.
// rows of up to two small tables added to the expanded row: out[r] += tab_a[idx_a[r]] (idx >= 0) + tab_b[idx_b[r]] -- ConcatInputer sums the
// token look-up with the special-id and the category look-up (concat_inputer.py:96-114); the [SEP] / category positions get theirs here
struct ExpandAdd { const float* tab_a; const int* idx_a; int ld_a; const float* tab_b; const int* idx_b; int ld_b; };

__global__ __launch_bounds__(256) void expand_rows_kernel(const float* __restrict__ src, int ld_src, const int* __restrict__ inv, int rows_cap,
                                                          const int* __restrict__ rows_dyn, int width, Dropout drop,
                                                          const int* __restrict__ rowinfo, ExpandAdd add, float* __restrict__ out, int ld_out) {
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    constexpr int U = 4;                                     // a wave owns FOUR CONSECUTIVE rows 4g .. 4g+3: they share one keep-bit word
    const int c = blockIdx.y * 512 + 4 * lane;               // lane moves columns c .. c+3 and c+256 .. c+259 (rows up to 512 wide
    const int c2 = c + 256;                                  // per blockIdx.y: a 300-float GloVe row is one wave's work)
    const bool in = c < width, in2 = c2 < width;
    const float dinv = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    const bool dropping = drop.p > 0.f;
    for (int g = wave; 4 * g < rows; g += n_waves) {
        const int r0 = 4 * g;
        f32x4 v[U], v2[U];
        uint32_t kw = 0x0f0f0f0fu, kw2 = 0x0f0f0f0fu;
        if (dropping) {
            if (drop.mask != nullptr) {
                if (in) kw = *reinterpret_cast<const uint32_t*>(drop.mask + (uint64_t)g * (uint64_t)width + (uint64_t)c);
                if (in2) kw2 = *reinterpret_cast<const uint32_t*>(drop.mask + (uint64_t)g * (uint64_t)width + (uint64_t)c2);
            } else {
                if (in) kw = dropout_draw4(drop, r0, c, width) | (dropout_draw4(drop, r0, c + 1, width) << 8) |
                             (dropout_draw4(drop, r0, c + 2, width) << 16) | (dropout_draw4(drop, r0, c + 3, width) << 24);
                if (in2) kw2 = dropout_draw4(drop, r0, c2, width) | (dropout_draw4(drop, r0, c2 + 1, width) << 8) |
                               (dropout_draw4(drop, r0, c2 + 2, width) << 16) | (dropout_draw4(drop, r0, c2 + 3, width) << 24);
            }
        }
        // the four rows' index words first, by lanes 0..3 in ONE round trip (row liveness, source row, the two added look-ups), then every
        // row load they address in a second one (one after the other they were five dependent round trips: 22 us for 31 k rows)
        int m_live = 0, m_inv = 0, m_ia = -1, m_ib = -1;
        {
            const int r = r0 + (lane & 3);
            if (r < rows) {
                m_inv = inv[r];
                m_live = rowinfo != nullptr ? ((rowinfo[r] & RI_LIVE) != 0 ? 1 : 0) : 1;
                if (add.idx_a != nullptr) m_ia = add.idx_a[r];
                if (add.idx_b != nullptr) m_ib = add.idx_b[r];
            }
        }
        f32x4 ta[U], ta2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int live = __builtin_amdgcn_readlane(m_live, u), iv = __builtin_amdgcn_readlane(m_inv, u);
            const int ia = __builtin_amdgcn_readlane(m_ia, u), ib = __builtin_amdgcn_readlane(m_ib, u);
            v[u] = v2[u] = ta[u] = ta2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live) {                                  // a row without the live bit (or past the end) contributes zeros
                const float* row = src + (size_t)iv * ld_src;
                if (in) v[u] = *reinterpret_cast<const f32x4*>(row + c);
                if (in2) v2[u] = *reinterpret_cast<const f32x4*>(row + c2);
            }
            if (ia >= 0) {
                if (in) ta[u] = *reinterpret_cast<const f32x4*>(add.tab_a + (size_t)ia * add.ld_a + c);
                if (in2) ta2[u] = *reinterpret_cast<const f32x4*>(add.tab_a + (size_t)ia * add.ld_a + c2);
            }
            if (ib >= 0) {
                if (in) ta[u] += *reinterpret_cast<const f32x4*>(add.tab_b + (size_t)ib * add.ld_b + c);
                if (in2) ta2[u] += *reinterpret_cast<const f32x4*>(add.tab_b + (size_t)ib * add.ld_b + c2);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u;
            if (r >= rows) break;
            float* dst = out + (size_t)r * ld_out;
            if (in) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (kw >> (8 * i + u)) & 1u ? v[u][i] * dinv : 0.f;
                *reinterpret_cast<f32x4*>(dst + c) = o + ta[u];
            }
            if (in2) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (kw2 >> (8 * i + u)) & 1u ? v2[u][i] * dinv : 0.f;
                *reinterpret_cast<f32x4*>(dst + c2) = o + ta2[u];
            }
        }
    }
}

// out[u, :] = sum of g[r, :] over the rows r with inv[r] == u.  `perm` lists the rows grouped by u; one wave owns 32
// consecutive positions of it, keeps the running sum of the current group in registers (lane = 4 columns) and flushes it when
// the group changes: with a plain store when the whole group lies inside the wave's 32 positions, with float atomics when the
// group spans waves (hot tokens) -- `out` rows [0, U) must be zero on entry (zero_rows_kernel, same entry point).
__global__ void zero_rows_kernel(float* out, int ld, int width, int rows_cap, const int* __restrict__ rows_dyn) {
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const long long total = (long long)rows * (width / 4);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / (width / 4)), c = (int)(e % (width / 4)) * 4;
        *reinterpret_cast<f32x4*>(out + (size_t)r * ld + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

constexpr int kSegRows = 32;         // positions of `perm` per wave, as two batches of 16 row loads in flight
constexpr int kSegBatch = 16;

// add a wave's f32x4-per-lane row (lane = 4 consecutive columns) to out_row with float atomics whose every instruction covers 256
// CONTIGUOUS bytes (instruction i: columns 64 i + lane; fetched from lane 16 i + lane / 4): the memory-side atomic units take a
// 256-B wave-instruction as four 64-B requests, the 16-B-per-lane layout as sixteen (MI355X_MICROARCH.md, "Global float atomics")
__device__ __forceinline__ void atomic_add_row(float* out_row, int col0, int width, const f32x4& acc, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int src = 16 * i + (lane >> 2);
        const float t0 = __shfl(acc[0], src, 64), t1 = __shfl(acc[1], src, 64), t2 = __shfl(acc[2], src, 64), t3 = __shfl(acc[3], src, 64);
        const int e = lane & 3;
        const float v = e == 0 ? t0 : (e == 1 ? t1 : (e == 2 ? t2 : t3));
        const int c = col0 + 64 * i + lane;
        if (c < width) atomicAdd(out_row + c, v);
    }
}
__device__ __forceinline__ void atomic_add_row(float* out_row, int col0, int width, const float& acc, int lane) {
    if (col0 + lane < width) atomicAdd(out_row + col0 + lane, acc);           // one column per lane: already 256 contiguous bytes
}

// VEC = columns per lane: 4 (a wave covers 256 columns of a row with one 16-byte load per lane) or 1 (64 columns, a dword per lane).
// Round 5: VEC = 1 is the default -- the launch has R / 32 waves per column slice, and with 256-column slices that is ~840 waves for a
// NAML batch (3 per CU: every wave waits out its two batches of 16 dependent-free row loads with nothing beside it); four times the
// slices = four times the waves for the same loads, groups and atomics per column (the VEC = 4 form stays for the tuning tools).
template <int VEC>
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const float* __restrict__ g, int ld_g, int width, const int* __restrict__ perm,
                                                               const int* __restrict__ inv, const int* __restrict__ sorted_keys,
                                                               int R_cap, const int* __restrict__ R_dyn, float* out, int ld_out,
                                                               const uint8_t* __restrict__ keep_mask, float keep_scale,
                                                               const int* __restrict__ rowinfo) {
    using V = std::conditional_t<VEC == 4, f32x4, float>;
    const int R = R_dyn != nullptr ? min(R_cap, *R_dyn) : R_cap;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int p0 = wave * kSegRows;
    if (p0 >= R) return;
    const int n = min(kSegRows, R - p0);
    const int col0 = blockIdx.y * 64 * VEC;
    const int c = col0 + VEC * lane;
    const bool in = c < width;
    // lanes 0 .. kSegRows + 1: position p0 - 1 + lane (one before and one past the chunk, for the two boundary decisions)
    const int pp = p0 - 1 + lane;
    int my_row = 0, my_u = -1, my_live = 1;
    if (lane < kSegRows + 2 && pp >= 0 && pp < R) {              // sorted_keys (the sort's key output) saves the dependent inv[] load
        my_row = perm[pp];
        my_u = sorted_keys != nullptr ? sorted_keys[pp] : inv[my_row];
        if (rowinfo != nullptr) my_live = (rowinfo[my_row] & RI_LIVE) != 0;      // ONE dependent round trip for the chunk's 32 rows
    }
    const int u_before = __shfl(my_u, 0, 64);                     // -1 when p0 == 0
    int cur = __shfl(my_u, 1, 64);
    bool began_here = cur != u_before;
    V acc = V{};
    for (int b0 = 0; b0 < n; b0 += kSegBatch) {
        V v[kSegBatch];
        int uu[kSegBatch];
#pragma unroll
        for (int j = 0; j < kSegBatch; ++j) {
            const int i = min(b0 + j, n - 1);
            const int row = __shfl(my_row, 1 + i, 64);
            uu[j] = __shfl(my_u, 1 + i, 64);
            const int live = __shfl(my_live, 1 + i, 64);                // by every lane: the source lane may have `in` == false
            const bool take = b0 + j < n && in && live != 0;            // a masked row adds nothing
            v[j] = take ? *reinterpret_cast<const V*>(g + (size_t)row * ld_g + c) : V{};
            if (keep_mask != nullptr && take) {      // Dropout backward of the row: byte [(row / 4) * width + col], bit row % 4
                if constexpr (VEC == 4) {
                    const uint32_t kw = *reinterpret_cast<const uint32_t*>(keep_mask + (uint64_t)(row >> 2) * (uint64_t)width + (uint64_t)c);
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) v[j][i2] = (kw >> (8 * i2 + (row & 3))) & 1u ? v[j][i2] * keep_scale : 0.f;
                } else {
                    const uint32_t kb = keep_mask[(uint64_t)(row >> 2) * (uint64_t)width + (uint64_t)c];
                    v[j] = (kb >> (row & 3)) & 1u ? v[j] * keep_scale : 0.f;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kSegBatch; ++j) {
            if (b0 + j < n) {
                if (uu[j] != cur) {                                // wave-uniform: the group ended inside the chunk
                    float* dst = out + (size_t)cur * ld_out;
                    if (began_here) { if (in) *reinterpret_cast<V*>(dst + c) = acc; }
                    else atomic_add_row(dst, col0, width, acc, lane);
                    acc = V{};
                    cur = uu[j];
                    began_here = true;
                }
                acc += v[j];
            }
        }
    }
    const int u_after = __shfl(my_u, 1 + n, 64);                   // -1 past the end of the rows
    float* dst = out + (size_t)cur * ld_out;
    if (began_here && (p0 + n >= R || u_after != cur)) { if (in) *reinterpret_cast<V*>(dst + c) = acc; }
    else atomic_add_row(dst, col0, width, acc, lane);
}

// NRMS sequence rows: the plan's row_tok word encodes token id (>= 0), SEP (-2) or category (-(3+cat))
__global__ void nrms_decode_rows_kernel(const int* __restrict__ row_tok, int R_cap, const int* __restrict__ R_dyn,
                                        int* idx_tok, int* idx_special, int* idx_cat, int* tokinfo) {
    const int R = R_dyn != nullptr ? min(R_cap, *R_dyn) : R_cap;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int v = row_tok[r];
    idx_tok[r] = v >= 0 ? v : -1;
    idx_special[r] = v == -2 ? 2 : -1;                 // [SEP] = 2 (concat_inputer.py:27-30)
    idx_cat[r] = v <= -3 ? -(v + 3) : -1;
    tokinfo[r] = v >= 0 ? RI_LIVE : 0;
}

// Key space of the per-key in-projection (engine.NrmsEngine.keyspace): every position of a ConcatInputer sequence is a function of ONE id --
// tokens keep theirs, [SEP] (-2 in row_tok) becomes V, category c (-(3 + c)) becomes V + 1 + c -- and, after lego_unique_tokens over that
// space, the distinct keys go back to per-table row indices (-1 = not this table's).  (Round 5: these were fifteen torch element-wise
// launches per planned batch on the prefetch stream.)
__global__ void nrms_key_rows_kernel(const int* __restrict__ row_tok, int R_cap, const int* __restrict__ R_dyn, int V, int* __restrict__ row_key) {
    const int R = R_dyn != nullptr ? min(R_cap, *R_dyn) : R_cap;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int v = row_tok[r];
    row_key[r] = v >= 0 ? v : (V - 2) - v;
}
__global__ void nrms_decode_keys_kernel(const int* __restrict__ uniq, int U_cap, const int* __restrict__ U_dyn, int V, int* __restrict__ idx_tok,
                                        int* __restrict__ idx_special, int* __restrict__ idx_cat, int* __restrict__ keyinfo) {
    const int U = U_dyn != nullptr ? min(U_cap, *U_dyn) : U_cap;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int k = uniq[u];
    idx_tok[u] = k < V ? k : -1;
    idx_special[u] = k == V ? 2 : -1;                   // [SEP] = 2 (concat_inputer.py:27-30)
    idx_cat[u] = k > V ? k - (V + 1) : -1;
    keyinfo[u] = k < V ? RI_LIVE : 0;
}

// Gradients of the two small embedding tables of ConcatInputer (concat_inputer.py:58-114) straight from the segment
// layout: an item's sequence is [title..., SEP, category, SEP], so of its L rows exactly three feed these tables -- rows
// L-3 and L-1 the [SEP] row of the special-id table, row L-2 the item's category row.  One workgroup per 32 items, one wave
// per item at a time, lane = 4 columns; SEP sums stay in registers, category sums in an LDS image of the table; one round of
// global atomics per workgroup.  (Round 1 ran the generic small-table scatter twice over ALL sequence rows: 2 x 62 us per
// NRMS step to find the three rows per item.)
constexpr int kSmallTableRows = 32;          // rows of a table that the LDS-accumulating scatter kernels can hold
constexpr int kSpecItems = 8;            // items per workgroup (two per wave): ~440 workgroups at the headline batch
__global__ __launch_bounds__(256) void nrms_special_grads_kernel(const int* __restrict__ seg_off, int n_cap, const int* __restrict__ n_dyn,
                                                                 const int* __restrict__ idx_cat, const float* __restrict__ g, int ld,
                                                                 int width, float* g_sep, float* g_cat, int ld_cat, int n_cat) {
    __shared__ float tab[kSmallTableRows][256];
    __shared__ float sep[4][256];
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int first = blockIdx.x * kSpecItems;
    if (first >= n) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: row indices and addresses stay scalar)
    const int c = blockIdx.y * 256 + 4 * lane;
    for (int e = threadIdx.x; e < n_cat * 256; e += 256) (&tab[0][0])[e] = 0.f;
    __syncthreads();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int i = first + wave; i < min(first + kSpecItems, n); i += 4) {
        const int beg = seg_off[i], L = seg_off[i + 1] - beg;
        if (L < 3 || c >= width) continue;
        const float* r = g + (size_t)(beg + L - 3) * ld + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(r), b = *reinterpret_cast<const f32x4*>(r + ld),
                    d = *reinterpret_cast<const f32x4*>(r + 2 * (size_t)ld);
        s += a + d;
        const int cat = idx_cat[beg + L - 2];
        if (cat >= 0 && cat < n_cat) {
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(&tab[cat][4 * lane + k], b[k]);
        }
    }
    *reinterpret_cast<f32x4*>(&sep[wave][4 * lane]) = s;
    __syncthreads();
    const int cc = blockIdx.y * 256 + threadIdx.x;
    if (cc < width) {
        atomicAdd(g_sep + cc, (sep[0][threadIdx.x] + sep[1][threadIdx.x]) + (sep[2][threadIdx.x] + sep[3][threadIdx.x]));
        for (int t = 0; t < n_cat; ++t) {
            const float v = tab[t][threadIdx.x];
            if (v != 0.f) atomicAdd(g_cat + (size_t)t * ld_cat + cc, v);
        }
    }
}

// x[r,:] *= live(rowinfo[r]) * dropout scale  (backward of a masked + dropped projection output)
__global__ void mask_dropout_rows_kernel(float* __restrict__ x, int ld, int R_cap, const int* __restrict__ R_dyn, int width,
                                         const int* __restrict__ rowinfo, Dropout drop) {
    const int R = R_dyn != nullptr ? min(R_cap, *R_dyn) : R_cap;
    // one thread = 4 rows x 1 column: ONE Philox call yields the four keep decisions of that group (the element-per-thread form
    // ran a whole call per element and threw three quarters of it away: 64-71 us per NRMS step)
    const long long total = (long long)((R + 3) / 4) * width;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r0 = (int)(e / width) * 4;
        const int c = (int)(e - (long long)(r0 / 4) * width);
        float ds[4];
        dropout_scale4(drop, r0, c, width, ds);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + i;
            if (r >= R) break;
            float* p = x + (size_t)r * ld + c;
            *p = (rowinfo != nullptr && !(rowinfo[r] & RI_LIVE)) ? 0.f : *p * ds[i];
        }
    }
}

// keep bits of one dropout site for rows [0, rows): byte [(row / 4) * cols + col], bit i = row % 4 -- exactly the
// decisions dropout_scale4 draws in the epilogues, generated ahead of time (prefetch stream) so that the GEMM
// epilogues read one byte instead of running Philox (a wave64 Philox call is ~500 cycles of VALU)
__global__ void dropout_mask_kernel(Dropout d, int rows_cap, const int* __restrict__ rows_dyn, int cols, uint8_t* __restrict__ mask) {
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const int groups4 = (rows + 3) >> 2;
    const long long total = (long long)((rows + 7) >> 3) * cols;          // one thread (one Philox call) per 8 rows x column
    d.mask = nullptr;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int g8 = (int)(e / cols);
        const int c = (int)(e - (long long)g8 * cols);
        const uint32_t bits = dropout_draw8(d, g8, c, cols);
        mask[(size_t)(2 * g8) * cols + c] = (uint8_t)(bits & 15u);
        if (2 * g8 + 1 < groups4) mask[(size_t)(2 * g8 + 1) * cols + c] = (uint8_t)(bits >> 4);
    }
}

// (row_lo, row_hi): only destination rows in [row_lo, row_hi) are touched -- the table gradient is produced bucket by
// bucket, so that bucket k is on the wire while bucket k+1 is still being scattered (train_step.TrainStep, world > 1)
__global__ void scatter_add_rows_kernel(float* grad_table, int ld_table, int width, const int* __restrict__ idx,
                                        int rows_cap, const int* __restrict__ rows_dyn, const float* __restrict__ g, int ld_g,
                                        int row_lo, int row_hi) {
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    const long long total = (long long)rows * width;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / width);
        const int c = (int)(e - (long long)r * width);
        const int i = idx[r];
        if (i >= row_lo && i < row_hi) atomicAdd(grad_table + (size_t)i * ld_table + c, g[(size_t)r * ld_g + c]);
    }
}

// Scatter-add into a SMALL table (category: 18 rows, ConcatInputer specials: 3 rows): every source row hits one of a
// handful of destination rows, so per-row global atomics serialise on a few cache lines.  One workgroup per 16 source
// rows x 256 columns: each wave loads its 4 rows at once (lane = columns lane, lane+64, ...: coalesced, and the LDS
// adds below are bank-conflict free), adds them into an LDS image of the table (ds_add_f32), then the touched table
// rows are added to memory ONCE per (workgroup, row, column).
constexpr int kSmallChunk = 16;
__global__ __launch_bounds__(256) void scatter_add_small_kernel(float* grad_table, int ld_table, int width, int table_rows,
                                                                const int* __restrict__ idx, int rows_cap,
                                                                const int* __restrict__ rows_dyn,
                                                                const float* __restrict__ g, int ld_g, int iters) {
    __shared__ float tab[kSmallTableRows][256];
    __shared__ int touched[kSmallTableRows];
    const int rows = rows_dyn != nullptr ? min(rows_cap, *rows_dyn) : rows_cap;
    // a workgroup folds `iters` chunks of 16 rows into its LDS image before it touches memory (the host picks `iters`
    // so that a long row range -- NRMS's ~50 k token rows, most of them not in this table -- still gives ~500 workgroups:
    // one LDS clear, one barrier pair and one round of global atomics per 16 rows was most of the kernel there)
    const int first = blockIdx.x * iters * kSmallChunk;
    if (first >= rows) return;
    const int c0 = blockIdx.y * 256;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: row indices and addresses stay scalar)
    for (int e = threadIdx.x; e < table_rows * 256; e += 256) (&tab[0][0])[e] = 0.f;
    if (threadIdx.x < kSmallTableRows) touched[threadIdx.x] = 0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const int r0 = first + it * kSmallChunk;
        if (r0 >= rows) break;
        int t[4];
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                  // all 16 loads of the wave are in flight before the first use
            const int r = r0 + wave * 4 + u;
            t[u] = r < rows ? idx[r] : -1;
            if (t[u] >= table_rows) t[u] = -1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + lane + 64 * i;
                v[u][i] = (t[u] >= 0 && c < width) ? g[(size_t)r * ld_g + c] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (t[u] < 0) continue;                    // wave-uniform
            if (lane == 0) touched[t[u]] = 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(&tab[t[u]][lane + 64 * i], v[u][i]);
        }
    }
    __syncthreads();
    for (int tr = 0; tr < table_rows; ++tr) {
        if (!touched[tr]) continue;
        const int cc = c0 + threadIdx.x;
        if (cc < width) atomicAdd(grad_table + (size_t)tr * ld_table + cc, tab[tr][threadIdx.x]);
    }
}

__global__ void gather_i32_kernel(const int* __restrict__ table, const int* __restrict__ idx, int n_cap,
                                  const int* __restrict__ n_dyn, int* out) {
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = table[idx[i]];
}

// rowinfo[i] = RI_LIVE iff segment i has rows (the live-mask epilogue of lego_linear_fwd then zeroes the output rows of empty segments)
__global__ void segment_live_kernel(const int* __restrict__ seg_off, int n_cap, const int* __restrict__ n_dyn, int* __restrict__ out) {
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = seg_off[i + 1] > seg_off[i] ? RI_LIVE : 0;
}

// out[c] += sum over rows; block = 64 columns x 4 row lanes, 256 rows per block
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ldx, int M_cap,
                                                     const int* __restrict__ M_dyn, const int* __restrict__ off_dyn,
                                                     int N, float* out) {
    __shared__ float part[4][64];
    const int M = M_dyn != nullptr ? min(M_cap, *M_dyn) : M_cap;
    const int off = off_dyn != nullptr ? *off_dyn : 0;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r0 = blockIdx.y * 256;
    if (r0 >= M) return;
    float s = 0.f;
    if (c < N)
        for (int r = r0 + ry; r < min(M, r0 + 256); r += 4) s += x[(size_t)(r + off) * ldx + c];
    part[ry][threadIdx.x & 63] = s;
    __syncthreads();
    if (ry == 0 && c < N) atomicAdd(out + c, (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// the same sums with 16-B loads: block = 64 float4 columns (256 columns) x 4 row lanes, 256 rows per block, four rows in
// flight per lane (the dword version above moved 1.7 TB/s on the 30 k x 768 QKV gradient of NRMS)
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ x, int ldx, int M_cap,
                                                      const int* __restrict__ M_dyn, const int* __restrict__ off_dyn,
                                                      int N, float* out) {
    __shared__ f32x4 part[4][64];
    const int M = M_dyn != nullptr ? min(M_cap, *M_dyn) : M_cap;
    const int off = off_dyn != nullptr ? *off_dyn : 0;
    const int lane = threadIdx.x & 63, ry = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * 256;
    if (r0 >= M) return;
    const int r_end = min(M, r0 + 256);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (c < N) {
        const float* p = x + (size_t)off * ldx + c;
        int r = r0 + ry;
        for (; r + 12 < r_end; r += 16) {
            s0 += *reinterpret_cast<const f32x4*>(p + (size_t)r * ldx);
            s1 += *reinterpret_cast<const f32x4*>(p + (size_t)(r + 4) * ldx);
            s2 += *reinterpret_cast<const f32x4*>(p + (size_t)(r + 8) * ldx);
            s3 += *reinterpret_cast<const f32x4*>(p + (size_t)(r + 12) * ldx);
        }
        for (; r < r_end; r += 4) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)r * ldx);
    }
    part[ry][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && c < N) {
        const f32x4 t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(out + c + i, t[i]);
    }
}

__global__ void conv3_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int Dout, int Din) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;      // over [tap][o][c]
    const int per = Dout * Din;
    if (e >= 3 * per) return;
    const int tap = e / per, oc = e - tap * per;
    wt[e] = w[(size_t)oc * 3 + tap];
}
__global__ void conv3_unpack_add_kernel(float* __restrict__ dwt, float* __restrict__ dw, int Dout, int Din) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;      // over [o][c][tap]
    const int per = Dout * Din;
    if (e >= 3 * per) return;
    const int oc = e / 3, tap = e - oc * 3;
    float* src = dwt + (size_t)tap * per + oc;
    atomicAdd(dw + e, *src);      // micro-batches accumulate into the same gradient concurrently
    *src = 0.f;                   // the split-K accumulator is handed back clean: no fill launch before the next backward
}

// ------------------------------------------------------------------ additive attention pooling
constexpr int kMaxChunks = 4;       // columns handled per lane: 4 floats x kMaxChunks  (D, A <= 1024)
constexpr int kMaxSegRows = 256;    // rows per segment incl. the extra row

__device__ __forceinline__ float dot_row(const float* __restrict__ a, const float* __restrict__ b, int n, int lane) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
        const int c = 4 * lane + 256 * j;
        if (c < n) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(a + c);
            const f32x4 y = *reinterpret_cast<const f32x4*>(b + c);
            s += (x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]);
        }
    }
    return wave_sum(s);
}

// One 256-thread workgroup per segment; the four waves take rows l = w, w+4, ... so 4 rows of the segment
// are in flight at once (a 31-row item segment = 8 dependent row steps per wave instead of 31), partial
// sums meet in LDS.  Row reductions are wave-shuffle sums (dot_row).
__global__ __launch_bounds__(256) void additive_pool_fwd_kernel(
    const float* __restrict__ t, int ldt, const float* __restrict__ x, int ldx, const float* __restrict__ w2,
    const int* __restrict__ seg_off, const int* __restrict__ rowinfo, const int* __restrict__ extra_off_dyn,
    int n_cap, const int* __restrict__ n_dyn, int D, int A, float* __restrict__ out, int ldo, float* __restrict__ wrow) {
    __shared__ float red_acc[4][kMaxChunks * 256];
    __shared__ float red_s[4];
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: row indices and addresses stay scalar)
    const int i = blockIdx.x;
    if (i >= n) return;
    const int beg = seg_off[i], len = seg_off[i + 1] - beg;
    const int extra = extra_off_dyn != nullptr ? *extra_off_dyn + i : -1;
    const int cnt = len + (extra >= 0 ? 1 : 0);
    f32x4 acc[kMaxChunks];
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float stash = 0.f, s = 0.f;
    for (int l = wave, it = 0; l < cnt; l += 4, ++it) {
        const int row = l < len ? beg + l : extra;
        const float a = dot_row(t + (size_t)row * ldt, w2, A, lane);
        const bool live = (l < len && rowinfo != nullptr) ? (rowinfo[row] & RI_LIVE) != 0 : true;
        const float e = live ? expf(a) : 0.f;
        s += e;
#pragma unroll
        for (int j = 0; j < kMaxChunks; ++j) {
            const int c = 4 * lane + 256 * j;
            if (c < D) acc[j] += e * *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + c);
        }
        if (it == lane) stash = e;
    }
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
        const int c = 4 * lane + 256 * j;
        if (c < D) *reinterpret_cast<f32x4*>(&red_acc[wave][c]) = acc[j];
    }
    if (lane == 0) red_s[wave] = s;
    __syncthreads();
    const float inv = 1.f / ((red_s[0] + red_s[1]) + (red_s[2] + red_s[3]) + kEps);
    for (int c = threadIdx.x; c < D; c += 256)
        out[(size_t)i * ldo + c] = ((red_acc[0][c] + red_acc[1][c]) + (red_acc[2][c] + red_acc[3][c])) * inv;
    const int l = wave + 4 * lane;                       // the row this lane stashed
    if (l < cnt) wrow[l < len ? beg + l : extra] = stash * inv;
}

// Fast forms for D, A <= 256 (lane = 4 columns, one f32x4 per row and lane): a wave loads ALL of its rows of a batch of 32 segment
// rows -- the tanh rows and the value rows, eight of each -- before the first reduction, instead of one dependent row after the
// other (a 31-row item segment was eight dependent global round trips per wave; these kernels sit on the latency-bound neck of the
// step between the item tower and the user tower).  Same arithmetic, same summation order across the waves.
constexpr int kPoolRW = 8;          // rows per wave and batch
__global__ __launch_bounds__(256) void additive_pool_fwd_fast_kernel(
    const float* __restrict__ t, int ldt, const float* __restrict__ x, int ldx, const float* __restrict__ w2,
    const int* __restrict__ seg_off, const int* __restrict__ rowinfo, const int* __restrict__ extra_off_dyn,
    int n_cap, const int* __restrict__ n_dyn, int D, int A, float* __restrict__ out, int ldo, float* __restrict__ wrow) {
    __shared__ float red_acc[4][256];
    __shared__ float red_s[4];
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: row indices and addresses stay scalar)
    const int i = blockIdx.x;
    if (i >= n) return;
    const int beg = seg_off[i], len = seg_off[i + 1] - beg;
    const int extra = extra_off_dyn != nullptr ? *extra_off_dyn + i : -1;
    const int cnt = len + (extra >= 0 ? 1 : 0);
    const int c = 4 * lane;
    const bool inD = c < D, inA = c < A;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 w2v = inA ? *reinterpret_cast<const f32x4*>(w2 + c) : zero4;
    f32x4 acc = zero4;
    float stash = 0.f, s = 0.f;
    int it = 0;
    for (int b0 = 0; b0 < cnt; b0 += 4 * kPoolRW) {
        f32x4 tv[kPoolRW], xv[kPoolRW];
        bool live[kPoolRW];
#pragma unroll
        for (int u = 0; u < kPoolRW; ++u) {
            const int l = b0 + wave + 4 * u;
            const bool ok = l < cnt;
            const int row = l < len ? beg + l : extra;
            tv[u] = (ok && inA) ? *reinterpret_cast<const f32x4*>(t + (size_t)row * ldt + c) : zero4;
            xv[u] = (ok && inD) ? *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + c) : zero4;
            live[u] = ok && ((l < len && rowinfo != nullptr) ? (rowinfo[row] & RI_LIVE) != 0 : true);
        }
#pragma unroll
        for (int u = 0; u < kPoolRW; ++u) {
            if (b0 + wave + 4 * u < cnt) {                   // wave-uniform
                const float a = wave_sum((tv[u][0] * w2v[0] + tv[u][1] * w2v[1]) + (tv[u][2] * w2v[2] + tv[u][3] * w2v[3]));
                const float e = live[u] ? expf(a) : 0.f;
                s += e;
                acc += e * xv[u];
                if (it == lane) stash = e;
                ++it;
            }
        }
    }
    if (inD) *reinterpret_cast<f32x4*>(&red_acc[wave][c]) = acc;
    if (lane == 0) red_s[wave] = s;
    __syncthreads();
    const float inv = 1.f / ((red_s[0] + red_s[1]) + (red_s[2] + red_s[3]) + kEps);
    for (int cc = threadIdx.x; cc < D; cc += 256)
        out[(size_t)i * ldo + cc] = ((red_acc[0][cc] + red_acc[1][cc]) + (red_acc[2][cc] + red_acc[3][cc])) * inv;
    const int l = wave + 4 * lane;                       // the row this lane stashed
    if (l < cnt) wrow[l < len ? beg + l : extra] = stash * inv;
}

constexpr int kPoolReplicas = 32;
__global__ __launch_bounds__(256) void additive_pool_bwd_kernel(
    float* __restrict__ t, int ldt, const float* __restrict__ x, int ldx, const float* __restrict__ w2,
    const int* __restrict__ seg_off, const int* __restrict__ extra_off_dyn, int n_cap, const int* __restrict__ n_dyn,
    int D, int A, const float* __restrict__ gout, int ldgo, const float* __restrict__ wrow,
    float* __restrict__ dx, int lddx, float* gw2, float* gb1, float* scratch) {
    __shared__ float red[2][4][kMaxChunks * 256];
    __shared__ float red_s[2][4];
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: row indices and addresses stay scalar)
    f32x4 aw2[kMaxChunks], ab1[kMaxChunks], w2v[kMaxChunks];
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
        aw2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = 4 * lane + 256 * j;
        w2v[j] = c < A ? *reinterpret_cast<const f32x4*>(w2 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int par = 0;
    for (int i = blockIdx.x; i < n; i += gridDim.x, par ^= 1) {
        const int beg = seg_off[i], len = seg_off[i + 1] - beg;
        const int extra = extra_off_dyn != nullptr ? *extra_off_dyn + i : -1;
        const int cnt = len + (extra >= 0 ? 1 : 0);
        const float* go = gout + (size_t)i * ldgo;
        float stash = 0.f, sdw = 0.f;
        for (int l = wave, it = 0; l < cnt; l += 4, ++it) {
            const int row = l < len ? beg + l : extra;
            const float dw = dot_row(go, x + (size_t)row * ldx, D, lane);
            sdw += wrow[row] * dw;
            if (it == lane) stash = dw;
        }
        if (lane == 0) red_s[par][wave] = sdw;
        __syncthreads();
        sdw = (red_s[par][0] + red_s[par][1]) + (red_s[par][2] + red_s[par][3]);
        for (int l = wave, it = 0; l < cnt; l += 4, ++it) {
            const int row = l < len ? beg + l : extra;
            const float dw = __shfl(stash, it, 64);
            const float w = wrow[row];
            const float da = w * (dw - sdw);
#pragma unroll
            for (int j = 0; j < kMaxChunks; ++j) {
                const int c = 4 * lane + 256 * j;
                if (c < D) *reinterpret_cast<f32x4*>(dx + (size_t)row * lddx + c) = w * *reinterpret_cast<const f32x4*>(go + c);
                if (c < A) {
                    f32x4* tp = reinterpret_cast<f32x4*>(t + (size_t)row * ldt + c);
                    const f32x4 tv = *tp;
                    const f32x4 dpre = da * w2v[j] * (1.f - tv * tv);
                    aw2[j] += da * tv;
                    ab1[j] += dpre;
                    *tp = dpre;
                }
            }
        }
    }
    // block reduction of the two parameter-gradient partials, then one atomic per column per block
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
        const int c = 4 * lane + 256 * j;
        *reinterpret_cast<f32x4*>(&red[0][wave][c]) = aw2[j];
        *reinterpret_cast<f32x4*>(&red[1][wave][c]) = ab1[j];
    }
    __syncthreads();
    // ~1000 workgroups finishing together would serialise their adds on the 2 x A words of gw2 / gb1 (the adds execute
    // at the memory side, one per address at a time): with a scratch they go to one of kPoolReplicas copies instead
    // and pool_replica_reduce_kernel folds the copies afterwards
    if (scratch != nullptr) {
        gw2 = scratch + (size_t)(blockIdx.x % kPoolReplicas) * 2 * A;
        gb1 = gw2 + A;
    }
    for (int c = threadIdx.x; c < A; c += 256) {
        atomicAdd(gw2 + c, (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]));
        atomicAdd(gb1 + c, (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]));
    }
}

// fast backward (D, A <= 256): segments of up to 32 rows keep every row of both passes in registers (one round of loads); longer
// segments (click histories of up to 50 items on the un-fused user side) run the row-by-row passes of the generic kernel
__global__ __launch_bounds__(256) void additive_pool_bwd_fast_kernel(
    float* __restrict__ t, int ldt, const float* __restrict__ x, int ldx, const float* __restrict__ w2,
    const int* __restrict__ seg_off, const int* __restrict__ extra_off_dyn, int n_cap, const int* __restrict__ n_dyn,
    int D, int A, const float* __restrict__ gout, int ldgo, const float* __restrict__ wrow,
    float* __restrict__ dx, int lddx, float* gw2, float* gb1, float* scratch) {
    __shared__ float red[2][4][256];
    __shared__ float red_s[2][4];
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: row indices and addresses stay scalar)
    const int c = 4 * lane;
    const bool inD = c < D, inA = c < A;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 w2v = inA ? *reinterpret_cast<const f32x4*>(w2 + c) : zero4;
    f32x4 aw2 = zero4, ab1 = zero4;
    int par = 0;
    for (int i = blockIdx.x; i < n; i += gridDim.x, par ^= 1) {
        const int beg = seg_off[i], len = seg_off[i + 1] - beg;
        const int extra = extra_off_dyn != nullptr ? *extra_off_dyn + i : -1;
        const int cnt = len + (extra >= 0 ? 1 : 0);
        const f32x4 gov = inD ? *reinterpret_cast<const f32x4*>(gout + (size_t)i * ldgo + c) : zero4;
        if (cnt <= 4 * kPoolRW) {                            // block-uniform
            f32x4 tv[kPoolRW], xv[kPoolRW];
            float wv[kPoolRW], dwv[kPoolRW];
            int rows[kPoolRW];
#pragma unroll
            for (int u = 0; u < kPoolRW; ++u) {
                const int l = wave + 4 * u;
                const bool ok = l < cnt;
                rows[u] = ok ? (l < len ? beg + l : extra) : -1;
                xv[u] = (ok && inD) ? *reinterpret_cast<const f32x4*>(x + (size_t)rows[u] * ldx + c) : zero4;
                tv[u] = (ok && inA) ? *reinterpret_cast<const f32x4*>(t + (size_t)rows[u] * ldt + c) : zero4;
                wv[u] = ok ? wrow[rows[u]] : 0.f;
            }
            float sdw = 0.f;
#pragma unroll
            for (int u = 0; u < kPoolRW; ++u) {
                dwv[u] = wave_sum((gov[0] * xv[u][0] + gov[1] * xv[u][1]) + (gov[2] * xv[u][2] + gov[3] * xv[u][3]));
                sdw += wv[u] * dwv[u];
            }
            if (lane == 0) red_s[par][wave] = sdw;
            __syncthreads();
            sdw = (red_s[par][0] + red_s[par][1]) + (red_s[par][2] + red_s[par][3]);
#pragma unroll
            for (int u = 0; u < kPoolRW; ++u) {
                if (rows[u] < 0) continue;                   // wave-uniform
                const float w = wv[u];
                const float da = w * (dwv[u] - sdw);
                if (inD) *reinterpret_cast<f32x4*>(dx + (size_t)rows[u] * lddx + c) = w * gov;
                if (inA) {
                    const f32x4 dpre = da * w2v * (1.f - tv[u] * tv[u]);
                    aw2 += da * tv[u];
                    ab1 += dpre;
                    *reinterpret_cast<f32x4*>(t + (size_t)rows[u] * ldt + c) = dpre;
                }
            }
        } else {
            float stash = 0.f, sdw = 0.f;
            for (int l = wave, it = 0; l < cnt; l += 4, ++it) {
                const int row = l < len ? beg + l : extra;
                const f32x4 xr = inD ? *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + c) : zero4;
                const float dw = wave_sum((gov[0] * xr[0] + gov[1] * xr[1]) + (gov[2] * xr[2] + gov[3] * xr[3]));
                sdw += wrow[row] * dw;
                if (it == lane) stash = dw;
            }
            if (lane == 0) red_s[par][wave] = sdw;
            __syncthreads();
            sdw = (red_s[par][0] + red_s[par][1]) + (red_s[par][2] + red_s[par][3]);
            for (int l = wave, it = 0; l < cnt; l += 4, ++it) {
                const int row = l < len ? beg + l : extra;
                const float dw = __shfl(stash, it, 64);
                const float w = wrow[row];
                const float da = w * (dw - sdw);
                if (inD) *reinterpret_cast<f32x4*>(dx + (size_t)row * lddx + c) = w * gov;
                if (inA) {
                    f32x4* tp = reinterpret_cast<f32x4*>(t + (size_t)row * ldt + c);
                    const f32x4 tvv = *tp;
                    const f32x4 dpre = da * w2v * (1.f - tvv * tvv);
                    aw2 += da * tvv;
                    ab1 += dpre;
                    *tp = dpre;
                }
            }
        }
    }
    *reinterpret_cast<f32x4*>(&red[0][wave][c]) = aw2;
    *reinterpret_cast<f32x4*>(&red[1][wave][c]) = ab1;
    __syncthreads();
    if (scratch != nullptr) {
        gw2 = scratch + (size_t)(blockIdx.x % kPoolReplicas) * 2 * A;
        gb1 = gw2 + A;
    }
    for (int cc = threadIdx.x; cc < A; cc += 256) {
        atomicAdd(gw2 + cc, (red[0][0][cc] + red[0][1][cc]) + (red[0][2][cc] + red[0][3][cc]));
        atomicAdd(gb1 + cc, (red[1][0][cc] + red[1][1][cc]) + (red[1][2][cc] + red[1][3][cc]));
    }
}

__global__ void pool_replica_reduce_kernel(float* __restrict__ scratch, int A, float* gw2, float* gb1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * A) return;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < kPoolReplicas; ++r) {
        s += scratch[(size_t)r * 2 * A + c];
        scratch[(size_t)r * 2 * A + c] = 0.f;                // handed back clean for the next call
    }
    atomicAdd((c < A ? gw2 : gb1) + (c < A ? c : c - A), s);
}

// ------------------------------------------------------------------ dot predictor + cross entropy(label 0)
constexpr int kMaxCand = 64;

__global__ __launch_bounds__(256) void dot_ce_fwd_kernel(const float* __restrict__ user, int ldu,
                                                         const float* __restrict__ items, int ldi, int B, int C, int D,
                                                         float* __restrict__ scores, float* loss) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float mine = -INFINITY;
    for (int c = 0; c < C; ++c) {
        const float s = dot_row(user + (size_t)b * ldu, items + (size_t)(b * C + c) * ldi, D, lane);
        if (lane == c) mine = s;
    }
    if (lane < C) scores[b * C + lane] = mine;
    if (loss != nullptr) {
        float mx = mine;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float ex = lane < C ? expf(mine - mx) : 0.f;
        const float se = wave_sum(ex);
        const float s0 = __shfl(mine, 0, 64);
        if (lane == 0) atomicAdd(loss, (logf(se) + mx - s0) / (float)B);
    }
}

__global__ __launch_bounds__(256) void dot_ce_bwd_kernel(const float* __restrict__ user, int ldu,
                                                         const float* __restrict__ items, int ldi,
                                                         const float* __restrict__ scores, int B, int C, int D, float gscale,
                                                         const float* __restrict__ gscale_dev,
                                                         float* __restrict__ guser, int ldgu, float* __restrict__ gitems, int ldgi) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    if (gscale_dev != nullptr) gscale *= *gscale_dev;          // the upstream gradient of the loss, read on the device (no host sync)
    const float mine = lane < C ? scores[b * C + lane] : -INFINITY;
    float mx = mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float ex = lane < C ? expf(mine - mx) : 0.f;
    const float se = wave_sum(ex);
    const float g_mine = (ex / se - (lane == 0 ? 1.f : 0.f)) * gscale;
    f32x4 gu[kMaxChunks];
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) gu[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
        const float g = __shfl(g_mine, c, 64);
#pragma unroll
        for (int j = 0; j < kMaxChunks; ++j) {
            const int d = 4 * lane + 256 * j;
            if (d < D) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(user + (size_t)b * ldu + d);
                const f32x4 it = *reinterpret_cast<const f32x4*>(items + (size_t)(b * C + c) * ldi + d);
                gu[j] += g * it;
                *reinterpret_cast<f32x4*>(gitems + (size_t)(b * C + c) * ldgi + d) = g * u;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
        const int d = 4 * lane + 256 * j;
        if (d < D) *reinterpret_cast<f32x4*>(guser + (size_t)b * ldgu + d) = gu[j];
    }
}

// row-wise dot product of two [n,D] matrices and its backward (DotPredictor.predict on pre-expanded pairs)
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ u, int ldu, const float* __restrict__ it, int ldi,
                                                         int n, int D, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float s = dot_row(u + (size_t)r * ldu, it + (size_t)r * ldi, D, lane);
    if (lane == 0) out[r] = s;
}
__global__ void rowdot_bwd_kernel(const float* __restrict__ u, int ldu, const float* __restrict__ it, int ldi,
                                  const float* __restrict__ g, int n, int D, float* __restrict__ gu, int ldgu,
                                  float* __restrict__ gi, int ldgi) {
    const long long total = (long long)n * D;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / D), c = (int)(e - (long long)r * D);
        const float gr = g[r];
        gu[(size_t)r * ldgu + c] = gr * it[(size_t)r * ldi + c];
        gi[(size_t)r * ldgi + c] = gr * u[(size_t)r * ldu + c];
    }
}
// g = ref > 0 ? g * scale : 0   (backward of ReLU -> mask -> dropout when y = ref is the stored output)
__global__ void relu_bwd_kernel(float* __restrict__ g, int ldg, const float* __restrict__ ref, int ldr, int rows, int width, float scale) {
    const long long total = (long long)rows * width;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / width), c = (int)(e - (long long)r * width);
        const float v = g[(size_t)r * ldg + c];
        g[(size_t)r * ldg + c] = ref[(size_t)r * ldr + c] > 0.f ? v * scale : 0.f;
    }
}

// ------------------------------------------------------------------ fused user tower (training)
// One workgroup per impression does, back to back and without leaving the CU, what used to be five dependent
// launches on the step's critical path: additive pool over the clicked items (AdaOperator), dot scores against the
// C candidates, cross-entropy(label 0) and its gradient, and the backward of the pool (dx direct part, dpre in
// place of t, parameter-gradient partials).  The tanh GEMM before it and the dpre.W1 / dpre^T.x GEMMs after it
// stay on the MFMA core.
// 16 waves per user: the row loops below are chains of dependent global loads (one history row per step and wave), so
// 16 waves make them <= 4 steps long for S = 50 instead of 13 with 4 waves (28 -> ~13 us on the critical path of the step)
constexpr int kUT_NW = 16;
__global__ __launch_bounds__(kUT_NW * 64) void user_tower_train_kernel(
    float* __restrict__ t, int ldt, const float* __restrict__ items, int ldi, const float* __restrict__ w2,
    const int* __restrict__ hist_off, int B, int C, int D, int A, float gscale,
    float* __restrict__ user, float* __restrict__ scores, float* loss, float* __restrict__ d_items, int lddi,
    float* gw2, float* gb1) {
    __shared__ float red[kMaxChunks * 256];             // partial sums of the 16 waves meet here through ds_add_f32
    __shared__ float uvec[kMaxChunks * 256], duvec[kMaxChunks * 256];
    __shared__ float wl[kMaxSegRows], dwl[kMaxSegRows], sc[kMaxCand], red_s[kUT_NW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: row indices and addresses stay scalar)
    const int b = blockIdx.x;
    const int BC = B * C;
    const int beg = hist_off[b], len = hist_off[b + 1] - beg;
    const float* xh = items + (size_t)(BC + beg) * ldi;          // this user's clicked-item vectors
    float* th = t + (size_t)beg * ldt;
    // ---- A: additive pool forward (model/common/attention.py:31-38)
    f32x4 acc[kMaxChunks];
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s = 0.f;
    for (int l = wave; l < len; l += kUT_NW) {
        const float e = expf(dot_row(th + (size_t)l * ldt, w2, A, lane));
        s += e;
#pragma unroll
        for (int j = 0; j < kMaxChunks; ++j) {
            const int c = 4 * lane + 256 * j;
            if (c < D) acc[j] += e * *reinterpret_cast<const f32x4*>(xh + (size_t)l * ldi + c);
        }
        if (lane == 0) wl[l] = e;
    }
    for (int c = threadIdx.x; c < kMaxChunks * 256; c += kUT_NW * 64) red[c] = 0.f;
    if (lane == 0) red_s[wave] = s;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
        const int c = 4 * lane + 256 * j;
        if (c < D) {
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(&red[c + i], acc[j][i]);
        }
    }
    __syncthreads();
    float ssum = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < kUT_NW; ++w_) ssum += red_s[w_];
    const float inv = 1.f / (ssum + kEps);
    for (int c = threadIdx.x; c < D; c += kUT_NW * 64) {
        const float u = red[c] * inv;
        uvec[c] = u;
        user[(size_t)b * D + c] = u;
    }
    for (int l = threadIdx.x; l < len; l += kUT_NW * 64) wl[l] *= inv;
    __syncthreads();
    // ---- B: dot scores + cross entropy with label 0 (dot_predictor.py:10, legommender.py:254,263)
    for (int c = wave; c < C; c += kUT_NW) {
        const float v = dot_row(uvec, items + (size_t)(b * C + c) * ldi, D, lane);
        if (lane == 0) { sc[c] = v; scores[b * C + c] = v; }
    }
    __syncthreads();
    float mx = -INFINITY, se = 0.f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, sc[c]);
    for (int c = 0; c < C; ++c) se += expf(sc[c] - mx);
    if (threadIdx.x == 0 && loss != nullptr) atomicAdd(loss, (logf(se) + mx - sc[0]) / (float)B);
    // ---- C: d scores -> d user, d candidates
    for (int d = threadIdx.x; d < D; d += kUT_NW * 64) {
        float du = 0.f;
        const float u = uvec[d];
        for (int c = 0; c < C; ++c) {
            const float g = (expf(sc[c] - mx) / se - (c == 0 ? 1.f : 0.f)) * gscale;
            du += g * items[(size_t)(b * C + c) * ldi + d];
            d_items[(size_t)(b * C + c) * lddi + d] = g * u;
        }
        duvec[d] = du;
    }
    __syncthreads();
    // ---- D: additive pool backward
    float sdw = 0.f;
    for (int l = wave; l < len; l += kUT_NW) {
        const float dw = dot_row(duvec, xh + (size_t)l * ldi, D, lane);
        sdw += wl[l] * dw;
        if (lane == 0) dwl[l] = dw;
    }
    if (lane == 0) red_s[wave] = sdw;
    __syncthreads();
    sdw = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < kUT_NW; ++w_) sdw += red_s[w_];
    f32x4 aw2[kMaxChunks], ab1[kMaxChunks];
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) { aw2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; ab1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int l = wave; l < len; l += kUT_NW) {
        const float w = wl[l];
        const float da = w * (dwl[l] - sdw);
#pragma unroll
        for (int j = 0; j < kMaxChunks; ++j) {
            const int c = 4 * lane + 256 * j;
            if (c < D)
                *reinterpret_cast<f32x4*>(d_items + (size_t)(BC + beg + l) * lddi + c) = w * *reinterpret_cast<const f32x4*>(duvec + c);
            if (c < A) {
                f32x4* tp = reinterpret_cast<f32x4*>(th + (size_t)l * ldt + c);
                const f32x4 tv = *tp;
                const f32x4 dpre = da * *reinterpret_cast<const f32x4*>(w2 + c) * (1.f - tv * tv);
                aw2[j] += da * tv;
                ab1[j] += dpre;
                *tp = dpre;
            }
        }
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {               // 0: d w2, 1: d b1
        for (int c = threadIdx.x; c < kMaxChunks * 256; c += kUT_NW * 64) red[c] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kMaxChunks; ++j) {
            const int c = 4 * lane + 256 * j;
            if (c < A) {
#pragma unroll
                for (int i = 0; i < 4; ++i) atomicAdd(&red[c + i], pass == 0 ? aw2[j][i] : ab1[j][i]);
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < A; c += kUT_NW * 64) atomicAdd((pass == 0 ? gw2 : gb1) + c, red[c]);
        __syncthreads();
    }
}

// Fast path of the fused user tower for D, A <= 256 and S <= 64 (the path's configurations): every history row has
// ONE owner wave (<= 4 rows per wave) that loads its item vector and tanh row once, all loads in flight together, and
// keeps them in registers through the forward pool, the loss and the backward pool -- the generic kernel above pays
// a dependent global load per row in each of its five passes (~30 us of latency for 64 workgroups).
__global__ __launch_bounds__(kUT_NW * 64) void user_tower_train_fast_kernel(
    float* __restrict__ t, int ldt, const float* __restrict__ items, int ldi, const float* __restrict__ w2,
    const int* __restrict__ hist_off, int B, int C, int D, int A, float gscale,
    float* __restrict__ user, float* __restrict__ scores, float* loss, float* __restrict__ d_items, int lddi,
    float* gw2, float* gb1) {
    constexpr int RW = 4;
    __shared__ float red[256], red2[256];
    __shared__ float uvec[256], duvec[256];
    __shared__ float sc[kMaxCand], red_s[kUT_NW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: row indices and addresses stay scalar)
    const int b = blockIdx.x;
    const int BC = B * C;
    const int beg = hist_off[b], len = hist_off[b + 1] - beg;
    const float* xh = items + (size_t)(BC + beg) * ldi;
    float* th = t + (size_t)beg * ldt;
    const int c4 = 4 * lane;
    const bool inD = c4 < D, inA = c4 < A;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 w2v = inA ? *reinterpret_cast<const f32x4*>(w2 + c4) : zero4;
    // candidate rows of this user (for the scores and d user), one per wave
    f32x4 cv = zero4;
    if (wave < C && inD) cv = *reinterpret_cast<const f32x4*>(items + (size_t)(b * C + wave) * ldi + c4);
    f32x4 xr[RW], tr[RW];
    bool own[RW];
#pragma unroll
    for (int u = 0; u < RW; ++u) {
        const int l = wave + kUT_NW * u;
        own[u] = l < len;
        xr[u] = (own[u] && inD) ? *reinterpret_cast<const f32x4*>(xh + (size_t)l * ldi + c4) : zero4;
        tr[u] = (own[u] && inA) ? *reinterpret_cast<const f32x4*>(th + (size_t)l * ldt + c4) : zero4;
    }
    for (int c = threadIdx.x; c < 256; c += kUT_NW * 64) { red[c] = 0.f; red2[c] = 0.f; }
    // ---- A: additive pool forward
    float e[RW], s = 0.f;
    f32x4 acc = zero4;
#pragma unroll
    for (int u = 0; u < RW; ++u) {
        const float a = wave_sum((tr[u][0] * w2v[0] + tr[u][1] * w2v[1]) + (tr[u][2] * w2v[2] + tr[u][3] * w2v[3]));
        e[u] = own[u] ? expf(a) : 0.f;
        s += e[u];
        acc += e[u] * xr[u];
    }
    if (lane == 0) red_s[wave] = s;
    __syncthreads();
    if (inD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(&red[c4 + i], acc[i]);
    }
    float ssum = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < kUT_NW; ++w_) ssum += red_s[w_];
    const float inv = 1.f / (ssum + kEps);
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += kUT_NW * 64) {
        const float uu = red[c] * inv;
        uvec[c] = uu;
        user[(size_t)b * D + c] = uu;
    }
    __syncthreads();
    // ---- B: dot scores + cross entropy with label 0
    const f32x4 uv = inD ? *reinterpret_cast<const f32x4*>(uvec + c4) : zero4;
    if (wave < C) {
        const float v = wave_sum((uv[0] * cv[0] + uv[1] * cv[1]) + (uv[2] * cv[2] + uv[3] * cv[3]));
        if (lane == 0) { sc[wave] = v; scores[b * C + wave] = v; }
    }
    __syncthreads();
    float mx = -INFINITY, se = 0.f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, sc[c]);
    for (int c = 0; c < C; ++c) se += expf(sc[c] - mx);
    if (threadIdx.x == 0 && loss != nullptr) atomicAdd(loss, (logf(se) + mx - sc[0]) / (float)B);
    // ---- C: d scores -> d candidates (this wave's), d user (sum over the candidate waves, through LDS)
    if (wave < C) {
        const float g = (expf(sc[wave] - mx) / se - (wave == 0 ? 1.f : 0.f)) * gscale;
        if (inD) {
            *reinterpret_cast<f32x4*>(d_items + (size_t)(b * C + wave) * lddi + c4) = g * uv;
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(&red2[c4 + i], g * cv[i]);
        }
    }
    __syncthreads();
    const f32x4 duv = inD ? *reinterpret_cast<const f32x4*>(red2 + c4) : zero4;
    // ---- D: additive pool backward on the cached rows
    float dw[RW], sdw = 0.f;
#pragma unroll
    for (int u = 0; u < RW; ++u) {
        dw[u] = wave_sum((duv[0] * xr[u][0] + duv[1] * xr[u][1]) + (duv[2] * xr[u][2] + duv[3] * xr[u][3]));
        sdw += e[u] * inv * dw[u];
    }
    if (lane == 0) red_s[wave] = sdw;
    for (int c = threadIdx.x; c < 256; c += kUT_NW * 64) { red[c] = 0.f; uvec[c] = 0.f; }      // reused: d w2 / d b1 partials
    __syncthreads();
    sdw = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < kUT_NW; ++w_) sdw += red_s[w_];
    f32x4 aw2 = zero4, ab1 = zero4;
#pragma unroll
    for (int u = 0; u < RW; ++u) {
        if (!own[u]) continue;                           // wave-uniform
        const int l = wave + kUT_NW * u;
        const float w = e[u] * inv;
        const float da = w * (dw[u] - sdw);
        if (inD) *reinterpret_cast<f32x4*>(d_items + (size_t)(BC + beg + l) * lddi + c4) = w * duv;
        if (inA) {
            const f32x4 dpre = da * w2v * (1.f - tr[u] * tr[u]);
            aw2 += da * tr[u];
            ab1 += dpre;
            *reinterpret_cast<f32x4*>(th + (size_t)l * ldt + c4) = dpre;
        }
    }
    if (inA) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { atomicAdd(&red[c4 + i], aw2[i]); atomicAdd(&uvec[c4 + i], ab1[i]); }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < A; c += kUT_NW * 64) { atomicAdd(gw2 + c, red[c]); atomicAdd(gb1 + c, uvec[c]); }
}

// ------------------------------------------------------------------ Adam
struct AdamCoef { float step_size, beta1, beta2, eps, inv_sqrt_bc2, gscale; int zero_grad; };

__device__ __forceinline__ void adam4(const AdamCoef& a, f32x4& p, f32x4& g, f32x4& m, f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float gr = g[i] * a.gscale;
        const float mi = a.beta1 * m[i] + (1.f - a.beta1) * gr;
        const float vi = a.beta2 * v[i] + (1.f - a.beta2) * gr * gr;
        m[i] = mi; v[i] = vi;
        p[i] -= a.step_size * (mi / (sqrtf(vi) * a.inv_sqrt_bc2 + a.eps));
        if (a.zero_grad) g[i] = 0.f;          // the next step accumulates into a clean buffer: no separate fill launch
    }
}

// 16 B per lane and array: the dense update of an embedding table is pure HBM streaming (4 reads + 4 writes per element)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n4, AdamCoef a) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        adam4(a, pp, gg, mm, vv);
        reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
        if (a.zero_grad) reinterpret_cast<f32x4*>(g)[i] = gg;
    }
}

// elements [first, n) one by one: the tail of a buffer whose length is not a multiple of 4, or an unaligned buffer
__global__ void adam_scalar_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                   long long first, long long n, AdamCoef a) {
    for (long long i = first + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gr = g[i] * a.gscale;
        if (a.zero_grad) g[i] = 0.f;
        const float mi = a.beta1 * m[i] + (1.f - a.beta1) * gr;
        const float vi = a.beta2 * v[i] + (1.f - a.beta2) * gr * gr;
        m[i] = mi; v[i] = vi;
        p[i] -= a.step_size * (mi / (sqrtf(vi) * a.inv_sqrt_bc2 + a.eps));
    }
}

// The same update over the rows of a [rows, width] table, SKIPPING rows whose "ever touched" bit is clear.  Exactly the dense
// result: a row that never received a gradient has g = m = v = 0, for which Adam's update is 0 and m, v stay 0 -- such a row
// is left alone, bit for bit, until its first gradient arrives; from then on it is updated every step like the dense rule
// demands (its momentum keeps moving it).  MIND-small titles use a small part of the 400 k GloVe vocabulary, so most of the
// 410 MB x 4 arrays is never read.  One wave per row.
__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int rows, int width,
                                                        const uint8_t* __restrict__ touched, AdamCoef a) {
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
        if (touched[r] == 0) continue;
        const size_t base = (size_t)r * width;
        for (int c = lane * 4; c < width; c += 256) {
            f32x4 pp = *reinterpret_cast<f32x4*>(p + base + c), gg = *reinterpret_cast<f32x4*>(g + base + c);
            f32x4 mm = *reinterpret_cast<f32x4*>(m + base + c), vv = *reinterpret_cast<f32x4*>(v + base + c);
            adam4(a, pp, gg, mm, vv);
            *reinterpret_cast<f32x4*>(p + base + c) = pp; *reinterpret_cast<f32x4*>(m + base + c) = mm;
            *reinterpret_cast<f32x4*>(v + base + c) = vv;
            if (a.zero_grad) *reinterpret_cast<f32x4*>(g + base + c) = gg;
        }
    }
}

__global__ void mark_rows_kernel(const int* __restrict__ idx, int n_cap, const int* __restrict__ n_dyn, int rows, uint8_t* touched) {
    const int n = n_dyn != nullptr ? min(n_cap, *n_dyn) : n_cap;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int i = idx[e];
        if (i >= 0 && i < rows) touched[i] = 1;             // racing writers store the same value
    }
}

// ------------------------------------------------------------------ negative sampling / history fetch
__global__ void sample_negatives_kernel(const int* __restrict__ row_user, const int* __restrict__ row_item,
                                        const int* __restrict__ neg_list, const int* __restrict__ neg_len, int neg_cap,
                                        int B, int K, int n_items, uint32_t seed_lo, uint32_t seed_hi, uint32_t step,
                                        uint32_t row_base, uint32_t row_stride, const int* __restrict__ row_pos, int* cand) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // the Philox stream of a row is keyed on its position in the GLOBAL batch of the step (rank r of W: r + b * W), so that
    // W ranks x B rows draw exactly what one device with batch W * B draws for the same rows
    // (row_pos: the position read from a table instead -- ranks that deal a global batch's rows by cost, DeviceData.balance)
    const uint32_t gb = row_pos != nullptr ? (uint32_t)row_pos[b] : row_base + (uint32_t)b * row_stride;
    const int u = row_user[b];
    const int L = min(max(neg_len[u], 0), neg_cap);
    const int n_true = min(K, L);
    int* out = cand + (size_t)b * (K + 1);
    out[0] = row_item[b];
    uint32_t ctr = 0;
    Philox4 r = philox4x32_10(gb, step, ctr++, 0x6e656773u, seed_lo, seed_hi);
    int have = 0;
    uint32_t pool[4] = {r.x, r.y, r.z, r.w};
    auto next = [&]() -> uint32_t {
        if (have == 4) {
            r = philox4x32_10(gb, step, ctr++, 0x6e656773u, seed_lo, seed_hi);
            pool[0] = r.x; pool[1] = r.y; pool[2] = r.z; pool[3] = r.w; have = 0;
        }
        return pool[have++];
    };
    int chosen[kMaxCand];
    for (int k = 0; k < n_true; ++k) {            // distinct POSITIONS of the true-negative list (random.sample)
        int pos;
        bool dup;
        do {
            pos = (int)(((unsigned long long)next() * (unsigned long long)L) >> 32);
            dup = false;
            for (int q = 0; q < k; ++q) dup |= (chosen[q] == pos);
        } while (dup);
        chosen[k] = pos;
        out[1 + k] = neg_list[(size_t)u * neg_cap + pos];
    }
    for (int k = n_true; k < K; ++k)              // random.randint(0, item_size - 1) fill
        out[1 + k] = (int)(((unsigned long long)next() * (unsigned long long)n_items) >> 32);
}

__global__ void gather_history_kernel(const int* __restrict__ row_user, const int* __restrict__ user_hist,
                                      const int* __restrict__ user_hist_len, int B, int S, int* hist, int* hist_len) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * S) return;
    const int b = gid / S, s = gid - b * S;
    const int u = row_user[b];
    hist[gid] = user_hist[(size_t)u * S + s];
    if (s == 0) hist_len[b] = user_hist_len[u];
}

}  // namespace lego

using namespace lego;
#define ST ((hipStream_t)stream)

extern "C" int lego_plan_batch(const int32_t* cand, const int32_t* hist, const int32_t* hist_len, int B, int C, int S,
                               const int32_t* title_tok, const int32_t* title_len, int T,
                               int32_t* counters, int32_t* inst_item, int32_t* seg_off, int32_t* hist_off,
                               int32_t* rowinfo, int32_t* row_tok, void* stream) {
    LEGO_REQUIRE(B > 0 && C >= 0 && S >= 0 && C + S > 0 && T > 0, "lego_plan_batch: bad sizes B=%d C=%d S=%d T=%d", B, C, S, T);
    LEGO_REQUIRE((long long)B * (C + S) < (1 << 23), "lego_plan_batch: too many item instances");
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, ST, cand, hist, hist_len, B, C, S, title_len,
                       counters, inst_item, seg_off, hist_off);
    const int NI_cap = B * (C + S);
    hipLaunchKernelGGL(plan_rows_kernel, dim3((NI_cap * 32 + 255) / 256), dim3(256), 0, ST, counters, inst_item, seg_off,
                       title_tok, title_len, T, NI_cap, rowinfo, row_tok);
    return check_launch("lego_plan_batch");
}

extern "C" int lego_plan_pairs(const int32_t* seg_off, int n_cap, const int32_t* n_dyn, int32_t* pair_info,
                               int32_t* n_pairs_out, void* stream) {
    LEGO_REQUIRE(n_cap > 0, "lego_plan_pairs: n_cap=%d", n_cap);
    hipLaunchKernelGGL(plan_pairs_kernel, dim3(1), dim3(1024), 0, ST, seg_off, n_cap, n_dyn, pair_info, n_pairs_out);
    return check_launch("lego_plan_pairs");
}

extern "C" int lego_conv3_wino_pack(const float* w, float* u, float* ut, int Dout, int Din, void* stream) {
    const int n = Dout * Din;
    hipLaunchKernelGGL(conv3_wino_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, w, u, ut, Dout, Din);
    return check_launch("lego_conv3_wino_pack");
}
extern "C" int lego_conv3_wino_unpack_add(float* du, int n_slabs, float* dw, int Dout, int Din, void* stream) {
    LEGO_REQUIRE(n_slabs >= 1, "lego_conv3_wino_unpack_add: n_slabs=%d", n_slabs);
    const int n = Dout * Din;
    LEGO_REQUIRE((n & 3) == 0, "lego_conv3_wino_unpack_add: Dout * Din = %d must be a multiple of 4", n);
    LEGO_REQUIRE((reinterpret_cast<uintptr_t>(dw) & 15) == 0 && (reinterpret_cast<uintptr_t>(du) & 15) == 0, "lego_conv3_wino_unpack_add: du / dw must be 16-byte aligned");
    hipLaunchKernelGGL(conv3_wino_unpack_add_kernel, dim3((n / 4 + 63) / 64), dim3(256), 0, ST, du, n_slabs, dw, Dout, Din);
    return check_launch("lego_conv3_wino_unpack_add");
}

extern "C" int lego_plan_dense(const int32_t* mask, int n, int L, int32_t* counters, int32_t* seg_off, int32_t* rowinfo,
                               void* stream) {
    LEGO_REQUIRE(n > 0 && L > 0, "lego_plan_dense: bad sizes n=%d L=%d", n, L);
    LEGO_REQUIRE(L + 1 <= kMaxSegRows, "lego_plan_dense: L=%d exceeds the %d-row segment limit", L, kMaxSegRows - 1);
    hipLaunchKernelGGL(plan_dense_kernel, dim3((n * L + n + 256) / 256), dim3(256), 0, ST, mask, n, L, counters, seg_off, rowinfo);
    return check_launch("lego_plan_dense");
}

extern "C" int lego_gather_rows(const float* table, int ld_table, int width, const int32_t* idx, int rows_cap,
                                const int32_t* rows_dyn, float* out, int ld_out, int accumulate, void* stream) {
    LEGO_REQUIRE((width & 3) == 0 && (ld_table & 3) == 0 && (ld_out & 3) == 0, "lego_gather_rows: width/ld must be multiples of 4");
    if (rows_cap <= 0) return 0;
    if (width >= 64 * 4 && width <= 128 * 4) {                       // rows of 1-2 KB: one wave per row, 2 rows in flight per wave, 16 waves per CU
        constexpr int U = 2;                                         // (2 / 4 / 8 rows in flight: 48.3 / 49.9 / 52.5 us on 105.6 k rows, profiles/r06_gather.txt)
        const int want_blocks = (rows_cap + 4 * U - 1) / (4 * U);    // 4 waves per block x U rows in flight
        const int blocks = want_blocks < 1024 ? (want_blocks > 0 ? want_blocks : 1) : 1024;
        hipLaunchKernelGGL((gather_rows_wave_kernel<U>), dim3(blocks), dim3(256), 0, ST, table, ld_table, width / 4, idx, rows_cap, rows_dyn, out, ld_out, accumulate);
        return check_launch("lego_gather_rows");
    }
    const long long total = (long long)rows_cap * (width / 4);
    const long long want = (total + 4 * 256 - 1) / (4 * 256);
    const int blocks = (int)(want < 8192 ? (want > 0 ? want : 1) : 8192);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, ST, table, ld_table, width / 4, idx, rows_cap, rows_dyn, out, ld_out, accumulate);
    return check_launch("lego_gather_rows");
}


extern "C" int lego_unique_tokens(const int32_t* row_tok, int R_cap, const int32_t* R_dyn, int V, uint32_t* stamp, uint32_t epoch,
                                  int32_t* rank, int32_t* bsum, int32_t* uniq, int32_t* inv, int32_t* cnt, int32_t* start,
                                  int32_t* perm, int32_t* sort_keys, int32_t* n_uniq, void* stream) {
    LEGO_REQUIRE(V > 0 && epoch != 0u, "lego_unique_tokens: V=%d epoch=%u (the stamp table is zero-initialised: epoch 0 is reserved)", V, epoch);
    if (R_cap <= 0) return 0;
    const int nblk = (V + kUqBlock - 1) / kUqBlock;
    const int rb = (R_cap + 255) / 256 < 1024 ? (R_cap + 255) / 256 : 1024;
    hipLaunchKernelGGL(uq_mark_kernel, dim3(rb), dim3(256), 0, ST, row_tok, R_cap, R_dyn, stamp, epoch, V);
    hipLaunchKernelGGL(uq_count_kernel, dim3(nblk), dim3(kUqBlock), 0, ST, stamp, V, epoch, bsum);
    hipLaunchKernelGGL(uq_scan_kernel, dim3(1), dim3(kUqBlock), 0, ST, bsum, nblk, (const int*)nullptr, n_uniq, (int*)nullptr);
    hipLaunchKernelGGL(uq_assign_kernel, dim3(nblk), dim3(kUqBlock), 0, ST, stamp, V, epoch, bsum, uniq, rank, cnt);
    // perm given: rows grouped by token through a counting sort (histogram + scan + atomic cursor: fine for tests and flat id
    // distributions; a Zipf head serialises its atomics on one address -- 2 x 64 us on the bench world -- so the engine passes
    // perm = NULL, sort_keys instead, and groups the rows with lego_sort_rows)
    hipLaunchKernelGGL(uq_inverse_kernel, dim3(rb), dim3(256), 0, ST, row_tok, R_cap, R_dyn, rank, V, inv, perm != nullptr ? cnt : (int*)nullptr,
                       sort_keys);
    if (perm != nullptr) {
        const int u_cap = R_cap < V ? R_cap : V;
        hipLaunchKernelGGL(uq_scan_kernel, dim3(1), dim3(kUqBlock), 0, ST, cnt, u_cap, n_uniq, (int*)nullptr, start);
        hipLaunchKernelGGL(uq_fill_kernel, dim3(rb), dim3(256), 0, ST, inv, R_cap, R_dyn, cnt, perm);
    }
    return check_launch("lego_unique_tokens");
}

extern "C" int lego_expand_rows(const float* src, int ld_src, const int32_t* inv, int rows_cap, const int32_t* rows_dyn, int width,
                                const lego_dropout* drop, const int32_t* rowinfo, const float* add_a, int ld_a, const int32_t* idx_a,
                                const float* add_b, int ld_b, const int32_t* idx_b, float* out, int ld_out, void* stream) {
    LEGO_REQUIRE((width & 3) == 0 && (ld_src & 3) == 0 && (ld_out & 3) == 0, "lego_expand_rows: width=%d must be a multiple of 4", width);
    LEGO_REQUIRE((add_a == nullptr) == (idx_a == nullptr) && (add_b == nullptr) == (idx_b == nullptr) && (add_a == nullptr || (ld_a & 3) == 0) &&
                 (add_b == nullptr || (ld_b & 3) == 0), "lego_expand_rows: an added table needs its index list and a row stride that is a multiple of 4");
    if (rows_cap <= 0) return 0;
    const int want = (rows_cap + 15) / 16;                 // four waves per block, four consecutive rows per wave and iteration
    hipLaunchKernelGGL(expand_rows_kernel, dim3(want < 2048 ? want : 2048, (width + 511) / 512), dim3(256), 0, ST, src, ld_src, inv, rows_cap, rows_dyn, width,
                       make_dropout(drop), rowinfo, ExpandAdd{add_a, idx_a, ld_a, add_b, idx_b, ld_b}, out, ld_out);
    return check_launch("lego_expand_rows");
}

extern "C" int lego_zero_rows(float* out, int ld_out, int width, int rows_cap, const int32_t* rows_dyn, void* stream) {
    LEGO_REQUIRE((width & 3) == 0 && (ld_out & 3) == 0, "lego_zero_rows: width=%d ld=%d must be multiples of 4", width, ld_out);
    if (rows_cap <= 0) return 0;
    const long long tot = (long long)rows_cap * (width / 4);
    hipLaunchKernelGGL(zero_rows_kernel, dim3((int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048)), dim3(256), 0, ST, out, ld_out, width, rows_cap, rows_dyn);
    return check_launch("lego_zero_rows");
}

extern "C" int lego_segment_sum_rows(const float* g, int ld_g, int width, const int32_t* perm, const int32_t* inv, int R_cap,
                                     const int32_t* sorted_keys, const int32_t* R_dyn, float* out, int ld_out, int U_cap,
                                     const int32_t* U_dyn, int zero_first, const lego_dropout* drop, const int32_t* rowinfo, void* stream) {
    LEGO_REQUIRE((width & 3) == 0 && (ld_g & 3) == 0 && (ld_out & 3) == 0, "lego_segment_sum_rows: width=%d must be a multiple of 4", width);
    LEGO_REQUIRE(drop == nullptr || drop->p <= 0.f || drop->mask != nullptr, "lego_segment_sum_rows: a dropout site needs its precomputed keep bits (lego_dropout_mask)");
    if (R_cap <= 0) return 0;
    const Dropout dr = make_dropout(drop);
    const bool dropping = dr.p > 0.f;
    if (zero_first) {
        const long long tot = (long long)U_cap * (width / 4);
        hipLaunchKernelGGL(zero_rows_kernel, dim3((int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048)), dim3(256), 0, ST, out, ld_out, width, U_cap, U_dyn);
    }
    const dim3 gx((R_cap + 4 * kSegRows - 1) / (4 * kSegRows), (width + 63) / 64);
    const uint8_t* km = dropping ? dr.mask : (const uint8_t*)nullptr;
    const float ks = dropping ? 1.f / (1.f - dr.p) : 1.f;
    hipLaunchKernelGGL(segment_sum_rows_kernel<1>, gx, dim3(256), 0, ST, g, ld_g, width, perm, inv, sorted_keys, R_cap, R_dyn, out, ld_out, km, ks, rowinfo);
    return check_launch("lego_segment_sum_rows");
}

extern "C" int lego_nrms_decode_rows(const int32_t* row_tok, int R_cap, const int32_t* R_dyn, int32_t* idx_tok,
                                     int32_t* idx_special, int32_t* idx_cat, int32_t* tokinfo, void* stream) {
    if (R_cap <= 0) return 0;
    hipLaunchKernelGGL(nrms_decode_rows_kernel, dim3((R_cap + 255) / 256), dim3(256), 0, ST, row_tok, R_cap, R_dyn, idx_tok,
                       idx_special, idx_cat, tokinfo);
    return check_launch("lego_nrms_decode_rows");
}

extern "C" int lego_nrms_key_rows(const int32_t* row_tok, int R_cap, const int32_t* R_dyn, int V, int32_t* row_key, void* stream) {
    if (R_cap <= 0) return 0;
    hipLaunchKernelGGL(nrms_key_rows_kernel, dim3((R_cap + 255) / 256), dim3(256), 0, ST, row_tok, R_cap, R_dyn, V, row_key);
    return check_launch("lego_nrms_key_rows");
}

extern "C" int lego_nrms_decode_keys(const int32_t* uniq, int U_cap, const int32_t* U_dyn, int V, int32_t* idx_tok, int32_t* idx_special,
                                     int32_t* idx_cat, int32_t* keyinfo, void* stream) {
    if (U_cap <= 0) return 0;
    hipLaunchKernelGGL(nrms_decode_keys_kernel, dim3((U_cap + 255) / 256), dim3(256), 0, ST, uniq, U_cap, U_dyn, V, idx_tok, idx_special, idx_cat, keyinfo);
    return check_launch("lego_nrms_decode_keys");
}

extern "C" int lego_mask_dropout_rows(float* x, int ld, int R_cap, const int32_t* R_dyn, int width, const int32_t* rowinfo,
                                      const lego_dropout* drop, float* colsum, void* stream) {
    if (R_cap <= 0) return 0;
    Dropout d = make_dropout(drop);
    const long long total = (long long)((R_cap + 3) / 4) * width;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(mask_dropout_rows_kernel, dim3(blocks), dim3(256), 0, ST, x, ld, R_cap, R_dyn, width, rowinfo, d);
    if (check_launch("lego_mask_dropout_rows") != 0) return 1;
    // the column sums are a second launch on purpose: folded into the mask pass they need one fp32 atomic per column and workgroup
    // on the same 1 KB of sums, and enough workgroups to feed HBM queue up on them (46-60 us against 20 + 18 us for the pair)
    return colsum != nullptr ? lego_colsum(x, ld, R_cap, R_dyn, nullptr, width, colsum, stream) : 0;
}

extern "C" int lego_dropout_mask(const lego_dropout* drop, int rows_cap, const int32_t* rows_dyn, int cols, uint8_t* mask,
                                 void* stream) {
    LEGO_REQUIRE(drop != nullptr && drop->p > 0.f && drop->p < 1.f, "lego_dropout_mask: needs 0 < p < 1");
    if (rows_cap <= 0) return 0;
    const long long total = (long long)((rows_cap + 7) / 8) * cols;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, ST, make_dropout(drop), rows_cap, rows_dyn, cols, mask);
    return check_launch("lego_dropout_mask");
}

extern "C" int lego_scatter_add_rows(float* grad_table, int ld_table, int width, int table_rows, const int32_t* idx,
                                     int rows_cap, const int32_t* rows_dyn, const float* g, int ld_g, void* stream) {
    if (rows_cap <= 0) return 0;
    if (table_rows > 0 && table_rows <= kSmallTableRows) {
        LEGO_REQUIRE((width & 3) == 0 && (ld_g & 3) == 0, "lego_scatter_add_rows: width=%d and ld_g=%d must be multiples of 4", width, ld_g);
        const int chunks = (rows_cap + kSmallChunk - 1) / kSmallChunk;
        int iters = chunks / 512;                      // ~512 workgroups for long row ranges, one chunk each for short ones
        iters = iters < 1 ? 1 : (iters > 16 ? 16 : iters);
        hipLaunchKernelGGL(scatter_add_small_kernel, dim3((chunks + iters - 1) / iters, (width + 255) / 256), dim3(256), 0, ST,
                           grad_table, ld_table, width, table_rows, idx, rows_cap, rows_dyn, g, ld_g, iters);
        return check_launch("lego_scatter_add_rows");
    }
    const long long total = (long long)rows_cap * width;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(blocks), dim3(256), 0, ST, grad_table, ld_table, width, idx, rows_cap, rows_dyn, g, ld_g,
                       0, 0x7fffffff);
    return check_launch("lego_scatter_add_rows");
}

extern "C" int lego_scatter_add_rows_range(float* grad_table, int ld_table, int width, const int32_t* idx, int rows_cap,
                                           const int32_t* rows_dyn, const float* g, int ld_g, int row_lo, int row_hi, void* stream) {
    LEGO_REQUIRE(row_lo >= 0 && row_hi >= row_lo, "lego_scatter_add_rows_range: bad row range [%d, %d)", row_lo, row_hi);
    if (rows_cap <= 0 || row_hi == row_lo) return 0;
    const long long total = (long long)rows_cap * width;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(blocks), dim3(256), 0, ST, grad_table, ld_table, width, idx, rows_cap, rows_dyn, g, ld_g,
                       row_lo, row_hi);
    return check_launch("lego_scatter_add_rows_range");
}

extern "C" int lego_nrms_special_grads(const int32_t* seg_off, int n_cap, const int32_t* n_dyn, const int32_t* idx_cat,
                                       const float* g, int ld, int width, float* g_sep, float* g_cat, int ld_cat, int n_cat,
                                       void* stream) {
    LEGO_REQUIRE((width & 3) == 0 && (ld & 3) == 0 && n_cat <= kSmallTableRows, "lego_nrms_special_grads: width=%d ld=%d must be multiples of 4, n_cat=%d <= %d",
                 width, ld, n_cat, kSmallTableRows);
    if (n_cap <= 0) return 0;
    hipLaunchKernelGGL(nrms_special_grads_kernel, dim3((n_cap + kSpecItems - 1) / kSpecItems, (width + 255) / 256), dim3(256), 0, ST,
                       seg_off, n_cap, n_dyn, idx_cat, g, ld, width, g_sep, g_cat, ld_cat, n_cat);
    return check_launch("lego_nrms_special_grads");
}

extern "C" int lego_gather_i32(const int32_t* table, const int32_t* idx, int n_cap, const int32_t* n_dyn, int32_t* out, void* stream) {
    if (n_cap <= 0) return 0;
    hipLaunchKernelGGL(gather_i32_kernel, dim3((n_cap + 255) / 256), dim3(256), 0, ST, table, idx, n_cap, n_dyn, out);
    return check_launch("lego_gather_i32");
}

extern "C" int lego_segment_live(const int32_t* seg_off, int n_cap, const int32_t* n_dyn, int32_t* rowinfo, void* stream) {
    if (n_cap <= 0) return 0;
    hipLaunchKernelGGL(segment_live_kernel, dim3((n_cap + 255) / 256), dim3(256), 0, ST, seg_off, n_cap, n_dyn, rowinfo);
    return check_launch("lego_segment_live");
}

extern "C" int lego_colsum(const float* x, int ldx, int M_cap, const int32_t* M_dyn, const int32_t* row_off_dyn,
                           int N, float* out, void* stream) {
    if (M_cap <= 0) return 0;
    if ((N & 3) == 0 && (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        hipLaunchKernelGGL(colsum4_kernel, dim3((N + 255) / 256, (M_cap + 255) / 256), dim3(256), 0, ST, x, ldx, M_cap, M_dyn,
                           row_off_dyn, N, out);
        return check_launch("lego_colsum");
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, (M_cap + 255) / 256), dim3(256), 0, ST, x, ldx, M_cap, M_dyn, row_off_dyn, N, out);
    return check_launch("lego_colsum");
}

extern "C" int lego_conv3_pack(const float* w, float* wt, int Dout, int Din, void* stream) {
    const int n = 3 * Dout * Din;
    hipLaunchKernelGGL(conv3_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, w, wt, Dout, Din);
    return check_launch("lego_conv3_pack");
}
extern "C" int lego_conv3_unpack_add(float* dwt, float* dw, int Dout, int Din, void* stream) {
    const int n = 3 * Dout * Din;
    hipLaunchKernelGGL(conv3_unpack_add_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, dwt, dw, Dout, Din);
    return check_launch("lego_conv3_unpack_add");
}

extern "C" int lego_additive_pool_fwd(const float* t, int ldt, const float* x, int ldx, const float* w2,
                                      const int32_t* seg_off, const int32_t* rowinfo, const int32_t* extra_off_dyn,
                                      int n_cap, const int32_t* n_dyn, int D, int A, float* out, int ldo, float* wrow,
                                      void* stream) {
    LEGO_REQUIRE((D & 3) == 0 && (A & 3) == 0 && D <= 256 * kMaxChunks && A <= 256 * kMaxChunks,
                 "lego_additive_pool_fwd: D=%d A=%d must be multiples of 4 and <= %d", D, A, 256 * kMaxChunks);
    LEGO_REQUIRE((ldt & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0, "lego_additive_pool_fwd: strides must be multiples of 4");
    if (n_cap <= 0) return 0;
    if (D <= 256 && A <= 256)                        // (wider rows: the row-by-row kernels)
        hipLaunchKernelGGL(additive_pool_fwd_fast_kernel, dim3(n_cap), dim3(256), 0, ST, t, ldt, x, ldx, w2, seg_off, rowinfo,
                           extra_off_dyn, n_cap, n_dyn, D, A, out, ldo, wrow);
    else
        hipLaunchKernelGGL(additive_pool_fwd_kernel, dim3(n_cap), dim3(256), 0, ST, t, ldt, x, ldx, w2, seg_off, rowinfo,
                           extra_off_dyn, n_cap, n_dyn, D, A, out, ldo, wrow);
    return check_launch("lego_additive_pool_fwd");
}

extern "C" int lego_additive_pool_bwd(float* t_dpre, int ldt, const float* x, int ldx, const float* w2,
                                      const int32_t* seg_off, const int32_t* extra_off_dyn, int n_cap, const int32_t* n_dyn,
                                      int D, int A, const float* gout, int ldgo, const float* wrow,
                                      float* dx, int lddx, float* gw2, float* gb1, float* scratch, void* stream) {
    LEGO_REQUIRE((D & 3) == 0 && (A & 3) == 0 && D <= 256 * kMaxChunks && A <= 256 * kMaxChunks,
                 "lego_additive_pool_bwd: D=%d A=%d must be multiples of 4 and <= %d", D, A, 256 * kMaxChunks);
    if (n_cap <= 0) return 0;
    const int cap = 1024;
    int blocks = n_cap;
    if (blocks > cap) blocks = cap;
    if (D <= 256 && A <= 256 && (ldt & 3) == 0 && (ldx & 3) == 0 && (lddx & 3) == 0 && (ldgo & 3) == 0)
        hipLaunchKernelGGL(additive_pool_bwd_fast_kernel, dim3(blocks), dim3(256), 0, ST, t_dpre, ldt, x, ldx, w2, seg_off, extra_off_dyn,
                           n_cap, n_dyn, D, A, gout, ldgo, wrow, dx, lddx, gw2, gb1, scratch);
    else
        hipLaunchKernelGGL(additive_pool_bwd_kernel, dim3(blocks), dim3(256), 0, ST, t_dpre, ldt, x, ldx, w2, seg_off, extra_off_dyn,
                           n_cap, n_dyn, D, A, gout, ldgo, wrow, dx, lddx, gw2, gb1, scratch);
    return check_launch("lego_additive_pool_bwd");
}

extern "C" int lego_additive_pool_bwd_fold(float* scratch, int A, float* gw2, float* gb1, void* stream) {
    LEGO_REQUIRE(scratch != nullptr && A > 0, "lego_additive_pool_bwd_fold: needs the scratch of lego_additive_pool_bwd");
    hipLaunchKernelGGL(pool_replica_reduce_kernel, dim3((2 * A + 255) / 256), dim3(256), 0, ST, scratch, A, gw2, gb1);
    return check_launch("lego_additive_pool_bwd_fold");
}

extern "C" int lego_dot_ce_fwd(const float* user, int ldu, const float* items, int ldi, int B, int C, int D,
                               float* scores, float* loss, void* stream) {
    LEGO_REQUIRE(C <= kMaxCand && (D & 3) == 0 && D <= 256 * kMaxChunks, "lego_dot_ce_fwd: C=%d D=%d unsupported", C, D);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(dot_ce_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, user, ldu, items, ldi, B, C, D, scores, loss);
    return check_launch("lego_dot_ce_fwd");
}
extern "C" int lego_dot_ce_bwd(const float* user, int ldu, const float* items, int ldi, const float* scores,
                               int B, int C, int D, float gscale, const float* gscale_dev, float* guser, int ldgu, float* gitems, int ldgi,
                               void* stream) {
    LEGO_REQUIRE(C <= kMaxCand && (D & 3) == 0 && D <= 256 * kMaxChunks, "lego_dot_ce_bwd: C=%d D=%d unsupported", C, D);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(dot_ce_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, user, ldu, items, ldi, scores, B, C, D, gscale, gscale_dev,
                       guser, ldgu, gitems, ldgi);
    return check_launch("lego_dot_ce_bwd");
}

extern "C" int lego_user_tower_train(float* t_dpre, int ldt, const float* items, int ldi, const float* w2,
                                     const int32_t* hist_off, int B, int C, int S, int D, int A, float gscale,
                                     float* user, float* scores, float* loss, float* d_items, int lddi,
                                     float* gw2, float* gb1, void* stream) {
    LEGO_REQUIRE((D & 3) == 0 && (A & 3) == 0 && D <= 256 * kMaxChunks && A <= 256 * kMaxChunks && C <= kMaxCand && S < kMaxSegRows,
                 "lego_user_tower_train: D=%d A=%d C=%d S=%d unsupported", D, A, C, S);
    if (B <= 0) return 0;
    if (D <= 256 && A <= 256 && S <= kUT_NW * 4 && C <= kUT_NW) {
        hipLaunchKernelGGL(user_tower_train_fast_kernel, dim3(B), dim3(kUT_NW * 64), 0, ST, t_dpre, ldt, items, ldi, w2, hist_off, B, C, D, A,
                           gscale, user, scores, loss, d_items, lddi, gw2, gb1);
        return check_launch("lego_user_tower_train");
    }
    hipLaunchKernelGGL(user_tower_train_kernel, dim3(B), dim3(kUT_NW * 64), 0, ST, t_dpre, ldt, items, ldi, w2, hist_off, B, C, D, A,
                       gscale, user, scores, loss, d_items, lddi, gw2, gb1);
    return check_launch("lego_user_tower_train");
}

extern "C" int lego_rowdot_fwd(const float* u, int ldu, const float* it, int ldi, int n, int D, float* out, void* stream) {
    LEGO_REQUIRE((D & 3) == 0 && D <= 256 * kMaxChunks, "lego_rowdot_fwd: D=%d unsupported", D);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(rowdot_fwd_kernel, dim3((n + 3) / 4), dim3(256), 0, ST, u, ldu, it, ldi, n, D, out);
    return check_launch("lego_rowdot_fwd");
}
extern "C" int lego_rowdot_bwd(const float* u, int ldu, const float* it, int ldi, const float* g, int n, int D,
                               float* gu, int ldgu, float* gi, int ldgi, void* stream) {
    if (n <= 0) return 0;
    const long long total = (long long)n * D;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(blocks), dim3(256), 0, ST, u, ldu, it, ldi, g, n, D, gu, ldgu, gi, ldgi);
    return check_launch("lego_rowdot_bwd");
}
extern "C" int lego_relu_bwd(float* g, int ldg, const float* ref, int ldr, int rows, int width, float scale, void* stream) {
    if (rows <= 0) return 0;
    const long long total = (long long)rows * width;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, ST, g, ldg, ref, ldr, rows, width, scale);
    return check_launch("lego_relu_bwd");
}

static AdamCoef adam_coef(float lr, float beta1, float beta2, float eps, int step, float grad_scale, int zero_grad) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    return AdamCoef{(float)((double)lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale, zero_grad};
}

extern "C" int lego_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, int step, float grad_scale, int zero_grad, void* stream) {
    LEGO_REQUIRE(step >= 1, "lego_adam_step: step is 1-based (got %d)", step);
    if (n <= 0) return 0;
    const AdamCoef a = adam_coef(lr, beta1, beta2, eps, step, grad_scale, zero_grad);
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const long long n4 = aligned ? n / 4 : 0;
    if (n4 > 0) {
        const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, ST, p, g, m, v, n4, a);
    }
    if (4 * n4 < n) {
        const long long rest = n - 4 * n4;
        const int blocks = (int)((rest + 255) / 256 < 4096 ? (rest + 255) / 256 : 4096);
        hipLaunchKernelGGL(adam_scalar_kernel, dim3(blocks), dim3(256), 0, ST, p, g, m, v, 4 * n4, (long long)n, a);
    }
    return check_launch("lego_adam_step");
}

extern "C" int lego_adam_step_rows(float* p, float* g, float* m, float* v, int rows, int width, const uint8_t* touched,
                                   float lr, float beta1, float beta2, float eps, int step, float grad_scale, int zero_grad,
                                   void* stream) {
    LEGO_REQUIRE(step >= 1, "lego_adam_step_rows: step is 1-based (got %d)", step);
    LEGO_REQUIRE((width & 3) == 0 && touched != nullptr, "lego_adam_step_rows: width=%d must be a multiple of 4, touched non-null", width);
    if (rows <= 0) return 0;
    const int blocks = (rows + 3) / 4 < 16384 ? (rows + 3) / 4 : 16384;
    hipLaunchKernelGGL(adam_rows_kernel, dim3(blocks), dim3(256), 0, ST, p, g, m, v, rows, width, touched,
                       adam_coef(lr, beta1, beta2, eps, step, grad_scale, zero_grad));
    return check_launch("lego_adam_step_rows");
}

extern "C" int lego_mark_rows(const int32_t* idx, int n_cap, const int32_t* n_dyn, int rows, uint8_t* touched, void* stream) {
    if (n_cap <= 0) return 0;
    const int blocks = (n_cap + 255) / 256 < 1024 ? (n_cap + 255) / 256 : 1024;
    hipLaunchKernelGGL(mark_rows_kernel, dim3(blocks), dim3(256), 0, ST, idx, n_cap, n_dyn, rows, touched);
    return check_launch("lego_mark_rows");
}

extern "C" int lego_sample_negatives(const int32_t* row_user, const int32_t* row_item, const int32_t* neg_list,
                                     const int32_t* neg_len, int neg_cap, int B, int K, int n_items, uint64_t seed,
                                     uint32_t step, uint32_t row_base, uint32_t row_stride, const int32_t* row_pos, int32_t* cand,
                                     void* stream) {
    LEGO_REQUIRE(K < kMaxCand && n_items > 0, "lego_sample_negatives: K=%d n_items=%d unsupported", K, n_items);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(sample_negatives_kernel, dim3((B + 127) / 128), dim3(128), 0, ST, row_user, row_item, neg_list, neg_len,
                       neg_cap, B, K, n_items, (uint32_t)seed, (uint32_t)(seed >> 32), step, row_base, row_stride, row_pos, cand);
    return check_launch("lego_sample_negatives");
}

extern "C" int lego_gather_history(const int32_t* row_user, const int32_t* user_hist, const int32_t* user_hist_len,
                                   int B, int S, int32_t* hist, int32_t* hist_len, void* stream) {
    if (B <= 0 || S <= 0) return 0;
    hipLaunchKernelGGL(gather_history_kernel, dim3((B * S + 255) / 256), dim3(256), 0, ST, row_user, user_hist, user_hist_len,
                       B, S, hist, hist_len);
    return check_launch("lego_gather_history");
}
