// Plain-row NT product  C[M, N] = epilogue(A[M, K] . B[N, K]^T),  second form (round 5) -- the k loop of wino2_kernel (gemm_wino2.hpp)
// without the Winograd sets, sized so that TWO workgroups share a CU.
//
// Why: the LDS-DMA row-strip kernel (gemm_dma.hpp) keeps its whole B panel [128 x K] in LDS -- 128 KB at K = 256, one workgroup per
// CU -- and is bulk-synchronous: every workgroup of the launch runs prologue, k loop and epilogue at the same time, so at the path's
// K = 256 the matrix pipe idles through 14 us of launch + prologue and 11-21 us of epilogue around 21 us of matrix instructions
// (profiles/r05_strip_ablation.txt: 0.53 / 0.48 busy).  Here
//   * the B fragments come straight from global memory (L2: the weight matrix is 256 KB) into the operand registers, one tile ahead,
//     as in wino2_kernel: a wave owns 16 output columns and nobody else reads its B rows;
//   * A rows are staged through two 20 KB LDS images with branch-free buffer loads (rows past the strip: an offset past the
//     descriptor's extent, the range check returns zeros), fragments are read two k at a time (ds_read_b64) so that the register
//     budget is <= 128 and four waves fit a SIMD;
//   * LDS is 63 KB (the row-major epilogue tile reuses the A images), the launch has 2 x #CU workgroups of <= 112 rows x 128 columns and
//     four waves per SIMD: one workgroup's loads, barriers and epilogue run under the other's matrix instructions.  (Starting the second
//     workgroup of every CU a few microseconds late, to put the two out of phase, measured 1-3 % slower: profiles/r05_rows2.txt.)
//   * B_MC: the NN form (data gradients, C = A . W with W [K][N] as the forward pass holds it): the lane's four reduction indices of a
//     k group are four rows of W -- eight dword loads per tile instead of two 16-byte ones, everything else unchanged.
// Measured (tools/rows2_check.py, 27.6 k x 256 x 256): NT + tanh 44.6 -> 41.1 us, NN accumulate + ReLU' + column sums 54.1 -> 45.5 us; in the
// NAML step 53 -> 45 and 58 -> 47-50 us, step 0.576 -> 0.555 ms.  Outputs wider than 256 columns (more column blocks re-reading every A
// strip) stay on the row-strip kernels: N = 768 158 against 133 us at two workgroups per CU.
// Epilogue kinds: bias + activation (none / ReLU / tanh), or accumulate onto C [+ ReLU' mask from a reference] + column sums.
// Needs K % 4 == 0 (a partial last k tile reads zeros), N % 4 == 0, 16-byte aligned rows, A of < 2 GiB, no dropout / live-bit epilogue; everything else stays on the
// row-strip kernels (gemm_ops.hip: launch_rows).
#pragma once
#include "gemm_epi.hpp"

namespace lego {

constexpr int R2_BN = 128;                                  // columns per workgroup (8 waves x 16)
constexpr int R2_BP = 112;                                  // rows per pass (7 fragments of 16)
constexpr int R2_LD = STRIP_KC_LD;                          // 40 floats per A image row
constexpr int R2_A_FLOATS = 128 * R2_LD;                    // one A stage: 128 x 40 floats (every thread stores, rows >= 112 unread)
constexpr int R2_EPI_LD = R2_BN + 4;                        // 132 floats per epilogue tile row
constexpr int R2_EPI_FLOATS = R2_BP * R2_EPI_LD;
constexpr size_t rows2_lds_bytes() { return (size_t)(R2_EPI_FLOATS + 8 * R2_BN) * sizeof(float); }
static_assert(2 * R2_A_FLOATS <= R2_EPI_FLOATS, "the A stages live inside the epilogue tile's region");

struct Rows2Args {
    const float* x; int ldx; unsigned x_bytes;              // A rows (x_bytes = M_cap * ldx * 4 < 2^31)
    const float* w; int ldw; unsigned w_bytes;              // B: [N][K] rows (NT), or -- B_MC -- [K][N] rows: the reduction index is the ROW (NN products,
    int M_cap; const int* M_dyn; int N, K;                  // data gradients: C = A . W with W as the forward pass holds it)
    int min_strip;                                          // fewest rows a workgroup takes (live rows / this = strips in use)
};

template <int NF, bool B_MC, bool ACCUM, bool RELUREF>
__device__ __forceinline__ void rows2_pass(const Rows2Args& w, const EpiArgs& e, float* lds, int r0, int r_end, int n0) {
    constexpr int AN = (NF * 16 + 63) / 64;                 // rows of the A tile per thread (1 or 2)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g4 = lane >> 4;
    const int N = w.N;
    const int K = w.K;
    const int KT = (K + BK - 1) / BK;                       // K % 4 == 0; a partial last tile reads zeros past K (offset selects below)
    float* const As0 = lds;

    // ---- A fetch stream (branch-free: the k loop is one basic block, hipcc's s_waitcnt vmcnt stays counted)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.x), 0, (int)w.x_bytes, 0x00020000);
    constexpr unsigned kNone = 0x80000000u;
    unsigned rowo[AN];
#pragma unroll
    for (int j = 0; j < AN; ++j) {
        const int r = r0 + (tid >> 3) + 64 * j;
        rowo[j] = r < r_end ? (unsigned)r * (unsigned)w.ldx * 4u + (unsigned)(tid & 7) * 16u : kNone;
    }
    int fkt = 0;
    f32x4 sa[AN];
    const int kposA = (tid & 7) * 4;                        // this thread's four reduction indices inside a tile
    auto fetchA = [&]() {
        const int kt_ = __builtin_amdgcn_readfirstlane(min(fkt, KT - 1));               // past the last tile: a re-read, never committed to a live stage
        const int ko = kt_ * (BK * 4);
        const bool in = kposA < K - kt_ * BK;               // (all true except in a partial last tile: there the offset becomes kNone -> zeros)
#pragma unroll
        for (int j = 0; j < AN; ++j) sa[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, in ? rowo[j] : kNone, ko, 0));
        ++fkt;
    };
    auto commit = [&](float* A_) {
#pragma unroll
        for (int j = 0; j < AN; ++j) *reinterpret_cast<f32x4*>(A_ + ((tid >> 3) + 64 * j) * R2_LD + (tid & 7) * 4) = sa[j];
    };
    // ---- B fetch stream: this wave's 16 columns (rows of w), per-lane offset fixed, k tile scalar
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.w), 0, (int)w.w_bytes, 0x00020000);
    const unsigned bcol = (unsigned)min(n0 + wave * 16 + l16, N - 1);
    const unsigned vb = B_MC ? (4u * g4 * (unsigned)w.ldw + bcol) * 4u : (bcol * (unsigned)w.ldw + 4u * g4) * 4u;
    const int ldw4 = w.ldw * 4;
    int bkt = 0;
    f32x4 sb0, sb1;
    auto fetchB = [&]() {
        if constexpr (!B_MC) {
            const int kt_ = __builtin_amdgcn_readfirstlane(min(bkt, KT - 1));
            const int ko = kt_ * (BK * 4);
            const int lim = K - kt_ * BK;
            sb0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, 4 * g4 < lim ? vb : kNone, ko, 0));
            sb1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, 16 + 4 * g4 < lim ? vb : kNone, ko + 64, 0));
        } else {        // (rows of w past K are past the descriptor's extent: zeros without a select)        // the lane's four reduction indices of a k group are four ROWS of w: eight dword loads per tile (each a 64-byte run per 16 lanes)
            const int ko = __builtin_amdgcn_readfirstlane(min(bkt, KT - 1)) * BK * ldw4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sb0[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vb, ko + j * ldw4, 0));
                sb1[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vb, ko + (16 + j) * ldw4, 0));
            }
        }
        ++bkt;
    };

    f32x4 acc[NF];
#pragma unroll
    for (int a = 0; a < NF; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef float f32x2_r __attribute__((ext_vector_type(2)));
    f32x2_r fr0[NF], fr1[NF];
    f32x4 fb0, fb1;
    // fragment reads two k at a time: half h = 0..3 of a tile = k group q = h >> 1, elements 2 (h & 1) .. +1 of the lane's four
    auto read_half = [&](const float* A_, int h, f32x2_r (&fr)[NF]) {
#pragma unroll
        for (int a = 0; a < NF; ++a)
            fr[a] = *reinterpret_cast<const f32x2_r*>(A_ + (a * 16 + l16) * R2_LD + 16 * (h >> 1) + 4 * g4 + 2 * (h & 1));
    };
    auto mfma2 = [&](const f32x2_r (&fr)[NF], float b0, float b1) {
#pragma unroll
        for (int a = 0; a < NF; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[a][0], b0, acc[a], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < NF; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[a][1], b1, acc[a], 0, 0, 0);
    };

    // ---- pipeline.  Top of tile t: LDS buf[t & 1] = A tile t, fr0 = its half 0, fb0 / fb1 = B tile t, staging registers = A tile t + 1
    fetchA();
    commit(As0);
    fetchA();
    fetchB();
    fb0 = sb0; fb1 = sb1;
    __syncthreads();
    read_half(As0, 0, fr0);
    int buf = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const float* A_ = As0 + buf * R2_A_FLOATS;
        float* An = As0 + (buf ^ 1) * R2_A_FLOATS;
        fetchB();
        read_half(A_, 1, fr1);
        mfma2(fr0, fb0[0], fb0[1]);
        commit(An);
        fetchA();
        read_half(A_, 2, fr0);
        mfma2(fr1, fb0[2], fb0[3]);
        read_half(A_, 3, fr1);
        mfma2(fr0, fb1[0], fb1[1]);
        __builtin_amdgcn_sched_barrier(0);          // the barrier stays BEHIND these MFMAs (they cover the stage stores' latency)
        __syncthreads();
        read_half(An, 0, fr0);
        mfma2(fr1, fb1[2], fb1[3]);
        fb0 = sb0; fb1 = sb1;
        buf ^= 1;
    }

    // ---- epilogue, row-major through LDS: the tile [NF * 16][132], then lane = (one of two rows, 4 consecutive columns)
    float* const tile = lds;
    float* const csc = lds + R2_EPI_FLOATS;         // [8][128] column-sum partials
    const int colw = wave * 16 + l16;
    const int hrow = lane >> 5, c4 = lane & 31;
    const int col = n0 + 4 * c4;
    const bool col_ok = col < N;
    const int colc = min(col, N - 4);
    const f32x4 b4 = e.bias != nullptr ? *reinterpret_cast<const f32x4*>(e.bias + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                // every wave is done with the A stages
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int v = 0; v < 4; ++v) tile[(a * 16 + 4 * g4 + v) * R2_EPI_LD + colw] = acc[a][v];
    // the accumulated-onto rows / ReLU references of all of this wave's rows first: independent loads, in flight across the barrier
    f32x4 old[NF], ref[NF];
#pragma unroll
    for (int I = 0; I < NF; ++I) {
        const int r = min(r0 + 16 * I + 2 * wave + hrow, r_end - 1) + e.row_off;
        old[I] = f32x4{0.f, 0.f, 0.f, 0.f};
        ref[I] = f32x4{1.f, 1.f, 1.f, 1.f};
        if constexpr (ACCUM) old[I] = *reinterpret_cast<const f32x4*>(e.C + (size_t)r * e.ldc + colc);
        if constexpr (RELUREF) ref[I] = *reinterpret_cast<const f32x4*>(e.relu_ref + (size_t)r * e.ld_ref + colc);
    }
    __syncthreads();
    f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int I = 0; I < NF; ++I) {
        const int lr = 16 * I + 2 * wave + hrow;    // row of the tile
        const int r = r0 + lr;
        f32x4 x = *reinterpret_cast<const f32x4*>(tile + lr * R2_EPI_LD + 4 * c4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = x[c] + b4[c];
            if (e.act == 1) v = fmaxf(v, 0.f);
            else if (e.act == 2) v = fast_tanh(v);
            if (ACCUM) v += old[I][c];
            if (RELUREF) v = ref[I][c] > 0.f ? v * e.relu_scale : 0.f;
            x[c] = v;
        }
        if (col_ok && r < r_end) {
            *reinterpret_cast<f32x4*>(e.C + (size_t)(r + e.row_off) * e.ldc + col) = x;
            cs += x;
        }
    }
    if (e.colsum != nullptr) {                      // kernel-uniform
#pragma unroll
        for (int c = 0; c < 4; ++c) cs[c] += __shfl_xor(cs[c], 32, 64);
        if (lane < 32) *reinterpret_cast<f32x4*>(csc + wave * R2_BN + 4 * c4) = cs;
        __syncthreads();
        if (tid < R2_BN) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += csc[q * R2_BN + tid];
            if (n0 + tid < N) atomicAdd(e.colsum + n0 + tid, s);
        }
    }
    __syncthreads();                                // the next pass refills the stages
}

template <bool B_MC, bool ACCUM, bool RELUREF>
__global__ __launch_bounds__(STRIP_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void rows2_kernel(Rows2Args w, EpiArgs e) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = w.M_dyn != nullptr ? min(w.M_cap, *w.M_dyn) : w.M_cap;
    if (M <= 0) return;
    e.row_off = e.row_off_dyn != nullptr ? *e.row_off_dyn : 0;
    // (strip, column block) dealing: the nblk blocks of a strip are 8 apart in blockIdx -- round-robin dispatch puts them on one XCD,
    // whose L2 then serves the strip's A rows to all of them
    const int nblk = (w.N + R2_BN - 1) / R2_BN;
    // strips: the grid's, but never shorter than min_strip rows -- a launch sized for a capacity with few live rows (the projection over the
    // ~4.5 k distinct tokens of a batch: capacity 105 k) would otherwise stream the whole B panel through 2 x #CU workgroups of 16 rows each
    const int G = max(min((int)gridDim.x / nblk, (M + w.min_strip - 1) / w.min_strip), 1);
    const int grp = blockIdx.x / (8 * nblk), in = blockIdx.x % (8 * nblk);
    const int strip = grp * 8 + (in & 7), blk = in >> 3;
    if (strip >= G) return;
    const int s = ((M + G - 1) / G + 15) & ~15;
    const int nsub = (s + R2_BP - 1) / R2_BP;
    const int sub = (((s + nsub - 1) / nsub) + 15) & ~15;
    const int strip0 = strip * s;
    if (strip0 >= M) return;
    const int strip_end = min(M, strip0 + s);
    const int n0 = blk * R2_BN;
    for (int r0 = strip0; r0 < strip_end; r0 += sub) {
        const int r_end = min(strip_end, r0 + sub);
        switch ((r_end - r0 + 15) >> 4) {                                   // block-uniform
            case 1: case 2: rows2_pass<2, B_MC, ACCUM, RELUREF>(w, e, smem, r0, r_end, n0); break;
            case 3: case 4: rows2_pass<4, B_MC, ACCUM, RELUREF>(w, e, smem, r0, r_end, n0); break;
            case 5: rows2_pass<5, B_MC, ACCUM, RELUREF>(w, e, smem, r0, r_end, n0); break;
            case 6: rows2_pass<6, B_MC, ACCUM, RELUREF>(w, e, smem, r0, r_end, n0); break;
            default: rows2_pass<7, B_MC, ACCUM, RELUREF>(w, e, smem, r0, r_end, n0); break;
        }
    }
}

}  // namespace lego
