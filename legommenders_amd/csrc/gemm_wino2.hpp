// Winograd F(2,3) conv over row pairs, second form (round 5).  Same arithmetic, work split and accumulators as wino_kernel<false>
// (round 3: a strip of <= 112 pairs x 128 columns per workgroup, 8 waves side by side over the columns, three accumulator
// sets, v_mfma_f32_16x16x4_f32), with the parts of its k loop that kept the matrix pipe at 0.56 busy rebuilt:
//
//   * the WEIGHT fragments never touch LDS.  A wave owns 16 output columns, so the B operand of its MFMAs is 16 rows x 32 k of the
//     transformed weight set per tile -- nobody else reads it.  Lane (l16, g4) loads U_s[col0 + l16][k0 + 16 q + 4 g4 .. +3] as one
//     f32x4 per k group (the same k permutation the A fragments use), one tile ahead, straight into the operand registers: no
//     staging store, no second read, half the LDS footprint, two global loads per lane and tile instead of two loads + two
//     ds_write_b128 + two ds_read_b128.
//   * the A staging has no zeroing selects and no 64-bit address arithmetic: the two source rows of a staged pair row are 32-bit
//     offsets into a buffer descriptor (a neighbour the plan says does not exist is an offset past the extent: the hardware range
//     check returns zeros), the k tile is the scalar offset.  The combination is one packed multiply-add per two elements.
//   * (tried and dropped: staggering the two waves of a SIMD -- waves 4-7 staging before the first k group's MFMAs, waves 0-3 after
//     them -- measured 4 % SLOWER here, 82.8 against 79.2 us; profiles/r05_wino2_ablation.txt.)
//   * the epilogue is row-major: the accumulators go through an LDS tile [2 x 64 pairs][128 + 4] and leave as 16-byte stores, a
//     wave instruction covering the 512 contiguous bytes of two output rows; bias, keep bits (one 32-bit load per 4 columns) and
//     the column sums are applied on that side.  The fragment-shaped epilogue stored 4 x 64 B per instruction, one dword per lane.
//
// Needs the keep bits precomputed (drop.mask) when dropout is on, N % 4 == 0, 16-byte aligned rows and an input of < 2 GiB; the
// entry points refuse anything else with the reason (wino2_why_not): round 6 removed the round-3 kernel this one superseded.
#pragma once
#include "gemm_epi.hpp"
#include "wino_common.hpp"

namespace lego {

constexpr int W2_LD = STRIP_KC_LD;                          // 40 floats per A image row
constexpr int W2_A_FLOATS = 128 * W2_LD;                    // one A stage: 128 x 40 floats = 20 480 B (every thread stores, rows >= 112 unread)
constexpr int W2_EPI_PAIRS = 64;                            // pairs per epilogue round
constexpr int W2_EPI_LD = WINO_BN + 4;                      // 132 floats per epilogue tile row
constexpr int W2_EPI_FLOATS = 2 * W2_EPI_PAIRS * W2_EPI_LD; // 128 rows
constexpr size_t wino2_lds_bytes() { return (size_t)(W2_EPI_FLOATS + 8 * WINO_BN) * sizeof(float); }
static_assert(2 * W2_A_FLOATS <= W2_EPI_FLOATS, "the A stages live inside the epilogue tile's region");


// ABL (tuning build only): ablation bits for timing what the k loop spends where -- 1 no global loads in the loop, 2 no LDS stores,
// 4 no barrier, 8 no epilogue, 16 no fragment reads, 32 no MFMAs, 64 the staging block pinned between the MFMA groups, 128 / 256 no A / B loads.  Results are wrong by construction; ABL = 0 is the product.
template <int NF, int ABL>
__device__ __forceinline__ void wino2_pass(const WinoArgs& w, const EpiArgs& e, float* lds, int p0, int p_end, int P, int n0) {
    constexpr int AN = (NF * 16 + 63) / 64;                 // pair rows of the A tile per thread (1 or 2)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g4 = lane >> 4;
    const int C = w.C, N = w.N;
    const int KT = C / BK;
    const size_t set_stride = (size_t)N * C;
    float* const As0 = lds;

    // ---- A fetch stream: (set, k tile) of the next tile to fetch.  Rows are read through a buffer descriptor: the two source rows of
    // a staged pair row are 32-bit byte offsets chosen per set, the k tile is the instruction's scalar offset, and a neighbour the
    // plan says does not exist is an offset past the descriptor's extent -- the hardware range check returns zeros for it.
    // The stream's state advances WITHOUT branches (selects on scalars): the k loop is one basic block, which is what lets hipcc
    // count its s_waitcnt vmcnt exactly (with branches in the loop it fell back to vmcnt(0) in front of every use of a loaded
    // register, i.e. the loads of tile t + 2 had to land half a tile after they were issued).
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.x), 0, (int)w.x_bytes, 0x00020000);
    constexpr unsigned kNone = 0x80000000u;                 // x_bytes < 2^31 (launcher)
    const unsigned ldb = (unsigned)w.ldx * 4u;
    // offsets per set, selected with scalar masks (no branch):  first source row = set 0: left, set 2: second, else the pair's row;
    // second source row = set 2: the pair's row, set 3: right2, else second
    unsigned row1[AN], second[AN], x_left[AN], x_second[AN], x_right2[AN];
#pragma unroll
    for (int j = 0; j < AN; ++j) {
        const int info = w.pair_info[min(p0 + (tid >> 3) + 64 * j, P - 1)];
        row1[j] = (unsigned)(info >> PI_ROW_SHIFT) * ldb + (unsigned)(tid & 7) * 16u;
        second[j] = (info & PI_HAS2) ? row1[j] + ldb : kNone;
        x_left[j] = ((info & PI_LEFT) ? row1[j] - ldb : kNone) ^ row1[j];
        x_second[j] = second[j] ^ row1[j];
        x_right2[j] = ((info & PI_RIGHT2) ? row1[j] + 2u * ldb : kNone) ^ second[j];
    }
    int fset = 0, fkt = 0;
    f32x4 sa1[AN], sa2[AN];
    float ssgn = -1.f;
    // set 0: d0 - d2   set 1: d1 + d2   set 2: d2 - d1   set 3: d1 - d3   (d1 = the pair's first row)
    auto fetchA = [&]() {
        const int s = __builtin_amdgcn_readfirstlane(min(fset, 3));    // past the last tile the stream re-reads set 3 (never committed to a live stage)
        const int ko = __builtin_amdgcn_readfirstlane(fkt) * (BK * 4);
        const unsigned m0 = s == 0 ? ~0u : 0u, m2 = s == 2 ? ~0u : 0u, m3 = s == 3 ? ~0u : 0u;
        ssgn = __builtin_bit_cast(float, 0xbf800000u ^ (s == 1 ? 0x80000000u : 0u));     // +1 for set 1, -1 otherwise
#pragma unroll
        for (int j = 0; j < AN; ++j) {
            const unsigned v1 = row1[j] ^ (m0 & x_left[j]) ^ (m2 & x_second[j]);
            const unsigned v2 = second[j] ^ (m2 & x_second[j]) ^ (m3 & x_right2[j]);
            sa1[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, v1, ko, 0));
            sa2[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, v2, ko, 0));
        }
        const int nk = fkt + 1;
        const int wrap = nk == KT ? 1 : 0;
        fkt = nk - wrap * KT;
        fset += wrap;
    };
    auto commit = [&](float* A_) {                          // the A image has AN * 64 rows: no guard (rows >= NF * 16 are never read)
#pragma unroll
        for (int j = 0; j < AN; ++j) {
            const f32x4 v = sa1[j] + ssgn * sa2[j];
            if constexpr (ABL & 2) asm volatile("" :: "v"(v));
            else *reinterpret_cast<f32x4*>(A_ + ((tid >> 3) + 64 * j) * W2_LD + (tid & 7) * 4) = v;
        }
    };

    // ---- B fetch stream: this wave's 16 columns of the set's weight matrix (per-lane offset fixed, set and k tile scalar)
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.u), 0, (int)(4 * set_stride * sizeof(float)), 0x00020000);
    const unsigned vb = ((unsigned)min(n0 + wave * 16 + l16, N - 1) * (unsigned)C + 4u * g4) * 4u;
    const int sbytes = (int)(set_stride * sizeof(float));
    int bset = 0, bkt = 0;
    f32x4 sb0, sb1;
    auto fetchB = [&]() {
        const int s = __builtin_amdgcn_readfirstlane(min(bset, 3));
        const int ws = s ^ (((s == 0 || s == 3) && w.swap) ? 3 : 0);   // backward-data: sets 0 and 3 swap their weights
        const int ko = ws * sbytes + __builtin_amdgcn_readfirstlane(bkt) * (BK * 4);
        sb0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, vb, ko, 0));
        sb1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, vb, ko + 64, 0));
        const int nk = bkt + 1;
        const int wrap = nk == KT ? 1 : 0;
        bkt = nk - wrap * KT;
        bset += wrap;
    };

    // y0 = M0 + M1 + M2, y1 = M1 - M2 - M3: sets 0 and 1 accumulate straight into y0 / y1, sets 2 and 3 into a temporary
    f32x4 y0a[NF], y1a[NF], tma[NF];
#pragma unroll
    for (int a = 0; a < NF; ++a) {
        y0a[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        y1a[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        tma[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 fa0[NF], fa1[NF], fb0, fb1;
    auto read_frags = [&](const float* A_, int q, f32x4 (&fa)[NF]) {
#pragma unroll
        for (int a = 0; a < NF; ++a)
            fa[a] = *reinterpret_cast<const f32x4*>(A_ + (a * 16 + l16) * W2_LD + 16 * q + 4 * g4);
    };
    auto mfma = [&](f32x4& c, float a, float b) {
        if constexpr (ABL & 32) asm volatile("" : "+v"(c) : "v"(a), "v"(b));
        else c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    };

    // ---- pipeline.  Top of tile t: LDS buf[t & 1] = A tile t, fa0 = its first k group, fb0 / fb1 = B tile t, staging registers =
    // A tile t + 1 (its loads were issued in the middle of tile t - 1).
    //   fetch B(t+1) | read F1(t) | MFMA F0 | commit A(t+1), fetch A(t+2) | MFMA F1 j=0..2 | barrier | read F0(t+1) | MFMA F1 j=3 |
    //   B registers <- B(t+1)  (counted wait: the A loads of tile t + 2 stay in flight)
    fetchA();
    commit(As0);
    fetchA();
    fetchB();
    fb0 = sb0; fb1 = sb1;
    __syncthreads();
    read_frags(As0, 0, fa0);
    read_frags(As0, 1, fa1);
    int buf = 0;
    auto run_set = [&](f32x4 (&ac)[NF]) {
        for (int kt = 0; kt < KT; ++kt) {
            const float* A_ = As0 + buf * W2_A_FLOATS;
            float* An = As0 + (buf ^ 1) * W2_A_FLOATS;
            if constexpr (!(ABL & (1 | 256))) fetchB();
            if constexpr (!(ABL & 16)) read_frags(A_, 1, fa1);
            if constexpr (ABL & 64) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < NF; ++a) mfma(ac[a], fa0[a][j], fb0[j]);
            if constexpr (ABL & 64) __builtin_amdgcn_sched_barrier(0);
            commit(An);
            if constexpr (!(ABL & (1 | 128))) fetchA();
            if constexpr (ABL & 64) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int a = 0; a < NF; ++a) mfma(ac[a], fa1[a][j], fb1[j]);
            if constexpr (!(ABL & 64)) {
                // ONE scheduling region from the top of the tile to the barrier: the staging work (address selects, the combination,
                // six loads, two stage stores -- ~40 vector instructions) is dealt out BETWEEN the 7 NF matrix instructions instead of
                // standing in a block between two groups of them, where both waves of a SIMD ran it at the same time with the matrix
                // pipe idle (tools/wino2_check.py ablations 1-3: 9 us of a 74 us launch).  0x008 MFMA, 0x002 VALU, 0x020 VMEM read,
                // 0x100 / 0x200 DS read / write.
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
#pragma unroll
                for (int i = 0; i < 7 * NF / 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    if (i >= 3 && i < 3 + 2 * AN) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (i >= 8 && i < 8 + AN) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 7 * NF - 3 * (7 * NF / 3), 0);
            }
            __builtin_amdgcn_sched_barrier(0);       // the barrier stays BEHIND these MFMAs (they cover the stage stores' latency); hipcc
                                                     // hoists it over register-only instructions otherwise
            if constexpr (ABL & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else __syncthreads();
            if constexpr (!(ABL & 16)) read_frags(An, 0, fa0);
#pragma unroll
            for (int a = 0; a < NF; ++a) mfma(ac[a], fa1[a][3], fb1[3]);
            fb0 = sb0; fb1 = sb1;
            buf ^= 1;
        }
    };
    run_set(y0a);                                    // M0
    run_set(y1a);                                    // M1
    run_set(tma);                                    // M2
#pragma unroll
    for (int a = 0; a < NF; ++a) {
        y0a[a] += y1a[a] + tma[a];
        y1a[a] -= tma[a];
        tma[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    run_set(tma);                                    // M3
#pragma unroll
    for (int a = 0; a < NF; ++a) y1a[a] -= tma[a];
    if constexpr (ABL & 8) {
#pragma unroll
        for (int a = 0; a < NF; ++a) asm volatile("" :: "v"(y0a[a]), "v"(y1a[a]));
        __syncthreads();
        return;
    }

    // ---- epilogue, row-major through LDS: rounds of 64 pairs = 128 tile rows (y0 row, y1 row of each pair)
    float* const tile = lds;
    float* const csc = lds + W2_EPI_FLOATS;          // [8][128] column-sum partials
    const int colw = wave * 16 + l16;                // this lane's accumulator column inside the workgroup's 128
    const int hrow = lane >> 5, c4 = lane & 31;      // store side: lane = (row of the pair, 4 consecutive columns)
    const int col = n0 + 4 * c4;
    const bool col_ok = col < N;
    const int colc = min(col, N - 4);
    const f32x4 b4 = e.bias != nullptr ? *reinterpret_cast<const f32x4*>(e.bias + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
    const bool dropping = e.drop.p > 0.f;
    const float dinv = dropping ? 1.f / (1.f - e.drop.p) : 1.f;
    f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
    // store-side row I of this wave = pair p0 + 8 I + wave.  Its pair_info word and keep bits are loaded for ALL rows up front, two
    // batches of independent loads (the first form walked the rows with a dependent pair_info -> keep-bits -> store chain per row:
    // 11 us of a 74 us launch, tools/wino2_check.py ablation 8)
    constexpr int NI = 2 * NF;
    int inf[NI];
    uint32_t kws[NI];
#pragma unroll
    for (int I = 0; I < NI; ++I) inf[I] = w.pair_info[min(p0 + 8 * I + wave, P - 1)];
#pragma unroll
    for (int I = 0; I < NI; ++I) {
        const int r = (inf[I] >> PI_ROW_SHIFT) + ((hrow == 1 && (inf[I] & PI_HAS2)) ? 1 : 0);   // lanes of a missing second row read row 1's bits
        kws[I] = 0x0f0f0f0fu;
        if (dropping) kws[I] = *reinterpret_cast<const uint32_t*>(e.drop.mask + (uint64_t)(r >> 2) * (uint64_t)e.drop_cols + (uint64_t)colc);
    }
#pragma unroll
    for (int g = 0; g < (NF + 3) / 4; ++g) {
        constexpr int per = W2_EPI_PAIRS / 16;       // fragments per round
        __syncthreads();                             // the tile region is free: last fragment reads / the previous round's row reads
#pragma unroll
        for (int a = per * g; a < NF && a < per * (g + 1); ++a)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float* t = tile + (2 * ((a - per * g) * 16 + 4 * g4 + v)) * W2_EPI_LD + colw;
                t[0] = y0a[a][v];
                t[W2_EPI_LD] = y1a[a][v];
            }
        __syncthreads();
        f32x4 xs[2 * per];
#pragma unroll
        for (int i = 0; i < 2 * per; ++i)
            if (2 * per * g + i < NI) xs[i] = *reinterpret_cast<const f32x4*>(tile + (2 * (8 * i + wave) + hrow) * W2_EPI_LD + 4 * c4);
#pragma unroll
        for (int i = 0; i < 2 * per; ++i) {
            const int I = 2 * per * g + i;
            if (I < NI) {
                const int info = inf[I];
                const bool has2 = (info & PI_HAS2) != 0;
                const int r = (info >> PI_ROW_SHIFT) + ((hrow == 1 && has2) ? 1 : 0);
                const bool ok = col_ok && (hrow == 0 || has2) && p0 + 8 * I + wave < p_end;
                const uint32_t kw = kws[I] >> (r & 3);
                f32x4 x = xs[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = x[c] + b4[c];
                    if (e.act == 1) v = fmaxf(v, 0.f);
                    v *= (kw >> (8 * c)) & 1u ? dinv : 0.f;
                    x[c] = v;
                }
                if (ok) {
                    *reinterpret_cast<f32x4*>(e.C + (size_t)r * e.ldc + col) = x;
                    cs += x;
                }
            }
        }
    }
    if (e.colsum != nullptr) {                       // kernel-uniform
#pragma unroll
        for (int c = 0; c < 4; ++c) cs[c] += __shfl_xor(cs[c], 32, 64);
        if (lane < 32) *reinterpret_cast<f32x4*>(csc + wave * WINO_BN + 4 * c4) = cs;
        __syncthreads();
        if (tid < WINO_BN) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += csc[q * WINO_BN + tid];
            if (n0 + tid < N) atomicAdd(e.colsum + n0 + tid, s);
        }
    }
    __syncthreads();                                 // the next pass refills the stages
}

template <int ABL>
__global__ __launch_bounds__(STRIP_THREADS) void wino2_kernel(WinoArgs w, EpiArgs e) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = w.P_dyn != nullptr ? min(w.P_cap, *w.P_dyn) : w.P_cap;
    if (P <= 0) return;
    // (strip, column half) dealing of wino_kernel: blockIdx and blockIdx + 8 share a strip and (round-robin dispatch) an XCD
    const int halves = (w.N + WINO_BN - 1) / WINO_BN;                       // 1 or 2
    const int G = max((int)gridDim.x / halves, 1);
    int strip, half;
    if (halves == 2) { half = (blockIdx.x >> 3) & 1; strip = (blockIdx.x & 7) + 8 * (blockIdx.x >> 4); }
    else { half = 0; strip = blockIdx.x; }
    if (strip >= G) return;
    int s = ((P + G - 1) / G + 15) & ~15;
    const int nsub = (s + WINO_BP - 1) / WINO_BP;
    const int sub = (((s + nsub - 1) / nsub) + 15) & ~15;
    const int strip0 = strip * s;
    if (strip0 >= P) return;
    const int strip_end = min(P, strip0 + s);
    const int n0 = half * WINO_BN;
    for (int p0 = strip0; p0 < strip_end; p0 += sub) {
        const int p_end = min(strip_end, p0 + sub);
        switch ((p_end - p0 + 15) >> 4) {                                   // block-uniform
            case 1: case 2: wino2_pass<2, ABL>(w, e, smem, p0, p_end, P, n0); break;
            case 3: case 4: wino2_pass<4, ABL>(w, e, smem, p0, p_end, P, n0); break;
            case 5: wino2_pass<5, ABL>(w, e, smem, p0, p_end, P, n0); break;
            case 6: wino2_pass<6, ABL>(w, e, smem, p0, p_end, P, n0); break;
            default: wino2_pass<7, ABL>(w, e, smem, p0, p_end, P, n0); break;
        }
    }
}

}  // namespace lego
