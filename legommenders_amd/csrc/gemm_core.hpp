// fp32 MFMA GEMM core for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
// One LDS-tiled main loop, specialised by operand loaders and an epilogue functor:
//   * operands are either K-contiguous in memory ("KC": rows of the tile are matrix rows,
//     LDS image [row][BK+4], fragments read with ds_read_b128) or M-contiguous ("MC": the
//     reduction index is the memory row, LDS image [BK][rows], fragments read with ds_read_b32);
//     NT = KC x KC, NN = KC x MC, TN = MC x MC -- no operand is ever transposed in memory.
//   * loaders express the ragged gathers of the path: plain rows, rows through an index,
//     conv taps (row r +/- 1 inside one item, predicate from the plan's rowinfo word).
//   * 256 threads = 4 waves as WM x WN, each wave TM x TN MFMA 32x32 sub-tiles, BK = 32,
//     register-staged double buffering: global loads for tile t+1 are issued before the
//     MFMAs of tile t and written to the other LDS buffer after them (one barrier per tile).
//   * the k order inside a BK step is permuted (lane half h takes k = 8q+4h+j for MFMA j)
//     identically for A and B, so one ds_read_b128 feeds four MFMAs.
#pragma once
#include <type_traits>
#include "common.hpp"

namespace lego {

constexpr int BK = 32;
constexpr int KC_LD = BK + 4;   // +16 B pad: the 16 lanes of a ds_read_b128 group hit 16 distinct slots

// ------------------------------------------------------------------ operand loaders
// Every loader has: ext (rows of a KC operand / columns of an MC operand), K (reduction bound, set by
// the kernel to the end of its k range), prepare(tap), called once in the kernel prologue, and tile(k0),
// called once per k tile.
// The element loads are BRANCH-FREE: load() reads from an address clamped into the operand and keep() says
// whether the value counts; the kernels apply the zeroing select when they write the staged registers to LDS
// (a tile later), so the k loop stays one basic block and nothing waits on a load where it is issued.  Rows / columns beyond ext are NOT zeroed: they
// only feed output elements that the epilogue never stores.  Requires ext >= 1 and K >= 4.
__device__ __forceinline__ f32x4 zero_unless(bool keep, f32x4 v) {
    return f32x4{keep ? v[0] : 0.f, keep ? v[1] : 0.f, keep ? v[2] : 0.f, keep ? v[3] : 0.f};
}

// KC loaders: tile row -> matrix row, k contiguous.  row(r) is evaluated once per thread.
struct KcRows {            // plain rows, optionally behind a device row offset
    const float* p; int ld; int ext; int K; const int* row_off_dyn;
    struct Row { const float* base; };
    __device__ __forceinline__ void prepare(int) { if (row_off_dyn != nullptr) p += (size_t)(*row_off_dyn) * ld; }
    __device__ __forceinline__ void tile(int) {}
    __device__ __forceinline__ Row row(int r) const { return {p + (size_t)min(r, ext - 1) * ld}; }
    __device__ __forceinline__ f32x4 load(const Row& R, int k) const { return *reinterpret_cast<const f32x4*>(R.base + min(k, K - 4)); }
    __device__ __forceinline__ bool keep(const Row&, int k) const { return k < K; }
};

struct KcConvA {           // logical K = 3*C: k -> (tap = k / C, c = k % C); source row r + dir*(tap-1)
    const float* p; int ld; int ext; int K; const int* rowinfo; int C; int dir;
    int need, koff, shift;                     // per-k-tile scalars (C % BK == 0: a tile never straddles two taps)
    struct Row { const float* base; int flags; };
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void tile(int k0) {
        const int tap = k0 / C;
        const int s = dir * (tap - 1);
        koff = tap * C;
        shift = s * ld;
        need = s == 0 ? 0 : (s < 0 ? RI_LEFT : RI_RIGHT);
    }
    __device__ __forceinline__ Row row(int r) const {
        const int rc = min(r, ext - 1);
        return {p + (size_t)rc * ld, rowinfo[rc]};
    }
    // keep: the plan says the neighbour row exists (same item)
    __device__ __forceinline__ bool keep(const Row& R, int) const { return need == 0 || (R.flags & need) != 0; }
    __device__ __forceinline__ f32x4 load(const Row& R, int k) const {
        return *reinterpret_cast<const f32x4*>(R.base + (keep(R, k) ? shift : 0) + (k - koff));
    }
};

struct KcTapW {            // B of the conv forward: B[o][tap*C + c] = wt[tap][o][c]  (tap-major packed weights)
    const float* p; int ld /*= C*/; int ext /*= Dout*/; int K; int C; size_t tap_stride /*= Dout*C*/;
    ptrdiff_t toff;                            // per-k-tile: tap * tap_stride - tap * C
    struct Row { const float* base; };
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void tile(int k0) {
        const int tap = k0 / C;
        toff = (ptrdiff_t)tap * (ptrdiff_t)tap_stride - (ptrdiff_t)tap * C;
    }
    __device__ __forceinline__ Row row(int r) const { return {p + (size_t)min(r, ext - 1) * ld}; }
    __device__ __forceinline__ f32x4 load(const Row& R, int k) const { return *reinterpret_cast<const f32x4*>(R.base + toff + k); }
    __device__ __forceinline__ bool keep(const Row&, int) const { return true; }
};

// MC loaders: the reduction index is the memory row, tile columns are contiguous (ext % 4 == 0).
struct McRows {
    const float* p; int ld; int ext; int K; const int* row_off_dyn;
    struct Row {};
    __device__ __forceinline__ void prepare(int) { if (row_off_dyn != nullptr) p += (size_t)(*row_off_dyn) * ld; }
    __device__ __forceinline__ void tile(int) {}
    __device__ __forceinline__ f32x4 load(int kk, int c, bool& keep) const {
        keep = kk < K;
        return *reinterpret_cast<const f32x4*>(p + (size_t)min(kk, K - 1) * ld + min(c, ext - 4));
    }
};

struct McShiftRows {       // row kk + (tap-1) of p, valid when the plan says row kk has that neighbour
    const float* p; int ld; int ext; int K; const int* rowinfo; int s;
    struct Row {};
    __device__ __forceinline__ void prepare(int tap) { s = tap - 1; }
    __device__ __forceinline__ void tile(int) {}
    __device__ __forceinline__ f32x4 load(int kk, int c, bool& keep) const {
        const int kc = min(kk, K - 1);
        const int f = rowinfo[kc];
        const bool ok = (s == 0) || (s < 0 ? (f & RI_LEFT) != 0 : (f & RI_RIGHT) != 0);
        keep = ok && kk < K;
        return *reinterpret_cast<const f32x4*>(p + (ptrdiff_t)(kc + (ok ? s : 0)) * ld + min(c, ext - 4));
    }
};

// loaders whose element is a combination of TWO loads (gemm_wino.hpp) declare kDual = true and provide
// load2(kk, c, v1, keep1, v2, keep2) + combine(v1, keep1, v2, keep2); MC operands only
template <class L, class = void> struct IsDual : std::false_type {};
template <class L> struct IsDual<L, std::void_t<decltype(L::kDual)>> : std::bool_constant<L::kDual> {};


// ------------------------------------------------------------------ split-bf16 operands (opt-in product mode, never the default)
// x = hi + lo + O(2^-18 |x|) with hi = bf16(x), lo = bf16(x - hi); a product a.b is taken as lo_a hi_b + hi_a lo_b + hi_a hi_b on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate): three matrix instructions of 32 cycles per 16 k against eight of 64 for exact f32,
// relative error ~2^-17 per term.  Loaders, tile configurations and epilogues are shared with the exact kernel; what changes is the
// LDS image: BOTH operand kinds are held row-major [row][BK + 4 slots] with the 16-byte chunk of 4 consecutive k of a row packed as
// [h0 h1][l0 l1][h2 h3][l2 l3], so the two ds_read_b128 of a lane for a 16-k step ARE its hi / lo operand registers (no unpacking,
// no per-element reads).  A K-contiguous operand packs the f32x4 it loaded; an M-contiguous one (reduction index = memory row) loads
// CONSECUTIVE k rows per thread (kN x 4 block), transposes it in registers and writes one chunk (or half chunk) per matrix row.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned split_pk(float a, float b) {
    bf16x2_t t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ u32x2 split_pair(float a, float b) {          // [ha hb][la lb]
    const unsigned h = split_pk(a, b);
    return u32x2{h, split_pk(a - __builtin_bit_cast(float, h << 16), b - __builtin_bit_cast(float, h & 0xffff0000u))};
}
__device__ __forceinline__ f32x4 split_pack_kc(f32x4 v) {
    const u32x2 p0 = split_pair(v[0], v[1]), p1 = split_pair(v[2], v[3]);
    const u32x4 w = {p0[0], p0[1], p1[0], p1[1]};
    return __builtin_bit_cast(f32x4, w);
}

// commit of an M-contiguous operand in split mode: thread (rg = tid % PER, kq = tid / PER) holds k rows kN * kq + j (j < kN) of the
// four matrix rows 4 rg + i; row i's kN values become kN / 4 packed chunks (kN = 2: half a chunk) of the row-major image
template <int KN, int PER>
__device__ __forceinline__ void split_commit_mc(float* T_, const f32x4 (&sv)[KN], const bool (&pv)[KN], int tid) {
    // the lanes of a write share kq and step through rg: rows 16 apart would land on the same banks (row stride 36 slots), so the
    // chunk index is XORed with bits 4-6 of the row -- uniform over the 16-lane groups of the fragment reads, which undo it
    const int rg = tid % PER, kq = tid / PER;
    const int g = (rg >> 2) & 7;                     // = ((4 rg + i) >> 4) & 7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* row = T_ + (4 * rg + i) * KC_LD;
        if constexpr (KN == 2) {
            *reinterpret_cast<u32x2*>(row + 4 * ((kq >> 1) ^ g) + 2 * (kq & 1)) = split_pair(pv[0] ? sv[0][i] : 0.f, pv[1] ? sv[1][i] : 0.f);
        } else {
#pragma unroll
            for (int c = 0; c < KN / 4; ++c) {
                const f32x4 v = {pv[4 * c] ? sv[4 * c][i] : 0.f, pv[4 * c + 1] ? sv[4 * c + 1][i] : 0.f,
                                 pv[4 * c + 2] ? sv[4 * c + 2][i] : 0.f, pv[4 * c + 3] ? sv[4 * c + 3][i] : 0.f};
                *reinterpret_cast<f32x4*>(row + 4 * ((kq * (KN / 4) + c) ^ g)) = split_pack_kc(v);
            }
        }
    }
}

// ------------------------------------------------------------------ the kernel
template <int BM, int BN, int WM, int WN, bool STAGGER = false>
struct TileCfg {
    static constexpr int kBM = BM, kBN = BN, kWM = WM, kWN = WN;
    static constexpr bool kStagger = STAGGER;     // upper half of the waves writes the next tile BEFORE its MFMAs
    static constexpr int kTM = BM / (WM * 32), kTN = BN / (WN * 32);
    static constexpr int kThreads = WM * WN * 64;
    static_assert(WM * WN == 4 || WM * WN == 8, "256- or 512-thread workgroups");
    static_assert(kTM >= 1 && kTN >= 1, "wave tile");
};

template <bool MC, int ROWS, int NT = 256>
struct Stage {             // global -> registers -> LDS staging of one operand tile
    static constexpr int kN = ROWS * 8 / NT;          // float4 per thread (ROWS x BK floats over NT threads)
    static_assert(kN >= 1, "tile too small for the workgroup");
    static constexpr int kLdsFloats = MC ? BK * ROWS : ROWS * KC_LD;
};

struct GemmDims {
    int M, N, K;            // static bounds
    const int* m_dyn;       // optional device scalar overriding M (NT/NN) ...
    const int* k_dyn;       // ... or K (TN: reduction over the dynamic row count)
    int split_k;            // TN only: gridDim.z / taps
    int abl = 0;            // tuning build only (LEGO_DMA_ABL): ablation bits of the LDS-DMA row-strip kernel, see gemm_dma.hpp
};

template <class Cfg, bool A_MC, bool B_MC, class ALoad, class BLoad, class Epi, bool SPLIT = false>
__global__ __launch_bounds__(Cfg::kThreads) void gemm_kernel(GemmDims dims, ALoad la, BLoad lb, Epi epi) {
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, TM = Cfg::kTM, TN = Cfg::kTN, NT = Cfg::kThreads, RS = NT / 8;
    using SA = Stage<A_MC && !SPLIT, BM, NT>;        // split mode: every image is row-major (see above)
    using SB = Stage<B_MC && !SPLIT, BN, NT>;
    static_assert(!SPLIT || !((A_MC && IsDual<ALoad>::value) || (B_MC && IsDual<BLoad>::value)), "no split form of the pair loaders");
    static_assert(!SPLIT || ((!A_MC || SA::kN % 2 == 0) && (!B_MC || SB::kN % 2 == 0)), "split mode: an M-contiguous operand needs >= 2 k rows per thread");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * SA::kLdsFloats;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / Cfg::kWN, wn = wave % Cfg::kWN;

    int M = dims.M, K = dims.K;
    if (dims.m_dyn != nullptr) M = min(M, *dims.m_dyn);
    if (dims.k_dyn != nullptr) K = min(K, *dims.k_dyn);
    const int N = dims.N;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= M || n0 >= N) return;

    // reduction range of this block (split-K along gridDim.z for the TN weight-gradient GEMMs)
    int kbeg = 0, kend = K;
    if (dims.split_k > 1) {
        const int z = blockIdx.z % dims.split_k;
        int chunk = (K + dims.split_k - 1) / dims.split_k;
        chunk = (chunk + BK - 1) / BK * BK;
        kbeg = z * chunk;
        kend = min(K, kbeg + chunk);
        if (kbeg >= kend) return;
    }
    const int tap = blockIdx.z / max(dims.split_k, 1);
    epi.setup(M, N, tap);
    if constexpr (!A_MC) la.ext = M;
    la.K = kend;
    lb.K = kend;
    la.prepare(tap);
    lb.prepare(tap);

    // per-thread row state for KC operands (constant over the k loop)
    typename ALoad::Row ra[SA::kN];
    typename BLoad::Row rb[SB::kN];
    if constexpr (!A_MC) {
#pragma unroll
        for (int j = 0; j < SA::kN; ++j) ra[j] = la.row(m0 + (tid >> 3) + RS * j);
    }
    if constexpr (!B_MC) {
#pragma unroll
        for (int j = 0; j < SB::kN; ++j) rb[j] = lb.row(n0 + (tid >> 3) + RS * j);
    }

    constexpr bool A2 = A_MC && IsDual<ALoad>::value, B2 = B_MC && IsDual<BLoad>::value;
    f32x4 sa[SA::kN], sb[SB::kN], sa2[A2 ? SA::kN : 1], sb2[B2 ? SB::kN : 1];
    bool pa[SA::kN], pb[SB::kN], pa2[A2 ? SA::kN : 1], pb2[B2 ? SB::kN : 1];
    auto fetch = [&](int k0) {
        la.tile(k0);
        lb.tile(k0);
        if constexpr (A_MC) {
            constexpr int PER = BM / 4, STEP = NT / PER;
#pragma unroll
            for (int j = 0; j < SA::kN; ++j) {
                if constexpr (A2) la.load2(k0 + tid / PER + STEP * j, m0 + (tid % PER) * 4, sa[j], pa[j], sa2[j], pa2[j]);
                else sa[j] = la.load(SPLIT ? k0 + (tid / PER) * SA::kN + j : k0 + tid / PER + STEP * j, m0 + (tid % PER) * 4, pa[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < SA::kN; ++j) { sa[j] = la.load(ra[j], k0 + (tid & 7) * 4); pa[j] = la.keep(ra[j], k0 + (tid & 7) * 4); }
        }
        if constexpr (B_MC) {
            constexpr int PER = BN / 4, STEP = NT / PER;
#pragma unroll
            for (int j = 0; j < SB::kN; ++j) {
                if constexpr (B2) lb.load2(k0 + tid / PER + STEP * j, n0 + (tid % PER) * 4, sb[j], pb[j], sb2[j], pb2[j]);
                else sb[j] = lb.load(SPLIT ? k0 + (tid / PER) * SB::kN + j : k0 + tid / PER + STEP * j, n0 + (tid % PER) * 4, pb[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < SB::kN; ++j) { sb[j] = lb.load(rb[j], k0 + (tid & 7) * 4); pb[j] = lb.keep(rb[j], k0 + (tid & 7) * 4); }
        }
    };
    auto commit = [&](float* A_, float* B_) {
        if constexpr (A_MC && SPLIT) {
            split_commit_mc<SA::kN, BM / 4>(A_, sa, pa, tid);
        } else if constexpr (A_MC) {
            constexpr int PER = BM / 4, STEP = NT / PER;
#pragma unroll
            for (int j = 0; j < SA::kN; ++j) {
                f32x4 v;
                if constexpr (A2) v = la.combine(sa[j], pa[j], sa2[j], pa2[j]); else v = zero_unless(pa[j], sa[j]);
                *reinterpret_cast<f32x4*>(A_ + (tid / PER + STEP * j) * BM + (tid % PER) * 4) = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SA::kN; ++j) {
                f32x4 v = zero_unless(pa[j], sa[j]);
                if constexpr (SPLIT) v = split_pack_kc(v);
                *reinterpret_cast<f32x4*>(A_ + ((tid >> 3) + RS * j) * KC_LD + (tid & 7) * 4) = v;
            }
        }
        if constexpr (B_MC && SPLIT) {
            split_commit_mc<SB::kN, BN / 4>(B_, sb, pb, tid);
        } else if constexpr (B_MC) {
            constexpr int PER = BN / 4, STEP = NT / PER;
#pragma unroll
            for (int j = 0; j < SB::kN; ++j) {
                f32x4 v;
                if constexpr (B2) v = lb.combine(sb[j], pb[j], sb2[j], pb2[j]); else v = zero_unless(pb[j], sb[j]);
                *reinterpret_cast<f32x4*>(B_ + (tid / PER + STEP * j) * BN + (tid % PER) * 4) = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SB::kN; ++j) {
                f32x4 v = zero_unless(pb[j], sb[j]);
                if constexpr (SPLIT) v = split_pack_kc(v);
                *reinterpret_cast<f32x4*>(B_ + ((tid >> 3) + RS * j) * KC_LD + (tid & 7) * 4) = v;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

    // stagger (MI355X_MICROARCH.md, two waves per SIMD running the same program): the second half of the waves
    // waits for its prefetch and writes the next LDS tile BEFORE its MFMAs, the first half after them, so on every
    // SIMD one wave is in its matrix phase while its partner is in its memory phase
    const bool early = Cfg::kStagger && __builtin_amdgcn_readfirstlane(wave) >= (NT / 128);
    fetch(kbeg);
    commit(As0, Bs0);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) fetch(k0 + BK);
        if (more && early) commit(As0 + (buf ^ 1) * SA::kLdsFloats, Bs0 + (buf ^ 1) * SB::kLdsFloats);
        const float* A_ = As0 + buf * SA::kLdsFloats;
        const float* B_ = Bs0 + buf * SB::kLdsFloats;
        if constexpr (SPLIT) {
#pragma unroll
            for (int s2 = 0; s2 < BK / 16; ++s2) {
                u32x4 ah[TM], al[TM], bh[TN], bl[TN];
                auto frag = [&](const float* T_, int idx, bool swz, u32x4& hi, u32x4& lo) {
                    const int g = swz ? (idx >> 4) & 7 : 0;         // split_commit_mc's chunk swizzle
                    const u32x4 c0 = *reinterpret_cast<const u32x4*>(T_ + idx * KC_LD + 4 * ((4 * s2 + lh) ^ g));
                    const u32x4 c1 = *reinterpret_cast<const u32x4*>(T_ + idx * KC_LD + 4 * ((4 * s2 + 2 + lh) ^ g));
                    hi = u32x4{c0[0], c0[2], c1[0], c1[2]};
                    lo = u32x4{c0[1], c0[3], c1[1], c1[3]};
                };
#pragma unroll
                for (int a = 0; a < TM; ++a) frag(A_, (wm * TM + a) * 32 + li, A_MC, ah[a], al[a]);
#pragma unroll
                for (int b = 0; b < TN; ++b) frag(B_, (wn * TN + b) * 32 + li, B_MC, bh[b], bl[b]);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[a]), __builtin_bit_cast(bf16x8, bh[b]), acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[a]), __builtin_bit_cast(bf16x8, bl[b]), acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[a]), __builtin_bit_cast(bf16x8, bh[b]), acc[a][b], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int row = (wm * TM + a) * 32 + li;
                if constexpr (A_MC) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[a][j] = A_[(8 * q + 4 * lh + j) * BM + row];
                } else {
                    fa[a] = *reinterpret_cast<const f32x4*>(A_ + row * KC_LD + 8 * q + 4 * lh);
                }
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = (wn * TN + b) * 32 + li;
                if constexpr (B_MC) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[b][j] = B_[(8 * q + 4 * lh + j) * BN + col];
                } else {
                    fb[b] = *reinterpret_cast<const f32x4*>(B_ + col * KC_LD + 8 * q + 4 * lh);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
        }
        }
        if (more && !early) commit(As0 + (buf ^ 1) * SA::kLdsFloats, Bs0 + (buf ^ 1) * SB::kLdsFloats);
        __syncthreads();
        buf ^= 1;
    }

    // epilogue: lane holds column (li) x rows {(v&3) + 8*(v>>2) + 4*lh} of each 32x32 sub-tile
    epi.template run<TM, TN>(acc, m0 + wm * TM * 32, n0 + wn * TN * 32, li, lh);
}

template <class Cfg, bool A_MC, bool B_MC, bool SPLIT = false>
constexpr size_t gemm_lds_bytes() {
    return 2 * (Stage<A_MC && !SPLIT, Cfg::kBM, Cfg::kThreads>::kLdsFloats + Stage<B_MC && !SPLIT, Cfg::kBN, Cfg::kThreads>::kLdsFloats) * sizeof(float);
}

}  // namespace lego
