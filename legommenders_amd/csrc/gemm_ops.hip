// GEMM-shaped entry points of liblego_hip.so: every dense product of the NAML / NRMS forward and
// backward pass runs on the fp32 MFMA core in gemm_core.hpp (exact f32; the path's parity bar is
// 1e-3 on fp32 logits, so no reduced-precision inputs are used) -- unless the caller opts into the
// split-bf16 product mode (lego_set_product_mode below; never the default).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/lego_hip.h"
#include "gemm_strip.hpp"
#include <atomic>
#include "gemm_oneshot.hpp"
#include "gemm_epi.hpp"

namespace lego {

static thread_local char g_err[512] = "";
int set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error("%s: %s", what, hipGetErrorString(e));
    return 0;
}
const char* last_error() { return g_err; }

// round 5: gemm_tnd.hpp (tnd_ops.hip) -- plain-row weight gradients with the operand fragments straight from global memory
int tndp_slabs(int Dout, int Din, int P_cap);
int launch_tndp(const float* gy, int ldg, const float* h, int ldh, const int* pair_info, int P_cap, const int* P_dyn, float* du,
                int Dout, int Din, hipStream_t st, const char* what);
bool tnd_ok(int M, int N, int K_cap, int lda, int ldb, int ldc);
int launch_tnd(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int M, int N, int K_cap, const int* k_dyn,
               const int* a_row_off, const int* b_row_off, hipStream_t st, const char* what);


}  // namespace lego
#include "wino_common.hpp"
#include "gemm_tn.hpp"
#include "gemm_dma.hpp"
namespace lego {

static Epi make_epi(float* C, int ldc) {
    Epi e;
    e.C = C; e.ldc = ldc; e.bias = nullptr; e.act = 0; e.rowinfo = nullptr; e.C2 = nullptr; e.ldc2 = 0;
    e.drop = make_dropout(nullptr); e.drop_cols = 1;
    e.relu_ref = nullptr; e.ld_ref = 0; e.relu_scale = 1.f; e.colsum = nullptr;
    e.tap_stride = 0; e.row_off_dyn = nullptr; e.M = e.N = e.row_off = 0;
    e.rows_form = 1;
    return e;
}
static void set_drop(Epi& e, const lego_dropout* d, int cols) {
    if (d != nullptr && d->p > 0.f) {
        e.drop = make_dropout(d);
        e.drop_cols = cols;
    }
}

template <class Cfg, bool A_MC, bool B_MC, class EK, bool SPLIT = false, class AL, class BL>
static int launch(const GemmDims& d, const AL& a, const BL& b, const Epi& e0, int tiles_m, int tiles_n, int gz,
                  hipStream_t st, const char* what) {
    EK e;
    static_cast<EpiArgs&>(e) = e0;
    auto k = gemm_kernel<Cfg, A_MC, B_MC, AL, BL, EK, SPLIT>;
    constexpr size_t lds = gemm_lds_bytes<Cfg, A_MC, B_MC, SPLIT>();
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(tiles_m, tiles_n, gz), dim3(Cfg::kThreads), lds, st, d, a, b, e);
    return check_launch(what);
}

using C128x128 = TileCfg<128, 128, 2, 4, true>;   // 8 waves of 64 x 32, staggered halves: best of tools/gemm_variants.py at the path's row counts
using C128x64 = TileCfg<128, 64, 4, 1>;
using C64x128 = TileCfg<64, 128, 1, 4>;
using C64x64 = TileCfg<64, 64, 2, 2>;

// NT / NN: rows x N output, BM = 128, BN by N
using EpiPlain = EpiT<false, false, false, false>;
using EpiLive = EpiT<true, false, false, false>;
using EpiAccum = EpiT<false, true, false, false>;
using EpiAccumRelu = EpiT<false, true, true, false>;
using EpiAtomic = EpiT<false, false, false, true>;
using EpiRef = EpiT<false, false, true, false>;       // a reference tensor without accumulation (GELU' in the FFN data gradient)

static int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// one block per CU, each a strip of ceil(M / #CU) rows x all N <= 256 columns (gemm_strip.hpp)
template <bool B_MC, class EK, class AL, class BL>
static int launch_strip(const GemmDims& d, const AL& a, const BL& b, const Epi& e0, hipStream_t st, const char* what) {
    EK e;
    static_cast<EpiArgs&>(e) = e0;
    auto k = strip_kernel<B_MC, AL, BL, EK>;
    constexpr size_t lds = strip_lds_bytes<B_MC>();
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int n_panels = (d.N + STRIP_BN - 1) / STRIP_BN;
    hipLaunchKernelGGL(k, dim3(num_cus() / n_panels * n_panels), dim3(STRIP_THREADS), lds, st, d, a, b, e);
    return check_launch(what);
}

// the same split with LDS-DMA operand staging (gemm_dma.hpp): plain-row operands only
template <bool B_MC, class EK, class BL, int NW = 8>
static int launch_dma_strip(const GemmDims& d, const KcRows& a, const BL& b, const Epi& e0, hipStream_t st, const char* what) {
    LEGO_REQUIRE(d.K >= 4 && d.K % 4 == 0 && a.ld % 4 == 0 && b.ld % 4 == 0,
                 "%s: the LDS-DMA row-strip kernel needs K %% 4 == 0 and 16-byte-aligned rows (K=%d, lda=%d, ldb=%d)", what, d.K, a.ld, b.ld);
    EK e;
    static_cast<EpiArgs&>(e) = e0;
#ifdef LEGO_TUNING_HOOKS
    GemmDims dd = d;
    { static int abl = -1; if (abl < 0) { const char* v = getenv("LEGO_DMA_ABL"); abl = v != nullptr ? atoi(v) : 0; } dd.abl = abl; }
    const GemmDims& d_ = dd;
#else
    const GemmDims& d_ = d;
#endif
    auto k = dma_strip_kernel<B_MC, NW, BL, EK>;
    constexpr size_t lds = dma_lds_bytes<B_MC>();
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int n_panels = (d.N + STRIP_BN - 1) / STRIP_BN;
    hipLaunchKernelGGL(k, dim3(num_cus() / n_panels * n_panels), dim3(NW * 64), lds, st, d_, a, b, e);
    return check_launch(what);
}

// small latency-bound products (gemm_oneshot.hpp: light_kernel): one wave per 16 x 32 outputs, every load of the wave in flight at once
template <bool B_MC, class EK, class AL, class BL>
static int launch_light(const GemmDims& d, const AL& a, const BL& b, const Epi& e0, hipStream_t st, const char* what) {
    EK e;
    static_cast<EpiArgs&>(e) = e0;
    // k groups of 16 in flight per wave: 4 (84-92 VGPRs; a K = 256 product is four dependent round trips) against 2 (56-68 VGPRs, eight):
    // NAML 0.6297 -> 0.6244 ms, NRMS 1.0363 -> 1.0328 ms (three alternating same-box runs each; 8 groups, 136 VGPRs: no further gain)
    hipLaunchKernelGGL((light_kernel<B_MC, AL, BL, EK, 4>), dim3((d.M + 15) / 16, (d.N + 31) / 32), dim3(64), 0, st, d, a, b, e);
    return check_launch(what);
}

// ---- product mode.  0 (default): exact f32 on v_mfma_f32_32x32x2 / 16x16x4 -- the path's parity contract.  1: split-bf16 (bf16 x 3,
// fp32 accumulate; gemm_core.hpp) for the large products -- an OPT-IN throughput mode that is not bit-compatible with the reference
// (relative error ~1e-5 per product); lego_set_product_mode() or LEGO_SPLIT_BF16=1 at load.  Small latency-bound products (user side,
// fold levels) stay exact in both modes.
// (process-wide, read at every launch, written by lego_set_product_mode from any thread: an atomic.  Engines cache which conv entry points
// they use when they are built -- engine.py refuses to step an engine built in the other mode)
static std::atomic<int> g_product_mode{-1};
static int product_mode() {
    int m = g_product_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("LEGO_SPLIT_BF16");
        m = (e != nullptr && e[0] == '1') ? 1 : 0;
        g_product_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
constexpr int SPLIT_MIN_ROWS = 2048;            // below this the products are latency-bound: nothing to gain, keep them exact

using C128x128s = TileCfg<128, 128, 4, 2, true>;   // split mode: 8 waves of 32 x 64
using C128x256s = TileCfg<128, 256, 2, 4>;

// round 5: plain-row NT / NN products on rows2_kernel (gemm_rows2.hpp, rows2_ops.hip)
bool rows2_ok(const float* x, int ldx, const float* w, int ldw, int M_cap, int N, int K, const EpiArgs& e, bool b_mc, bool accum, bool reluref);
int launch_rows2(const float* x, int ldx, const float* w, int ldw, int M_cap, const int* M_dyn, int N, int K, const EpiArgs& e, bool b_mc, bool accum,
                 bool reluref, hipStream_t st, const char* what);

template <bool B_MC, class EK, class AL, class BL>
static int launch_rows(const GemmDims& d, const AL& a, const BL& b, const Epi& e, hipStream_t st, const char* what) {
    const int tm = (d.M + 127) / 128;
    if constexpr (std::is_same<AL, KcRows>::value && (std::is_same<BL, KcRows>::value || std::is_same<BL, McRows>::value) &&
                  (std::is_same<EK, EpiPlain>::value || std::is_same<EK, EpiAccum>::value || std::is_same<EK, EpiAccumRelu>::value)) {
        constexpr bool acc_ = !std::is_same<EK, EpiPlain>::value, ref_ = std::is_same<EK, EpiAccumRelu>::value;
        static_assert(B_MC == std::is_same<BL, McRows>::value, "an M-contiguous B operand is the NN form");
        if (product_mode() == 0 && a.row_off_dyn == nullptr && b.row_off_dyn == nullptr && e.row_off_dyn == nullptr && d.k_dyn == nullptr &&
            rows2_ok(a.p, a.ld, b.p, b.ld, d.M, d.N, d.K, e, B_MC, acc_, ref_))
            return launch_rows2(a.p, a.ld, b.p, b.ld, d.M, d.m_dyn, d.N, d.K, e, B_MC, acc_, ref_, st, what);
    }
    if (product_mode() == 1 && d.M >= SPLIT_MIN_ROWS) {
        // tools/gemm_variants.py 20-26 (round 4): 128 x 128 on 4 x 2 staggered waves at the path's widths (214 / 154 TFLOP/s-equivalent
        // at K = 768 / 256 over 26 k rows, against 95 / 83 exact), 128 x 256 for the BERT widths (253-266 against 113-123)
        if (d.N >= 512) return launch<C128x256s, false, B_MC, EK, true>(d, a, b, e, tm, (d.N + 255) / 256, 1, st, what);
        if (d.N > 64) return launch<C128x128s, false, B_MC, EK, true>(d, a, b, e, tm, (d.N + 127) / 128, 1, st, what);
        return launch<C128x64, false, B_MC, EK, true>(d, a, b, e, tm, (d.N + 63) / 64, 1, st, what);
    }
    if (d.M >= 32 * num_cus() && d.N <= 4 * STRIP_BN) {
        if constexpr (std::is_same<AL, KcRows>::value && (std::is_same<BL, KcRows>::value || std::is_same<BL, McRows>::value))
            // the DMA staging moves 16-byte chunks clamped to K - 4 and zeroes tails at 4-element granularity: K % 4 == 0 and
            // 16-byte-aligned rows on both sides, else the register-staged strip kernel
            if (d.K >= 4 && d.K % 4 == 0 && a.ld % 4 == 0 && b.ld % 4 == 0 && (!B_MC || d.N % 4 == 0))
                return launch_dma_strip<B_MC, EK>(d, a, b, e, st, what);
        return launch_strip<B_MC, EK>(d, a, b, e, st, what);
    }
    if constexpr (std::is_same<AL, KcRows>::value)
        if (d.K <= 4 * ONE_KMAX && d.K % 4 == 0 &&
            ((d.M + ONE_BM - 1) / ONE_BM) * ((d.N + ONE_BN - 1) / ONE_BN) <= 3 * num_cus()) { // few rounds of whole-CU blocks by
                                                                                              // CAPACITY: the user side fills 40 %
            return launch_light<B_MC, EK>(d, a, b, e, st, what);
        }
    if (tm * ((d.N + 127) / 128) < 128) {         // few row tiles (user / category side): 64-row tiles fill more CUs
        if (d.N > 64) return launch<C64x128, false, B_MC, EK>(d, a, b, e, (d.M + 63) / 64, (d.N + 127) / 128, 1, st, what);
        return launch<C64x64, false, B_MC, EK>(d, a, b, e, (d.M + 63) / 64, (d.N + 63) / 64, 1, st, what);
    }
    if (d.N > 64) return launch<C128x128, false, B_MC, EK>(d, a, b, e, tm, (d.N + 127) / 128, 1, st, what);
    return launch<C128x64, false, B_MC, EK>(d, a, b, e, tm, (d.N + 63) / 64, 1, st, what);
}
// TN (weight gradients): small [M,N] output, reduction over the (ragged) rows split along gridDim.z.
//   * long reductions (>= TN_LONG rows of capacity): tn_kernel of gemm_tn.hpp, one workgroup of 128 x 128 per CU, loads two
//     k tiles ahead, split chosen so that the grid is one round of CUs;
//   * short ones (user side, category column): 64 x 64 tiles of the generic tile kernel, ~1024 workgroups (more tiles =
//     more CUs busy for a reduction of a few thousand rows).
constexpr int TN_LONG = 8192;

static int pick_split_small(int rows_cap, int M, int N, int small_target = 4096) {
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
    // Outputs of several hundred tiles (BERT's 768 x 3072 FFN weights: 576) take ~2 300 workgroups --
    // 1024 / 576 floors to 1 and leaves 44 % of the workgroup slots empty; more, shorter workgroups also hide each other's load
    // latency
    // measured on [3072 x 768] over 29.6 k rows (tools/bert_shapes_bench.py): split 1 / 2 / 3 / 4 / 8 = 82 / 98 / 104 / 107 / 105 TFLOP/s
    // small outputs: ~4096 SHORT workgroups rather than ~1024 long ones.  Alone the launch is no faster (more atomics), but these products
    // run on the side stream next to the main chain's latency-bound kernels, which can only start on a CU when a workgroup retires:
    // NRMS step 1.035 -> 1.025 ms with the in-projection weight gradient (48 tiles) at 80-128 splits instead of 16, NAML unchanged
    // (tools/r04_tnsplit.sh, two alternating same-box runs)
    int s = tiles > 256 ? (2304 + tiles - 1) / tiles : small_target / tiles;
    const int max_s = (rows_cap + 127) / 128;       // at least 128 reduction rows per block
    if (s > max_s) s = max_s;
    if (s >= 16) s &= ~7;                           // a multiple of 8: the k splits can then be dealt to the 8 XCDs (gemm_tn.hpp)
    return s < 1 ? 1 : s;
}

static int tn_split(int M, int N, int K_cap, int taps) {     // k splits of the long-reduction kernel: one round of CUs
    const int tm = (M + TN_BM - 1) / TN_BM, tn = (N + TN_BN - 1) / TN_BN;
    int split = num_cus() / (tm * tn * taps);
    const int max_s = (K_cap + 255) / 256;          // at least 8 k tiles per workgroup
    if (split > max_s) split = max_s;
    return split < 1 ? 1 : split;
}

// SLAB: the partial tile of every k split goes to its own [taps][M][N] slab with plain stores (e.C = slab 0) instead of fp32
// atomics into one buffer; the caller folds the slabs (conv: lego_conv3_wino_unpack_add)
template <bool SLAB, class AL, class BL>
static int launch_tn_long(int M, int N, int K_cap, const int* k_dyn, const AL& a, const BL& b, const Epi& e, int taps, hipStream_t st,
                          const char* what) {
    const int tm = (M + TN_BM - 1) / TN_BM, tn = (N + TN_BN - 1) / TN_BN;
    const int split = tn_split(M, N, K_cap, taps);
    const int deal = split % 8 == 0;
    TnDims d{M, N, K_cap, k_dyn, split, taps, (size_t)taps * e.tap_stride, tm, tn, deal};
    auto k = tn_kernel<AL, BL, SLAB>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tn_lds_bytes());
        attr_done = true;
    }
    hipLaunchKernelGGL(k, deal ? dim3(tm * tn * taps * split) : dim3(tm, tn, taps * split), dim3(TN_THREADS), tn_lds_bytes(), st, d, a, b, e);
    return check_launch(what);
}

// plain row operands: 64 x 64 tiles, ~1024 workgroups, fp32 atomics.  Measured against the 128 x 128 kernel above on the
// path's own shapes (profiles/r02_tn_modes.md): 29.5 us per launch against 56.9 -- four small workgroups per CU hide each
// other's barriers and load latency better than one large one, and these operands need no index look-up
template <class AL, class BL>
static int launch_tn(int M, int N, int K_cap, const int* k_dyn, const AL& a, const BL& b, const Epi& e, int taps, hipStream_t st,
                     const char* what) {
    const int split = pick_split_small(K_cap, M, N);
    if (product_mode() == 1 && K_cap >= SPLIT_MIN_ROWS) {
        if constexpr (!IsDual<AL>::value && !IsDual<BL>::value) {
            // tools/tn_variants.py 20-22 (round 4).  Outputs of >= 36 tiles of 128 x 128 (the BERT block weights): 128 x 128 on 8 waves,
            // ~1152 workgroups -- 212-250 TFLOP/s-equivalent over 29.6 k rows against 110-115 exact; the path's 256-wide weights: 64 x 64
            // tiles (the 128-wide ones leave most CUs without a workgroup), 90-108 against 78
            const int t128 = ((M + 127) / 128) * ((N + 127) / 128);
            if (t128 * taps >= 36) {
                int s128 = 1152 / (t128 * taps);
                const int max_s = (K_cap + 255) / 256;
                s128 = s128 < 1 ? 1 : (s128 > max_s ? max_s : s128);
                GemmDims d{M, N, K_cap, nullptr, k_dyn, s128};
                return launch<TileCfg<128, 128, 4, 2>, true, true, EpiAtomic, true>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, taps * s128, st, what);
            }
            // (~1024 workgroups here: the split kernel's 64 x 64 tiles are L2-bound, and four times the workgroups cost it 10 % of the
            // step -- NAML 122.5 k -> 109.8 k impressions/s with the exact kernels' ~4096)
            GemmDims d{M, N, K_cap, nullptr, k_dyn, pick_split_small(K_cap, M, N, 1024)};
            return launch<C64x64, true, true, EpiAtomic, true>(d, a, b, e, (M + 63) / 64, (N + 63) / 64, taps * d.split_k, st, what);
        }
    }
    if constexpr (!IsDual<AL>::value && !IsDual<BL>::value) {
        // 64 x 64 tiles, loads two k tiles ahead (tn_kernel), four workgroups per CU: 27.7 us per launch against 29.7 for the
        // generic tile kernel (one tile ahead) on the path's shapes
        const int tm = (M + 63) / 64, tn = (N + 63) / 64;
        const int deal = split % 8 == 0;
        TnDims d{M, N, K_cap, k_dyn, split, taps, 0, tm, tn, deal};
        if (K_cap >= TN_LONG) d.min_chunk = 256;       // (the de-duplicated projection: 4.6 k distinct tokens under a 105 k capacity)
        auto k = tn_kernel<AL, BL, false, 2, 2, 1>;
        constexpr size_t lds = tn_lds_bytes(TN_BM_S, TN_BN_S, false);
        hipLaunchKernelGGL(k, deal ? dim3(tm * tn * taps * split) : dim3(tm, tn, taps * split), dim3(TN_THREADS_S), lds, st, d, a, b, e);
        return check_launch(what);
    }
    GemmDims d{M, N, K_cap, nullptr, k_dyn, split};
    return launch<C64x64, true, true, EpiAtomic>(d, a, b, e, (M + 63) / 64, (N + 63) / 64, taps * d.split_k, st, what);
}

}  // namespace lego

using namespace lego;

extern "C" const char* lego_last_error(void) { return lego::last_error(); }
extern "C" int lego_abi_version(void) { return LEGO_ABI_VERSION; }
extern "C" int lego_set_product_mode(int mode) {
    LEGO_REQUIRE(mode == 0 || mode == 1, "lego_set_product_mode: mode=%d (0 = exact f32, 1 = split-bf16)", mode);
    lego::g_product_mode.store(mode, std::memory_order_relaxed);
    return 0;
}
extern "C" int lego_get_product_mode(void) { return lego::product_mode(); }

#define CHECK4(x) LEGO_REQUIRE(((x) & 3) == 0, "%s: " #x "=%d must be a multiple of 4", __func__, (int)(x))

extern "C" int lego_linear_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias,
                               float* out, int ldo, int M_cap, const int32_t* M_dyn, int N, int K, int act,
                               const int32_t* rowinfo, const lego_dropout* drop,
                               const int32_t* x_row_off_dyn, const int32_t* out_row_off_dyn, void* stream) {
    CHECK4(ldx); CHECK4(ldw); CHECK4(K);
    if (M_cap <= 0) return 0;
    GemmDims d{M_cap, N, K, M_dyn, nullptr, 1};
    KcRows a{x, ldx, M_cap, K, x_row_off_dyn};
    KcRows b{W, ldw, N, K, nullptr};
    Epi e = make_epi(out, ldo);
    e.bias = bias; e.act = act; e.rowinfo = rowinfo; e.row_off_dyn = out_row_off_dyn;
    set_drop(e, drop, N);
    if (rowinfo != nullptr) return launch_rows<false, EpiLive>(d, a, b, e, (hipStream_t)stream, "lego_linear_fwd");
    return launch_rows<false, EpiPlain>(d, a, b, e, (hipStream_t)stream, "lego_linear_fwd");
}

extern "C" int lego_linear_bwd_data(const float* g, int ldg, const float* W, int ldw, float* dx, int lddx,
                                    int M_cap, const int32_t* M_dyn, int N, int K, int accumulate,
                                    const float* relu_ref, int ld_ref, float relu_scale,
                                    const int32_t* rowinfo, const lego_dropout* drop, float* colsum,
                                    const int32_t* g_row_off_dyn, const int32_t* dx_row_off_dyn, void* stream) {
    CHECK4(ldg); CHECK4(ldw); CHECK4(N); CHECK4(K);
    if (M_cap <= 0) return 0;
    // dx[M,K] (+)= g[M,N] . W[N,K]: NN product, reduction over N; W rows are the reduction index (MC)
    GemmDims d{M_cap, /*N=*/K, /*K=*/N, M_dyn, nullptr, 1};
    KcRows a{g, ldg, M_cap, N, g_row_off_dyn};
    McRows b{W, ldw, K, N, nullptr};
    Epi e = make_epi(dx, lddx);
    e.relu_ref = relu_ref; e.ld_ref = ld_ref; e.relu_scale = relu_scale;
    e.rowinfo = rowinfo; e.colsum = colsum; e.row_off_dyn = dx_row_off_dyn;
    set_drop(e, drop, K);
    hipStream_t st = (hipStream_t)stream;
    if (rowinfo != nullptr) {
        LEGO_REQUIRE(!accumulate && relu_ref == nullptr, "lego_linear_bwd_data: rowinfo cannot be combined with accumulate / relu_ref");
        return launch_rows<true, EpiLive>(d, a, b, e, st, "lego_linear_bwd_data");
    }
    if (relu_ref != nullptr) {
        LEGO_REQUIRE(accumulate, "lego_linear_bwd_data: relu_ref requires accumulate=1");
        return launch_rows<true, EpiAccumRelu>(d, a, b, e, st, "lego_linear_bwd_data");
    }
    if (accumulate) return launch_rows<true, EpiAccum>(d, a, b, e, st, "lego_linear_bwd_data");
    return launch_rows<true, EpiPlain>(d, a, b, e, st, "lego_linear_bwd_data");
}

// ---- BERT feed-forward with the GELU in the product epilogues (config 5; modeling_bert.py BertIntermediate / BertOutput).  The tile kernel's
// epilogue only (act 3 / 4 of gemm_epi.hpp): these products have N = 3072 outputs and run on 128 x 128 tiles whatever the row count.
extern "C" int lego_linear_gelu_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias, float* z, int ldz, float* g, int ldg,
                                    int M, int N, int K, void* stream) {
    CHECK4(ldx); CHECK4(ldw); CHECK4(K);
    LEGO_REQUIRE(z != nullptr && g != nullptr, "lego_linear_gelu_fwd: both outputs (pre-activation z, gelu(z)) are required");
    if (M <= 0) return 0;
    GemmDims d{M, N, K, nullptr, nullptr, 1};
    KcRows a{x, ldx, M, K, nullptr};
    KcRows b{W, ldw, N, K, nullptr};
    Epi e = make_epi(g, ldg);
    e.bias = bias; e.act = 3; e.C2 = z; e.ldc2 = ldz;
    const int tm = (M + 127) / 128;
    if (product_mode() == 1 && M >= SPLIT_MIN_ROWS) {
        // the opt-in split-bf16 tiles (128 x 256, 64 x 64 per wave) have no registers to spare for the erf in their epilogue (measured: 248 -> 206
        // TFLOP/s-equivalent fused): that mode keeps the product and the GELU as two passes
        LEGO_REQUIRE(ldz == N && ldg == N, "lego_linear_gelu_fwd: the split-bf16 mode needs contiguous outputs (ldz=%d ldg=%d N=%d)", ldz, ldg, N);
        if (lego_linear_fwd(x, ldx, W, ldw, bias, z, ldz, M, nullptr, N, K, 0, nullptr, nullptr, nullptr, nullptr, stream) != 0) return 1;
        return lego_gelu_fwd(z, g, (int64_t)M * N, stream);
    }
    return launch<C128x128, false, false, EpiPlain>(d, a, b, e, tm, (N + 127) / 128, 1, (hipStream_t)stream, "lego_linear_gelu_fwd");
}

/* dz[M,K] = (dy[M,N] . W[N,K]) * gelu'(z[M,K]) */
extern "C" int lego_linear_bwd_data_gelu(const float* dy, int ldy, const float* W, int ldw, const float* z, int ldz, float* dz, int lddz,
                                         int M, int N, int K, void* stream) {
    CHECK4(ldy); CHECK4(ldw); CHECK4(N); CHECK4(K);
    LEGO_REQUIRE(z != nullptr, "lego_linear_bwd_data_gelu: the forward's pre-activation z is required");
    if (M <= 0) return 0;
    GemmDims d{M, /*N=*/K, /*K=*/N, nullptr, nullptr, 1};
    KcRows a{dy, ldy, M, N, nullptr};
    McRows b{W, ldw, K, N, nullptr};
    Epi e = make_epi(dz, lddz);
    e.relu_ref = z; e.ld_ref = ldz; e.act = 4;
    const int tm = (M + 127) / 128;
    if (product_mode() == 1 && M >= SPLIT_MIN_ROWS) {              // (see lego_linear_gelu_fwd: fused, the split tile spilled -- 238 -> 36 TFLOP/s-equivalent)
        LEGO_REQUIRE(ldz == K && lddz == K, "lego_linear_bwd_data_gelu: the split-bf16 mode needs contiguous z / dz (ldz=%d lddz=%d K=%d)", ldz, lddz, K);
        if (lego_linear_bwd_data(dy, ldy, W, ldw, dz, lddz, M, nullptr, N, K, 0, nullptr, 0, 1.f, nullptr, nullptr, nullptr, nullptr, nullptr, stream) != 0) return 1;
        return lego_gelu_bwd(dz, z, dz, (int64_t)M * K, stream);
    }
    return launch<C128x128, false, true, EpiRef>(d, a, b, e, tm, (K + 127) / 128, 1, (hipStream_t)stream, "lego_linear_bwd_data_gelu");
}

extern "C" int lego_linear_bwd_weight(const float* g, int ldg, const float* x, int ldx, float* dW, int lddw,
                                      int M_cap, const int32_t* M_dyn, int N, int K,
                                      const int32_t* g_row_off_dyn, const int32_t* x_row_off_dyn, void* stream) {
    CHECK4(ldg); CHECK4(ldx); CHECK4(N); CHECK4(K);
    if (M_cap <= 0) return 0;
    // dW[N,K] += sum_r g[r,:]^T x[r,:]: TN product, reduction over the rows
    if (product_mode() == 0 && tnd_ok(N, K, M_cap, ldg, ldx, lddw))        // round 5: operand fragments straight from global memory (gemm_tnd.hpp)
        return launch_tnd(g, ldg, x, ldx, dW, lddw, N, K, M_cap, M_dyn, g_row_off_dyn, x_row_off_dyn, (hipStream_t)stream, "lego_linear_bwd_weight");
    // Round 6: outputs beyond that kernel's window (BERT: 3072 x 768, 2304 x 768) as 768-wide CHUNKS inside it.  One launch of the tile kernel
    // has too few tiles to split the reduction (576 tiles, split 1: 114-117 TFLOP/s; 2304 x 768: 108); the chunks are disjoint blocks of the
    // output over the same operands -- the same bytes from L2 -- at 135 TFLOP/s each (profiles/r06_bert_wgrad_chunks.txt).  Exact.
    constexpr int CH = 768;
    if (product_mode() == 0 && (long long)N * K > (1ll << 20)) {
        if (N % CH == 0 && N > CH && tnd_ok(CH, K, M_cap, ldg, ldx, lddw)) {
            for (int r0 = 0; r0 < N; r0 += CH)
                if (launch_tnd(g + r0, ldg, x, ldx, dW + (size_t)r0 * lddw, lddw, CH, K, M_cap, M_dyn, g_row_off_dyn, x_row_off_dyn, (hipStream_t)stream,
                               "lego_linear_bwd_weight") != 0) return 1;
            return 0;
        }
        if (K % CH == 0 && K > CH && tnd_ok(N, CH, M_cap, ldg, ldx, lddw)) {
            for (int c0 = 0; c0 < K; c0 += CH)
                if (launch_tnd(g, ldg, x + c0, ldx, dW + c0, lddw, N, CH, M_cap, M_dyn, g_row_off_dyn, x_row_off_dyn, (hipStream_t)stream,
                               "lego_linear_bwd_weight") != 0) return 1;
            return 0;
        }
    }
    McRows a{g, ldg, N, M_cap, g_row_off_dyn};
    McRows b{x, ldx, K, M_cap, x_row_off_dyn};
    Epi e = make_epi(dW, lddw);
    return launch_tn(/*M=*/N, /*N=*/K, /*K=*/M_cap, M_dyn, a, b, e, 1, (hipStream_t)stream, "lego_linear_bwd_weight");
}

extern "C" int lego_conv3_fwd(const float* h, int ldh, const float* wt, const float* bias, const int32_t* rowinfo,
                              float* y, int ldy, int R_cap, const int32_t* R_dyn, int Dout, int Din,
                              const lego_dropout* drop, int mask_rows, void* stream) {
    CHECK4(ldh);
    LEGO_REQUIRE(Din % BK == 0, "lego_conv3_fwd: Din=%d must be a multiple of %d", Din, BK);
    if (R_cap <= 0) return 0;
    // y[r,o] = relu(sum_tap sum_c h[r+tap-1,c] wt[tap][o][c] + b[o]): NT product with K = 3*Din
    GemmDims d{R_cap, Dout, 3 * Din, R_dyn, nullptr, 1};
    KcConvA a{h, ldh, R_cap, 3 * Din, rowinfo, Din, +1, 0, 0, 0};
    KcTapW b{wt, Din, Dout, 3 * Din, Din, (size_t)Dout * Din, 0};
    Epi e = make_epi(y, ldy);
    e.bias = bias; e.act = 1; e.rowinfo = rowinfo;
    set_drop(e, drop, Dout);
    if (!mask_rows) return launch_rows<false, EpiPlain>(d, a, b, e, (hipStream_t)stream, "lego_conv3_fwd");
    return launch_rows<false, EpiLive>(d, a, b, e, (hipStream_t)stream, "lego_conv3_fwd");
}

extern "C" int lego_conv3_bwd_data(const float* gy, int ldg, const float* wt, const int32_t* rowinfo,
                                   float* dh, int lddh, int R_cap, const int32_t* R_dyn, int Dout, int Din,
                                   const lego_dropout* drop_in, float* colsum, int mask_rows, void* stream) {
    CHECK4(ldg); CHECK4(Din);
    LEGO_REQUIRE(Dout % BK == 0, "lego_conv3_bwd_data: Dout=%d must be a multiple of %d", Dout, BK);
    if (R_cap <= 0) return 0;
    // dh[r,c] = sum_tap sum_o gy[r-(tap-1),o] wt[tap][o][c]: NN product, K = 3*Dout, wt is [3*Dout][Din] row-major
    GemmDims d{R_cap, Din, 3 * Dout, R_dyn, nullptr, 1};
    KcConvA a{gy, ldg, R_cap, 3 * Dout, rowinfo, Dout, -1, 0, 0, 0};
    McRows b{wt, Din, Din, 3 * Dout, nullptr};
    Epi e = make_epi(dh, lddh);
    e.rowinfo = rowinfo; e.colsum = colsum;
    set_drop(e, drop_in, Din);
    if (!mask_rows) return launch_rows<true, EpiPlain>(d, a, b, e, (hipStream_t)stream, "lego_conv3_bwd_data");
    return launch_rows<true, EpiLive>(d, a, b, e, (hipStream_t)stream, "lego_conv3_bwd_data");
}

extern "C" int lego_conv3_bwd_weight(const float* gy, int ldg, const float* h, int ldh, const int32_t* rowinfo,
                                     float* dwt, int R_cap, const int32_t* R_dyn, int Dout, int Din, void* stream) {
    CHECK4(ldg); CHECK4(ldh); CHECK4(Dout); CHECK4(Din);
    if (R_cap <= 0) return 0;
    // dwt[tap][o][c] += sum_r gy[r,o] h[r+tap-1,c]: three TN products (gridDim.z = 3 * split)
    McRows a{gy, ldg, Dout, R_cap, nullptr};
    McShiftRows b{h, ldh, Din, R_cap, rowinfo, 0};
    Epi e = make_epi(dwt, Din);
    e.tap_stride = (size_t)Dout * Din;
    return launch_tn(Dout, Din, R_cap, R_dyn, a, b, e, 3, (hipStream_t)stream, "lego_conv3_bwd_weight");
}


// ---- Winograd F(2,3) form of the conv over row pairs (wino_common.hpp; kernel: gemm_wino2.hpp / wino2_ops.hip -- weight fragments
// straight from global memory, branch-free A staging through buffer descriptors, row-major epilogue)
namespace lego {
const char* wino2_why_not(const WinoArgs& w, const EpiArgs& e);
int launch_wino2(const WinoArgs& w, const EpiArgs& e, hipStream_t st, const char* what);
}

static int launch_wino(const WinoArgs& w, const Epi& e, hipStream_t st, const char* what) {
    const char* why = wino2_why_not(w, e);
    LEGO_REQUIRE(why == nullptr, "%s: %s (the direct three-tap entry points lego_conv3_* take every shape)", what, why);
    return launch_wino2(w, e, st, what);
}

extern "C" int lego_conv3_wino_fwd(const float* h, int ldh, const float* u, const float* bias, const int32_t* pair_info,
                                   int P_cap, const int32_t* P_dyn, float* y, int ldy, int Dout, int Din,
                                   const lego_dropout* drop, void* stream) {
    CHECK4(ldh);
    LEGO_REQUIRE(Din % BK == 0 && Dout % 4 == 0 && Dout <= STRIP_BN, "lego_conv3_wino_fwd: Din=%d must be a multiple of %d, Dout=%d a multiple of 4 and <= %d", Din, BK, Dout, STRIP_BN);
    if (P_cap <= 0) return 0;
    WinoArgs w{h, ldh, u, Din, Dout, pair_info, P_cap, P_dyn, 0};
    Epi e = make_epi(y, ldy);
    e.bias = bias; e.act = 1;
    set_drop(e, drop, Dout);
    return launch_wino(w, e, (hipStream_t)stream, "lego_conv3_wino_fwd");
}

extern "C" int lego_conv3_wino_bwd_data(const float* gy, int ldg, const float* u, const float* ut, const int32_t* pair_info,
                                        int P_cap, const int32_t* P_dyn, float* dh, int lddh, int Dout, int Din,
                                        const lego_dropout* drop_in, float* colsum, void* stream) {
    CHECK4(ldg);
    LEGO_REQUIRE(Dout % BK == 0 && Din % 4 == 0 && Din <= STRIP_BN, "lego_conv3_wino_bwd_data: Dout=%d must be a multiple of %d, Din=%d a multiple of 4 and <= %d", Dout, BK, Din, STRIP_BN);
    if (P_cap <= 0) return 0;
    Epi e = make_epi(dh, lddh);
    e.colsum = colsum;
    set_drop(e, drop_in, Din);
    LEGO_REQUIRE(ut != nullptr, "lego_conv3_wino_bwd_data: needs the transposed weight sets `ut` (lego_conv3_wino_pack writes both)");
    (void)u;
    WinoArgs w{gy, ldg, ut, Dout, Din, pair_info, P_cap, P_dyn, 1};      // transposed sets: the weight panel is K-contiguous like the forward's
    return launch_wino(w, e, (hipStream_t)stream, "lego_conv3_wino_bwd_data");
}

extern "C" int lego_conv3_wino_du_slabs(int Dout, int Din, int P_cap) {
    if (P_cap < TN_LONG) return 1;                  // short reductions accumulate with atomics into ONE cleared buffer
    if (product_mode() == 0 && tndp_slabs(Dout, Din, P_cap) > 0) return tndp_slabs(Dout, Din, P_cap);      // round 5: gemm_tnd.hpp
    return tn_split(Dout, Din, P_cap, 4);
}

extern "C" int lego_conv3_wino_bwd_weight(const float* gy, int ldg, const float* h, int ldh, const int32_t* pair_info,
                                          int P_cap, const int32_t* P_dyn, float* du, int n_slabs, int Dout, int Din, void* stream) {
    CHECK4(ldg); CHECK4(ldh); CHECK4(Dout); CHECK4(Din);
    if (P_cap <= 0) return 0;
    // the slab count is a function of (Dout, Din, P_cap) AND the process-wide product mode at call time: the caller states how many
    // slabs `du` holds, and a launch that would write another number is refused instead of running past the buffer
    const int want = lego_conv3_wino_du_slabs(Dout, Din, P_cap);
    LEGO_REQUIRE(n_slabs == want, "lego_conv3_wino_bwd_weight: du holds %d slab(s), this launch writes %d (lego_conv3_wino_du_slabs; did the "
                 "product mode change after the buffer was sized?)", n_slabs, want);
    // du[set][o][c] += sum_pairs dM_set[o] * A_set[c]: four TN products over the pair rows (gridDim.z = 4 * split)
    if (P_cap >= TN_LONG && product_mode() == 0 && tndp_slabs(Dout, Din, P_cap) > 0)
        return launch_tndp(gy, ldg, h, ldh, pair_info, P_cap, P_dyn, du, Dout, Din, (hipStream_t)stream, "lego_conv3_wino_bwd_weight");
    McPair a{gy, ldg, Dout, P_cap, pair_info, 1, 0};
    McPair b{h, ldh, Din, P_cap, pair_info, 0, 0};
    Epi e = make_epi(du, Din);
    e.tap_stride = (size_t)Dout * Din;
    if (want > 1)                                   // long reduction: one slab per k split, plain stores
        return launch_tn_long<true>(Dout, Din, P_cap, P_dyn, a, b, e, 4, (hipStream_t)stream, "lego_conv3_wino_bwd_weight");
    return launch_tn(Dout, Din, P_cap, P_dyn, a, b, e, 4, (hipStream_t)stream, "lego_conv3_wino_bwd_weight");
}

#ifdef LEGO_TUNING_HOOKS   // `make tune` only (liblego_hip_tune.so, tools/*_variants.py): never in the product library
namespace lego { __device__ unsigned long long g_clock_probe[4]; }
// accumulated (core cycles, 100 MHz ticks) of slot 0 = Winograd kernel, 1 = row-strip kernel; reset != 0 clears them
extern "C" int lego_debug_clock(unsigned long long* out4, int reset) {
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(lego::g_clock_probe), sizeof(unsigned long long) * 4) != hipSuccess) return set_error("clock probe read");
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(lego::g_clock_probe), z, sizeof(z)); }
    return 0;
}
// ---- internal tuning hook (not part of the public ABI): plain NT product with a selectable tile config
extern "C" int lego_debug_gemm_nt(int variant, const float* x, const float* W, const float* bias, float* out,
                                  int M, int N, int K, void* stream) {
    GemmDims d{M, N, K, nullptr, nullptr, 1};
    KcRows a{x, K, M, K, nullptr};
    KcRows b{W, K, N, K, nullptr};
    Epi e = make_epi(out, N);
    e.bias = bias;
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 0: return launch<TileCfg<128, 128, 2, 2>, false, false, EpiPlain>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg0");
        case 1: return launch<TileCfg<128, 128, 2, 4>, false, false, EpiPlain>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg1");
        case 2: return launch<TileCfg<256, 128, 4, 2>, false, false, EpiPlain>(d, a, b, e, (M + 255) / 256, (N + 127) / 128, 1, st, "dbg2");
        case 3: return launch<TileCfg<128, 256, 2, 4>, false, false, EpiPlain>(d, a, b, e, (M + 127) / 128, (N + 255) / 256, 1, st, "dbg3");
        case 4: return launch<TileCfg<64, 128, 1, 4>, false, false, EpiPlain>(d, a, b, e, (M + 63) / 64, (N + 127) / 128, 1, st, "dbg4");
        case 5: return launch<TileCfg<128, 128, 4, 2>, false, false, EpiPlain>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg5");
        case 6: return launch<TileCfg<64, 256, 1, 4>, false, false, EpiPlain>(d, a, b, e, (M + 63) / 64, (N + 255) / 256, 1, st, "dbg6");
        case 7: return launch<TileCfg<128, 128, 4, 2, true>, false, false, EpiPlain>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg7");
        case 8: return launch<TileCfg<128, 128, 2, 4, true>, false, false, EpiPlain>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg8");
        case 9: return launch_strip<false, EpiPlain>(d, a, b, e, st, "dbg9");
        case 10: return launch_dma_strip<false, EpiPlain>(d, a, b, e, st, "dbg10");
        case 20: return launch<TileCfg<128, 128, 2, 4, true>, false, false, EpiPlain, true>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg20");
        case 21: return launch<TileCfg<256, 128, 4, 2>, false, false, EpiPlain, true>(d, a, b, e, (M + 255) / 256, (N + 127) / 128, 1, st, "dbg21");
        case 22: return launch<TileCfg<128, 256, 2, 4>, false, false, EpiPlain, true>(d, a, b, e, (M + 127) / 128, (N + 255) / 256, 1, st, "dbg22");
        case 23: return launch<TileCfg<128, 128, 2, 2>, false, false, EpiPlain, true>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg23");
        case 24: return launch<TileCfg<256, 128, 4, 2, true>, false, false, EpiPlain, true>(d, a, b, e, (M + 255) / 256, (N + 127) / 128, 1, st, "dbg24");
        case 25: return launch<TileCfg<64, 128, 1, 4>, false, false, EpiPlain, true>(d, a, b, e, (M + 63) / 64, (N + 127) / 128, 1, st, "dbg25");
        case 26: return launch<TileCfg<128, 128, 4, 2, true>, false, false, EpiPlain, true>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg26");
        default: return set_error("lego_debug_gemm_nt: unknown variant %d", variant);
    }
}

extern "C" int lego_debug_gemm_tn(int variant, int split, const float* g, const float* x, float* dW,
                                  int R, int N, int K, void* stream) {
    GemmDims d{N, K, R, nullptr, nullptr, split};
    McRows a{g, N, N, R, nullptr};
    McRows b{x, K, K, R, nullptr};
    Epi e = make_epi(dW, K);
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 0: return launch<TileCfg<128, 128, 4, 2>, true, true, EpiAtomic>(d, a, b, e, (N + 127) / 128, (K + 127) / 128, split, st, "tn0");
        case 1: return launch<TileCfg<64, 64, 2, 2>, true, true, EpiAtomic>(d, a, b, e, (N + 63) / 64, (K + 63) / 64, split, st, "tn1");
        case 2: return launch<TileCfg<128, 64, 4, 1>, true, true, EpiAtomic>(d, a, b, e, (N + 127) / 128, (K + 63) / 64, split, st, "tn2");
        case 3: return launch<TileCfg<64, 128, 1, 4>, true, true, EpiAtomic>(d, a, b, e, (N + 63) / 64, (K + 127) / 128, split, st, "tn3");
        case 4: return launch<TileCfg<128, 128, 2, 2>, true, true, EpiAtomic>(d, a, b, e, (N + 127) / 128, (K + 127) / 128, split, st, "tn4");
        case 20: return launch<TileCfg<128, 128, 4, 2>, true, true, EpiAtomic, true>(d, a, b, e, (N + 127) / 128, (K + 127) / 128, split, st, "tn20");
        case 21: return launch<TileCfg<64, 64, 2, 2>, true, true, EpiAtomic, true>(d, a, b, e, (N + 63) / 64, (K + 63) / 64, split, st, "tn21");
        case 22: return launch<TileCfg<128, 128, 2, 2>, true, true, EpiAtomic, true>(d, a, b, e, (N + 127) / 128, (K + 127) / 128, split, st, "tn22");
        default: return set_error("lego_debug_gemm_tn: unknown variant %d", variant);
    }
}
#endif
