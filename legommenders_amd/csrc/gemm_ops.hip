// GEMM-shaped entry points of liblego_hip.so: every dense product of the NAML / NRMS forward and
// backward pass runs on the fp32 MFMA core in gemm_core.hpp (exact f32; the path's parity bar is
// 1e-3 on fp32 logits, so no reduced-precision inputs are used).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/lego_hip.h"
#include "gemm_core.hpp"

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

// One run-time configurable epilogue (the k-loop dominates; the branches here are noise).
struct Epi {
    float* C; int ldc;
    const float* bias;        // [N] or null
    int act;                  // 0 none, 1 relu, 2 tanh
    const int* rowinfo;       // null or live-bit source (indexed by absolute row)
    Dropout drop;             // p == 0: off
    int drop_cols;            // column count of the dropout counter space
    int accumulate;           // add the previous C value
    const float* relu_ref; int ld_ref; float relu_scale;   // backward of ReLU(+dropout): ref>0 ? x*scale : 0
    int atomic;               // split-K: atomicAdd into C
    float* colsum;            // += column sums of the stored values (bias gradients)
    size_t tap_stride;        // C offset per tap (TN conv weight gradient)
    const int* row_off_dyn;   // device row offset of C / rowinfo / relu_ref rows
    int M, N, row_off;

    __device__ __forceinline__ void setup(int M_, int N_, int tap) {
        M = M_; N = N_;
        row_off = row_off_dyn != nullptr ? *row_off_dyn : 0;
        C += (size_t)tap * tap_stride;
    }
    __device__ __forceinline__ void apply4(int r0, int c, float (&v)[4]) {
        const float b = bias != nullptr ? bias[c] : 0.f;
        float ds[4];
        dropout_scale4(drop, r0 + row_off, c, drop_cols, ds);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + i;
            if (r >= M) { v[i] = 0.f; continue; }
            const int ra = r + row_off;
            float x = v[i] + b;
            if (act == 1) x = fmaxf(x, 0.f);
            else if (act == 2) x = tanhf(x);
            if (rowinfo != nullptr && !(rowinfo[ra] & RI_LIVE)) x = 0.f;
            x *= ds[i];
            const size_t o = (size_t)ra * ldc + c;
            if (accumulate) x += C[o];
            if (relu_ref != nullptr) x = relu_ref[(size_t)ra * ld_ref + c] > 0.f ? x * relu_scale : 0.f;
            if (atomic) atomicAdd(C + o, x); else C[o] = x;
            v[i] = x;
        }
    }
};

static Epi make_epi(float* C, int ldc) {
    Epi e;
    e.C = C; e.ldc = ldc; e.bias = nullptr; e.act = 0; e.rowinfo = nullptr;
    e.drop = Dropout{0.f, 0u, 0u, 0u}; e.drop_cols = 1; e.accumulate = 0;
    e.relu_ref = nullptr; e.ld_ref = 0; e.relu_scale = 1.f; e.atomic = 0; e.colsum = nullptr;
    e.tap_stride = 0; e.row_off_dyn = nullptr; e.M = e.N = e.row_off = 0;
    return e;
}
static void set_drop(Epi& e, const lego_dropout* d, int cols) {
    if (d != nullptr && d->p > 0.f) {
        e.drop = Dropout{d->p, (uint32_t)d->seed, (uint32_t)(d->seed >> 32), d->site};
        e.drop_cols = cols;
    }
}

template <class Cfg, bool A_MC, bool B_MC, class AL, class BL>
static int launch(const GemmDims& d, const AL& a, const BL& b, const Epi& e, int tiles_m, int tiles_n, int gz,
                  hipStream_t st, const char* what) {
    auto k = gemm_kernel<Cfg, A_MC, B_MC, AL, BL, Epi>;
    constexpr size_t lds = gemm_lds_bytes<Cfg, A_MC, B_MC>();
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(tiles_m, tiles_n, gz), dim3(Cfg::kThreads), lds, st, d, a, b, e);
    return check_launch(what);
}

using C128x128 = TileCfg<128, 128, 4, 2>;     // 8 waves, each 32 x 64: +5-20 % over 4 waves of 64 x 64 (scratch/gemm_variants.py)
using C128x64 = TileCfg<128, 64, 4, 1>;
using C64x128 = TileCfg<64, 128, 1, 4>;
using C64x64 = TileCfg<64, 64, 2, 2>;

// NT / NN: rows x N output, BM = 128, BN by N
template <bool B_MC, class AL, class BL>
static int launch_rows(const GemmDims& d, const AL& a, const BL& b, const Epi& e, hipStream_t st, const char* what) {
    const int tm = (d.M + 127) / 128;
    if (tm * ((d.N + 127) / 128) < 128) {         // few row tiles (user / category side): 64-row tiles fill more CUs
        if (d.N > 64) return launch<C64x128, false, B_MC>(d, a, b, e, (d.M + 63) / 64, (d.N + 127) / 128, 1, st, what);
        return launch<C64x64, false, B_MC>(d, a, b, e, (d.M + 63) / 64, (d.N + 63) / 64, 1, st, what);
    }
    if (d.N > 64) return launch<C128x128, false, B_MC>(d, a, b, e, tm, (d.N + 127) / 128, 1, st, what);
    return launch<C128x64, false, B_MC>(d, a, b, e, tm, (d.N + 63) / 64, 1, st, what);
}
// TN: small [M,N] output, reduction over the (ragged) rows split along gridDim.z
template <class AL, class BL>
static int launch_tn(const GemmDims& d, const AL& a, const BL& b, const Epi& e, int taps, hipStream_t st, const char* what) {
    const int gz = taps * d.split_k;
    if (d.M > 64) {
        if (d.N > 64) return launch<C128x128, true, true>(d, a, b, e, (d.M + 127) / 128, (d.N + 127) / 128, gz, st, what);
        return launch<C128x64, true, true>(d, a, b, e, (d.M + 127) / 128, (d.N + 63) / 64, gz, st, what);
    }
    if (d.N > 64) return launch<C64x128, true, true>(d, a, b, e, (d.M + 63) / 64, (d.N + 127) / 128, gz, st, what);
    return launch<C64x64, true, true>(d, a, b, e, (d.M + 63) / 64, (d.N + 63) / 64, gz, st, what);
}

static int pick_split(int rows_cap, int M, int N, int taps) {
    // (tile x split) blocks fill the 256 CUs twice but never spill into a third round; >= 128 reduction rows per block
    const int bm = M > 64 ? 128 : 64, bn = N > 64 ? 128 : 64;
    const int tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * taps;
    int s = 512 / tiles;
    const int max_s = (rows_cap + 127) / 128;
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return s;
}

}  // namespace lego

using namespace lego;

extern "C" const char* lego_last_error(void) { return lego::last_error(); }
extern "C" int lego_abi_version(void) { return LEGO_ABI_VERSION; }

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
    return launch_rows<false>(d, a, b, e, (hipStream_t)stream, "lego_linear_fwd");
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
    e.accumulate = accumulate; e.relu_ref = relu_ref; e.ld_ref = ld_ref; e.relu_scale = relu_scale;
    e.rowinfo = rowinfo; e.colsum = colsum; e.row_off_dyn = dx_row_off_dyn;
    set_drop(e, drop, K);
    return launch_rows<true>(d, a, b, e, (hipStream_t)stream, "lego_linear_bwd_data");
}

extern "C" int lego_linear_bwd_weight(const float* g, int ldg, const float* x, int ldx, float* dW, int lddw,
                                      int M_cap, const int32_t* M_dyn, int N, int K,
                                      const int32_t* g_row_off_dyn, const int32_t* x_row_off_dyn, void* stream) {
    CHECK4(ldg); CHECK4(ldx); CHECK4(N); CHECK4(K);
    if (M_cap <= 0) return 0;
    // dW[N,K] += sum_r g[r,:]^T x[r,:]: TN product, reduction over the rows
    GemmDims d{/*M=*/N, /*N=*/K, /*K=*/M_cap, nullptr, M_dyn, pick_split(M_cap, N, K, 1)};
    McRows a{g, ldg, N, M_cap, g_row_off_dyn};
    McRows b{x, ldx, K, M_cap, x_row_off_dyn};
    Epi e = make_epi(dW, lddw);
    e.atomic = 1;
    return launch_tn(d, a, b, e, 1, (hipStream_t)stream, "lego_linear_bwd_weight");
}

extern "C" int lego_conv3_fwd(const float* h, int ldh, const float* wt, const float* bias, const int32_t* rowinfo,
                              float* y, int ldy, int R_cap, const int32_t* R_dyn, int Dout, int Din,
                              const lego_dropout* drop, void* stream) {
    CHECK4(ldh);
    LEGO_REQUIRE(Din % BK == 0, "lego_conv3_fwd: Din=%d must be a multiple of %d", Din, BK);
    if (R_cap <= 0) return 0;
    // y[r,o] = relu(sum_tap sum_c h[r+tap-1,c] wt[tap][o][c] + b[o]): NT product with K = 3*Din
    GemmDims d{R_cap, Dout, 3 * Din, R_dyn, nullptr, 1};
    KcConvA a{h, ldh, R_cap, 3 * Din, rowinfo, Din, +1};
    KcTapW b{wt, Din, Dout, 3 * Din, Din, (size_t)Dout * Din};
    Epi e = make_epi(y, ldy);
    e.bias = bias; e.act = 1; e.rowinfo = rowinfo;
    set_drop(e, drop, Dout);
    return launch_rows<false>(d, a, b, e, (hipStream_t)stream, "lego_conv3_fwd");
}

extern "C" int lego_conv3_bwd_data(const float* gy, int ldg, const float* wt, const int32_t* rowinfo,
                                   float* dh, int lddh, int R_cap, const int32_t* R_dyn, int Dout, int Din,
                                   const lego_dropout* drop_in, float* colsum, void* stream) {
    CHECK4(ldg); CHECK4(Din);
    LEGO_REQUIRE(Dout % BK == 0, "lego_conv3_bwd_data: Dout=%d must be a multiple of %d", Dout, BK);
    if (R_cap <= 0) return 0;
    // dh[r,c] = sum_tap sum_o gy[r-(tap-1),o] wt[tap][o][c]: NN product, K = 3*Dout, wt is [3*Dout][Din] row-major
    GemmDims d{R_cap, Din, 3 * Dout, R_dyn, nullptr, 1};
    KcConvA a{gy, ldg, R_cap, 3 * Dout, rowinfo, Dout, -1};
    McRows b{wt, Din, Din, 3 * Dout, nullptr};
    Epi e = make_epi(dh, lddh);
    e.rowinfo = rowinfo; e.colsum = colsum;
    set_drop(e, drop_in, Din);
    return launch_rows<true>(d, a, b, e, (hipStream_t)stream, "lego_conv3_bwd_data");
}

extern "C" int lego_conv3_bwd_weight(const float* gy, int ldg, const float* h, int ldh, const int32_t* rowinfo,
                                     float* dwt, int R_cap, const int32_t* R_dyn, int Dout, int Din, void* stream) {
    CHECK4(ldg); CHECK4(ldh); CHECK4(Dout); CHECK4(Din);
    if (R_cap <= 0) return 0;
    // dwt[tap][o][c] += sum_r gy[r,o] h[r+tap-1,c]: three TN products (gridDim.z = 3 * split)
    GemmDims d{Dout, Din, R_cap, nullptr, R_dyn, pick_split(R_cap, Dout, Din, 3)};
    McRows a{gy, ldg, Dout, R_cap, nullptr};
    McShiftRows b{h, ldh, Din, R_cap, rowinfo, 0};
    Epi e = make_epi(dwt, Din);
    e.atomic = 1; e.tap_stride = (size_t)Dout * Din;
    return launch_tn(d, a, b, e, 3, (hipStream_t)stream, "lego_conv3_bwd_weight");
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
        case 0: return launch<TileCfg<128, 128, 2, 2>, false, false>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg0");
        case 1: return launch<TileCfg<128, 128, 2, 4>, false, false>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg1");
        case 2: return launch<TileCfg<256, 128, 4, 2>, false, false>(d, a, b, e, (M + 255) / 256, (N + 127) / 128, 1, st, "dbg2");
        case 3: return launch<TileCfg<128, 256, 2, 4>, false, false>(d, a, b, e, (M + 127) / 128, (N + 255) / 256, 1, st, "dbg3");
        case 4: return launch<TileCfg<64, 128, 1, 4>, false, false>(d, a, b, e, (M + 63) / 64, (N + 127) / 128, 1, st, "dbg4");
        case 5: return launch<TileCfg<128, 128, 4, 2>, false, false>(d, a, b, e, (M + 127) / 128, (N + 127) / 128, 1, st, "dbg5");
        case 6: return launch<TileCfg<64, 256, 1, 4>, false, false>(d, a, b, e, (M + 63) / 64, (N + 255) / 256, 1, st, "dbg6");
        default: return set_error("lego_debug_gemm_nt: unknown variant %d", variant);
    }
}
