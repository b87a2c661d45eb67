// Launcher of the round-5 Winograd conv kernel (gemm_wino2.hpp); its own translation unit so that the kernel builds in
// seconds (gemm_ops.hip instantiates every other product of the library).
#include <stdlib.h>
#include "gemm_wino2.hpp"

namespace lego {

static int w2_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

#ifdef LEGO_TUNING_HOOKS
static int wino2_abl() {        // LEGO_WINO2_ABL=<bits>: the ablation variants of gemm_wino2.hpp (timing only, wrong results)
    static int v = -1;
    if (v < 0) { const char* e = getenv("LEGO_WINO2_ABL"); v = e != nullptr ? atoi(e) : 0; }
    return v;
}
#endif

// nullptr when the kernel takes the launch, else the reason it cannot (the entry points report it: there is no second Winograd kernel)
const char* wino2_why_not(const WinoArgs& w, const EpiArgs& e) {
    if (w.C % BK != 0 || (w.N & 3) != 0 || (w.ldx & 3) != 0 || (e.ldc & 3) != 0) return "channel counts / row strides must be multiples of 32 / 4";
    if ((reinterpret_cast<uintptr_t>(w.x) | reinterpret_cast<uintptr_t>(w.u) | reinterpret_cast<uintptr_t>(e.C)) & 15) return "operands must be 16-byte aligned";
    if (e.bias != nullptr && (reinterpret_cast<uintptr_t>(e.bias) & 15)) return "the bias must be 16-byte aligned";
    if (e.drop.p > 0.f && (e.drop.mask == nullptr || (e.drop_cols & 3) != 0 || (reinterpret_cast<uintptr_t>(e.drop.mask) & 3)))
        return "a dropout site needs its keep bits drawn ahead of time (lego_dropout_mask)";
    if (2ull * (unsigned long long)w.P_cap * (unsigned long long)w.ldx * 4ull >= 0x7FFFFFF0ull) return "the input exceeds 2 GB (31-bit row offsets)";
    if (e.act == 2) return "tanh epilogue not built";
    return nullptr;
}

int launch_wino2(const WinoArgs& w0, const EpiArgs& e, hipStream_t st, const char* what) {
    WinoArgs w = w0;
    w.x_bytes = (unsigned)(2ull * (unsigned long long)w.P_cap * (unsigned long long)w.ldx * 4ull);    // rows < 2 * P_cap; kNone (2^31) is past it
    const dim3 grid(w2_num_cus() / 16 * 16);            // (strip, half) dealing needs a multiple of 16
    auto go = [&](auto k) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino2_lds_bytes());
        hipLaunchKernelGGL(k, grid, dim3(STRIP_THREADS), wino2_lds_bytes(), st, w, e);
    };
#ifdef LEGO_TUNING_HOOKS
    switch (wino2_abl()) {
        case 0: break;
        case 1: go(wino2_kernel<1>); return check_launch(what);
        case 2: go(wino2_kernel<2>); return check_launch(what);
        case 3: go(wino2_kernel<3>); return check_launch(what);
        case 4: go(wino2_kernel<4>); return check_launch(what);
        case 7: go(wino2_kernel<7>); return check_launch(what);
        case 8: go(wino2_kernel<8>); return check_launch(what);
        case 15: go(wino2_kernel<15>); return check_launch(what);
        case 16: go(wino2_kernel<16>); return check_launch(what);
        case 31: go(wino2_kernel<31>); return check_launch(what);
        case 40: go(wino2_kernel<40>); return check_launch(what);
        case 47: go(wino2_kernel<47>); return check_launch(what);
        case 64: go(wino2_kernel<64>); return check_launch(what);
        case 72: go(wino2_kernel<72>); return check_launch(what);
        case 128: go(wino2_kernel<128>); return check_launch(what);
        case 256: go(wino2_kernel<256>); return check_launch(what);
        default: return set_error("LEGO_WINO2_ABL=%d is not an instantiated variant", wino2_abl());
    }
#endif
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino2_lds_bytes());
        attr_done = true;
    }
    hipLaunchKernelGGL(wino2_kernel<0>, grid, dim3(STRIP_THREADS), wino2_lds_bytes(), st, w, e);
    return check_launch(what);
}

}  // namespace lego
