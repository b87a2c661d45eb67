// Launcher of the round-5 plain-row NT product kernel (gemm_rows2.hpp); its own translation unit (see wino2_ops.hip).
#include <stdlib.h>
#include "gemm_rows2.hpp"

namespace lego {

static int r2_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// Dispatch window (measured, tools/rows2_check.py / profiles/r05_rows2.txt): launches sized for >= 8 192 rows (fewer are latency-bound: the
// single-wave-tile kernels serve them); outputs of <= 256 columns (wider ones re-read every A strip once per column block: N = 768 takes 158 us
// against 133 for the row-strip kernel at two workgroups per CU); strips of >= 64 rows (16 / 64 / 112: 17.6 / 16.5 / 23.8 us on the projection over
// 4.5 k live of 105 k capacity rows); two workgroups per CU for one or two column blocks, three for more
constexpr int R2_MIN_ROWS = 8192, R2_MAX_N = 256, R2_MIN_STRIP = 64;
static int rows2_wgs_per_cu(int nblk) { return nblk <= 2 ? 2 : 3; }

// x [M, K] rows (no row offset), w [N, K] rows; e: C, bias, act, [accumulate is the template kind], relu_ref, colsum
bool rows2_ok(const float* x, int ldx, const float* w, int ldw, int M_cap, int N, int K, const EpiArgs& e, bool b_mc, bool accum, bool reluref) {
    if (M_cap < R2_MIN_ROWS || N > R2_MAX_N) return false;
    if ((K & 3) != 0 || K < 4 || (N & 3) != 0 || (ldx & 3) != 0 || (ldw & 3) != 0 || (e.ldc & 3) != 0) return false;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(e.C)) & 15) return false;
    if (e.bias != nullptr && (reinterpret_cast<uintptr_t>(e.bias) & 15)) return false;
    if (e.drop.p > 0.f || e.rowinfo != nullptr || e.tap_stride != 0) return false;
    if (reluref && (e.relu_ref == nullptr || (e.ld_ref & 3) != 0 || (reinterpret_cast<uintptr_t>(e.relu_ref) & 15))) return false;
    if (!reluref && e.relu_ref != nullptr) return false;
    (void)accum;
    if ((unsigned long long)M_cap * (unsigned long long)ldx * 4ull >= 0x7FFFFFF0ull) return false;      // row offsets are 31-bit
    if ((unsigned long long)(b_mc ? K : N) * (unsigned long long)ldw * 4ull >= 0x7FFFFFF0ull) return false;
    return true;
}

int launch_rows2(const float* x, int ldx, const float* w, int ldw, int M_cap, const int* M_dyn, int N, int K, const EpiArgs& e, bool b_mc, bool accum,
                 bool reluref, hipStream_t st, const char* what) {
    Rows2Args a{x, ldx, (unsigned)((unsigned long long)M_cap * (unsigned long long)ldx * 4ull), w, ldw,
                (unsigned)((unsigned long long)(b_mc ? K : N) * (unsigned long long)ldw * 4ull), M_cap, M_dyn, N, K, R2_MIN_STRIP};
    const int nblk = (N + R2_BN - 1) / R2_BN;
    // strips: rows2_wgs_per_cu workgroups per CU over the launch, in groups of 8 strips x nblk column blocks (XCD dealing); never more
    // strips than 16-row groups of the capacity
    int strips = r2_num_cus() * rows2_wgs_per_cu(nblk) / nblk;
    const int max_strips = (M_cap + 15) / 16;
    if (strips > max_strips) strips = max_strips;
    strips = (strips + 7) / 8 * 8;
    const dim3 grid(strips * nblk);
    auto go = [&](auto k) {
        static bool attr_done = false;              // (one flag per instantiation of this lambda)
        if (!attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows2_lds_bytes());
            attr_done = true;
        }
        hipLaunchKernelGGL(k, grid, dim3(STRIP_THREADS), rows2_lds_bytes(), st, a, e);
    };
    if (b_mc) {
        if (accum && reluref) go(rows2_kernel<true, true, true>);
        else if (accum) go(rows2_kernel<true, true, false>);
        else go(rows2_kernel<true, false, false>);
    } else {
        if (accum && reluref) go(rows2_kernel<false, true, true>);
        else if (accum) go(rows2_kernel<false, true, false>);
        else go(rows2_kernel<false, false, false>);
    }
    return check_launch(what);
}

}  // namespace lego
