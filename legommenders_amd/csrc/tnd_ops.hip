// Launcher of the LDS-free weight-gradient kernel (gemm_tnd.hpp); its own translation unit (builds in seconds).
#include <stdlib.h>
#include "gemm_tnd.hpp"

namespace lego {

static int tnd_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// tnd_kernel takes plain-row weight gradients whose OUTPUT has 128 K .. 1 M elements over >= 2 048 rows.  Measured (tools/tnd_check.py,
// profiles/r05_tnd.txt): 768 x 256 over 30.7 k rows (NRMS in-projection) 135 -> 106-125 us, 768 x 768 over 29.6 k (BERT) 318 -> 265-288 us, NRMS step
// 1.048 -> 1.020 ms; the 256 x 256 / 256 x 300 gradients of NAML are 15-20 % faster alone (45 -> 37 us) but the step is not (its
// side-stream launches then hold every CU's registers and LDS while the main stream's next kernel waits): they keep the tile kernels;
// 3072 x 768 (BERT FFN) is 6 % slower (the 64 x 64 wave tiles re-read the operands 48 x 12 times from L2).
constexpr int TND_MIN_ROWS = 2048, TND_MIN_NK = 1 << 17, TND_MAX_NK = 1 << 20;

bool tnd_ok(int M, int N, int K_cap, int lda, int ldb, int ldc) {
    if (K_cap < TND_MIN_ROWS || M < 4 || N < 4 || (long long)M * N > TND_MAX_NK || (long long)M * N < TND_MIN_NK) return false;
    if ((lda & 3) || (ldb & 3) || (M & 3) || (N & 3)) return false;
    const unsigned long long lim = 0x7FFF0000ull;          // 32-bit byte offsets, with room for the ring's reads past the range
    return ((unsigned long long)K_cap + 64) * (unsigned long long)lda * 4ull < lim && ((unsigned long long)K_cap + 64) * (unsigned long long)ldb * 4ull < lim;
}

int launch_tnd(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int M, int N, int K_cap, const int* k_dyn,
               const int* a_row_off, const int* b_row_off, hipStream_t st, const char* what) {
    // four per CU: at the shapes the dispatch window lets in, 2 per CU is the WORST count (profiles/r05_tnd.txt: 768 x 256 over 30.7 k rows
    // 106 / 125 / 104 us at 1 / 2 / 4 per CU, 768 x 768 425 / 288 / 266); NRMS step 0.990 -> 0.977-0.981 ms
    const int wgs = 4 * tnd_num_cus();
    const int tm = (M + TND_T - 1) / TND_T, tn = (N + TND_T - 1) / TND_T;
    int split = wgs / (tm * tn);
    const int max_s = (K_cap + 255) / 256;                  // at least 32 reduction rows per wave at capacity
    if (split > max_s) split = max_s;
    if (split >= 8) split &= ~7;
    if (split < 1) split = 1;
    TndArgs t{a, lda, b, ldb, c, ldc, M, N, K_cap, k_dyn, a_row_off, b_row_off, tm, tn, split, split % 8 == 0 ? 1 : 0, 0};
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tnd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tnd_lds_bytes());
        attr_done = true;
    }
    hipLaunchKernelGGL(tnd_kernel<4>, dim3(tm * tn * split), dim3(TND_THREADS), tnd_lds_bytes(), st, t);
    return check_launch(what);
}


// ---- the Winograd conv's weight gradient (tndp_kernel)

// slabs the kernel writes for (Dout, Din, P_cap); 0 = not its case (the caller keeps the tile kernel).  Decided from these three
// alone: lego_conv3_wino_du_slabs sizes the slab buffer before the launch sees the row strides.
int tndp_slabs(int Dout, int Din, int P_cap) {
    if (P_cap < 8192 || Dout % TNDP_TM != 0 || Din % TNDP_TN != 0) return 0;
    const unsigned long long widest = (unsigned long long)(Dout > Din ? Dout : Din);
    if (2ull * (unsigned long long)P_cap * widest * 4ull * 4ull >= 0x7FFF0000ull) return 0;     // 31-bit row offsets, strides up to 4 x the width
    const int wgs = tnd_num_cus();
    const int tiles = (Dout / TNDP_TM) * (Din / TNDP_TN);
    int split = wgs / tiles;
    const int max_s = P_cap / 1024 > 0 ? P_cap / 1024 : 1;                  // at least 128 pairs per wave at capacity
    if (split > max_s) split = max_s;
    if (split >= 8) split &= ~7;
    return split < 1 ? 1 : split;
}

int launch_tndp(const float* gy, int ldg, const float* h, int ldh, const int* pair_info, int P_cap, const int* P_dyn, float* du,
                int Dout, int Din, hipStream_t st, const char* what) {
    const int split = tndp_slabs(Dout, Din, P_cap);
    const unsigned long long gb = 2ull * (unsigned long long)P_cap * (unsigned long long)ldg * 4ull, hb = 2ull * (unsigned long long)P_cap * (unsigned long long)ldh * 4ull;
    if (split < 1 || gb >= 0x7FFF0000ull || hb >= 0x7FFF0000ull)
        return set_error("%s: the row strides (%d, %d) exceed what the slab count was sized for", what, ldg, ldh);
    TndpArgs t{gy, ldg, h, ldh, pair_info, P_cap, P_dyn, du, Dout, Din, split, split % 8 == 0 ? 1 : 0, (unsigned)gb, (unsigned)hb};
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tndp_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tndp_lds_bytes());
        attr_done = true;
    }
    hipLaunchKernelGGL(tndp_kernel<3>, dim3((Dout / TNDP_TM) * (Din / TNDP_TN) * split), dim3(TND_THREADS), tndp_lds_bytes(), st, t);
    return check_launch(what);
}

}  // namespace lego
