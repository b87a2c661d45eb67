// Shared device helpers for the lego_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lego {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// rowinfo word of the token-row space (built by the plan kernels, plan.hip)
//   bit0 = row has a left neighbour inside its item, bit1 = right neighbour,
//   bit2 = row is live (mask == 1), bits 8.. = item-instance index
constexpr int RI_LEFT = 1, RI_RIGHT = 2, RI_LIVE = 4, RI_INST_SHIFT = 8;

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based RNG for the three dropout sites; the mask is never stored, forward and
// backward regenerate it from (seed, site, element counter).
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // (as ONE 32 x 32 -> 64 product each: v_mad_u64_u32; written as __umulhi + a 32-bit product hipcc emits v_mul_hi_u32 + v_mul_lo_u32)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

struct Dropout {
    float p;            // drop probability; 0 disables
    uint32_t seed_lo, seed_hi;
    uint32_t site;      // dropout site id (distinct streams per site / step)
    const uint8_t* mask;   // optional: keep bits precomputed by dropout_mask_kernel (same bits the Philox path draws):
                           // byte [(row / 4) * ncols + col], bit i = row % 4 -- the epilogues then skip the RNG
};

// The keep decisions of a dropout site: ONE Philox4x32-10 call per (group of 8 rows, column) -- counter (row / 8) * ncols + col --
// yields eight 16-bit fields; row 8 * (row / 8) + f takes the low (f < 4) or high (f >= 4) half of word f & 3 and is kept iff that
// field >= floor(p * 65536) (P(drop) = p to within 1.5e-5; the kept values are scaled by 1 / (1 - p) as nn.Dropout does).  Round 1
// spent a whole call on 4 rows and compared 32-bit words: the keep-bit kernels of the prefetch stream were 2 x 17 us of VALU work
// per NAML step.
__device__ __forceinline__ uint32_t dropout_draw8(const Dropout& d, int g8, int c, int ncols) {      // bit f = row 8 * g8 + f kept
    const uint64_t ctr = (uint64_t)g8 * (uint64_t)ncols + (uint64_t)c;
    const Philox4 r = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), d.site, 0u, d.seed_lo, d.seed_hi);
    const uint32_t thr = (uint32_t)(d.p * 65536.0f);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    uint32_t bits = 0u;
#pragma unroll
    for (int f = 0; f < 8; ++f) bits |= (((w[f & 3] >> (16 * (f >> 2))) & 0xFFFFu) >= thr ? 1u : 0u) << f;
    return bits;
}
// the four decisions of rows r0 .. r0 + 3 (r0 % 4 == 0): the same call, one half of every word (a shift, no select: the Winograd
// epilogue has no registers to spare)
__device__ __forceinline__ uint32_t dropout_draw4(const Dropout& d, int r0, int c, int ncols) {     // bit i = row r0 + i kept
    const uint64_t ctr = (uint64_t)(r0 >> 3) * (uint64_t)ncols + (uint64_t)c;
    const Philox4 r = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), d.site, 0u, d.seed_lo, d.seed_hi);
    const uint32_t thr = (uint32_t)(d.p * 65536.0f);
    const uint32_t sh = (uint32_t)(r0 & 4) << 2;          // 0 or 16
    return (((r.x >> sh) & 0xFFFFu) >= thr ? 1u : 0u) | (((r.y >> sh) & 0xFFFFu) >= thr ? 2u : 0u) |
           (((r.z >> sh) & 0xFFFFu) >= thr ? 4u : 0u) | (((r.w >> sh) & 0xFFFFu) >= thr ? 8u : 0u);
}

// keep-scales for the 4 elements (rows r0..r0+3, r0 % 4 == 0, column c) of a [*, ncols] matrix
__device__ __forceinline__ void dropout_scale4(const Dropout& d, int r0, int c, int ncols, float (&s)[4]) {
    if (d.p <= 0.f) { s[0] = s[1] = s[2] = s[3] = 1.f; return; }
    const uint32_t bits = d.mask != nullptr ? (uint32_t)d.mask[(uint64_t)(r0 >> 2) * (uint64_t)ncols + (uint64_t)c] : dropout_draw4(d, r0, c, ncols);
    const float inv = 1.f / (1.f - d.p);
    s[0] = (bits & 1u) ? inv : 0.f; s[1] = (bits & 2u) ? inv : 0.f;
    s[2] = (bits & 4u) ? inv : 0.f; s[3] = (bits & 8u) ? inv : 0.f;
}

// the same decisions as 4 keep bits (bit i = row r0 + i); with a precomputed mask this is ONE byte load, which the
// epilogues issue for their whole tile up front (a load placed between the stores of a tile cannot be hoisted by
// the compiler and would pay a full memory latency per fragment)
__device__ __forceinline__ uint32_t dropout_bits4(const Dropout& d, int r0, int c, int ncols) {
    if (d.p <= 0.f) return 15u;
    if (d.mask != nullptr) return d.mask[(uint64_t)(r0 >> 2) * (uint64_t)ncols + (uint64_t)c];
    return dropout_draw4(d, r0, c, ncols);
}

__device__ __forceinline__ float dropout_scale1(const Dropout& d, int r, int c, int ncols) {
    float s[4];
    dropout_scale4(d, r & ~3, c, ncols, s);
    return s[r & 3];
}

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp / reciprocal: absolute error ~1e-7 (the additive attention's hidden
// layer; libm's tanhf costs ~5x the instructions in the epilogue of a 27 k x 200 product).  Saturates cleanly: exp -> 0
// gives -1, exp -> inf gives 1.
__device__ __forceinline__ float fast_tanh(float x) {
    const float t = __expf(2.f * x);
    return 1.f - 2.f * __frcp_rn(t + 1.f);
}

// exact GELU (torch.nn.functional.gelu, approximate='none'; BertConfig.hidden_act = "gelu"): g = z Phi(z), g' = Phi(z) + z phi(z).
// One definition for the stand-alone kernels (bert_ops.hip) and the fused product epilogues (gemm_epi.hpp): the two forms agree bit for bit.
__device__ __forceinline__ float gelu_exact(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_exact_grad(float z) {
    const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);
    return cdf + z * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace lego

// ---------------------------------------------------------------- in-kernel clock probe (tuning build only: `make tune`)
// MI355X lowers its clock under MFMA-dense load (MI355X_MICROARCH.md 'DVFS give-back'): the clock a kernel really ran at is
// delta(s_memtime) / delta(s_memrealtime) x 100 MHz.  The product library compiles these to nothing.
#ifdef LEGO_TUNING_HOOKS
namespace lego { extern __device__ unsigned long long g_clock_probe[4]; }
#define LEGO_CLOCK_BEGIN unsigned long long lego_c0 = __builtin_amdgcn_s_memtime(), lego_r0 = __builtin_amdgcn_s_memrealtime();
#define LEGO_CLOCK_END(slot) if (threadIdx.x == 0) { \
        atomicAdd(&::lego::g_clock_probe[2 * (slot)], __builtin_amdgcn_s_memtime() - lego_c0); \
        atomicAdd(&::lego::g_clock_probe[2 * (slot) + 1], __builtin_amdgcn_s_memrealtime() - lego_r0); }
#else
#define LEGO_CLOCK_BEGIN
#define LEGO_CLOCK_END(slot)
#endif

// ---------------------------------------------------------------- host-side error plumbing
extern "C" const char* lego_last_error(void);
namespace lego {
int set_error(const char* fmt, ...);
int check_launch(const char* what);
}
#include "../../include/lego_hip.h"
namespace lego {
inline Dropout make_dropout(const lego_dropout* d) {
    if (d != nullptr && d->p > 0.f) return Dropout{d->p, (uint32_t)d->seed, (uint32_t)(d->seed >> 32), d->site, d->mask};
    return Dropout{0.f, 0u, 0u, 0u, nullptr};
}
}
#define LEGO_REQUIRE(cond, ...) do { if (!(cond)) return ::lego::set_error(__VA_ARGS__); } while (0)
