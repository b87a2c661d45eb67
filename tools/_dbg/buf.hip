#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* p, float* q, int ld, int L, float* res) {
    int lane = threadIdx.x;
    int bytes = ((L - 1) * ld + 8) * 4;
    // 1: shifted descriptor, rows k = 0..3, lane reads column lane%8 of row k + 4*(lane/8 & 1)
    for (int kk = 0; kk < 4; ++kk) {
        int off_f = kk * ld;
        auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + off_f), 0, max(bytes - off_f * 4, 0), 0x00020000);
        int voff = ((4 * ((lane >> 3) & 1)) * ld + (lane & 7)) * 4;
        unsigned b = __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
        res[kk * 64 + lane] = __builtin_bit_cast(float, b);
        auto w = __builtin_amdgcn_make_buffer_rsrc(q + off_f, 0, max(bytes - off_f * 4, 0), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 100.f + kk * 10 + lane), w, voff, 0, 0);
    }
}
int main() {
    int ld = 16, L = 6, rows = 12;
    std::vector<float> h(rows * ld);
    for (int i = 0; i < rows * ld; ++i) h[i] = i;
    float *p, *q, *res;
    hipMalloc(&p, h.size() * 4); hipMalloc(&q, h.size() * 4); hipMalloc(&res, 4 * 64 * 4);
    hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(q, 0, h.size() * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, q, ld, L, res);
    std::vector<float> r(256), qq(h.size());
    hipMemcpy(r.data(), res, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(qq.data(), q, h.size() * 4, hipMemcpyDeviceToHost);
    for (int kk = 0; kk < 4; ++kk) { printf("k=%d:", kk); for (int l = 0; l < 16; ++l) printf(" %g", r[kk * 64 + l]); printf("\n"); }
    for (int rr = 0; rr < rows; ++rr) { printf("q row %d:", rr); for (int c = 0; c < 10; ++c) printf(" %g", qq[rr * ld + c]); printf("\n"); }
    return 0;
}
