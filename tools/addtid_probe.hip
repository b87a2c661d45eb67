// what "TID" is in ds_read_addtid_b32 on gfx950: the lane of the wave or the work-item of the workgroup?  (address = M0 + offset + TID * 4)
//   hipcc --offload-arch=gfx950 -O2 tools/addtid_probe.hip -o /tmp/addtid_probe && /tmp/addtid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) sm[i] = (float)i;
    __syncthreads();
    float a;
    int o = 0;
    asm volatile("s_mov_b32 m0, %1\n s_nop 0\n ds_read_addtid_b32 %0 offset:0\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "s"(o) : "memory");
    out[threadIdx.x] = a;
    float b;
    int o2 = 70000 * 4 / 4 * 1;     // a byte offset past 64 KB: does M0 carry more than 16 bits?
    asm volatile("s_mov_b32 m0, %1\n s_nop 0\n ds_read_addtid_b32 %0 offset:0\n s_waitcnt lgkmcnt(0)" : "=v"(b) : "s"(o2) : "memory");
    out[256 + threadIdx.x] = b;
}
int main() {
    float* d; hipMalloc(&d, 1024 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 32768 * 4, 0, d);
    float h[512]; hipMemcpy(h, d, 512 * 4, hipMemcpyDeviceToHost);
    printf("M0 = 70000 bytes: lane 0 of wave 0 read float %g (17500 = full M0; 1116 = M0 mod 65536)\n", h[256]);
    printf("lane 0 of waves 0..3 read floats %g %g %g %g ; lane 5 of wave 2: %g\n", h[0], h[64], h[128], h[192], h[128 + 5]);
    printf(h[64] == 0.f ? "TID = lane of the wave\n" : (h[64] == 64.f ? "TID = work-item of the workgroup\n" : "TID = something else\n"));
    return 0;
}
