// Which XCD does workgroup i of a 1-D grid land on?  (HW_REG_XCC_ID, gfx940+: bits 3:0 of hardware register 20.)
// hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o tools/xcc_probe && tools/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {
    if (threadIdx.x == 0) {
        unsigned v = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);   // size-1 = 3, offset 0, id 20 (XCC_ID)
        unsigned cu = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);   // HW_ID
        out[2 * blockIdx.x] = (int)v;
        out[2 * blockIdx.x + 1] = (int)cu;
    }
}
int main() {
    for (int threads : {64, 256, 512}) {
        const int n = 64;
        int* d; hipMalloc(&d, 2 * n * sizeof(int));
        probe<<<n, threads>>>(d);
        std::vector<int> h(2 * n); hipMemcpy(h.data(), d, 2 * n * sizeof(int), hipMemcpyDeviceToHost);
        printf("block size %d: XCC of workgroups 0..%d:", threads, n - 1);
        for (int i = 0; i < n; ++i) printf(" %d", h[2 * i]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
