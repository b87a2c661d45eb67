// Stable key sort of row indices (rocPRIM radix sort through hipCUB): groups the token rows of a batch by their distinct-token
// index for the de-duplicated projection's weight gradient (misc_ops.hip, lego_segment_sum_rows).  A utility on the prefetch
// stream, off the step's critical path; kept in its own translation unit because the library headers are slow to compile.
#include <hipcub/hipcub.hpp>
#include "common.hpp"

namespace lego {
__global__ void iota_kernel(int* v, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i;
}
}  // namespace lego

extern "C" int64_t lego_sort_rows_temp_bytes(int n) {
    size_t bytes = 0;
    if (n <= 0) return 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr, n) != hipSuccess)
        return -1;
    return (int64_t)(bytes + (size_t)n * sizeof(int) + 256);       // + the iota value array
}

extern "C" int lego_sort_rows(const int32_t* keys, int n, int32_t* keys_sorted, int32_t* perm, void* temp, int64_t temp_bytes, void* stream) {
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t iota_bytes = ((size_t)n * sizeof(int) + 255) & ~(size_t)255;
    LEGO_REQUIRE(temp != nullptr && temp_bytes >= (int64_t)iota_bytes, "lego_sort_rows: temp too small");
    int* iota = reinterpret_cast<int*>(temp);
    hipLaunchKernelGGL(lego::iota_kernel, dim3((n + 255) / 256 < 512 ? (n + 255) / 256 : 512), dim3(256), 0, st, iota, n);
    size_t bytes = (size_t)temp_bytes - iota_bytes;
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(reinterpret_cast<char*>(temp) + iota_bytes, bytes, keys, keys_sorted, iota, perm, n, 0, 32, st);
    if (e != hipSuccess) return lego::set_error("lego_sort_rows: %s", hipGetErrorString(e));
    return lego::check_launch("lego_sort_rows");
}
