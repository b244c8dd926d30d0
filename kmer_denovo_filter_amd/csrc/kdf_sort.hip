// kdf_sort.hip -- ascending key order for kdf_export_ge (deterministic dump
// order).  Not on the hot path: rocPRIM's radix sort is used as a library sort.
// Narrow keys: one pair sort (key -> count).  Wide keys: LSD over the two
// words with an index permutation (rocPRIM's radix sort is stable).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <string>

namespace {

__global__ void iota_kernel(uint32_t *idx, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)i;
}
template <typename T>
__global__ void gather_kernel(const T *__restrict__ src, const uint32_t *__restrict__ idx, T *__restrict__ dst, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

struct Bufs {
    void *p[8] = {};
    ~Bufs() { for (void *q : p) if (q) (void)hipFree(q); }
};

#define SCHK(call)                                                         \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) {                   \
        err = std::string(#call) + ": " + hipGetErrorString(e_); return 1; } } while (0)

}  // namespace

int kdf_sort_pairs_device(uint64_t *d_lo, uint64_t *d_hi, uint32_t *d_cnt, uint64_t n,
                          hipStream_t stream, std::string &err) {
    if (n < 2) return 0;
    if (n > 0xFFFFFFFFull && d_hi) { err = "wide export larger than 2^32 entries"; return 1; }
    Bufs b;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (!d_hi) {
        SCHK(hipMalloc(&b.p[0], n * 8));
        SCHK(hipMalloc(&b.p[1], n * 4));
        size_t tmp = 0;
        SCHK(rocprim::radix_sort_pairs(nullptr, tmp, d_lo, (uint64_t *)b.p[0], d_cnt, (uint32_t *)b.p[1], n, 0, 64, stream));
        SCHK(hipMalloc(&b.p[2], tmp ? tmp : 8));
        SCHK(rocprim::radix_sort_pairs(b.p[2], tmp, d_lo, (uint64_t *)b.p[0], d_cnt, (uint32_t *)b.p[1], n, 0, 64, stream));
        SCHK(hipMemcpyAsync(d_lo, b.p[0], n * 8, hipMemcpyDeviceToDevice, stream));
        SCHK(hipMemcpyAsync(d_cnt, b.p[1], n * 4, hipMemcpyDeviceToDevice, stream));
        SCHK(hipStreamSynchronize(stream));
        return 0;
    }
    uint32_t *idx0, *idx1; uint64_t *k0, *k1;
    SCHK(hipMalloc(&b.p[0], n * 4)); idx0 = (uint32_t *)b.p[0];
    SCHK(hipMalloc(&b.p[1], n * 4)); idx1 = (uint32_t *)b.p[1];
    SCHK(hipMalloc(&b.p[2], n * 8)); k0 = (uint64_t *)b.p[2];
    SCHK(hipMalloc(&b.p[3], n * 8)); k1 = (uint64_t *)b.p[3];
    hipLaunchKernelGGL(iota_kernel, dim3(blocks), dim3(256), 0, stream, idx0, n);
    size_t tmp = 0;
    SCHK(rocprim::radix_sort_pairs(nullptr, tmp, d_lo, k0, idx0, idx1, n, 0, 64, stream));
    SCHK(hipMalloc(&b.p[4], tmp ? tmp : 8));
    // pass 1: by lo
    SCHK(rocprim::radix_sort_pairs(b.p[4], tmp, d_lo, k0, idx0, idx1, n, 0, 64, stream));
    // pass 2: by hi (stable), carrying the permutation
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(blocks), dim3(256), 0, stream, (const uint64_t *)d_hi, (const uint32_t *)idx1, k0, n);
    SCHK(rocprim::radix_sort_pairs(b.p[4], tmp, k0, k1, idx1, idx0, n, 0, 64, stream));
    // k1 = sorted hi, idx0 = final permutation
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(blocks), dim3(256), 0, stream, (const uint64_t *)d_lo, (const uint32_t *)idx0, k0, n);
    SCHK(hipMalloc(&b.p[5], n * 4));
    hipLaunchKernelGGL(gather_kernel<uint32_t>, dim3(blocks), dim3(256), 0, stream, (const uint32_t *)d_cnt, (const uint32_t *)idx0, (uint32_t *)b.p[5], n);
    SCHK(hipMemcpyAsync(d_lo, k0, n * 8, hipMemcpyDeviceToDevice, stream));
    SCHK(hipMemcpyAsync(d_hi, k1, n * 8, hipMemcpyDeviceToDevice, stream));
    SCHK(hipMemcpyAsync(d_cnt, b.p[5], n * 4, hipMemcpyDeviceToDevice, stream));
    SCHK(hipStreamSynchronize(stream));
    return 0;
}
