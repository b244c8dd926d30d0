// micro-benchmark: per-CU throughput of random-address LDS operations (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(512) void k(uint32_t *out, int iters, uint32_t seed) {
    __shared__ uint64_t tl[4096];
    __shared__ uint32_t tc[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) { tl[i] = ~0ull; tc[i] = 0; }
    __syncthreads();
    uint32_t x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            const uint32_t sl = (x >> 12) & 4095;
            if (OP == 0) acc += tl[sl];                                         // ds_read_b64
            if (OP == 1) atomicAdd(&tc[sl], 1u);                                // ds_add_u32 (no return)
            if (OP == 2) acc += atomicAdd(&tc[sl], 1u);                         // ds_add_rtn_u32
            if (OP == 3) acc += atomicCAS((unsigned long long *)&tl[sl], ~0ull, (unsigned long long)x);   // ds_cmpst_rtn_b64
            if (OP == 4) { acc += tl[sl]; atomicAdd(&tc[sl], 1u); }             // read + add
            if (OP == 5) tc[sl] = x;                                            // ds_write_b32
            if (OP == 6) acc += tc[sl];                                         // ds_read_b32
        }
    }
    __syncthreads();
    if (acc == 0x1234567) out[0] = (uint32_t)acc;
    if (threadIdx.x == 0) out[1 + (blockIdx.x & 1023)] = tc[blockIdx.x & 4095];
}

template <int OP>
int run(const char *name, uint32_t *d) {
    const int iters = 2000, blocks = 256 * 3 * 4;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d, 10, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d, iters, 7u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * 512 * iters * 8;
    const double wave_ops_per_cu = ops / 64 / 256;
    printf("%-22s %8.3f ms  %7.1f Gop/s chip  %6.2f clk/wave-instr/CU (2.4 GHz)\n", name, ms, ops / ms / 1e6,
           ms * 1e-3 * 2.4e9 / wave_ops_per_cu);
    return 0;
}

int main() {
    uint32_t *d; CHECK(hipMalloc(&d, 8192));
    run<0>("ds_read_b64", d); run<6>("ds_read_b32", d); run<5>("ds_write_b32", d);
    run<1>("ds_add_u32", d); run<2>("ds_add_rtn_u32", d); run<3>("ds_cmpst_rtn_b64", d); run<4>("read_b64+add_u32", d);
    return 0;
}
