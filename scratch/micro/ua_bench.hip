// micro-benchmark (GPU box): kernel C's entry gather with 8-byte entries (aligned dwordx2) against 6-byte entries read
// as one UNALIGNED dwordx2 per lane (2-byte aligned; the compiler emits global_load_dwordx2 for it), as dword + short,
// and as one 12-byte load per lane for two entries.  Pattern: a wave-instruction reads 64 consecutive entries, 12 in flight
// per lane, runs of ~64 entries scattered over a 9 GB ring like the sorted pieces are.
// build: hipcc --offload-arch=gfx950 -O3 scratch/micro/ua_bench.hip -o gpurun_out/ua_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int EPB = 12, CT = 768;
// run r of a "bucket" b: starts at entry (r * NB + b) * RUN  (pieces interleave the buckets)
template <int MODE>
__global__ __launch_bounds__(CT) void gather(const char *__restrict__ ring, uint64_t *__restrict__ out, uint32_t nb, uint32_t runs, uint32_t run_len) {
    const uint32_t b = blockIdx.x;
    uint64_t acc = 0;
    const uint32_t total = runs * run_len;
    for (uint32_t e0 = 0; e0 < total; e0 += CT * EPB) {
        uint64_t v[EPB];
        const uint32_t wbase = e0 + (threadIdx.x >> 6) * (64 * EPB) + (threadIdx.x & 63);
#pragma unroll
        for (int q = 0; q < EPB; ++q) {
            const uint32_t ei = wbase + 64 * q;
            v[q] = 0;
            if (ei < total) {
                const uint32_t r = ei / run_len, o = ei % run_len;
                const uint64_t ent = ((uint64_t)r * nb + b) * run_len + o;
                if (MODE == 0) v[q] = *(const uint64_t *)(ring + ent * 8);
                else if (MODE == 1) { uint64_t t; __builtin_memcpy(&t, ring + ent * 6, 8); v[q] = t & 0xFFFFFFFFFFFFull; }
                else if (MODE == 2) { uint32_t lo; uint16_t hi; __builtin_memcpy(&lo, ring + ent * 6, 4); __builtin_memcpy(&hi, ring + ent * 6 + 4, 2); v[q] = lo | ((uint64_t)hi << 32); }
            }
        }
#pragma unroll
        for (int q = 0; q < EPB; ++q) acc += v[q] * (q + 1);
    }
    if (acc == 0x1234567) out[b] = acc;
    if (threadIdx.x == 0 && (acc & 0xFFFFF) == 7) out[b] = acc;
}
__global__ void fill(uint64_t *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull; }
// correctness of the unaligned load: entry i of a 6-byte packed array must read back as the 48 bits written
__global__ void check(const char *ring, size_t n, unsigned long long *bad) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t t; __builtin_memcpy(&t, ring + i * 6, 8);
        uint32_t lo; uint16_t hi; __builtin_memcpy(&lo, ring + i * 6, 4); __builtin_memcpy(&hi, ring + i * 6 + 4, 2);
        if ((t & 0xFFFFFFFFFFFFull) != (lo | ((uint64_t)hi << 32))) atomicAdd(bad, 1ull);
    }
}
int main() {
    const uint32_t nb = 131072, runs = 96, run_len = 64;         // 8.9 K entries per bucket, as the bench pass
    const size_t n_ent = (size_t)nb * runs * run_len;            // 805 M entries
    char *ring; uint64_t *out; unsigned long long *bad;
    CK(hipMalloc(&ring, n_ent * 8 + 64)); CK(hipMalloc(&out, nb * 8)); CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    fill<<<4096, 256>>>((uint64_t *)ring, n_ent + 8);
    check<<<4096, 256>>>(ring, n_ent, bad);
    unsigned long long hb = 1; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("unaligned dwordx2 loads vs dword+short on %zu entries: %llu mismatches\n", n_ent, hb);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(a));
            if (mode == 0) gather<0><<<nb, CT>>>(ring, out, nb, runs, run_len);
            if (mode == 1) gather<1><<<nb, CT>>>(ring, out, nb, runs, run_len);
            if (mode == 2) gather<2><<<nb, CT>>>(ring, out, nb, runs, run_len);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        const double bytes = (double)n_ent * (mode == 0 ? 8 : 6);
        printf("mode %d (%s): %.3f ms, %.2f TB/s of entry bytes, %.1f G entries/s\n", mode, mode == 0 ? "8 B aligned" : mode == 1 ? "6 B, one unaligned dwordx2" : "6 B, dword + short",
               best, bytes / best / 1e9, n_ent / best / 1e6);
    }
    // run length: the same bytes in runs of 16 ... 256 entries (8-byte entries, aligned loads; runs start 8-byte aligned only: + 1 entry of skew per run)
    for (uint32_t rl = 16; rl <= 256; rl *= 2) {
        const uint32_t rn = runs * run_len / rl;
        float best = 1e9;
        for (int it = 0; it < 3; ++it) {
            CK(hipEventRecord(a));
            gather<0><<<nb, CT>>>(ring + 8 * (rl == run_len ? 0 : 1), out, nb, rn, rl - (rl == run_len ? 0 : 1));
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        const double bytes = (double)nb * rn * (rl - (rl == run_len ? 0 : 1)) * 8;
        printf("runs of %u entries (%u B): %.3f ms, %.2f TB/s\n", rl - (rl == run_len ? 0 : 1), 8 * (rl - (rl == run_len ? 0 : 1)), best, bytes / best / 1e9);
    }
    return 0;
}
