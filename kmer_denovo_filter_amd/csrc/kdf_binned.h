// kdf_binned.h -- the LDS-staged-bucket count pipeline ("binned" path).
//
// The table (kdf_device.h) is an array of buckets of 2^bucket_bits slots; a
// key's bucket is the top bits of its hash.  Random probes into an HBM table
// move a 64-128 B sector per 8 useful bytes and pay a device atomic per window.
// Instead, a batch of reads is processed as
//
//   A0  histogram of the coarse bin (top c1 hash bits) of every valid window
//   A1  extract canonical k-mers again and scatter them to their coarse bin:
//       LDS counting sort per slab -> coalesced run writes
//   B   every CHUNK-entry chunk of a coarse bin is sorted IN PLACE by the next
//       c2 hash bits; a per-chunk offset table locates each fine run
//       (no global fine histogram, immune to multiplicity skew)
//   C   one workgroup per table bucket: bucket slice (keys+counts) lives in
//       LDS, the bucket's runs are gathered from all chunks of its coarse bin
//       and inserted / probed with LDS atomics, the slice is written back once.
//       Transactional per bucket: a bucket that overflows is left untouched in
//       HBM and flagged; the host grows the table and replays those buckets
//       through the global-atomic path (D).
//
// All of it is placement independent: no workgroup reads another workgroup's
// output inside a launch.
#pragma once
#include "kdf_device.h"

#define KB_THREADS   1024
#define KB_F_BITS    8                   // preferred fine radix (level 2)
#define KB_F_BITS_MAX 9                  // used only when the table has more buckets than 2^(C1_MAX+8)
#define KB_F         (1 << KB_F_BITS_MAX)  // LDS array size for the fine histogram
#define KB_C1_MAX    10                  // coarse bins <= 1024
// Bucket kernel: 768 threads = 12 waves per workgroup, two workgroups per CU (LDS) = 6 waves per SIMD,
// which needs <= 80 VGPRs (amdgpu_waves_per_eu below; 78 used with 12 entries in flight per lane).  Measured on
// the bench pass: 512 threads x 20 entries (4 waves per SIMD) 6.16 ms, 768 x 12 5.4 ms, 1024 x 8 (8 per SIMD,
// spills) 6.0 ms; thread counts whose waves do not divide evenly over the four SIMDs (640, 896) leave one
// workgroup per CU (9-10 ms).  12 x 768 = 9216 entries per batch also covers the bench's ~8.9 K entries per
// bucket in one batch with 11.6 of the 12 waves busy.  Past six waves nothing more comes: with two keys resolved
// together instead of four the kernel fits 64 VGPRs without spilling, and 1024 threads x 10 or 12 entries at 8 waves
// per SIMD then take 5.7-5.8 ms; lookahead 1 / 3 instead of 2: 5.9 / 5.4-5.5 ms.
#ifndef KB_C_THREADS
#define KB_C_THREADS 768
#endif
#ifndef KB_C_EPB_N
#define KB_C_EPB_N 12                    // VAR 1, narrow keys: entries per thread and batch (a multiple of 4)
#endif
#ifndef KB_C_EPB_W
#define KB_C_EPB_W 8                     // VAR 1, wide keys (4 measured the same, 12 spills)
#endif
#ifndef KB_C_THREADS_W
#define KB_C_THREADS_W 512                // wide keys: 2048-slot buckets hold ~3 K entries; 512 x 8 covers them and three workgroups fit a CU
#endif
#ifndef KB_C_WQ_W
#define KB_C_WQ_W 48                      // wide keys: queue entries per wave (18 B each)
#endif
#ifndef KB_C_WPE
#define KB_C_WPE 6                       // waves per SIMD the register allocation aims at
#endif
static_assert(KB_C_EPB_N % 4 == 0 && KB_C_EPB_W % 4 == 0, "kernel C resolves entries four at a time");
static_assert(KB_C_THREADS % 256 == 0 && KB_C_THREADS <= 1024 && KB_C_THREADS_W % 256 == 0 && KB_C_THREADS_W <= KB_C_THREADS, "whole waves on every SIMD");
#define KB_C_CT(KW) ((KW) == 2 ? KB_C_THREADS_W : KB_C_THREADS)
#define KB_C_RUNS    256                 // runs (chunks of the coarse bin) staged per round

template <int KW> struct KbCfg;
template <> struct KbCfg<1> { static constexpr int WPT = 16, CHUNK = 16384; };   // 8-byte entries: 128 KB of LDS
// (WPT = 8 for narrow keys -- 8 K slabs, two workgroups per CU at 8 waves per SIMD -- was measured at 8.6 ms for A1
// against 4.6: the runs halve and 64 VGPRs spill.)
template <> struct KbCfg<2> { static constexpr int WPT = 8,  CHUNK = 8192;  };   // 16-byte entries
// Wide entries travel as 16-byte (lo, hi) structs: one dwordx4 / ds_*_b128 per entry instead
// of two 8-byte accesses to two arrays (runs are short: 9 entries in A1, 16 in C).
struct __attribute__((aligned(16))) KbEnt2 { uint64_t lo, hi; };

struct KbPlan {
    uint32_t c1;            // coarse bits
    uint32_t c2;            // fine bits (<= KB_F_BITS)
    uint32_t sub_bits;      // table buckets per partition bucket = 2^sub_bits
    uint32_t log2cap, bucket_bits;
    uint32_t off_stride;    // entries per row of chunk_off = 2^c2 + 1
    uint32_t key_parts, key_part;   // KdfTable::key_parts: windows of other key-space slices are dropped in A0 / A1
    uint32_t dbg;           // experiments only (bucket kernel, plain-loop variant): 1 skip LDS insert, 4 skip write-back
    uint32_t cells;         // 1: the partition was built without a histogram pass (fixed cells); the windows A1 counted wait in totals[5]
    uint32_t cell_stride;   // cells: entries from one cell's base to the next (CHUNK + a pad: cells that are exactly 128 KB apart
                            // advance in lockstep through the same HBM channels)
};

// device scratch shared by the kernels of one pass
struct KbScratch {
    unsigned long long *hist1;      // [2^c1]
    unsigned long long *bin_start;  // [2^c1 + 1]
    uint32_t *hist_wg;              // [n_wg][2^c1] per-workgroup coarse histogram
    uint32_t *wg_base;              // [n_wg][2^c1] exclusive prefix of hist_wg over the workgroups
    unsigned long long *chunk_first;// [2^c1 + 1]
    unsigned long long *totals;     // [4]: n_entries, n_chunks, n_failed, claimed
    unsigned int *failed_flag;      // [1]: set when the scatter pass disagrees with the histogram pass
    uint32_t *chunk_off;            // [n_chunks][2^c2 + 1]
    uint32_t *failed;               // bitmap over TABLE buckets (2^(c1+c2+sub_bits) bits)
    uint64_t *ent_lo;               // entries (keys); wide keys: an array of (lo, hi) pairs, 16 B each (kb_ent2)
    // pool variant of the scatter (kb_scatter2_kernel): no histogram pass, the runs go to 4 KB chunks taken from a pool
    uint64_t *pool;                 // [max_chunks][KB_PCH entries]
    uint32_t *chunk_bin, *chunk_pos, *chunk_fill, *chunk_list;   // [max_chunks]
    uint32_t *bin_nchunks;          // [2^c1]
    uint32_t *bin_chunk_start;      // [2^c1 + 1]
    uint32_t *pool_ctr;             // [0] chunks taken
    uint32_t max_chunks, pad2;
    // heavy buckets of a skewed pass (kb_heavy_slice_kernel): [0] how many, their ids, staged (key, count) pairs
    uint32_t *hv_ctr;               // [4]
    uint32_t *hv_bucket, *hv_n, *hv_failed;   // [KB_HV_MAX]
    uint64_t *hv_key;               // [KB_HV_MAX][KB_HV_SLICES << 12]
    uint32_t *hv_cnt;
};
#ifndef KB_HV_MAX
#define KB_HV_MAX    64u                             // heavy buckets split per pass (further ones are processed the ordinary way)
#endif
#define KB_HV_SLICES 32u                             // workgroups that share one heavy bucket's runs
#ifndef KB_C_HEAVY
#define KB_C_HEAVY   65536u                          // entries (first 256 runs) from which a bucket counts as heavy: ~7x a bucket's share at bench load
#endif
#define KB_GROUP 32                                  // pool chunks per fine-sort group
#define KB_PCH(KW) (KbCfg<KW>::CHUNK / KB_GROUP)     // entries per pool chunk: 4 KB for either key width
#define KB_NOCHUNK 0xFFFFFFFFu

__device__ __forceinline__ uint32_t kb_coarse(const KbPlan &p, uint64_t h) {
    return p.c1 ? (uint32_t)(h >> (64 - p.c1)) : 0u;
}
__device__ __forceinline__ uint32_t kb_fine(const KbPlan &p, uint64_t h) {
    return p.c2 ? (uint32_t)((h >> (64 - p.c1 - p.c2)) & ((1u << p.c2) - 1)) : 0u;
}

// block-wide exclusive scan of n <= KB_THREADS uint32 values held one per
// thread (threads >= n pass 0); returns the exclusive prefix, total via *tot.
__device__ __forceinline__ uint32_t kb_block_exscan(uint32_t v, uint32_t *wsum /* >= 16 words LDS */, uint32_t *tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t s = lane < nw ? wsum[lane] : 0, si = s;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { uint32_t t = __shfl_up(si, o); if (lane >= o) si += t; }
        if (lane < nw) wsum[lane] = si - s;
        if (lane == nw - 1) wsum[16] = si;
    }
    __syncthreads();
    const uint32_t r = wsum[wave] + inc - v;
    if (tot) *tot = wsum[16];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------
// A: thread -> (tile, part).  A tile = 64 window starts; it is split over
// 64/WPT threads so a 1024-thread workgroup covers a slab of 1024*WPT positions.
template <int KW>
struct KbWindows {
    static constexpr int WPT = KbCfg<KW>::WPT;
    static constexpr int NE = KW == 1 ? 2 : 3;
    uint64_t e[NE];       // the stream from this thread's first position on (bit 0 = its first base)
    uint32_t valid;       // WPT bits
    uint64_t kmask;
    int k;
    // The words are pre-shifted once so that window u is taken with COMPILE-TIME
    // shifts (a runtime position would index the word array dynamically).
    // Two-step load so it can be used as a prefetch: issue() only starts the
    // global loads (all independent: no load waits on the mask), finish() turns
    // the raw words into e[] / valid.  The compiler places the vmcnt wait at the
    // first USE, i.e. in finish().
    uint64_t raw[NE + 1], m0, m1;
    uint64_t g0, g1;      // narrow keys: the span with its 2-bit groups reversed, pre-shifted (forward k-mers)
    uint64_t gw[3], hmask;   // wide keys: the same for the 192-bit span, and the mask of the key's high word
    int p0_;
    bool in_range;
    __device__ __forceinline__ void issue(const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid,
                                          uint64_t tile, uint64_t n_tiles, int part, int k_) {
        k = k_; p0_ = part * WPT;
        kmask = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1);
        in_range = tile < n_tiles;
        // unconditional loads (a branch around them would make the compiler wait for
        // them at the join): out-of-range threads read the last tile and are masked
        // off through in_range in finish()
        const uint64_t t = in_range ? tile : n_tiles - 1;
        const uint64_t *src = packed + t * 2 + (p0_ >> 5);
        m0 = invalid[t]; m1 = invalid[t + 1];
#pragma unroll
        for (int i = 0; i <= NE; ++i) raw[i] = src[i];       // within the padded tail (kdf_stream_words)
    }
    __device__ __forceinline__ void finish() {
        const int sh = (p0_ & 31) * 2;
#pragma unroll
        for (int i = 0; i < NE; ++i) e[i] = kdf_funnel(raw[i], raw[i + 1], sh);
        const uint64_t v = kdf_valid_windows(m0, m1, k);
        valid = in_range ? (uint32_t)((v >> p0_) & ((1ull << WPT) - 1)) : 0u;
        if constexpr (KW == 1) {
            // Forward k-mer of window u (MSB-first code) = bits [2(64-u-k), +2k) of F, the
            // 128-bit span with its 2-bit groups reversed: ONE reversal per thread instead of
            // one per window.  F is pre-shifted by the runtime part of that offset so that
            // key(u) shifts by the compile-time 2(WPT-1-u).
            const uint64_t fhi = kdf_rev2(e[0]), flo = kdf_rev2(e[1]);
            const int base = 2 * (64 - k - (WPT - 1));             // 34 (k = 32) .. 118
            g0 = base >= 64 ? (fhi >> (base - 64)) : kdf_funnel(flo, fhi, base);
            g1 = base >= 64 ? 0ull : (fhi >> base);
        } else {
            // Wide keys, the same idea over the 192-bit span (round 1 reversed two words and shifted by runtime amounts
            // PER WINDOW: 127 vector instructions per window in A0 against 39 for narrow keys).  R = the span with its 96
            // two-bit groups reversed = (rev2(e0) : rev2(e1) : rev2(e2)); base j sits at bits 2 (95 - j) of R, so the
            // forward k-mer of window u is (R >> 2 (96 - k - u)) & mask(2k).  R is shifted ONCE by the runtime part
            // 2 (96 - k - (WPT - 1)); key(u) then shifts by the compile-time 2 (WPT - 1 - u).
            hmask = (1ull << (2 * k - 64)) - 1;
            const uint64_t r0 = kdf_rev2(e[2]), r1 = kdf_rev2(e[1]), r2 = kdf_rev2(e[0]);
            const int sh0 = 2 * (96 - k - (WPT - 1));              // 52 (k = 63) .. 112 (k = 33)
            const bool w1 = sh0 >= 64; const int b = sh0 & 63;
            const uint64_t s0 = w1 ? r1 : r0, s1 = w1 ? r2 : r1, s2 = w1 ? 0ull : r2;
            gw[0] = kdf_funnel(s0, s1, b); gw[1] = kdf_funnel(s1, s2, b); gw[2] = b ? (s2 >> b) : s2;
        }
    }
    __device__ __forceinline__ void load(const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid,
                                         uint64_t tile, uint64_t n_tiles, int part, int k_) {
        issue(packed, invalid, tile, n_tiles, part, k_);
        finish();
    }
    __device__ __forceinline__ void key(int u, uint64_t &lo, uint64_t &hi) const {   // u: compile-time
        if constexpr (KW == 1) {
            const uint64_t rc = ~kdf_funnel(e[0], e[1], 2 * u) & kmask;      // reverse complement: ~E (kdf_device.h)
            const uint64_t fwd = kdf_funnel(g0, g1, 2 * (WPT - 1 - u)) & kmask;
            lo = fwd < rc ? fwd : rc; hi = 0;
        } else {
            const uint64_t rlo = ~kdf_funnel(e[0], e[1], 2 * u), rhi = ~kdf_funnel(e[1], e[2], 2 * u) & hmask;   // reverse complement: ~E
            const uint64_t flo = kdf_funnel(gw[0], gw[1], 2 * (WPT - 1 - u)), fhi = kdf_funnel(gw[1], gw[2], 2 * (WPT - 1 - u)) & hmask;
            const bool fw = (fhi < rhi) || (fhi == rhi && flo < rlo);
            lo = fw ? flo : rlo; hi = fw ? fhi : rhi;
        }
    }
};

// A0: persistent workgroups.  Workgroup w owns slabs [w*spw, (w+1)*spw) in BOTH
// passes; it accumulates its coarse-bin histogram in LDS over all its slabs and
// writes ONE row hist_wg[w][bin] (no global atomics).
template <int KW, bool SLICED>
__global__ __launch_bounds__(KB_THREADS) void kb_hist1_kernel(
    const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid, uint64_t n_tiles, int k,
    KbPlan plan, KbScratch s, uint32_t slabs_per_wg)
{
    __shared__ uint32_t hist[(1 << KB_C1_MAX) + 1];           // last = dummy counter of invalid windows
    constexpr int WPT = KbCfg<KW>::WPT, TPT = 64 / WPT;      // threads per tile
    constexpr uint32_t TILES_PER_SLAB = KB_THREADS / TPT;
    const int nb = 1 << plan.c1;
    for (int i = threadIdx.x; i < nb; i += KB_THREADS) hist[i] = 0;
    __syncthreads();
    const uint64_t slab0 = (uint64_t)blockIdx.x * slabs_per_wg;
    for (uint32_t sl = 0; sl < slabs_per_wg; ++sl) {
        const uint64_t tile = (slab0 + sl) * TILES_PER_SLAB + threadIdx.x / TPT;
        if ((slab0 + sl) * TILES_PER_SLAB >= n_tiles) break;
        KbWindows<KW> win;
        win.load(packed, invalid, tile, n_tiles, threadIdx.x % TPT, k);
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            uint64_t lo, hi; win.key(u, lo, hi);
            const uint64_t hsh = kdf_hash(lo, hi);
            const bool ok = ((win.valid >> u) & 1) && (!SLICED || kdf_slice(hsh, plan.key_parts) == plan.key_part);
            const uint32_t bin = ok ? kb_coarse(plan, hsh) : (uint32_t)(1 << KB_C1_MAX);   // dummy counter
            atomicAdd(&hist[bin], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += KB_THREADS) s.hist_wg[(uint64_t)blockIdx.x * nb + i] = hist[i];
}

// column scan: block b = coarse bin b; exclusive prefix over the workgroups ->
// wg_base[w][b] (offset of workgroup w inside bin b) and the bin total hist1[b]
__global__ __launch_bounds__(256) void kb_colscan_kernel(KbPlan plan, KbScratch s, uint32_t n_wg) {
    __shared__ uint32_t wsum[32];
    const int nb = 1 << plan.c1;
    const uint32_t b = blockIdx.x;
    unsigned long long carry = 0;
    for (uint32_t w0 = 0; w0 < n_wg; w0 += 256) {
        const uint32_t w = w0 + threadIdx.x;
        const uint32_t v = w < n_wg ? s.hist_wg[(uint64_t)w * nb + b] : 0;
        uint32_t tot = 0;
        const uint32_t ex = kb_block_exscan(v, wsum, &tot);
        if (w < n_wg) s.wg_base[(uint64_t)w * nb + b] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) s.hist1[b] = carry;
}

// single workgroup: bin starts, chunk layout
__global__ __launch_bounds__(KB_THREADS) void kb_scan1_kernel(KbPlan plan, KbScratch s, uint32_t chunk, KdfCtl *ctl) {
    __shared__ unsigned long long a[(1 << KB_C1_MAX) + 1], c[(1 << KB_C1_MAX) + 1];
    const int nb = 1 << plan.c1;
    if (threadIdx.x == 0) {
        unsigned long long acc = 0, cacc = 0;
        for (int i = 0; i < nb; ++i) {
            a[i] = acc; c[i] = cacc;
            const unsigned long long n = s.hist1[i];
            acc += n; cacc += (n + chunk - 1) / chunk;
        }
        a[nb] = acc; c[nb] = cacc;
        unsigned long long mx = 0;
        for (int i = 0; i < nb; ++i) mx = s.hist1[i] > mx ? s.hist1[i] : mx;
        // skewed: one coarse bin holds more than twice its share (a uniform hash keeps the bins within a few per cent)
        s.totals[7] = (nb > 1 && mx * (unsigned long long)nb > 2 * acc + 65536ull * nb) ? 1ull : 0ull;
        s.totals[0] = acc; s.totals[1] = cacc; s.totals[2] = 0; s.totals[3] = 0; s.totals[4] = 0; s.failed_flag[0] = 0;
        for (int i = 9; i < 16; ++i) s.totals[i] = 0;          // diagnostic stamps
        if (acc) atomicAdd(&ctl->windows[0], acc);
    }
    if (s.hv_ctr) {
        if (threadIdx.x == 0) s.hv_ctr[0] = 0;
        if (threadIdx.x < KB_HV_MAX) { s.hv_n[threadIdx.x] = 0; s.hv_failed[threadIdx.x] = 0; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= nb; i += KB_THREADS) { s.bin_start[i] = a[i]; s.chunk_first[i] = c[i]; }
}

// A1: same slab ownership as A0.  Each workgroup keeps a private cursor per bin
// (bin_start + wg_base, advanced slab by slab): no global atomics, deterministic
// layout.  Per slab: rank the windows with LDS atomics, counting-sort them into
// an LDS image, then copy out run by run (a half-wave per bin) so that every
// (workgroup, bin) run is one contiguous global write.
// Barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0),
// i.e. waits for every outstanding global store; in the persistent scatter
// kernel (one workgroup per CU) that serialises the copy-out's HBM writes with
// the next slab's compute.  The barriers there protect LDS data only.
__device__ __forceinline__ void kb_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// CELLS: no histogram pass ran.  Every (bin, workgroup) pair owns a CELL of CHUNK entries -- chunk number
// bin * gridDim.x + workgroup of the entry buffer -- and the workgroup's cursor of the bin starts at the cell's base; the
// cell's fill goes to hist_wg[chunk] at the end and the valid windows are counted here.  A cell that would overflow
// (a bin far above the mean inside one workgroup's slabs) raises failed_flag: the later stages then do nothing and the
// host redoes the pass with the exact layout (A0 first).
template <int KW, bool SLICED, bool CELLS = false>
__global__ __launch_bounds__(KB_THREADS) void kb_scatter1_kernel(
    const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid, uint64_t n_tiles, int k,
    KbPlan plan, KbScratch s, uint32_t slabs_per_wg, KdfCtl *ctl = nullptr)
{
    constexpr int WPT = KbCfg<KW>::WPT, TPT = 64 / WPT, SLAB = KB_THREADS * WPT;
    constexpr uint32_t TILES_PER_SLAB = KB_THREADS / TPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *slo = (uint64_t *)smem;                                   // [SLAB + 1]: last = trash slot
    KbEnt2 *s2 = (KbEnt2 *)smem;                                        // [SLAB + 1] wide: the image holds (lo, hi) pairs
    KbEnt2 *const ent2 = (KbEnt2 *)s.ent_lo;
    unsigned long long *gcur = (unsigned long long *)(smem + (size_t)(SLAB + 2) * 8 * KW);   // next free entry of this WG per bin
    unsigned long long *gend = gcur + (1 << KB_C1_MAX);                 // [512] end of this WG's range (guard)
    uint32_t *hist = (uint32_t *)(gend + (1 << KB_C1_MAX));             // [bins + 1]: last = dummy counter of invalid windows
    uint32_t *offs = hist + (1 << KB_C1_MAX) + 32;                      // [bins + 1]: offs[DUMMY] = trash slot
    uint32_t *wsum = offs + (1 << KB_C1_MAX) + 32;                      // [32]
    constexpr int DUMMY = 1 << KB_C1_MAX;
    const int nb = 1 << plan.c1;
    if (threadIdx.x == 0) { hist[DUMMY] = 0; offs[DUMMY] = (uint32_t)SLAB; }
    if (CELLS && threadIdx.x == 0) wsum[31] = 0, wsum[30] = 0;          // valid windows of this workgroup (64-bit, lane 63 adds)
    for (int i = threadIdx.x; i < nb; i += KB_THREADS) {
        hist[i] = 0;
        if constexpr (CELLS) {
            const unsigned long long st = ((unsigned long long)i * gridDim.x + blockIdx.x) * plan.cell_stride;
            gcur[i] = st; gend[i] = st + KbCfg<KW>::CHUNK;
        } else {
            const unsigned long long st = s.bin_start[i] + s.wg_base[(uint64_t)blockIdx.x * nb + i];
            gcur[i] = st;
            gend[i] = st + s.hist_wg[(uint64_t)blockIdx.x * nb + i];
        }
    }
    __syncthreads();
    // CELLS: the workgroups are persistent (one per CU) and CLAIM batches of slabs_per_wg slabs from a counter, so a CU
    // that runs slower takes fewer batches (a static split ran as long as the slowest CU: +20 % on the narrow scatter)
    uint64_t slab0 = (uint64_t)blockIdx.x * slabs_per_wg;
    auto claim = [&]() {
        if (threadIdx.x == 0) wsum[28] = (uint32_t)atomicAdd(&s.totals[6], 1ull);
        __syncthreads();
        slab0 = (uint64_t)wsum[28] * slabs_per_wg;
        __syncthreads();
    };
    if constexpr (CELLS) claim();
    constexpr int GL = 2 * WPT;                                         // lanes that copy one bin's run = its mean length (wide keys: 16 lanes instead of 32 took A1 from 8.9 to 7.9 ms)
    const int half = threadIdx.x / GL, lane32 = threadIdx.x % GL;
    constexpr int NHALF = KB_THREADS / GL;
    // Four barriers per slab (a 16-wave workgroup alone on its CU pays the skew of
    // its slowest wave at every barrier): rank | scan by ONE wave | LDS scatter |
    // copy-out + cursor update.  The next slab's input words are fetched before
    // the current slab is processed, so waves do not reach the first barrier
    // skewed by global-load latency.
    for (;;) {
    if (slab0 * TILES_PER_SLAB >= n_tiles) break;                      // uniform
    KbWindows<KW> win;
    win.load(packed, invalid, slab0 * TILES_PER_SLAB + threadIdx.x / TPT, n_tiles, threadIdx.x % TPT, k);
    for (uint32_t sl = 0; sl < slabs_per_wg; ++sl) {
        if ((slab0 + sl) * TILES_PER_SLAB >= n_tiles) break;          // uniform
        KbWindows<KW> nxt;
        {
            // prefetch: loads only; (tile >= n_tiles handles "no next slab")
            const bool more = sl + 1 < slabs_per_wg;
            nxt.issue(packed, invalid, more ? (slab0 + sl + 1) * TILES_PER_SLAB + threadIdx.x / TPT : n_tiles,
                      n_tiles, threadIdx.x % TPT, k);
        }
        // Branch-free ranking: invalid windows (~4 %) go to a dummy counter
        // hist[DUMMY], so the WPT returning LDS atomics issue back to back with
        // ONE wait instead of WPT serialized round trips inside exec branches.
        uint64_t klo[WPT], khi[KW == 2 ? WPT : 1];
        uint32_t br[WPT];                       // bin << 16 | rank  (rank < SLAB <= 16384)
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            uint64_t lo, hi; win.key(u, lo, hi);
            klo[u] = lo; if constexpr (KW == 2) khi[u] = hi;
            const uint64_t hsh = kdf_hash(lo, hi);
            const bool ok = ((win.valid >> u) & 1) && (!SLICED || kdf_slice(hsh, plan.key_parts) == plan.key_part);
            const uint32_t bin = ok ? kb_coarse(plan, hsh) : (uint32_t)DUMMY;
            br[u] = bin << 16;
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) br[u] |= atomicAdd(&hist[br[u] >> 16], 1u) & 0xFFFFu;
        kb_lds_barrier();                                               // B1: all ranks taken
        if (threadIdx.x < 64) {
            // exclusive scan of hist[0..nb) by one wave: each lane owns a contiguous strip
            const int per = (nb + 63) >> 6;                             // 1..16
            const int b0 = threadIdx.x * per;
            uint32_t sum = 0;
            for (int i = 0; i < per; ++i) sum += (b0 + i < nb) ? hist[b0 + i] : 0;
            uint32_t inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if ((int)threadIdx.x >= o) inc += t; }
            uint32_t run = inc - sum;
            for (int i = 0; i < per; ++i) if (b0 + i < nb) { offs[b0 + i] = run; run += hist[b0 + i]; }
            if (CELLS && threadIdx.x == 63) atomicAdd((unsigned long long *)&wsum[30], (unsigned long long)inc);   // lane 63: the slab's valid windows
        }
        kb_lds_barrier();                                               // B2: offsets ready
        {
            // invalid windows land on the trash slot (offs[DUMMY] = SLAB, rank masked off)
            uint32_t pos[WPT];
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const uint32_t bin = br[u] >> 16;
                pos[u] = offs[bin] + ((bin == (uint32_t)DUMMY) ? 0u : (br[u] & 0xFFFF));
            }
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                if constexpr (KW == 2) s2[pos[u]] = KbEnt2{klo[u], khi[u]};
                else slo[pos[u]] = klo[u];
            }
        }
        // retire the prefetched words of the next slab BEFORE any store is issued:
        // vmcnt retires in order, so a later wait for these loads would also wait
        // for every store issued in between
        nxt.finish();
        asm volatile("" :: "v"(nxt.e[0]), "v"(nxt.e[1]), "v"(nxt.valid));
        kb_lds_barrier();                                               // B3: sorted image complete
        for (int bin = half; bin < nb; bin += NHALF) {
            const uint32_t n = hist[bin], o = offs[bin];
            const unsigned long long g = gcur[bin];
            if (g + n > gend[bin]) {           // the stream changed between the passes: never write past the range
                if (lane32 == 0 && n) s.failed_flag[0] = 1;
            } else {
                for (uint32_t i = lane32; i < n; i += GL) {
                    if constexpr (KW == 2) ent2[g + i] = s2[o + i];
                    else s.ent_lo[g + i] = slo[o + i];
                }
            }
            if (lane32 == 0) { gcur[bin] = g + n; hist[bin] = 0; }      // this half-wave owns the bin
        }
        if (threadIdx.x == 0) hist[DUMMY] = 0;
        kb_lds_barrier();                                               // B4: hist is zero, image free (stores still draining)
        win = nxt;
    }
    if constexpr (!CELLS) break;
    claim();
    }
    if constexpr (CELLS) {
        for (int i = threadIdx.x; i < nb; i += KB_THREADS) {
            const unsigned long long chunk = (unsigned long long)i * gridDim.x + blockIdx.x;
            s.hist_wg[chunk] = (uint32_t)(gcur[i] - chunk * plan.cell_stride);
        }
        if (threadIdx.x == 0) {
            // not into ctl->windows yet: if a cell overflowed somewhere the pass is redone (kernel C adds totals[5])
            const unsigned long long w = *(unsigned long long *)&wsum[30];
            if (w) atomicAdd(&s.totals[5], w);
        }
    }
}

// cells: the arrays kernel C reads, for nbins x n_wg cells of CHUNK entries (chunk j of the bins starts at j * CHUNK)
__global__ __launch_bounds__(KB_THREADS) void kb_cellscan_kernel(KbPlan plan, KbScratch s, uint32_t chunk_entries, uint32_t n_wg) {
    const int nb = 1 << plan.c1;
    for (int i = threadIdx.x; i <= nb; i += KB_THREADS) { s.chunk_first[i] = (unsigned long long)i * n_wg; s.bin_start[i] = (unsigned long long)i * n_wg * chunk_entries; }
    if (threadIdx.x == 0) {
        s.totals[0] = 0; s.totals[1] = (unsigned long long)nb * n_wg; s.totals[2] = 0; s.totals[3] = 0; s.totals[4] = 0; s.totals[5] = 0; s.totals[6] = 0; s.totals[7] = 0;
        for (int i = 9; i < 16; ++i) s.totals[i] = 0;             // (totals[8] holds failed_flag: the host cleared it)
    }
}

// B: one workgroup per chunk; in-place sort by fine bin + offset table
template <int KW, bool CELLS = false>
__global__ __launch_bounds__(KB_THREADS) void kb_finesort_kernel(KbPlan plan, KbScratch s)
{
    constexpr int CHUNK = KbCfg<KW>::CHUNK, EPT = CHUNK / KB_THREADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *slo = (uint64_t *)smem;
    KbEnt2 *s2 = (KbEnt2 *)smem;
    KbEnt2 *const ent2 = (KbEnt2 *)s.ent_lo;
    uint32_t *hist = (uint32_t *)(smem + (size_t)CHUNK * 8 * KW);       // [256]
    uint32_t *offs = hist + KB_F;                                        // [256]
    uint32_t *wsum = offs + KB_F;                                        // [32]
    unsigned long long &sh_start = *(unsigned long long *)(wsum + 32);
    uint32_t &sh_len = *(uint32_t *)(wsum + 34);
    const uint64_t chunk = blockIdx.x;
    if (CELLS && s.failed_flag[0]) return;                  // a cell overflowed: the host redoes the pass, nothing may be used
    if (chunk >= s.totals[1]) return;                       // the grid covers the largest possible number of chunks
    if (CELLS) {
        if (threadIdx.x == 0) { sh_start = chunk * (unsigned long long)plan.cell_stride; sh_len = s.hist_wg[chunk]; }
    } else
    if (threadIdx.x == 0) {
        // locate the coarse bin of this chunk: chunk_first is ascending
        const int nb = 1 << plan.c1;
        int lo = 0, hi = nb;            // largest c with chunk_first[c] <= chunk
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s.chunk_first[mid] <= chunk) lo = mid; else hi = mid; }
        const unsigned long long st = s.bin_start[lo] + (chunk - s.chunk_first[lo]) * (unsigned long long)CHUNK;
        const unsigned long long en = s.bin_start[lo + 1];
        sh_start = st;
        sh_len = (uint32_t)((en - st) < (unsigned long long)CHUNK ? (en - st) : (unsigned long long)CHUNK);
    }
    const int nf = 1 << plan.c2;
    for (int i = threadIdx.x; i < KB_F; i += KB_THREADS) hist[i] = 0;
    __syncthreads();
    const unsigned long long start = sh_start;
    const uint32_t len = sh_len;
    uint64_t klo[EPT], khi[KW == 2 ? EPT : 1];
    uint32_t br[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = e * KB_THREADS + threadIdx.x;
        if (i < len) {
            if constexpr (KW == 2) { const KbEnt2 v = ent2[start + i]; klo[e] = v.lo; khi[e] = v.hi; }
            else klo[e] = s.ent_lo[start + i];
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = e * KB_THREADS + threadIdx.x;
        if (i < len) {
            const uint32_t f = kb_fine(plan, kdf_hash(klo[e], KW == 2 ? khi[e] : 0));
            br[e] = (f << 16) | atomicAdd(&hist[f], 1u);
        }
    }
    __syncthreads();
    {
        const uint32_t v = threadIdx.x < nf ? hist[threadIdx.x] : 0;
        const uint32_t ex = kb_block_exscan(v, wsum, nullptr);
        if (threadIdx.x < nf) {
            offs[threadIdx.x] = ex;
            s.chunk_off[chunk * plan.off_stride + threadIdx.x] = ex;
        }
        if (threadIdx.x == 0) s.chunk_off[chunk * plan.off_stride + nf] = len;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = e * KB_THREADS + threadIdx.x;
        if (i < len) {
            const uint32_t pos = offs[br[e] >> 16] + (br[e] & 0xFFFF);
            if constexpr (KW == 2) s2[pos] = KbEnt2{klo[e], khi[e]};
            else slo[pos] = klo[e];
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < len; i += KB_THREADS) {
        if constexpr (KW == 2) {
            // the sorted chunk goes back as chunk-local structure of arrays -- len lo words, then len hi
            // words, in the same 16 * len bytes -- because kernel C's gathers run faster on two 8-byte
            // streams than on 16-byte entries (measured: 15.5 vs 17.0 ms at k = 63)
            const KbEnt2 v = s2[i];
            // (cells may be partly filled anywhere in a bin: kernel C then finds the hi words at + CHUNK, see kb_finesort2)
            s.ent_lo[2 * start + i] = v.lo; s.ent_lo[2 * start + (CELLS ? (uint32_t)CHUNK : len) + i] = v.hi;
        } else s.ent_lo[start + i] = slo[i];
    }
}


// ---------------------------------------------------------------------------
// A1 without A0.  The histogram pass existed only to give every (workgroup, bin) run an exact place in a contiguous
// bin.  Here a workgroup writes its runs of a bin into 4 KB chunks it takes from a global pool (one thread per bin
// takes the chunks a slab needs, all bins at once, next to the scan), so one pass over the stream is enough; the
// fine sort then works on groups of KB_GROUP chunks of one bin (kb_poolscan / kb_chunklist / kb_finesort2 below) and
// leaves the arrays kernel C reads -- chunk_first, bin_start, chunk_off, ent_lo -- with the meaning they always had
// (a "chunk" of C is a group: bin_start[c] = chunk_first[c] * CHUNK, so chunk j of the bins starts at j * CHUNK).
// (the kernel takes only the pointers it needs: the whole KbScratch costs ~40 SGPRs that spill into VGPR lanes,
// and this kernel sits at the 128-VGPR limit of a 1024-thread workgroup)
struct KbPool { uint64_t *pool; uint32_t *chunk_bin, *chunk_pos, *chunk_fill, *bin_nchunks, *pool_ctr; uint32_t max_chunks, c1, key_parts, key_part; };
template <int KW, bool SLICED>
__global__ __launch_bounds__(KB_THREADS) void kb_scatter2_kernel(
    const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid, uint64_t n_tiles, int k,
    KbPool s, uint32_t slabs_per_wg, KdfCtl *ctl)
{
    KbPlan plan{}; plan.c1 = s.c1; plan.key_parts = s.key_parts; plan.key_part = s.key_part;
    constexpr int WPT = KbCfg<KW>::WPT, TPT = 64 / WPT, SLAB = KB_THREADS * WPT, PCH = KB_PCH(KW);
    constexpr uint32_t TILES_PER_SLAB = KB_THREADS / TPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *slo = (uint64_t *)smem;                                   // [SLAB + 1]: last = trash slot
    KbEnt2 *s2 = (KbEnt2 *)smem;
    KbEnt2 *const pool2 = (KbEnt2 *)s.pool;
    uint32_t *cur_chunk = (uint32_t *)(smem + (size_t)(SLAB + 2) * 8 * KW);   // [bins] this workgroup's open chunk of the bin
    uint32_t *cur_fill = cur_chunk + (1 << KB_C1_MAX);                  // [bins]
    uint32_t *nxt = cur_fill + (1 << KB_C1_MAX);                        // [bins] first of the chunks taken for the bin this slab
    uint32_t *hist = nxt + (1 << KB_C1_MAX);                            // [bins + 1]: last = dummy counter of invalid windows
    uint32_t *offs = hist + (1 << KB_C1_MAX) + 32;                      // [bins + 1]: offs[DUMMY] = trash slot
    constexpr int DUMMY = 1 << KB_C1_MAX;
    const int nb = 1 << plan.c1;
    if (threadIdx.x == 0) { hist[DUMMY] = 0; offs[DUMMY] = (uint32_t)SLAB; }
    for (int i = threadIdx.x; i < nb; i += KB_THREADS) { hist[i] = 0; cur_chunk[i] = KB_NOCHUNK; cur_fill[i] = 0; }
    __syncthreads();
    const uint64_t slab0 = (uint64_t)blockIdx.x * slabs_per_wg;
    constexpr int GL = 2 * WPT;                                         // lanes that copy one bin's run = its mean length
    const int grp = threadIdx.x / GL, lane_g = threadIdx.x % GL;
    constexpr int NGRP = KB_THREADS / GL;
    unsigned long long nwin = 0;
    KbWindows<KW> win;
    if (slab0 * TILES_PER_SLAB < n_tiles)
        win.load(packed, invalid, slab0 * TILES_PER_SLAB + threadIdx.x / TPT, n_tiles, threadIdx.x % TPT, k);
    for (uint32_t sl = 0; sl < slabs_per_wg; ++sl) {
        if ((slab0 + sl) * TILES_PER_SLAB >= n_tiles) break;          // uniform
        KbWindows<KW> nx;
        {
            const bool more = sl + 1 < slabs_per_wg;
            nx.issue(packed, invalid, more ? (slab0 + sl + 1) * TILES_PER_SLAB + threadIdx.x / TPT : n_tiles,
                     n_tiles, threadIdx.x % TPT, k);
        }
        uint64_t klo[WPT], khi[KW == 2 ? WPT : 1];
        uint32_t br[WPT];                       // bin << 16 | rank
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            uint64_t lo, hi; win.key(u, lo, hi);
            klo[u] = lo; if constexpr (KW == 2) khi[u] = hi;
            const uint64_t hsh = kdf_hash(lo, hi);
            const bool ok = ((win.valid >> u) & 1) && (!SLICED || kdf_slice(hsh, plan.key_parts) == plan.key_part);
            br[u] = (ok ? kb_coarse(plan, hsh) : (uint32_t)DUMMY) << 16;
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) br[u] |= atomicAdd(&hist[br[u] >> 16], 1u) & 0xFFFFu;
        kb_lds_barrier();                                               // B1: all ranks taken
        if (threadIdx.x < 64) {
            const int per = (nb + 63) >> 6;
            const int b0 = threadIdx.x * per;
            uint32_t sum = 0;
            for (int i = 0; i < per; ++i) sum += (b0 + i < nb) ? hist[b0 + i] : 0;
            uint32_t inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if ((int)threadIdx.x >= o) inc += t; }
            uint32_t run = inc - sum;
            for (int i = 0; i < per; ++i) if (b0 + i < nb) { offs[b0 + i] = run; run += hist[b0 + i]; }
            if (threadIdx.x == 63) nwin += inc;                         // the slab's valid windows (lane 63 holds the total)
        }
        kb_lds_barrier();                                               // B2: offsets ready
        {
            uint32_t pos[WPT];
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const uint32_t bin = br[u] >> 16;
                pos[u] = offs[bin] + ((bin == (uint32_t)DUMMY) ? 0u : (br[u] & 0xFFFF));
            }
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                if constexpr (KW == 2) s2[pos[u]] = KbEnt2{klo[u], khi[u]};
                else slo[pos[u]] = klo[u];
            }
        }
        nx.finish();
        asm volatile("" :: "v"(nx.e[0]), "v"(nx.e[1]), "v"(nx.valid));
        // a thread per bin takes the chunks the bin's run needs beyond its open chunk: two independent atomics, all bins
        // at once -- after the LDS scatter, when the slab's keys no longer occupy the registers (the kernel sits at the
        // 128-VGPR limit of a 1024-thread workgroup).  The pool holds one entry per stream position plus every
        // workgroup's open chunks: it cannot run out.
        for (int bin = (int)threadIdx.x; bin < nb; bin += KB_THREADS) {
            const uint32_t n = hist[bin], ch = cur_chunk[bin];
            const uint32_t room = ch == KB_NOCHUNK ? 0u : (uint32_t)PCH - cur_fill[bin];
            uint32_t id0 = KB_NOCHUNK;
            if (n > room) {
                const uint32_t need = (n - room + PCH - 1) / PCH;
                id0 = atomicAdd(&s.pool_ctr[0], need);
                const uint32_t pos0 = atomicAdd(&s.bin_nchunks[bin], need);
                for (uint32_t q = 0; q < need; ++q) if (id0 + q < s.max_chunks) { s.chunk_bin[id0 + q] = (uint32_t)bin; s.chunk_pos[id0 + q] = pos0 + q; }
            }
            nxt[bin] = id0;
        }
        kb_lds_barrier();                                               // B3: sorted image complete, chunks taken
        for (int bin = grp; bin < nb; bin += NGRP) {
            const uint32_t n = hist[bin], o = offs[bin];
            if (n == 0) continue;
            uint32_t ch = cur_chunk[bin], fl = cur_fill[bin], nxc = nxt[bin], done = 0;
            while (done < n) {
                if (ch == KB_NOCHUNK || fl == (uint32_t)PCH) {
                    if (lane_g == 0 && ch != KB_NOCHUNK && ch < s.max_chunks) s.chunk_fill[ch] = PCH;
                    ch = nxc++; fl = 0;
                }
                const uint32_t take = min(n - done, (uint32_t)PCH - fl);
                if (ch < s.max_chunks) {
                    const size_t dst = (size_t)ch * PCH + fl;
                    for (uint32_t i = lane_g; i < take; i += GL) {
                        if constexpr (KW == 2) pool2[dst + i] = s2[o + done + i];
                        else s.pool[dst + i] = slo[o + done + i];
                    }
                }
                done += take; fl += take;
            }
            if (lane_g == 0) { cur_chunk[bin] = ch; cur_fill[bin] = fl; hist[bin] = 0; }
        }
        if (threadIdx.x == 0) hist[DUMMY] = 0;
        kb_lds_barrier();                                               // B4: hist is zero, image free
        win = nx;
    }
    for (int i = threadIdx.x; i < nb; i += KB_THREADS) {
        const uint32_t ch = cur_chunk[i];
        if (ch != KB_NOCHUNK && ch < s.max_chunks) s.chunk_fill[ch] = cur_fill[i];
    }
    if (threadIdx.x == 63 && nwin) atomicAdd(&ctl->windows[(blockIdx.x % KDF_SHARDS) * 16], nwin);
}

// bins -> chunk lists, groups of KB_GROUP chunks; leaves chunk_first / bin_start as kernel C reads them
__global__ __launch_bounds__(KB_THREADS) void kb_poolscan_kernel(KbPlan plan, KbScratch s, uint32_t chunk_entries) {
    __shared__ uint32_t a[(1 << KB_C1_MAX) + 1];
    __shared__ unsigned long long g[(1 << KB_C1_MAX) + 1];
    const int nb = 1 << plan.c1;
    if (threadIdx.x == 0) {
        uint32_t acc = 0; unsigned long long gacc = 0;
        for (int i = 0; i < nb; ++i) {
            a[i] = acc; g[i] = gacc;
            const uint32_t n = s.bin_nchunks[i];
            acc += n; gacc += (n + KB_GROUP - 1) / KB_GROUP;
        }
        a[nb] = acc; g[nb] = gacc;
        s.totals[7] = 0; s.totals[4] = 0; s.totals[0] = (unsigned long long)acc * (chunk_entries / KB_GROUP); s.totals[1] = gacc; s.totals[2] = 0; s.totals[3] = 0; s.failed_flag[0] = 0;
        for (int i = 9; i < 16; ++i) s.totals[i] = 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= nb; i += KB_THREADS) { s.bin_chunk_start[i] = a[i]; s.chunk_first[i] = g[i]; s.bin_start[i] = g[i] * chunk_entries; }
}
__global__ __launch_bounds__(256) void kb_chunklist_kernel(KbScratch s) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= min(s.pool_ctr[0], s.max_chunks)) return;
    s.chunk_list[s.bin_chunk_start[s.chunk_bin[c]] + s.chunk_pos[c]] = c;
}

// B on groups of pool chunks: gather the group's entries, sort them by fine bin in LDS, write the sorted group to
// ent_lo[group * CHUNK ...] (wide keys: lo words, then at + CHUNK the hi words) and its offset table
template <int KW>
__global__ __launch_bounds__(KB_THREADS) void kb_finesort2_kernel(KbPlan plan, KbScratch s)
{
    constexpr int CHUNK = KbCfg<KW>::CHUNK, EPT = CHUNK / KB_THREADS, PCH = KB_PCH(KW);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *slo = (uint64_t *)smem;
    KbEnt2 *s2 = (KbEnt2 *)smem;
    const KbEnt2 *const pool2 = (const KbEnt2 *)s.pool;
    uint32_t *hist = (uint32_t *)(smem + (size_t)CHUNK * 8 * KW);       // [KB_F]
    uint32_t *offs = hist + KB_F;                                        // [KB_F]
    uint32_t *wsum = offs + KB_F;                                        // [40]
    uint32_t *cid = wsum + 40, *cfl = cid + KB_GROUP;                    // [KB_GROUP] chunk ids, fills
    const uint64_t grp = blockIdx.x;
    if (grp >= s.totals[1]) return;                                     // the grid covers the largest possible number of groups
    const int nf = 1 << plan.c2;
    for (int i = threadIdx.x; i < KB_F; i += KB_THREADS) hist[i] = 0;
    if (threadIdx.x < KB_GROUP) {
        const int nbn = 1 << plan.c1;
        int lo = 0, hi = nbn;            // largest bin with chunk_first[bin] <= grp
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s.chunk_first[mid] <= grp) lo = mid; else hi = mid; }
        const uint32_t lc = s.bin_chunk_start[lo] + (uint32_t)(grp - s.chunk_first[lo]) * KB_GROUP + threadIdx.x;
        const bool ok = lc < s.bin_chunk_start[lo + 1];
        const uint32_t id = ok ? s.chunk_list[lc] : 0u;
        cid[threadIdx.x] = id; cfl[threadIdx.x] = ok ? s.chunk_fill[id] : 0u;
    }
    __syncthreads();
    uint64_t klo[EPT], khi[KW == 2 ? EPT : 1];
    uint32_t br[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = e * KB_THREADS + threadIdx.x, c = i / PCH, o = i % PCH;
        br[e] = KB_NOCHUNK;
        if (o < cfl[c]) {
            const size_t src = (size_t)cid[c] * PCH + o;
            if constexpr (KW == 2) { const KbEnt2 v = pool2[src]; klo[e] = v.lo; khi[e] = v.hi; }
            else klo[e] = s.pool[src];
            br[e] = 0;
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        if (br[e] != KB_NOCHUNK) {
            const uint32_t f = kb_fine(plan, kdf_hash(klo[e], KW == 2 ? khi[e] : 0));
            br[e] = (f << 16) | atomicAdd(&hist[f], 1u);
        }
    }
    __syncthreads();
    {
        const uint32_t v = threadIdx.x < nf ? hist[threadIdx.x] : 0;
        uint32_t len = 0;
        const uint32_t ex = kb_block_exscan(v, wsum, &len);
        if (threadIdx.x < nf) { offs[threadIdx.x] = ex; s.chunk_off[grp * plan.off_stride + threadIdx.x] = ex; }
        if (threadIdx.x == 0) { s.chunk_off[grp * plan.off_stride + nf] = len; wsum[39] = len; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        if (br[e] != KB_NOCHUNK) {
            const uint32_t pos = offs[br[e] >> 16] + (br[e] & 0xFFFF);
            if constexpr (KW == 2) s2[pos] = KbEnt2{klo[e], khi[e]};
            else slo[pos] = klo[e];
        }
    }
    __syncthreads();
    const uint32_t len = wsum[39];
    const uint64_t start = grp * (uint64_t)CHUNK;
    for (uint32_t i = threadIdx.x; i < len; i += KB_THREADS) {
        if constexpr (KW == 2) {
            // chunk-local structure of arrays with a FIXED distance between a key's words (a group may be partly
            // filled; kernel C takes min(entries left in the bin, CHUNK) as that distance, and bin_start makes it CHUNK)
            const KbEnt2 v = s2[i];
            s.ent_lo[2 * start + i] = v.lo; s.ent_lo[2 * start + CHUNK + i] = v.hi;
        } else s.ent_lo[start + i] = slo[i];
    }
}

// ---------------------------------------------------------------------------
// C: one workgroup per TABLE bucket.
enum { KB_MODE_INSERT = 0, KB_MODE_FILTERED = 1, KB_MODE_REPLAY = 2 };

__device__ __forceinline__ void kb_lds_sat_add(uint32_t *p, uint32_t add) {
    uint32_t old = atomicAdd(p, add);
    if (old + add < old || old + add == 0xFFFFFFFFu) atomicMax(p, 0xFFFFFFFFu);
}

// count += 1 at LDS slot sl for the lanes with `hit` (the whole wave calls it together).  AGG: when every hit lane names
// the SAME slot -- a key of enormous multiplicity: a homopolymer k-mer took 1.4 % of all windows of a repeat-rich genome, all
// of them in one workgroup -- one lane adds the lot instead of 64 adds serialising on one LDS address (kernel C 14.6 -> 10.2
// ms there, pass 24.8 -> 20.4 ms).  The test costs every wave ~8 instructions per key (+3 % on the kernel for a uniform
// genome), so it lives in its own instantiation of the kernel (VAR 2).  kb_scan1_kernel reports a skewed coarse histogram
// (totals[7]) and the host launches VAR 2 for the passes that FOLLOW a skewed one (a sample is many passes; launching both
// instantiations and letting the device pick cost 0.08 ms per pass for 131 K workgroups that return at once).
template <bool AGG>
__device__ __forceinline__ void kb_count_hits(uint32_t *tcnt, uint32_t sl, bool hit) {
    if constexpr (AGG) {
        const unsigned long long hm = __ballot(hit);
        if (hm == 0) return;
        const uint32_t s0 = (uint32_t)__shfl((int)sl, __ffsll(hm) - 1);
        if (__ballot(hit && sl != s0) == 0) {
            if ((threadIdx.x & 63u) == (uint32_t)(__ffsll(hm) - 1)) atomicAdd(&tcnt[s0], (uint32_t)__popcll(hm));
            return;
        }
    }
    if (hit) atomicAdd(&tcnt[sl], 1u);
}

// MODE_INSERT / MODE_FILTERED: bucket slice staged in LDS.
// MODE_REPLAY: only buckets flagged in s.failed, inserted through the global
// atomic path into table t (which the host has grown since the failed pass;
// `old_plan` is the plan the partition was built with).
// narrow keys, one key: linear probing in the LDS slice from slot `sl`
template <int MODE>
__device__ __forceinline__ void kb_probe_narrow(uint64_t *tlo, uint32_t *tcnt, uint32_t bmask, uint64_t klo, uint32_t sl,
                                                uint32_t &claimed, bool &failed) {
    // FOUR slots per iteration, their reads in flight together: a wave runs as many iterations as its longest probe, and
    // every iteration costs scalar exec-mask bookkeeping on the CU's one scalar unit (round 2: the one-slot loop was 40 %
    // of the super-k-mer bucket kernel's scalar instructions)
    for (uint32_t n = 0; n <= bmask;) {
        uint64_t c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = tlo[(sl + i) & bmask];
        uint32_t f = 4; bool isk = false;
#pragma unroll
        for (int i = 3; i >= 0; --i) { const bool k_ = c[i] == klo, e_ = c[i] == KDF_EMPTY; if (k_ || e_) { f = (uint32_t)i; isk = k_; } }
        if (f == 4) { sl = (sl + 4) & bmask; n += 4; continue; }
        const uint32_t at = (sl + f) & bmask;
        if (!isk) {
            if (MODE != KB_MODE_INSERT) return;                  // FILTERED: absent
            const uint64_t old = atomicCAS((unsigned long long *)&tlo[at], KDF_EMPTY, klo);
            if (old == KDF_EMPTY) { claimed++; isk = true; }
            else if (old == klo) isk = true;
        }
        if (isk) { atomicAdd(&tcnt[at], 1u); return; }
        sl = (at + 1) & bmask; n += f + 1;
    }
    failed = true;
}

// wide keys, one ATTEMPT from slot `sl`: 0 done, 1 bucket full, 2 blocked by a slot another lane is publishing.
// The caller retries blocked keys under a wave-uniform loop: no lane ever waits inside a divergent loop.
template <int MODE>
__device__ __forceinline__ int kb_probe_wide_once(uint64_t *tlo, uint64_t *thi, uint32_t *tcnt, uint32_t bmask,
                                                  uint64_t klo, uint64_t khi, uint32_t sl, uint32_t &claimed) {
    for (uint32_t n = 0; n <= bmask; ++n) {
        uint64_t chi = __hip_atomic_load(&thi[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (chi == KDF_EMPTY && MODE == KB_MODE_INSERT) {
            chi = atomicCAS((unsigned long long *)&thi[sl], KDF_EMPTY, khi | KDF_PENDING);
            if (chi == KDF_EMPTY) {
                __hip_atomic_store(&tlo[sl], klo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&thi[sl], khi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                claimed++;
                atomicAdd(&tcnt[sl], 1u);
                return 0;
            }
        }
        if (chi == KDF_EMPTY) return 0;                           // FILTERED: absent
        if ((chi & ~KDF_PENDING) == khi) {
            if (chi & KDF_PENDING) return 2;
            const uint64_t clo = __hip_atomic_load(&tlo[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (clo == klo) { atomicAdd(&tcnt[sl], 1u); return 0; }
        }
        sl = (sl + 1) & bmask;
    }
    return 1;
}
// all lanes of the wave call this together (`todo`: this lane has a key); returns when every key is placed
template <int MODE>
__device__ __forceinline__ void kb_probe_wide_wave(uint64_t *tlo, uint64_t *thi, uint32_t *tcnt, uint32_t bmask, bool todo,
                                                   uint64_t klo, uint64_t khi, uint32_t sl, uint32_t &claimed, bool &failed) {
    while (__any(todo)) {
        if (todo) {
            const int res = kb_probe_wide_once<MODE>(tlo, thi, tcnt, bmask, klo, khi, sl, claimed);
            if (res != 2) { todo = false; if (res == 1) failed = true; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// VAR 0: every lane probes its key in a loop (a wave pays the longest probe of its
// 64 lanes for every key).  VAR 1 (INSERT / FILTERED): the first KB_C_LA slots of
// the probe sequence are read at once and resolved in straight-line code; the keys
// that need more go to a wave-private queue in LDS (ballot + mbcnt, no atomics, no
// barrier) and are probed densely, one per lane, after the batch.  Measured on the
// bench pass: kernel C 7.6 -> 6.5 ms at k = 31, 16.0 -> 12.7 ms at k = 63.
#ifndef KB_C_LA
#define KB_C_LA    2                   // VAR 1: slots of the probe sequence read up front
#endif
#define KB_C_QCAPK(KW) (((KW) == 2 ? KB_C_WQ_W : 128) * (KB_C_CT(KW) / 64))   // VAR 1: queue entries per workgroup
#define KB_C_QEXTRA(VAR, KW) ((VAR) ? (KB_C_QCAPK(KW) * ((KW) == 2 ? 18 : 10) + 16 + (KB_C_RUNS + 4) * 4) : 0)   // LDS bytes VAR 1 adds
template <int KW, int MODE, int VAR>
__global__ __launch_bounds__(KB_C_CT(KW)) __attribute__((amdgpu_waves_per_eu(KB_C_WPE, KB_C_WPE))) void kb_bucket_kernel(
    KbPlan plan, KbScratch s, KdfTable t, KdfCtl *ctl, int table_nonempty)
{
    constexpr int CHUNK = KbCfg<KW>::CHUNK;
    constexpr uint32_t CT = KB_C_CT(KW), QCAP = KB_C_QCAPK(KW);      // threads and queue entries per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t B = 1u << plan.bucket_bits;
    uint64_t *tlo = (uint64_t *)smem;                         // [B]
    uint64_t *thi = KW == 2 ? tlo + B : nullptr;              // [B] wide
    uint32_t *tcnt = (uint32_t *)(smem + (size_t)B * 8 * KW); // [B]
    uint32_t &sh_failed = tcnt[B], &sh_claimed = tcnt[B + 1];
    uint32_t *wsum = tcnt + B + 2;                            // [32]
    uint32_t *run_pref = wsum + 32;                           // [KB_C_RUNS] exclusive prefix of run lengths
    unsigned long long *run_first = (unsigned long long *)(run_pref + KB_C_RUNS);   // [KB_C_RUNS] (B + 34 + KB_C_RUNS is even: 8-aligned)
    uint32_t *run_hi = (uint32_t *)(run_first + KB_C_RUNS);   // [KB_C_RUNS] wide keys only: distance (words) from a run's lo words to its hi words
    uint64_t *qk = (uint64_t *)(run_hi + (KW == 2 ? KB_C_RUNS : 0));   // [QCAP] VAR 1: keys whose probe goes past the lookahead (per wave: QCAP / 8)
    uint16_t *qs = (uint16_t *)(qk + QCAP);              // [QCAP] slot to go on from
    uint64_t *qk2 = (uint64_t *)((uint32_t *)(qs + QCAP) + 2 + KB_C_RUNS + 4);   // [QCAP] wide keys: hi words of the queued keys
    uint32_t *rpw = (uint32_t *)(qs + QCAP) + 2;                                  // [KB_C_RUNS + 4] VAR >= 1: run_pref shifted by one, padded with `total`

    // `plan` describes the table the partition was built for.  In MODE_REPLAY
    // that is the OLD geometry (the host has grown the table since) and `t` is
    // the new table.
    // Workgroups are dealt to the 8 XCDs round robin by blockIdx, and each XCD has its own L2.  Neighbouring buckets
    // (f, f + 1 of one coarse bin) read neighbouring runs of the same chunks -- they share the cache line at every run
    // boundary and the lines of the offset table -- so an XCD takes a contiguous eighth of the buckets, in order.
    if (s.failed_flag[0]) return;                              // the partition is not usable (a cell overflowed / the stream changed): the host knows

    if (plan.cells && MODE != KB_MODE_REPLAY && blockIdx.x == 0 && threadIdx.x == 0 && s.totals[5])
        atomicAdd(&ctl->windows[0], s.totals[5]);              // the valid windows the cell scatter counted
    const uint32_t nbk = gridDim.x;
    const uint64_t bucket = (nbk & 7) ? blockIdx.x : (uint64_t)(blockIdx.x & 7) * (nbk >> 3) + (blockIdx.x >> 3);   // table bucket of `plan`
    const uint64_t pb = bucket >> plan.sub_bits;              // partition bucket holding its entries
    const uint32_t c = (uint32_t)(pb >> plan.c2), f = (uint32_t)(pb & ((1u << plan.c2) - 1));
    if constexpr (MODE == KB_MODE_REPLAY) {
        if (!((s.failed[bucket >> 5] >> (bucket & 31)) & 1)) return;
    }
    const uint64_t slot0 = bucket << plan.bucket_bits;        // first slot of the bucket in HBM
    if (threadIdx.x == 0) { sh_failed = 0; sh_claimed = 0; }
    if constexpr (MODE != KB_MODE_REPLAY) {
        if (table_nonempty) {
            for (uint32_t i = threadIdx.x; i < B; i += CT) {
                tlo[i] = t.lo[slot0 + i];
                if constexpr (KW == 2) thi[i] = t.hi[slot0 + i];
                tcnt[i] = t.cnt[slot0 + i];
            }
        } else {
            // two slots per lane and step: 16-byte LDS writes (B is even, the arrays are 16-byte aligned)
            const ulonglong2 e2 = {KDF_EMPTY, KDF_EMPTY};
            for (uint32_t i = threadIdx.x; i < B / 2; i += CT) {
                ((ulonglong2 *)tlo)[i] = e2;
                if constexpr (KW == 2) ((ulonglong2 *)thi)[i] = e2;
                ((uint2 *)tcnt)[i] = uint2{0u, 0u};
            }
        }
    }
    __syncthreads();

    constexpr uint32_t WQ = QCAP / (CT / 64);
    uint64_t *wqk = qk + (threadIdx.x >> 6) * WQ;
    uint64_t *wqk2 = qk2 + (threadIdx.x >> 6) * WQ;
    uint16_t *wqs = qs + (threadIdx.x >> 6) * WQ;
    uint32_t wq_n = 0;
    const unsigned long long j0 = s.chunk_first[c], j1 = s.chunk_first[c + 1];
    const unsigned long long bstart = s.bin_start[c], bend = s.bin_start[c + 1];
    const uint32_t bmask = B - 1;
    uint32_t claimed = 0;
    bool failed = false;
    // Runs of this bucket: one per chunk of its coarse bin.  Their bounds are
    // fetched by all threads at once (one global latency, not one per run) and
    // laid out in LDS as a flat work list; threads then take entries round
    // robin, so every lane is busy whatever the run lengths are.
    for (unsigned long long jb = j0; jb < j1; jb += KB_C_RUNS) {
        uint32_t len = 0, hioff = 0; unsigned long long first = 0;
        {
            const unsigned long long j = jb + threadIdx.x;
            if (threadIdx.x < KB_C_RUNS && j < j1) {
                const uint32_t r0 = s.chunk_off[j * plan.off_stride + f], r1 = s.chunk_off[j * plan.off_stride + f + 1];
                len = r1 - r0;
                const unsigned long long cs = bstart + (j - j0) * (unsigned long long)(plan.cells ? plan.cell_stride : (uint32_t)CHUNK);      // first entry of the chunk
                if constexpr (KW == 2) {                      // wide: word index of the run's lo words; the hi words follow the chunk's lo words
                    const unsigned long long left = bend - cs;
                    hioff = (uint32_t)(left < (unsigned long long)CHUNK ? left : (unsigned long long)CHUNK);
                    first = 2 * cs + r0;
                } else first = cs + r0;
            }
        }
        uint32_t total = 0;
        const uint32_t ex = kb_block_exscan(len, wsum, &total);
        if constexpr (VAR == 2 && KW == 1 && MODE == KB_MODE_INSERT) {
            // A heavy bucket (a few keys of enormous multiplicity) is not for ONE workgroup: it is left untouched here, as
            // a failed bucket would be, and KB_HV_SLICES workgroups share its runs afterwards (kb_heavy_slice_kernel).
            if (jb == j0 && total > KB_C_HEAVY && s.hv_ctr && !plan.cells && plan.sub_bits == 0 && plan.bucket_bits == 12) {   // (the host launches the heavy kernels under the same conditions)
                if (threadIdx.x == 0) {
                    const uint32_t idx = atomicAdd(&s.hv_ctr[0], 1u);
                    sh_failed = idx;                          // (borrowed as a broadcast word; restored below)
                    if (idx < KB_HV_MAX) s.hv_bucket[idx] = (uint32_t)bucket;
                }
                __syncthreads();
                const bool taken = sh_failed < KB_HV_MAX;
                __syncthreads();
                if (threadIdx.x == 0) sh_failed = 0;
                if (taken) {
                    if (!table_nonempty)
                        for (uint32_t i = threadIdx.x; i < B; i += CT) { t.lo[slot0 + i] = KDF_EMPTY; t.cnt[slot0 + i] = 0; }
                    return;
                }
                __syncthreads();
            }
        }
        if (threadIdx.x < KB_C_RUNS) { run_pref[threadIdx.x] = ex; run_first[threadIdx.x] = first; if constexpr (KW == 2) run_hi[threadIdx.x] = hioff; }
        if constexpr (VAR >= 1) {
            if (threadIdx.x < KB_C_RUNS + 3) rpw[threadIdx.x + 1] = ex;     // threads past the last run hold ex == total
            if (threadIdx.x == 0) rpw[0] = 0;
        }
        __syncthreads();
        const uint32_t nruns = (uint32_t)((j1 - jb) < (unsigned long long)KB_C_RUNS ? (j1 - jb) : (unsigned long long)KB_C_RUNS);
        constexpr int EPB = VAR >= 1 ? (KW == 2 ? KB_C_EPB_W : KB_C_EPB_N) : 12;    // entries per thread per batch: EPB (x KW) loads in flight per lane
        // (ei * inv_total) >> 32 ~= ei * nruns / total
        const unsigned long long inv_total = total ? (((unsigned long long)nruns << 32) / total) : 0;
        for (uint32_t e0 = 0; e0 < total; e0 += CT * EPB) {   // wave-uniform trip count
          uint64_t bklo[EPB], bkhi[KW == 2 ? EPB : 1];
          // Flat index of this thread's q-th entry of the batch.  Narrow keys (VAR 1): a WAVE takes 64 * EPB
          // consecutive entries, lane l's q-th entry is wbase + 64 q -- one load instruction still reads 64
          // consecutive entries, and a lane's consecutive entries are about one run (64 entries) further on,
          // so the run is searched once per batch and then only advanced.  The per-entry search was a
          // quarter of this kernel's vector instructions, and the kernel is bound by those (6.45 -> 6.2 ms).
          // Wide keys have 16-entry runs (four runs per step): they keep the per-entry windowed search.
          constexpr bool WAVE_SPANS = VAR >= 1 && KW == 1;
          const uint32_t wbase = WAVE_SPANS ? e0 + (threadIdx.x >> 6) * (64 * EPB) + (threadIdx.x & 63) : e0 + threadIdx.x;
          constexpr uint32_t QSTEP = WAVE_SPANS ? 64u : (uint32_t)CT;      // flat-index distance between a thread's consecutive entries
          if constexpr (WAVE_SPANS) {
            uint32_t cr = 0, cpf = 0, cnx = 0;                 // current run: index, first flat entry, first entry of the next run
            if (wbase < total) {
                const uint32_t ei = wbase;
                uint32_t gu = (uint32_t)(((unsigned long long)ei * inv_total) >> 32);
                gu = gu < nruns ? gu : nruns - 1;
                const uint32_t w0 = rpw[gu], w1 = rpw[gu + 1], w2 = rpw[gu + 2], w3 = rpw[gu + 3];
                const uint32_t cnt = (w1 <= ei) + (w2 <= ei) + (w3 <= ei);
                uint32_t lo_ = gu + cnt - 1;                                 // cnt == 0: the run before the guess
                const bool sure = cnt == 0 ? (w0 <= ei && gu > 0) : cnt < 3;
                if (!sure) {                                                  // outside the window (rare): walk
                    lo_ = gu;
                    while (run_pref[lo_] > ei) --lo_;
                    while (lo_ + 1 < nruns && run_pref[lo_ + 1] <= ei) ++lo_;
                }
                cr = lo_; cpf = rpw[cr + 1]; cnx = rpw[cr + 2];
            }
            unsigned long long cf = run_first[cr];
#pragma unroll
            for (int q = 0; q < EPB; ++q) {
                const uint32_t ei = wbase + QSTEP * q;
                bklo[q] = 0;
                if (ei < total) {
                    if (ei >= cnx) {                                          // (entries beyond the last run read `total`: the loop ends)
                        do { ++cr; cpf = cnx; cnx = rpw[cr + 2]; } while (ei >= cnx);
                        cf = run_first[cr];
                    }
                    bklo[q] = s.ent_lo[cf + (ei - cpf)];
                }
            }
          } else
          if constexpr (VAR >= 1) {
            // windowed search: the guess is within a run or two of the answer, so read
            // run_pref[guess-1 .. guess+2] for four entries at once and count -- two LDS
            // round trips per four entries instead of a dependent probe chain per entry
#pragma unroll
            for (int q0 = 0; q0 < EPB; q0 += 4) {
                if (e0 + q0 * CT >= total) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) { bklo[q0 + g] = 0; if constexpr (KW == 2) bkhi[q0 + g] = 0; }
                    continue;
                }
                uint32_t w[4][4], gs[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t ei = wbase + QSTEP * (q0 + g);
                    uint32_t gu = (uint32_t)(((unsigned long long)ei * inv_total) >> 32);
                    gu = ei < total ? (gu < nruns ? gu : nruns - 1) : 0;
                    gs[g] = gu;
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[g][i] = rpw[gu + i];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(w[g][i]));
                uint32_t lo4[4], pf4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t ei = wbase + QSTEP * (q0 + g);
                    const uint32_t cnt = (w[g][1] <= ei) + (w[g][2] <= ei) + (w[g][3] <= ei);
                    uint32_t lo_ = gs[g] + cnt - 1;                       // cnt == 0: the run before the guess
                    uint32_t pf = cnt == 0 ? w[g][0] : cnt == 1 ? w[g][1] : cnt == 2 ? w[g][2] : w[g][3];
                    const bool sure = cnt == 0 ? (w[g][0] <= ei && gs[g] > 0) : cnt < 3;
                    if (ei < total && !sure) {                             // outside the window (rare): walk
                        lo_ = gs[g];
                        while (run_pref[lo_] > ei) --lo_;
                        while (lo_ + 1 < nruns && run_pref[lo_ + 1] <= ei) ++lo_;
                        pf = run_pref[lo_];
                    }
                    lo4[g] = ei < total ? lo_ : 0; pf4[g] = pf;
                }
                unsigned long long rf[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) rf[g] = run_first[lo4[g]];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t ei = wbase + QSTEP * (q0 + g);
                    bklo[q0 + g] = 0; if constexpr (KW == 2) bkhi[q0 + g] = 0;
                    if (ei < total) {
                        const unsigned long long src = rf[g] + (ei - pf4[g]);
                        bklo[q0 + g] = s.ent_lo[src];
                        if constexpr (KW == 2) bkhi[q0 + g] = s.ent_lo[src + run_hi[lo4[g]]];
                    }
                }
            }
          } else {
#pragma unroll
          for (int q = 0; q < EPB; ++q) {
            const uint32_t ei = e0 + q * CT + threadIdx.x;
            bklo[q] = 0; if constexpr (KW == 2) bkhi[q] = 0;
            if (ei < total) {
                // largest r with run_pref[r] <= ei.  Runs of a bucket have nearly equal
                // lengths (hash-uniform), so interpolate and correct by a step or two.
                uint32_t lo_ = (uint32_t)(((unsigned long long)ei * inv_total) >> 32);
                if (lo_ >= nruns) lo_ = nruns - 1;
                while (run_pref[lo_] > ei) --lo_;
                while (lo_ + 1 < nruns && run_pref[lo_ + 1] <= ei) ++lo_;
                const unsigned long long src = run_first[lo_] + (ei - run_pref[lo_]);
                if constexpr (KW == 2) { bklo[q] = s.ent_lo[src]; bkhi[q] = s.ent_lo[src + run_hi[lo_]]; }
                else bklo[q] = s.ent_lo[src];
            }
          }
          }
          if constexpr (VAR >= 1 && KW == 1 && MODE != KB_MODE_REPLAY) {
            constexpr int G = 4;                                 // keys resolved together: G * KB_C_LA LDS reads in flight
#pragma unroll
            for (int q0 = 0; q0 < EPB; q0 += G) {
                if (wbase - (threadIdx.x & 63) + QSTEP * q0 >= total) break;      // wave-uniform: nothing left for this wave in this batch
                uint64_t cur[G][KB_C_LA]; uint32_t sl0[G]; bool td[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t ei = wbase + QSTEP * (q0 + g);
                    const uint64_t home = kdf_hash(bklo[q0 + g], 0) >> (64 - plan.log2cap);
                    td[g] = ei < total && !(plan.sub_bits && (home >> plan.bucket_bits) != bucket);
                    sl0[g] = (uint32_t)home & bmask;
#pragma unroll
                    for (int i = 0; i < KB_C_LA; ++i) cur[g][i] = tlo[(sl0[g] + i) & bmask];
                }
                // pin the loads here: all G * KB_C_LA reads are issued before the first
                // key is resolved (the compiler would otherwise sink each into its use)
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < KB_C_LA; ++i) asm volatile("" : "+v"(cur[g][i]));
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint64_t klo = bklo[q0 + g];
                    // first slot of the lookahead that holds the key or is empty.  A slot
                    // read as EMPTY may have been taken since: the CAS tells.  A slot read
                    // as taken stays as it is (nothing is ever removed).
                    uint32_t r = KB_C_LA; bool hit = false;
#pragma unroll
                    for (int i = KB_C_LA - 1; i >= 0; --i) {
                        const bool m = cur[g][i] == klo, e = cur[g][i] == KDF_EMPTY;
                        if (m || e) { r = (uint32_t)i; hit = m; }
                    }
                    // (no per-lane `continue`: the wave-queue counter below must stay wave-uniform)
                    if (!__any(td[g])) continue;
                    uint32_t sl = (sl0[g] + r) & bmask;
                    bool more = td[g] && r == KB_C_LA;
                    hit = hit && td[g];
                    if (td[g] && !more && !hit) {
                        if constexpr (MODE == KB_MODE_INSERT) {
                            const uint64_t old = atomicCAS((unsigned long long *)&tlo[sl], KDF_EMPTY, klo);
                            if (old == KDF_EMPTY) { claimed++; hit = true; }
                            else if (old == klo) hit = true;
                            else { more = true; sl = (sl + 1) & bmask; }
                        }                                            // FILTERED: absent, nothing to do
                    }
                    kb_count_hits<VAR == 2>(tcnt, sl, hit);
                    {
                        const unsigned long long mk = __ballot(more);
                        if (mk) {
                            const uint32_t at = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                            if (more) {
                                if (at < WQ) { wqk[at] = klo; wqs[at] = (uint16_t)sl; }
                                else kb_probe_narrow<MODE>(tlo, tcnt, bmask, klo, sl, claimed, failed);
                            }
                            wq_n += (uint32_t)__popcll(mk);
                        }
                    }
                }
            }
            // drain this wave's queue: dense probing, one queued key per lane (wave-private: no barrier)
            {
                const uint32_t nq = wq_n < WQ ? wq_n : WQ;
                for (uint32_t i = threadIdx.x & 63; i < nq; i += 64)
                    kb_probe_narrow<MODE>(tlo, tcnt, bmask, wqk[i], wqs[i], claimed, failed);
                wq_n = 0;
            }
          } else if constexpr (VAR >= 1 && KW == 2 && MODE != KB_MODE_REPLAY) {
            // Two-word keys, same scheme.  The hi words of the lookahead slots are read BEFORE their lo
            // words (LDS serves a wave's instructions in order and a claimer writes lo before the final
            // hi), so a slot whose hi reads as the key's hi without PENDING has its lo in place.  A slot
            // seen PENDING with this hi may become this key: it goes to the queue.  The straight-line
            // code never waits; the queue is drained under the wave-uniform retry loop.
            constexpr int G = 2;
#pragma unroll
            for (int q0 = 0; q0 < EPB; q0 += G) {
                if (e0 + q0 * CT >= total) break;
                uint64_t chi[G][KB_C_LA], clo[G][KB_C_LA]; uint32_t sl0[G]; bool td[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t ei = wbase + QSTEP * (q0 + g);
                    const uint64_t home = kdf_hash(bklo[q0 + g], bkhi[q0 + g]) >> (64 - plan.log2cap);
                    td[g] = ei < total && !(plan.sub_bits && (home >> plan.bucket_bits) != bucket);
                    sl0[g] = (uint32_t)home & bmask;
#pragma unroll
                    for (int i = 0; i < KB_C_LA; ++i) chi[g][i] = __hip_atomic_load(&thi[(sl0[g] + i) & bmask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < KB_C_LA; ++i) asm volatile("" : "+v"(chi[g][i]) :: "memory");      // hi words first ...
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < KB_C_LA; ++i) clo[g][i] = __hip_atomic_load(&tlo[(sl0[g] + i) & bmask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < KB_C_LA; ++i) asm volatile("" : "+v"(clo[g][i]) :: "memory");      // ... then the lo words
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint64_t klo = bklo[q0 + g], khi = bkhi[q0 + g];
                    uint32_t r = KB_C_LA; int kind = 0;                      // 1 the key, 2 empty, 3 this hi, still PENDING
#pragma unroll
                    for (int i = KB_C_LA - 1; i >= 0; --i) {
                        const uint64_t h = chi[g][i];
                        const bool same = (h & ~KDF_PENDING) == khi;
                        const bool m = h == khi && clo[g][i] == klo, e = h == KDF_EMPTY, pd = same && (h & KDF_PENDING);
                        if (m || e || pd) { r = (uint32_t)i; kind = m ? 1 : (e ? 2 : 3); }
                    }
                    if (!__any(td[g])) continue;
                    uint32_t sl = (sl0[g] + r) & bmask;
                    bool hit = td[g] && kind == 1;
                    bool more = td[g] && (r == KB_C_LA || kind == 3);
                    if (td[g] && kind == 2) {
                        if constexpr (MODE == KB_MODE_INSERT) {
                            const uint64_t old = atomicCAS((unsigned long long *)&thi[sl], KDF_EMPTY, khi | KDF_PENDING);
                            if (old == KDF_EMPTY) {
                                __hip_atomic_store(&tlo[sl], klo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_store(&thi[sl], khi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                claimed++; hit = true;
                            } else {
                                more = true;                                 // taken meanwhile: by this key (same hi) or by another
                                if ((old & ~KDF_PENDING) != khi) sl = (sl + 1) & bmask;
                            }
                        }                                                    // FILTERED: absent
                    }
                    kb_count_hits<VAR == 2>(tcnt, sl, hit);
                    const unsigned long long mk = __ballot(more);
                    if (mk) {
                        const uint32_t at = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                        const bool queued = more && at < WQ;
                        if (queued) { wqk[at] = klo; wqk2[at] = khi; wqs[at] = (uint16_t)sl; }
                        wq_n += (uint32_t)__popcll(mk);
                        // queue full: place the key now, under the wave-uniform retry loop
                        kb_probe_wide_wave<MODE>(tlo, thi, tcnt, bmask, more && !queued, klo, khi, sl, claimed, failed);
                    }
                }
            }
            {
                const uint32_t nq = wq_n < WQ ? wq_n : WQ;
                for (uint32_t b0 = 0; b0 < nq; b0 += 64) {                  // wave-uniform trip count
                    const uint32_t i = b0 + (threadIdx.x & 63);
                    const bool todo = i < nq;
                    const uint64_t klo = todo ? wqk[i] : 0, khi = todo ? wqk2[i] : 0;
                    kb_probe_wide_wave<MODE>(tlo, thi, tcnt, bmask, todo, klo, khi, todo ? (uint32_t)wqs[i] : 0u, claimed, failed);
                }
                wq_n = 0;
            }
          } else {
#pragma unroll
          for (int q = 0; q < EPB; ++q) {
            const uint32_t ei = e0 + q * CT + threadIdx.x;
            bool todo = ei < total;
            const uint64_t klo = bklo[q], khi = KW == 2 ? bkhi[q] : 0;
            const uint64_t h = kdf_hash(klo, khi);
            const uint64_t home = h >> (64 - plan.log2cap);
            if (plan.sub_bits && (home >> plan.bucket_bits) != bucket) todo = false;   // sibling bucket's entry
            if constexpr (MODE == KB_MODE_REPLAY) {
                const uint64_t slot = kdf_home(t, h);
                bool ok = true;
                if constexpr (KW == 1) { if (todo) ok = kdf_add_narrow<true>(t, klo, 1u, slot, t.lo[slot], claimed); }
                else ok = kdf_add_wide<true>(t, todo, klo, khi, 1u, slot, claimed);
                if (!ok) failed = true;
                continue;
            }
            if constexpr (KW == 1) {
                if (!todo) continue;
                if (plan.dbg & 1) { claimed += (uint32_t)(klo >> 61); continue; }
                kb_probe_narrow<MODE>(tlo, tcnt, bmask, klo, (uint32_t)home & bmask, claimed, failed);
            } else {
                // wide: claim hi with PENDING, publish lo, then the final hi (all in LDS).
                // No lane waits inside a divergent loop (kdf_device.h): a lane that meets
                // a PENDING slot retries in the next pass of a wave-uniform loop.
                kb_probe_wide_wave<MODE>(tlo, thi, tcnt, bmask, todo, klo, khi, (uint32_t)home & bmask, claimed, failed);
            }
          }
        }
          }
        __syncthreads();       // run_pref / run_first are rewritten by the next round
    }
    if (failed) atomicOr(&sh_failed, 1u);
    if (claimed) atomicAdd(&sh_claimed, claimed);
    __syncthreads();
    if constexpr (MODE == KB_MODE_REPLAY) {
        if (threadIdx.x == 0) {
            if (sh_failed) atomicOr(&ctl->error, 1u);
            if (sh_claimed) atomicAdd(&ctl->distinct[(bucket % KDF_SHARDS) * 16], (unsigned long long)sh_claimed);
        }
        return;
    }
    if (sh_failed) {
        // leave the bucket as it was in HBM; flag it for replay.  A lazily
        // cleared table holds garbage there: write an empty slice instead.
        if (threadIdx.x == 0) {
            atomicOr(&s.failed[bucket >> 5], 1u << (bucket & 31));
            atomicAdd(&s.totals[2], 1ull);
        }
        if (!table_nonempty) {
            for (uint32_t i = threadIdx.x; i < B; i += CT) {
                t.lo[slot0 + i] = KDF_EMPTY;
                if constexpr (KW == 2) t.hi[slot0 + i] = KDF_EMPTY;
                t.cnt[slot0 + i] = 0;
            }
        }
        return;
    }
    if (plan.dbg & 4) return;
    // LDS counts were advanced with plain (wrapping, non-returning) adds.  A pass
    // adds fewer than 2^32 to a slot, so a slot wrapped iff its new value is below
    // the value it had in HBM: saturate those (Jellyfish's 4-byte counter).
    // two slots per lane and step: 16-byte LDS reads and HBM stores for the keys, 8-byte ones for the counts
    // (slot0 is a multiple of B, B is even: everything stays aligned)
    for (uint32_t i = threadIdx.x; i < B / 2; i += CT) {
        if constexpr (MODE == KB_MODE_INSERT) {
            ((ulonglong2 *)(t.lo + slot0))[i] = ((const ulonglong2 *)tlo)[i];
            if constexpr (KW == 2) ((ulonglong2 *)(t.hi + slot0))[i] = ((const ulonglong2 *)thi)[i];
        }
        uint2 c = ((const uint2 *)tcnt)[i];
        if (table_nonempty) {
            const uint2 o = ((const uint2 *)(t.cnt + slot0))[i];
            if (c.x < o.x) c.x = 0xFFFFFFFFu;
            if (c.y < o.y) c.y = 0xFFFFFFFFu;
        }
        ((uint2 *)(t.cnt + slot0))[i] = c;
    }
    if (threadIdx.x == 0 && sh_claimed)
        atomicAdd(&ctl->distinct[(bucket % KDF_SHARDS) * 16], (unsigned long long)sh_claimed);
}


// ---------------------------------------------------------------------------
// Heavy buckets of a skewed pass (narrow keys, insert mode; see kb_count_hits).  kb_bucket_kernel<.., VAR 2> lists the
// buckets whose first 256 runs alone hold more than KB_C_HEAVY entries and leaves them untouched.  Here KB_HV_SLICES
// workgroups share such a bucket's runs (run r to slice r % KB_HV_SLICES), each counting into a PRIVATE empty LDS table,
// and stage their distinct (key, count) pairs; kb_heavy_combine_kernel then folds the staged pairs into the bucket the way
// kernel C would have: slice in LDS, transactional (no room: the bucket is flagged for the replay pass and stays as it was).
// Measured on the repeat-rich 100 Mbp genome (38 heavy buckets, the heaviest 15.9 M entries): kernel C 10.0 -> 6.2 ms, the
// pass 20.1 -> 16.0 ms.
__global__ __launch_bounds__(256) void kb_heavy_slice_kernel(KbPlan plan, KbScratch s) {
    constexpr int CHUNK = KbCfg<1>::CHUNK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!s.hv_ctr || s.failed_flag[0]) return;
    const uint32_t nh = min(s.hv_ctr[0], KB_HV_MAX), h = blockIdx.y, slice = blockIdx.x;
    if (h >= nh) return;
    const uint32_t B = 1u << plan.bucket_bits, bmask = B - 1;
    uint64_t *tlo = (uint64_t *)smem;
    uint32_t *tcnt = (uint32_t *)(smem + (size_t)B * 8);
    __shared__ uint32_t sh_fail, sh_n, sh_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t i = tid; i < B; i += 256) { tlo[i] = KDF_EMPTY; tcnt[i] = 0; }
    if (tid == 0) { sh_fail = 0; sh_n = 0; }
    __syncthreads();
    const uint64_t bucket = s.hv_bucket[h];
    const uint32_t c = (uint32_t)(bucket >> plan.c2), f = (uint32_t)(bucket & ((1u << plan.c2) - 1));
    const unsigned long long j0 = s.chunk_first[c], j1 = s.chunk_first[c + 1], bstart = s.bin_start[c];
    uint32_t claimed = 0; bool failed = false;
    for (unsigned long long j = j0 + slice; j < j1; j += KB_HV_SLICES) {
        const uint32_t r0 = s.chunk_off[j * plan.off_stride + f], r1 = s.chunk_off[j * plan.off_stride + f + 1];
        const uint64_t *ent = s.ent_lo + bstart + (j - j0) * (unsigned long long)CHUNK + r0;
        const uint32_t n = r1 - r0;
        constexpr int U = 8;                                       // entries per thread in flight: one workgroup has to cover the HBM latency alone
        for (uint32_t i0 = 0; i0 < n; i0 += 256 * U) {             // whole waves: the hit counting is a wave operation
            uint64_t keys[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const uint32_t i = i0 + u * 256 + tid; keys[u] = i < n ? ent[i] : 0; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u * 256 >= n) break;                      // (uniform)
                const bool todo = i0 + u * 256 + tid < n;
                const uint64_t key = keys[u];
                const uint32_t sl = (uint32_t)(kdf_hash(key, 0) >> (64 - plan.log2cap)) & bmask;
                const bool hit = todo && tlo[sl] == key;           // the heavy key sits in its home slot after its first insertion
                kb_count_hits<true>(tcnt, sl, hit);
                if (todo && !hit) {
                    // (one slot per step is enough here: almost every entry took the branch above)
                    uint32_t at = sl; bool done = false;
                    for (uint32_t p_ = 0; p_ <= bmask && !done; ++p_) {
                        uint64_t cur = tlo[at];
                        if (cur == KDF_EMPTY) {
                            cur = atomicCAS((unsigned long long *)&tlo[at], KDF_EMPTY, key);
                            if (cur == KDF_EMPTY) { ++claimed; cur = key; }
                        }
                        if (cur == key) { atomicAdd(&tcnt[at], 1u); done = true; }
                        at = (at + 1) & bmask;
                    }
                    if (!done) failed = true;
                }
            }
        }
    }
    if (failed) sh_fail = 1;
    __syncthreads();
    if (sh_fail) { if (tid == 0) s.hv_failed[h] = 1; return; }
    // stage the private table's pairs: one reservation per workgroup
    uint32_t mine = 0;
    for (uint32_t i = tid; i < B; i += 256) mine += tlo[i] != KDF_EMPTY ? 1u : 0u;
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
    uint32_t wbase = 0;
    if (lane == 63) wbase = atomicAdd(&sh_n, inc);
    wbase = __shfl(wbase, 63);
    __syncthreads();
    if (tid == 0) sh_base = atomicAdd(&s.hv_n[h], sh_n);
    __syncthreads();
    uint32_t pos = sh_base + wbase + inc - mine;
    const size_t room = (size_t)KB_HV_SLICES << plan.bucket_bits;
    uint64_t *ok_ = s.hv_key + (size_t)h * room; uint32_t *oc_ = s.hv_cnt + (size_t)h * room;
    for (uint32_t i = tid; i < B; i += 256)
        if (tlo[i] != KDF_EMPTY) { ok_[pos] = tlo[i]; oc_[pos] = tcnt[i]; ++pos; }
}

__global__ __launch_bounds__(256) void kb_heavy_combine_kernel(KbPlan plan, KbScratch s, KdfTable t, KdfCtl *ctl, int table_nonempty) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!s.hv_ctr || s.failed_flag[0]) return;
    const uint32_t nh = min(s.hv_ctr[0], KB_HV_MAX), h = blockIdx.x;
    if (h >= nh) return;
    if (h == 0 && threadIdx.x == 0) s.totals[4] = s.hv_ctr[0];     // (statistics: heavy buckets of this pass; the first KB_HV_MAX were split)
    const uint32_t B = 1u << plan.bucket_bits, bmask = B - 1;
    uint64_t *tlo = (uint64_t *)smem;
    uint32_t *tcnt = (uint32_t *)(smem + (size_t)B * 8);
    __shared__ uint32_t sh_fail, sh_claimed;
    const uint32_t tid = threadIdx.x;
    const uint64_t bucket = s.hv_bucket[h];
    const uint64_t slot0 = bucket << plan.bucket_bits;
    for (uint32_t i = tid; i < B; i += 256) {
        tlo[i] = table_nonempty ? t.lo[slot0 + i] : KDF_EMPTY;
        tcnt[i] = table_nonempty ? t.cnt[slot0 + i] : 0u;
    }
    if (tid == 0) { sh_fail = s.hv_failed[h]; sh_claimed = 0; }
    __syncthreads();
    const size_t room = (size_t)KB_HV_SLICES << plan.bucket_bits;
    const uint64_t *ik = s.hv_key + (size_t)h * room; const uint32_t *ic = s.hv_cnt + (size_t)h * room;
    const uint32_t n = s.hv_n[h];
    uint32_t claimed = 0; bool failed = false;
    if (!sh_fail)
        for (uint32_t i = tid; i < n; i += 256) {
            const uint64_t key = ik[i]; const uint32_t add = ic[i];
            uint32_t at = (uint32_t)(kdf_hash(key, 0) >> (64 - plan.log2cap)) & bmask; bool done = false;
            for (uint32_t p_ = 0; p_ <= bmask && !done; ++p_) {
                uint64_t cur = tlo[at];
                if (cur == KDF_EMPTY) {
                    cur = atomicCAS((unsigned long long *)&tlo[at], KDF_EMPTY, key);
                    if (cur == KDF_EMPTY) { ++claimed; cur = key; }
                }
                if (cur == key) { kb_lds_sat_add(&tcnt[at], add); done = true; }
                at = (at + 1) & bmask;
            }
            if (!done) failed = true;
        }
    if (failed) sh_fail = 1;
    if (claimed) atomicAdd(&sh_claimed, claimed);
    __syncthreads();
    if (sh_fail) {                                                 // as kernel C: untouched in HBM, flagged for the replay pass
        if (tid == 0) { atomicOr(&s.failed[bucket >> 5], 1u << (bucket & 31)); atomicAdd(&s.totals[2], 1ull); }
        return;                                                    // (kernel C already wrote an empty slice into a lazily cleared table)
    }
    for (uint32_t i = tid; i < B; i += 256) { t.lo[slot0 + i] = tlo[i]; t.cnt[slot0 + i] = tcnt[i]; }
    if (tid == 0 && sh_claimed) atomicAdd(&ctl->distinct[(bucket % KDF_SHARDS) * 16], (unsigned long long)sh_claimed);
}

