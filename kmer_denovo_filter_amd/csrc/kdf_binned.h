// kdf_binned.h -- the LDS-staged-bucket count pipeline ("binned" path).
//
// The table (kdf_device.h) is an array of buckets of 2^bucket_bits slots; a
// key's bucket is the top bits of its hash.  Random probes into an HBM table
// move a 64-128 B sector per 8 useful bytes and pay a device atomic per window.
// Instead, a batch of reads is PARTITIONED, reading the stream ONCE:
//
//   A   kb_slabsort_kernel: every slab of SLAB stream positions is turned into the STORED FORMS h of its canonical
//       k-mers (kdf_device.h: the hash is a bijection, so the entry IS the hash and no later stage evaluates it
//       again), counting-sorted in LDS by coarse bin (top c1 hash bits) and written out as ONE contiguous block,
//       with the slab's bin offsets (a row of u16).  No histogram pass, no global cursors, no partial cache lines.
//   P   kb_groupsum_kernel / kb_plan_kernel: G consecutive slabs form a GROUP; the entries of bin c in group g are a
//       PIECE of ~0.95 CHUNK entries (more under skew: then several pieces).  Exact piece sizes from the offset rows
//       (190 MB for 1.5 G positions), piece -> row of the offset table, compact piece -> ring position, per-bin piece lists.
//   B   kb_piecesort_pipe_kernel: a piece's runs (one per slab of the group) are gathered, sorted in LDS by the next c2
//       hash bits and appended, with the piece's offset row, to the engine's entry RING -- by one persistent workgroup
//       per CU that keeps four pieces in flight at different stages (kb_piecesort_kernel: a workgroup per piece, for
//       comparison; kb_piecesort_more_kernel: the further pieces of a skewed pair).
//
// Rounds 1-2 read the stream twice (a histogram pass A0 fixed an exact place for every (workgroup, bin) run before the
// scatter A1 could copy its runs out bin by bin); A0 + A1 took 5.9 ms of a 14.3 ms pass.
//
// Pass after pass is appended to the ring (a streamed sample is many batches into one table); the ring is APPLIED to the
// table only when something needs the table (dump / query / stats ...) or the ring is full:
//
//   C   kb_bucket_kernel: one workgroup per table bucket: bucket slice (keys+counts) lives in
//       LDS, the bucket's runs are gathered from all pieces of its coarse bin IN EVERY PENDING PASS
//       and inserted / probed with LDS atomics, the slice is written back once.
//       Transactional per bucket: a bucket that overflows is left untouched in
//       HBM and flagged; the host grows the table and replays those buckets
//       through the global-atomic path (kb_replay_kernel).
//
// Kernel C reads and rewrites every bucket of the table whatever the batches hold (0.5 ms per GB of table): deferring
// it over the pending passes is what makes a streamed sample run at the single-batch rate.
// All of it is placement independent: no workgroup reads another workgroup's
// output inside a launch.
#pragma once
#include "kdf_device.h"

#define KB_THREADS   1024
#define KB_F_BITS    8                   // fine radix (level 2) of tables of up to 2^16 buckets
#define KB_F_BITS_MAX 9                  // ... of the others (kb_make_plan: 8 + 8, then 8 + 9, 9 + 9, 10 + 9 bits)
#define KB_F         (1 << KB_F_BITS_MAX)  // LDS array size for the fine histogram
#define KB_C1_MAX    10                  // coarse bins <= 1024
#define KB_MAX_PASS  64                  // pending passes one kernel C can apply
// Bucket kernel: 768 threads = 12 waves per workgroup, two workgroups per CU (LDS) = 6 waves per SIMD,
// which needs <= 80 VGPRs (amdgpu_waves_per_eu below).  Measured on
// the bench pass: 512 threads x 20 entries (4 waves per SIMD) 6.16 ms, 768 x 12 5.4 ms, 1024 x 8 (8 per SIMD,
// spills) 6.0 ms; thread counts whose waves do not divide evenly over the four SIMDs (640, 896) leave one
// workgroup per CU (9-10 ms).  12 x 768 = 9216 entries per batch also covers the bench's ~8.9 K entries per
// bucket in one batch with 11.6 of the 12 waves busy.
#ifndef KB_C_THREADS
#define KB_C_THREADS 768
#endif
#ifndef KB_C_EPB_N
#define KB_C_EPB_N 12                    // narrow keys: entries per thread and batch (a multiple of 4)
#endif
#ifndef KB_C_EPB_W
#define KB_C_EPB_W 8                     // wide keys (4 measured the same, 12 spills)
#endif
#ifndef KB_C_THREADS_W
#define KB_C_THREADS_W 512                // wide keys: 2048-slot buckets hold ~3 K entries; 512 x 8 covers them and three workgroups fit a CU
#endif
#ifndef KB_C_WQ_W
#define KB_C_WQ_W 40                      // wide keys: queue entries per wave (18 B each): with 40, three workgroups' LDS (52.8 KB each) fit a CU; 48 + the
                                          // run index of round 3 = 53.9 KB left room for two (kernel C 10.3 -> 13.8 ms at k = 63)
#endif
#ifndef KB_C_WPE
#define KB_C_WPE 6                       // waves per SIMD the register allocation aims at
#endif
static_assert(KB_C_EPB_N % 4 == 0 && KB_C_EPB_W % 4 == 0, "kernel C resolves entries four at a time");
static_assert(KB_C_THREADS % 256 == 0 && KB_C_THREADS <= 1024 && KB_C_THREADS_W % 256 == 0 && KB_C_THREADS_W <= KB_C_THREADS, "whole waves on every SIMD");
#define KB_C_CT(KW) ((KW) == 2 ? KB_C_THREADS_W : KB_C_THREADS)
// BIG: tables of 2^32 slots and more have buckets of twice the slots (kdf_engine.hip: table_alloc) -- 96 / 80 KB of LDS per
// bucket, one workgroup of 1024 threads per CU.  Beyond 2^19 buckets the partition (10 + 9 bits) leaves sub-buckets to
// kernel C, which reads every run once per sub-bucket; twice the bucket halves that.  Measured: 8 bench batches into 2^32
// slots, C 9.05 -> 8.44 ms per batch; 64 small batches into 2^33 slots 38 -> 41.7 Gk-mer/s; but a 2^31-slot table is
// better off with two 4096-slot workgroups per CU and 10 + 9 bits (77 against 70 Gk-mer/s), hence the threshold.
#define KB_C_CT_BIG 1024
#define KB_C_CTB(KW, BIG) ((BIG) ? KB_C_CT_BIG : KB_C_CT(KW))
#define KB_BB_SMALL(KW) ((KW) == 1 ? 12u : 11u)         // bucket bits of tables below the threshold
#define KB_C_RUNS    256                 // runs (pieces of the coarse bin) staged per round
#define KB_C_RUNS_BIG 1024               // ... by the big-bucket instantiation: a streamed sample has hundreds of runs per bucket (8 bench
                                         // batches into 2^32 slots: 736), and every round costs two global latencies that the one
                                         // workgroup a CU holds there has nothing to hide behind
#ifndef KB_C_RUNS_N
#define KB_C_RUNS_N  512                 // narrow keys, ordinary buckets: 73.6 KB of LDS, still two workgroups per CU (wide keys have no room: three per CU)
#endif
#define KB_C_RUNS_T(KW, BIG) ((BIG) ? KB_C_RUNS_BIG : (KW) == 1 ? KB_C_RUNS_N : KB_C_RUNS)

// Threads of the slab kernel: 1024 x 16 windows = 16 K-entry slabs (132 KB of LDS, one workgroup per CU).  512 (8 K slabs,
// two workgroups per CU) measured 4.03 ms against 3.14 for the slab kernel and 6.7 against 5.5 for the piece kernel, whose
// gathers are runs of SLAB / bins entries: 128 bytes instead of 256.
#ifndef KB_A_THREADS
#define KB_A_THREADS 1024
#endif
#ifndef KB_A_SCAN
#define KB_A_SCAN 0                      // 1: every wave scans its own 64 bins (measured: slab kernel 3.17 against 3.13 ms); 0: one wave scans them all
#endif
#ifndef KB_A_B4
#define KB_A_B4 1                        // 1: a fourth barrier per slab after the write-out.  Not needed for correctness, but without it the
                                         // slab kernel is SLOWER (3.24 against 3.13 ms, same box): waves that run ahead into the next slab's
                                         // ranking atomics slow the write-out of the others down
#endif
template <int KW> struct KbCfg;
// WPT windows per thread of the slab kernel; SLAB stream positions (= entries at most) per slab; CHUNK = capacity of a piece
#ifndef KB_CHUNK_N
#define KB_CHUNK_N 16384
#endif
template <> struct KbCfg<1> { static constexpr int WPT = 16, SLAB = KB_A_THREADS * 16, CHUNK = KB_CHUNK_N; };       // 8-byte entries: a piece is 128 KB of LDS
template <> struct KbCfg<2> { static constexpr int WPT = 8,  SLAB = KB_A_THREADS * 8,  CHUNK = KB_CHUNK_N / 2;  };   // 16-byte entries
// Wide entries travel as 16-byte (h, hi) structs: one dwordx4 / ds_*_b128 per entry instead
// of two 8-byte accesses to two arrays (runs are short).
struct __attribute__((aligned(16))) KbEnt2 { uint64_t lo, hi; };
#ifndef KB_G_MAX
#define KB_G_MAX 2048                    // slabs per group: a thread of the piece kernels takes the runs of two consecutive slabs.  (1024,
                                         // one slab per thread, capped the groups of tables with 1024 coarse bins at 84 % full pieces)
#endif

struct KbPlan {
    uint32_t c1;            // coarse bits
    uint32_t c2;            // fine bits (<= KB_F_BITS_MAX)
    uint32_t sub_bits;      // table buckets per partition bucket = 2^sub_bits (a table that grew since the partition: more)
    uint32_t log2cap, bucket_bits;
    uint32_t off_stride;    // words per row of chunk_off = 2^KB_F_BITS_MAX + 1 (fixed: the rows of passes of any geometry share the array)
    uint32_t key_parts, key_part;   // KdfTable::key_parts: windows of other key-space slices are dropped in A
    uint32_t dbg;           // experiments only
    uint32_t n_pass;        // pending passes kernel C applies (1 .. KB_MAX_PASS)
    uint32_t group;         // G: slabs per group
    uint32_t n_groups, n_slabs;
    uint32_t dump_min;      // the DUMP instantiations of kernel C (insert mode) also dump every slot with count >= dump_min >= 1 while they hold the bucket (KbScratch::dump_*)
};

// One partitioned pass in the ring (device resident, written by kb_binfirst_kernel).  Pieces are numbered BIN-MAJOR: the
// pieces of bin c are rows row_base + binrow_first[c] .. binrow_first[c + 1]) and lie one after the other in the ring.
struct KbPass {
    unsigned long long ent_base;                          // first entry of the pass in the ring (entries)
    unsigned long long row_base;                          // first row of the pass (chunk_off, row_ent, row_len)
    unsigned long long n_entries, n_rows;
    uint32_t binrow_first[(1 << KB_C1_MAX) + 1];          // rows of the bins before bin c
    unsigned long long binent_first[(1 << KB_C1_MAX) + 1];// entries of the bins before bin c
};

// device scratch shared by the kernels of the binned path
struct KbScratch {
    // the pass being partitioned (reused by the next pass)
    uint64_t *tmp;                  // [n_slabs][SLAB] slab-sorted entries (wide: 16-byte pairs)
    uint16_t *off;                  // [n_slabs][2^c1 + 1] bin offsets inside each slab (last = its valid windows)
    uint32_t *gn;                   // [n_groups][2^c1] entries of the pair (group g, bin c)
    uint32_t *gpre_row;             // [n_groups][2^c1] pieces of bin c in the groups before g
    unsigned long long *gpre_ent;   // [n_groups][2^c1] entries of bin c in the groups before g
    uint32_t *bin_rows;             // [2^c1] pieces of bin c
    unsigned long long *bin_ent;    // [2^c1] entries of bin c
    uint32_t *ovf;                  // [0] = pieces beyond the first of their pair (skew), then (pair, piece) couples
    uint32_t ovf_cap, pad_;
    // the ring
    uint64_t *ent;                  // entries: stored forms h; wide keys: (h, hi) pairs, 16 B each
    uint32_t *chunk_off;            // [rows][off_stride] fine-run offsets of every sorted piece
    unsigned long long *row_ent;    // [rows] first entry of the piece (absolute)
    uint32_t *row_len;              // [rows] its entries
    KbPass *pass;                   // [KB_MAX_PASS]
    unsigned long long *totals;     // [16]: 0 entries (all pending passes), 2 failed buckets, 4 heavy buckets, 7 skew flag
    uint32_t *failed;               // bitmap over TABLE buckets
    // heavy buckets of a skewed flush (kb_heavy_slice_kernel): [0] how many, their ids, staged (key, count) pairs
    uint32_t *hv_ctr;               // [4]
    uint32_t *hv_bucket, *hv_n, *hv_failed;   // [KB_HV_MAX]
    uint64_t *hv_key;               // [KB_HV_MAX][KB_HV_SLICES << bucket bits] stored forms (wide: + hv_khi, the keys' high words)
    uint64_t *hv_khi;
    uint32_t *hv_cnt;
    uint64_t *trash;                // [64] words nobody reads: where the pipelined piece sort stores when it has nothing to store
    // `dump -L` fused into the flush (KbPlan::dump_min): keys (as KEYS, not stored forms) and counts go out through
    // ctl->cursor, one reservation per bucket; nothing is written at or past dump_cap
    uint64_t *dump_lo, *dump_hi; uint32_t *dump_cnt; unsigned long long dump_cap;
};
#ifndef KB_HV_MAX
#define KB_HV_MAX    64u                             // heavy buckets split per flush (further ones are processed the ordinary way)
#endif
#define KB_HV_SLICES 32u                             // workgroups that share one heavy bucket's runs
#ifndef KB_C_HEAVY
#define KB_C_HEAVY   65536u                          // entries (first 256 runs) from which a bucket counts as heavy: ~7x a bucket's share at bench load
#endif

// (both live in the high half of the hash -- c1 + c2 <= 19: 32-bit shifts)
#ifndef KDF_FUNNEL64
__device__ __forceinline__ uint32_t kb_coarse(const KbPlan &p, uint64_t h) {
    return p.c1 ? (uint32_t)(h >> 32) >> (32 - p.c1) : 0u;
}
__device__ __forceinline__ uint32_t kb_fine(const KbPlan &p, uint64_t h) {
    return p.c2 ? ((uint32_t)(h >> 32) >> (32 - p.c1 - p.c2)) & ((1u << p.c2) - 1) : 0u;
}
#else
__device__ __forceinline__ uint32_t kb_coarse(const KbPlan &p, uint64_t h) {
    return p.c1 ? (uint32_t)(h >> (64 - p.c1)) : 0u;
}
__device__ __forceinline__ uint32_t kb_fine(const KbPlan &p, uint64_t h) {
    return p.c2 ? (uint32_t)((h >> (64 - p.c1 - p.c2)) & ((1u << p.c2) - 1)) : 0u;
}
#endif

// Ablation flags (debug_flags 256 / 512 / 1024: a kernel without its write-out / with its gather served from L2 / without rank
// return values -- WRONG results, timing only; scratch/ablate2.py) exist in variant builds only (-DKB_ABLATE): in the product
// kernels the tests on the flags cost 1-3 % of the instructions.
#ifdef KB_ABLATE
#define KB_ABL(plan, bit) (((plan).dbg & (bit)) != 0)
#else
#define KB_ABL(plan, bit) (false)
#endif

// Phase timing of a kernel, for variant builds only (-DKB_TIMING; scratch/build_variant.sh): thread 0 of every workgroup
// adds the cycles since its last stamp to trash[8 + n] (read back through kdf_get_stat "trashN").
#ifdef KB_TIMING
#define KB_T_INIT unsigned long long kb_t_prev = __builtin_readcyclecounter()
#define KB_T(tr, n) do { if (threadIdx.x == 0) { const unsigned long long kb_t_now = __builtin_readcyclecounter(); \
                          atomicAdd((unsigned long long *)&(tr)[8 + (n)], kb_t_now - kb_t_prev); kb_t_prev = kb_t_now; } } while (0)
#else
#define KB_T_INIT do {} while (0)
#define KB_T(tr, n) do {} while (0)
#endif

// block-wide exclusive scan of n <= KB_THREADS uint32 values held one per
// thread (threads >= n pass 0); returns the exclusive prefix, total via *tot.
__device__ __forceinline__ uint32_t kb_block_exscan(uint32_t v, uint32_t *wsum /* >= 17 words LDS */, uint32_t *tot) {
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
            // Wide keys, the same idea over the 192-bit span.  R = the span with its 96
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
            static_assert(2 * (WPT - 1) < 32, "the per-window shifts fit kdf_funnel32");
            const uint64_t rc = ~kdf_funnel32(e[0], e[1], 2 * u) & kmask;      // reverse complement: ~E (kdf_device.h)
            const uint64_t fwd = kdf_funnel32(g0, g1, 2 * (WPT - 1 - u)) & kmask;
            lo = fwd < rc ? fwd : rc; hi = 0;
        } else {
            const uint64_t rlo = ~kdf_funnel32(e[0], e[1], 2 * u), rhi = ~kdf_funnel32(e[1], e[2], 2 * u) & hmask;   // reverse complement: ~E
            const uint64_t flo = kdf_funnel32(gw[0], gw[1], 2 * (WPT - 1 - u)), fhi = kdf_funnel32(gw[1], gw[2], 2 * (WPT - 1 - u)) & hmask;
            const bool fw = (fhi < rhi) || (fhi == rhi && flo < rlo);
            lo = fw ? flo : rlo; hi = fw ? fhi : rhi;
        }
    }
    // stored form of window u: h (and the key's high word as it is)
    __device__ __forceinline__ void stored(int u, uint64_t &h, uint64_t &hi) const {
        uint64_t lo; key(u, lo, hi);
        h = kdf_hash(lo, hi);
    }
};

// Barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0),
// i.e. waits for every outstanding global store; in the slab kernel that would
// serialise a slab's write-out with the next slab's compute.  The barriers there protect LDS data only.
__device__ __forceinline__ void kb_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// A: the stream is read ONCE.  A workgroup takes slabs_per_wg consecutive slabs; per slab: stored form + coarse bin of
// every window, rank by LDS atomics, counting sort into an LDS image, the image written out CONTIGUOUSLY to
// tmp[slab * SLAB ...] and the slab's offset row to off[slab][0 .. nbins] (off[slab][nbins] = its valid windows).
template <int KW, bool SLICED>
__global__ __launch_bounds__(KB_A_THREADS) void kb_slabsort_kernel(
    const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid, uint64_t n_tiles, int k,
    KbPlan plan, KbScratch s, uint32_t slabs_per_wg)
{
    constexpr int WPT = KbCfg<KW>::WPT, TPT = 64 / WPT, SLAB = KbCfg<KW>::SLAB, NT = KB_A_THREADS;
    constexpr uint32_t TILES_PER_SLAB = NT / TPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *slo = (uint64_t *)smem;                                   // [SLAB + 64]: [SLAB + lane] = the lane's trash slot
    KbEnt2 *s2 = (KbEnt2 *)smem;                                        // wide: the image holds (h, hi) pairs
    // Invalid windows (N, read ends: 22 % of the window slots of 150 bp reads) are ranked like the others, branch-free, on
    // DUMMY counters -- ONE PER LANE: with a single dummy counter a quarter of a wave's lanes hit the same LDS word in every
    // rank instruction and the same trash slot in every scatter (SQ_LDS_ADDR_CONFLICT: 318 M of the kernel's 831 M active
    // LDS cycles, profiles/r03b_lds_counters.txt).
    uint32_t *hist = (uint32_t *)(smem + (size_t)(SLAB + 64) * 8 * KW); // [bins + 1 ..]: [DUMMY + lane] = counters of invalid windows (never read)
    uint32_t *offs = hist + (1 << KB_C1_MAX) + 96;                      // [bins + 1 ..]: offs[DUMMY + lane] = SLAB + lane, the lane's trash slot
    constexpr int DUMMY = (1 << KB_C1_MAX) + 1;             // (index 2^c1 <= 1024 holds the slab's total)
    const int nb = 1 << plan.c1;
    for (int i = threadIdx.x; i <= DUMMY; i += NT) hist[i] = 0;
    if (threadIdx.x < 64) offs[DUMMY + threadIdx.x] = (uint32_t)SLAB + threadIdx.x;
    __syncthreads();
    const uint64_t slab0 = (uint64_t)blockIdx.x * slabs_per_wg;
    if (slab0 * TILES_PER_SLAB >= n_tiles) return;                     // uniform
    KbWindows<KW> win;
    win.load(packed, invalid, slab0 * TILES_PER_SLAB + threadIdx.x / TPT, n_tiles, threadIdx.x % TPT, k);
    KB_T_INIT;
    for (uint32_t sl = 0; sl < slabs_per_wg; ++sl) {
        const uint64_t slab = slab0 + sl;
        if (slab * TILES_PER_SLAB >= n_tiles) break;                  // uniform
        KB_T(s.trash, 18);                                               // (loop turn-around: B4, win = nxt)
        KbWindows<KW> nxt;
        {
            // prefetch of the next slab's words: loads only; (tile >= n_tiles handles "no next slab")
            const bool more = sl + 1 < slabs_per_wg;
            nxt.issue(packed, invalid, more ? (slab + 1) * TILES_PER_SLAB + threadIdx.x / TPT : n_tiles,
                      n_tiles, threadIdx.x % TPT, k);
        }
        // Branch-free ranking: invalid windows (~4 %) go to a dummy counter
        // hist[DUMMY], so the WPT returning LDS atomics issue back to back with
        // ONE wait instead of WPT serialized round trips inside exec branches.
        uint64_t klo[WPT], khi[KW == 2 ? WPT : 1];
        uint32_t br[WPT];                       // bin << 16 | rank  (rank < SLAB <= 16384)
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            uint64_t hsh, hi; win.stored(u, hsh, hi);
            klo[u] = hsh; if constexpr (KW == 2) khi[u] = hi;
            const bool ok = ((win.valid >> u) & 1) && (!SLICED || kdf_slice(hsh, plan.key_parts) == plan.key_part);
            const uint32_t bin = ok ? kb_coarse(plan, hsh) : (uint32_t)DUMMY + (threadIdx.x & 63u);
            br[u] = bin << 16;
        }
        if (KB_ABL(plan, 1024)) {                                          // (ablation: counts without ranks -- WRONG results, timing only)
#pragma unroll
            for (int u = 0; u < WPT; ++u) atomicAdd(&hist[br[u] >> 16], 1u);
        } else {
#pragma unroll
        for (int u = 0; u < WPT; ++u) br[u] |= atomicAdd(&hist[br[u] >> 16], 1u) & 0xFFFFu;
        }
        KB_T(s.trash, 10);                                               // keys, bins, rank atomics issued
        kb_lds_barrier();                                               // B1: all ranks taken
        KB_T(s.trash, 11);                                               // ... ranks back, barrier
#if KB_A_SCAN == 0
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
            if (threadIdx.x == 63) offs[nb] = inc;                      // the slab's valid windows
        }
#else
        // exclusive scan of hist[0..nb): wave w owns the bins [64 w, 64 w + 64); what lies below them is summed by the
        // wave itself (lane l adds hist[l + 64 j], j < w: independent reads, one wave reduction) -- no extra barrier, and
        // no wave walks 16 dependent LDS round trips while fifteen wait (the one-wave scan: ~0.5 us per slab)
        if ((int)threadIdx.x < ((nb + 63) & ~63)) {                     // whole waves
            const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
            uint32_t below = 0;
            for (int j = 0; j < w; ++j) below += hist[l + 64 * j];      // (w is wave-uniform)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
            const uint32_t v = (int)threadIdx.x < nb ? hist[threadIdx.x] : 0u;
            uint32_t inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (l >= o) inc += t; }
            if ((int)threadIdx.x < nb) offs[threadIdx.x] = below + inc - v;
            if ((int)threadIdx.x == nb - 1) offs[nb] = below + inc;   // the slab's valid windows
        }
#endif
        KB_T(s.trash, 12);                                               // scan (wave 0)
        kb_lds_barrier();                                               // B2: offsets ready
        KB_T(s.trash, 13);
        {
            // invalid windows land on their lane's trash slot (offs[DUMMY + lane] = SLAB + lane, rank masked off)
            uint32_t pos[WPT];
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const uint32_t bin = br[u] >> 16;
                pos[u] = offs[bin] + ((bin >= (uint32_t)DUMMY) ? 0u : (br[u] & 0xFFFF));
            }
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                if constexpr (KW == 2) s2[pos[u]] = KbEnt2{klo[u], khi[u]};
                else slo[pos[u]] = klo[u];
            }
        }
        // the slab's offset row (coalesced) while the image settles; the histogram is zeroed for the next slab
        {
            uint16_t *orow = s.off + slab * (uint64_t)(nb + 1);
            for (int i = threadIdx.x; i <= nb; i += NT) { orow[i] = (uint16_t)offs[i]; hist[i] = 0; }
            // (the dummy counters are never read: they may run on)
        }
        // (Measured and dropped, round 3: the NEXT slab's keys computed here, between scatter and B3, with its words
        // requested before the write-out -- 3.93 against 3.15 ms at k = 31, 6.95 against 5.61 at k = 63: the key arithmetic at
        // the top of the loop is what the previous slab's 128 KB of stores drain under.)
        // retire the prefetched words of the next slab BEFORE the write-out is issued:
        // vmcnt retires in order, so a later wait for these loads would also wait
        // for every store issued in between
        KB_T(s.trash, 14);                                               // scatter + offset row
        nxt.finish();
        asm volatile("" :: "v"(nxt.e[0]), "v"(nxt.e[1]), "v"(nxt.valid));
        KB_T(s.trash, 15);                                               // next slab's words retired
        kb_lds_barrier();                                               // B3: sorted image complete
        KB_T(s.trash, 16);
        {
            // ONE contiguous block per slab: 16 bytes per lane and step
            const uint32_t nv = KB_ABL(plan, 256) ? 0u : offs[nb];          // (ablation 256: no write-out -- timing only)
            if constexpr (KW == 2) {
                KbEnt2 *dst = (KbEnt2 *)s.tmp + slab * (uint64_t)SLAB;
                for (uint32_t i = threadIdx.x; i < nv; i += NT) dst[i] = s2[i];
            } else {
                ulonglong2 *dst = (ulonglong2 *)(s.tmp + slab * (uint64_t)SLAB);
                const ulonglong2 *src = (const ulonglong2 *)slo;
                for (uint32_t i = threadIdx.x; i < (nv + 1) / 2; i += NT) dst[i] = src[i];      // (SLAB is even: the odd tail stays inside the slab's block)
            }
        }
        KB_T(s.trash, 17);                                               // write-out issued
#ifdef KB_TIMING
        if (threadIdx.x == 0) atomicAdd((unsigned long long *)&s.trash[8 + 19], 1ull);
#endif
#if KB_A_B4
        kb_lds_barrier();                                               // B4: image free (stores still draining)
#endif
        win = nxt;
    }
}

// P1: entries of every (group, bin) pair: column sums over the group's offset rows
// (grid: groups x ceil(bins / 64); a workgroup = 64 bins x 4 quarters of the group's rows, eight rows in flight per
// thread: the kernel is a chain of dependent-latency loads otherwise -- 0.66 ms for 190 workgroups of 952 rows each)
__global__ __launch_bounds__(256) void kb_groupsum_kernel(KbPlan plan, KbScratch s) {
    __shared__ uint32_t part[256];
    const uint32_t nb = 1u << plan.c1, g = blockIdx.x;
    const uint64_t s0 = (uint64_t)g * plan.group;
    const uint32_t ns = (uint32_t)min((uint64_t)plan.group, (uint64_t)plan.n_slabs - s0);
    const uint32_t c = blockIdx.y * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    uint32_t acc = 0;
    if (c < nb) {
        const uint32_t i0 = (uint32_t)((uint64_t)ns * q / 4), i1 = (uint32_t)((uint64_t)ns * (q + 1) / 4);
        const uint16_t *o = s.off + (s0 + i0) * (nb + 1) + c;
        uint32_t i = i0;
        for (; i + 8 <= i1; i += 8) {
            uint32_t a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[u] = o[(size_t)u * (nb + 1)]; b[u] = o[(size_t)u * (nb + 1) + 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += b[u] - a[u];
            o += (size_t)8 * (nb + 1);
        }
        for (; i < i1; ++i, o += nb + 1) acc += (uint32_t)o[1] - (uint32_t)o[0];
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (q == 0 && c < nb) s.gn[(uint64_t)g * nb + c] = part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192];
    if (g == 0 && blockIdx.y == 0 && threadIdx.x == 0) s.ovf[0] = 0;
}

// P2: workgroup = bin c: prefix over the groups of the bin's pieces and entries (a pair (g, c) of n entries is
// ceil(n / CHUNK) pieces: one, unless the input is skewed); the pieces beyond a pair's first are listed for B's second launch
template <int CHUNK>
__global__ __launch_bounds__(256) void kb_binscan_kernel(KbPlan plan, KbScratch s) {
    __shared__ uint32_t wsum[32];
    __shared__ unsigned long long esum[256];
    const uint32_t nb = 1u << plan.c1, c = blockIdx.x;
    uint32_t carry_r = 0; unsigned long long carry_e = 0;
    for (uint32_t g0 = 0; g0 < plan.n_groups; g0 += 256) {
        const uint32_t g = g0 + threadIdx.x;
        const uint64_t pair = (uint64_t)g * nb + c;
        const uint32_t n = g < plan.n_groups ? s.gn[pair] : 0u, np = (n + CHUNK - 1) / CHUNK;
        uint32_t tot = 0;
        const uint32_t ex = kb_block_exscan(np, wsum, &tot);
        // entries: 64-bit sums, a serial prefix over the 4 waves' totals after a wave scan
        unsigned long long inc = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(inc, o); if ((int)(threadIdx.x & 63) >= o) inc += t; }
        esum[threadIdx.x] = inc;
        __syncthreads();
        unsigned long long wpre = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wpre += esum[w * 64 + 63];
        const unsigned long long etot = esum[63] + esum[127] + esum[191] + esum[255];
        if (g < plan.n_groups) {
            s.gpre_row[pair] = carry_r + ex;
            s.gpre_ent[pair] = carry_e + wpre + inc - n;
            if (np > 1) {
                const uint32_t at = atomicAdd(&s.ovf[0], np - 1);
                for (uint32_t p = 1; p < np; ++p) if (at + p - 1 < s.ovf_cap) { s.ovf[1 + 2 * (at + p - 1)] = (uint32_t)pair; s.ovf[2 + 2 * (at + p - 1)] = p; }
            }
        }
        carry_r += tot; carry_e += etot;
        __syncthreads();
    }
    if (threadIdx.x == 0) { s.bin_rows[c] = carry_r; s.bin_ent[c] = carry_e; }
}

// P3 (one workgroup): prefix over the bins -> the pass descriptor.  The pass's entries start at ent_base of the ring and
// its rows at row_base (chosen by the host from upper bounds: no host round trip).
__global__ __launch_bounds__(KB_THREADS) void kb_binfirst_kernel(KbPlan plan, KbScratch s, uint32_t pass_idx, unsigned long long ent_base,
                                                                 unsigned long long row_base, KdfCtl *ctl) {
    __shared__ uint32_t wsum[32];
    __shared__ unsigned long long esum[KB_THREADS];
    const uint32_t nb = 1u << plan.c1;
    KbPass *P = s.pass + pass_idx;
    const uint32_t r = threadIdx.x < nb ? s.bin_rows[threadIdx.x] : 0u;
    const unsigned long long e = threadIdx.x < nb ? s.bin_ent[threadIdx.x] : 0ull;
    uint32_t tot_r = 0;
    const uint32_t ex = kb_block_exscan(r, wsum, &tot_r);
    unsigned long long inc = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(inc, o); if ((int)(threadIdx.x & 63) >= o) inc += t; }
    esum[threadIdx.x] = inc;
    __syncthreads();
    unsigned long long wpre = 0, tot_e = 0;
    for (uint32_t w = 0; w < KB_THREADS / 64; ++w) { const unsigned long long v = esum[w * 64 + 63]; if (w < (threadIdx.x >> 6)) wpre += v; tot_e += v; }
    if (threadIdx.x < nb) {
        P->binrow_first[threadIdx.x] = ex; P->binent_first[threadIdx.x] = wpre + inc - e;
        // skewed: one coarse bin holds 6 % + 65 536 entries more than its share (a uniform hash keeps a bin of n entries
        // within a few sqrt(n) of it).  (Rounds 2-3 asked for TWICE the share, which a table with four coarse bins cannot
        // show unless one key is half of all windows.)
        if (nb > 1 && e * (unsigned long long)nb > tot_e + tot_e / 16 + 65536ull * nb) s.totals[7] = 1ull;
    }
    if (threadIdx.x == 0) {
        P->binrow_first[nb] = tot_r; P->binent_first[nb] = tot_e;
        P->ent_base = ent_base; P->row_base = row_base; P->n_entries = tot_e; P->n_rows = tot_r;
        s.totals[0] += tot_e;
        if (tot_e) atomicAdd(&ctl->windows[0], tot_e);
    }
}

// B: one piece: gather its runs (one per slab of its group), sort by fine bin, append to the ring with
// the piece's offset row.  All threads of the workgroup call it.
// The gather is FLAT: thread t takes the piece's entries t', t' + 64, ... of its wave's 64 * EPT consecutive entries, so a
// load instruction reads 64 consecutive entries (two or three runs of ~32) and all EPT loads of a thread are in flight
// together -- one global latency per piece.  The run of an entry is found by interpolation for a lane's first entry and by
// advancing for the next ones (a lane's consecutive entries are ~2 runs apart).  (A first version gave every run to a
// group of 32 lanes and staged the gathered entries in LDS: runs longer than 32 took a second, dependent load, a group of
// more than 512 slabs a second round of loads, and the kernel ran at 5.5-6.7 ms against 3.9 for round 2's in-place sort.)
template <int KW>
__device__ __forceinline__ void kb_sort_piece(const KbPlan &plan, const KbScratch &s, const KbPass *P, char *smem, uint32_t pair, uint32_t p)
{
    constexpr int CHUNK = KbCfg<KW>::CHUNK, EPT = CHUNK / KB_THREADS, SLAB = KbCfg<KW>::SLAB;
    uint64_t *slo = (uint64_t *)smem;
    KbEnt2 *s2 = (KbEnt2 *)smem;
    uint32_t *hist = (uint32_t *)(smem + (size_t)CHUNK * 8 * KW);       // [KB_F]
    uint32_t *offs = hist + KB_F;                                        // [KB_F]
    uint32_t *wsum = offs + KB_F;                                        // [32]
    unsigned long long *rsrc = (unsigned long long *)(wsum + 32);       // [KB_G_MAX] entry index in tmp of the run's first entry MINUS the run's position in the pair
    uint32_t *rpre = (uint32_t *)(rsrc + KB_G_MAX);                     // [KB_G_MAX + 4] position of the run's first entry in the pair's sequence; padded with the total
    const uint32_t nb = 1u << plan.c1;
    const uint32_t g = pair / nb, c = pair % nb;
    const uint32_t n_pair = s.gn[pair];
    if (n_pair <= p * (uint32_t)CHUNK) return;                          // an empty pair has no piece (uniform)
    const uint32_t lo_ = p * (uint32_t)CHUNK, len = min((uint32_t)CHUNK, n_pair - lo_);   // the piece is [lo_, lo_ + len) of the pair's entries
    const uint64_t row = P->row_base + P->binrow_first[c] + s.gpre_row[pair] + p;
    const unsigned long long dst0 = P->ent_base + P->binent_first[c] + s.gpre_ent[pair] + lo_;
    const int nf = 1 << plan.c2;
    KB_T_INIT;
    for (int i = threadIdx.x; i < KB_F; i += KB_THREADS) hist[i] = 0;
    // the run of slab s0 + t in bin c
    const uint64_t s0 = (uint64_t)g * plan.group;
    const uint32_t ns = (uint32_t)min((uint64_t)plan.group, (uint64_t)plan.n_slabs - s0);
    static_assert(KB_G_MAX == 2 * KB_THREADS, "a thread takes the runs of two consecutive slabs");
    uint32_t o0 = 0, rl0 = 0, o1 = 0, rl1 = 0;
    const uint32_t sa = 2 * threadIdx.x;
    if (sa < ns) {
        const uint16_t *o = s.off + (s0 + sa) * (uint64_t)(nb + 1) + c;
        o0 = o[0]; rl0 = (uint32_t)o[1] - o0;
        if (sa + 1 < ns) { o1 = o[nb + 1]; rl1 = (uint32_t)o[nb + 2] - o1; }
    }
    KB_T(s.trash, 0);                                                   // offsets loaded (the scan below waits for them)
    const uint32_t pre = kb_block_exscan(rl0 + rl1, wsum, nullptr);     // (barriers inside: hist is zeroed)
    KB_T(s.trash, 1);
    if (sa < ns) {
        rsrc[sa] = (s0 + sa) * (unsigned long long)SLAB + o0 - pre;
        rpre[sa] = pre;
        if (sa + 1 < ns) { rsrc[sa + 1] = (s0 + sa + 1) * (unsigned long long)SLAB + o1 - (pre + rl0); rpre[sa + 1] = pre + rl0; }
    }
    if (threadIdx.x < 4) rpre[ns + threadIdx.x] = n_pair;               // padding: the run search below stops there
    if (threadIdx.x == 0) { s.row_ent[row] = dst0; s.row_len[row] = len; }
    __syncthreads();
    KB_T(s.trash, 2);
    uint64_t klo[EPT], khi[KW == 2 ? EPT : 1];
    uint32_t br[EPT];
    {
        const KbEnt2 *tmp2 = (const KbEnt2 *)s.tmp;
        const uint32_t wbase = (threadIdx.x >> 6) * (64 * EPT) + (threadIdx.x & 63);   // this lane's first entry of the piece
        uint32_t cr = 0, cnx = 0;                                       // current run and the position (in the pair) where the next one starts
        unsigned long long cs = 0;
        if (wbase < len) {
            const uint32_t e = lo_ + wbase;
            uint32_t gu = (uint32_t)(((unsigned long long)e * ns) / n_pair);           // runs have nearly equal lengths: interpolate, then walk
            gu = gu < ns ? gu : ns - 1;
            while (rpre[gu] > e) --gu;
            while (rpre[gu + 1] <= e) ++gu;                             // (rpre[ns] = n_pair > e: stops at the last run)
            cr = gu; cnx = rpre[cr + 1]; cs = rsrc[cr];
        }
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const uint32_t i = wbase + 64 * q;
            klo[q] = 0; if constexpr (KW == 2) khi[q] = 0;
            if (i < len) {
                const uint32_t e = lo_ + i;
                if (e >= cnx) {
                    do { ++cr; cnx = rpre[cr + 1]; } while (e >= cnx);
                    cs = rsrc[cr];
                }
                const unsigned long long src = KB_ABL(plan, 256) ? ((cs + e) & 0x3FFFFull) : cs + e;      // (ablation 256: gather from 2 MB, L2 resident -- timing only)
                if constexpr (KW == 2) { const KbEnt2 v = tmp2[src]; klo[q] = v.lo; khi[q] = v.hi; }
                else klo[q] = s.tmp[src];
            }
        }
#ifdef KB_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KB_T(s.trash, 3);                                               // gather: addresses, issue, arrival
#endif
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const uint32_t i = wbase + 64 * q;
            if (i < len) {
                const uint32_t f = kb_fine(plan, klo[q]);               // the entry is the hash
                if (KB_ABL(plan, 1024)) { atomicAdd(&hist[f], 1u); br[q] = f << 16; }       // (ablation 1024: no ranks -- timing only)
                else br[q] = (f << 16) | atomicAdd(&hist[f], 1u);
            }
        }
    }
    __syncthreads();
    KB_T(s.trash, 4);                                                   // ranks
    {
        const uint32_t v = threadIdx.x < nf ? hist[threadIdx.x] : 0;
        const uint32_t ex = kb_block_exscan(v, wsum, nullptr);
        if (threadIdx.x < nf) {
            offs[threadIdx.x] = ex;
            s.chunk_off[row * plan.off_stride + threadIdx.x] = ex;
        }
        if (threadIdx.x == 0) s.chunk_off[row * plan.off_stride + nf] = len;
    }
    __syncthreads();
    KB_T(s.trash, 5);                                                   // scan
    {
        const uint32_t wbase = (threadIdx.x >> 6) * (64 * EPT) + (threadIdx.x & 63);
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            if (wbase + 64 * q < len) {
                const uint32_t pos = offs[br[q] >> 16] + (br[q] & 0xFFFF);
                if constexpr (KW == 2) s2[pos] = KbEnt2{klo[q], khi[q]};
                else slo[pos] = klo[q];
            }
        }
    }
    __syncthreads();
    KB_T(s.trash, 6);                                                   // scatter
    if (KB_ABL(plan, 512)) return;                                         // (ablation 512: no write-out -- timing only)
    if constexpr (KW == 2) {
        // the sorted piece is stored as piece-local structure of arrays -- len h words, then len hi
        // words, in the same 16 * len bytes -- because kernel C's gathers run faster on two 8-byte
        // streams than on 16-byte entries (measured: 15.5 vs 17.0 ms at k = 63)
        for (uint32_t i = threadIdx.x; i < len; i += KB_THREADS) {
            const KbEnt2 v = s2[i];
            s.ent[2 * dst0 + i] = v.lo; s.ent[2 * dst0 + len + i] = v.hi;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < len; i += KB_THREADS) s.ent[dst0 + i] = slo[i];
    }
#ifdef KB_TIMING
    KB_T(s.trash, 7);                                                   // write-out issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KB_T(s.trash, 8);                                                   // ... and drained
    if (threadIdx.x == 0) atomicAdd((unsigned long long *)&s.trash[8 + 9], 1ull);
#endif
}

// first launch: workgroup = pair (g, c), its first piece.  Workgroups are dealt to the 8 XCDs round robin by blockIdx
// and each XCD has its own L2: an XCD takes a contiguous eighth of the pairs, in (group, bin) order, so a group's offset
// rows (~1 MB) are fetched into one L2 once and hit there by the group's other pieces.  (grid = 8 * ceil(pairs / 8))
template <int KW>
__global__ __launch_bounds__(KB_THREADS) void kb_piecesort_kernel(KbPlan plan, KbScratch s, uint32_t pass_idx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t n_pairs = plan.n_groups << plan.c1, eighth = gridDim.x >> 3;
    const uint32_t pair = (blockIdx.x & 7) * eighth + (blockIdx.x >> 3);
    if (pair >= n_pairs) return;
    kb_sort_piece<KW>(plan, s, s.pass + pass_idx, smem, pair, 0);
}
// first launch, PIPELINED (the default): one persistent workgroup per CU walks its share of the pairs, three pieces in
// flight at different stages --
//   piece i + 3: the loads of its slabs' bin offsets and of its place in the ring are issued;
//   piece i + 2: run table from those offsets (block scan), then all EPT gather loads of every lane are issued into a
//                SECOND set of registers;
//   piece i + 1: its entries have arrived and wait in registers;
//   piece i:     rank by fine bin, scan, scatter into the LDS image, write-out.
// With one workgroup per CU (the image is 128 KB) kb_piecesort_kernel pays, piece after piece, a global latency for the
// offsets, another for the gather and the drain of 128 KB of stores, with the vector units idle meanwhile (7.6 us of issue
// in 24.6 us per piece); here the gather of the next piece is in flight under the LDS phases of this one and the stores
// drain under the next.  Barriers order LDS traffic only (kb_lds_barrier): a __syncthreads() would wait for the loads in
// flight.  An XCD still takes a contiguous eighth of the pairs, its workgroups interleaved over them, so a group's offset
// rows are fetched into one L2 once.
__device__ __forceinline__ uint32_t kb_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long kb_uni64(unsigned long long v) {
    return ((unsigned long long)kb_uni((uint32_t)(v >> 32)) << 32) | kb_uni((uint32_t)v);
}
__device__ __forceinline__ uint32_t kb_block_exscan_lds(uint32_t v, uint32_t *wsum /* >= 17 words LDS */, uint32_t tid) {
    const int lane = tid & 63, wave = tid >> 6, nw = KB_THREADS >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wave] = inc;
    kb_lds_barrier();
    if (wave == 0) {
        uint32_t sv = lane < nw ? wsum[lane] : 0, si = sv;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { uint32_t t = __shfl_up(si, o); if (lane >= o) si += t; }
        if (lane < nw) wsum[lane] = si - sv;
    }
    kb_lds_barrier();
    const uint32_t r = wsum[wave] + inc - v;
    kb_lds_barrier();                                  // (not needed by the pipelined piece sort, whose stages put barriers of their own before
                                                       // wsum is written again -- but without it, and without the barrier that ends an iteration,
                                                       // the kernel measured 4.65-4.67 against 4.59 ms: waves that stay in phase share the LDS better)
    return r;
}

// One iteration of the pipelined piece sort (see kb_piecesort_pipe_kernel) on ONE of its two register sets: piece `it`,
// whose entries were requested into (klo, khi) two iterations ago, is sorted and written out, and the entries of piece
// it + 2 are requested into the same registers.  EVERY global load and store of the body is unconditional -- lanes and
// whole iterations that have nothing to do read a valid dummy address and re-store a value to where it already is (or
// to the trash words) -- so that the number of vector-memory instructions between a request and its use is the same on
// every path: vmcnt retires in issue order, and only then can the compiler wait for "all but the N youngest" instead of
// for everything (which would include the gather just issued).
template <int KW>
struct KbPipe {
    static constexpr int CHUNK = KbCfg<KW>::CHUNK, EPT = CHUNK / KB_THREADS, SLAB = KbCfg<KW>::SLAB;
    // LDS
    uint64_t *slo; KbEnt2 *s2; uint32_t *hist, *offs, *wsum, *rpre; unsigned long long *rsrc;
    // loop invariants
    const KbPass *P; uint32_t nb, n_pairs, eighth, xcd, j0, nj; int n_it, nf; unsigned long long row_base, ent_base;
    // O -> R: what was loaded for the piece whose run table comes next (raw: made scalars where they are used)
    uint32_t o_pair = 0, o_npair = 0, o_raw = 0, o_raw2 = 0, o_rowoff = 0, o_binrow = 0; bool o_ok = false;
    unsigned long long o_entoff = 0, o_binent = 0;

    __device__ __forceinline__ void iter(const KbPlan &plan, const KbScratch &s, int it, uint64_t (&klo)[EPT], uint64_t (&khi)[KW == 2 ? EPT : 1],
                                         uint32_t &m_len, unsigned long long &m_row, unsigned long long &m_dst0) {
        // (the thread index is made opaque once per iteration: the compiler would otherwise keep every per-lane address of
        // the loop body in a register of its own across the whole loop -- ~50 VGPRs, i.e. spills at the 128 this kernel has)
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const uint32_t wbase = (tid >> 6) * (64 * EPT) + (tid & 63);      // this lane's first entry of a piece
        // ---- (a) LDS phases of piece `it`
        const uint32_t len = m_len;
        if (len) {                                                        // (uniform; LDS only)
            uint32_t rk[EPT / 2];                                         // ranks inside the fine bin, two to a register
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                uint32_t r = 0;
                if (wbase + 64 * q < len) r = atomicAdd(&hist[kb_fine(plan, klo[q])], 1u);
                if (q & 1) rk[q >> 1] |= r << 16; else rk[q >> 1] = r;
            }
            kb_lds_barrier();
            {
                const uint32_t v = (int)tid < nf ? hist[tid] : 0;
                const uint32_t ex = kb_block_exscan_lds(v, wsum, tid);
                if ((int)tid < nf) { offs[tid] = ex; hist[tid] = 0; }    // (hist: for the next piece)
            }
            kb_lds_barrier();
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                if (wbase + 64 * q < len) {
                    const uint32_t pos = offs[kb_fine(plan, klo[q])] + ((rk[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu);
                    if constexpr (KW == 2) s2[pos] = KbEnt2{klo[q], khi[q]};
                    else slo[pos] = klo[q];
                }
            }
            kb_lds_barrier();
        }
        const unsigned long long w_row = m_row, w_dst0 = m_dst0;
        // ---- (b) run table of piece it + 2 (its offsets were requested one iteration ago; LDS only)
        m_len = 0;
        uint32_t r_ns = 1, r_npair = 1;
        {
            const uint32_t npair = o_ok ? kb_uni(o_npair) : 0u;
            if (npair) {                                                  // (uniform) an empty pair has no piece
                const uint32_t g = o_pair / nb;
                const uint64_t s0 = (uint64_t)g * plan.group;
                const uint32_t ns = (uint32_t)min((uint64_t)plan.group, (uint64_t)plan.n_slabs - s0);
                const uint32_t sa = 2 * tid;                              // this thread's two consecutive slabs
                const uint32_t o_a0 = o_raw & 0xFFFFu, o_a1 = o_raw >> 16, o_b0 = o_raw2 & 0xFFFFu, o_b1 = o_raw2 >> 16;
                const uint32_t rl0 = sa < ns ? o_a1 - o_a0 : 0u, rl1 = sa + 1 < ns ? o_b1 - o_b0 : 0u;
                const uint32_t pre = kb_block_exscan_lds(rl0 + rl1, wsum, tid);
                if (sa < ns) {
                    rsrc[sa] = (s0 + sa) * (unsigned long long)SLAB + o_a0 - pre;
                    rpre[sa] = pre;
                }
                if (sa + 1 < ns) {
                    rsrc[sa + 1] = (s0 + sa + 1) * (unsigned long long)SLAB + o_b0 - (pre + rl0);
                    rpre[sa + 1] = pre + rl0;
                }
                if (tid < 4) rpre[ns + tid] = npair;
                m_row = row_base + kb_uni(o_binrow) + kb_uni(o_rowoff);
                m_dst0 = ent_base + kb_uni64(o_binent) + kb_uni64(o_entoff);
                m_len = min((uint32_t)CHUNK, npair);                     // its first piece; the others: kb_piecesort_more_kernel
                r_ns = ns; r_npair = npair;
                kb_lds_barrier();
            }
        }
        // ---- (c) write-out of piece `it`: its row of run offsets, its place in the ring, the image
        {
            uint32_t *co = len ? s.chunk_off + w_row * plan.off_stride : (uint32_t *)s.trash;
            const uint32_t ci = len ? min(tid, (uint32_t)nf) : 0u;
            co[ci] = (int)tid < nf ? offs[ci] : len;                     // (lanes past the row all store its last word)
            unsigned long long *re = len ? s.row_ent + w_row : (unsigned long long *)s.trash + 1;
            uint32_t *rl_ = len ? s.row_len + w_row : (uint32_t *)s.trash + 4;
            *re = w_dst0; *rl_ = len;
            const uint32_t last = len ? len - 1 : 0u;
            if constexpr (KW == 2) {
                // (the piece goes out as len h words, then len hi words -- see kb_sort_piece -- each array in 16-byte
                // stores of two words; the hi words start on an odd word when len is odd: their pairs start one further,
                // and the words a pair leaves out -- last h, first and last hi -- go out by themselves, every lane the same)
                uint64_t *d = len ? s.ent + 2 * w_dst0 : s.trash + 4;
                const uint64_t *sw = (const uint64_t *)s2;              // word 2 i = h of entry i, word 2 i + 1 = its hi
                const uint32_t ah = len & 1;
                const uint32_t npl = len >> 1, nph = len > ah ? (len - ah) >> 1 : 0u;
                ulonglong2 *dl = npl ? (ulonglong2 *)d : (ulonglong2 *)(s.trash + 4);
                ulonglong2 *dh = nph ? (ulonglong2 *)(d + len + ah) : (ulonglong2 *)(s.trash + 6);
                const uint32_t lastl = npl ? npl - 1 : 0u, lasth = nph ? nph - 1 : 0u, ahh = nph ? ah : 0u;
#pragma unroll
                for (int q = 0; q < EPT / 2; ++q) {
                    const uint32_t pl = min(tid + KB_THREADS * q, lastl), ph = min(tid + KB_THREADS * q, lasth);
                    dl[pl] = ulonglong2{sw[4 * pl], sw[4 * pl + 2]};
                    dh[ph] = ulonglong2{sw[2 * (ahh + 2 * ph) + 1], sw[2 * (ahh + 2 * ph) + 3]};
                }
                d[last] = sw[2 * last]; d[len] = sw[1]; d[len + last] = sw[2 * last + 1];
            } else {
                // 16 bytes per lane and store (8-byte stores run at 0.54-0.70 of that rate, MI355X_MICROARCH.md; the
                // write-out ISSUE was 14.7 % of the one-piece-per-workgroup kernel's time): the pairs start at the first
                // entry whose place in the ring is 16-byte aligned; the piece's first and last entry go out once more by
                // themselves (one store each, every lane the same word), whichever of them a pair has left out
                uint64_t *d = len ? s.ent + w_dst0 : s.trash + 4;
                const uint32_t a = len ? (uint32_t)(w_dst0 & 1) : 0u;
                const uint32_t npairs = len > a ? (len - a) >> 1 : 0u;
                ulonglong2 *dp = npairs ? (ulonglong2 *)(d + a) : (ulonglong2 *)(s.trash + 4);
                const uint64_t *sp = npairs ? slo + a : slo;
                const uint32_t lastp = npairs ? npairs - 1 : 0u;
#pragma unroll
                for (int q = 0; q < EPT / 2; ++q) {
                    const uint32_t pi = min(tid + KB_THREADS * q, lastp); // (lanes past the piece store its last pair again)
                    dp[pi] = ulonglong2{sp[2 * pi], sp[2 * pi + 1]};
                }
                d[0] = slo[0]; d[last] = slo[last];
            }
        }
        // ---- (e) offsets of piece it + 3
        {
            const int i3 = it + 3;
            const uint32_t jj = j0 + (uint32_t)i3 * nj;
            const uint32_t pair = xcd * eighth + jj;
            o_ok = i3 >= 0 && i3 < n_it && jj < eighth && pair < n_pairs;
            const uint32_t pc = o_ok ? pair : 0u;                         // (nothing to do: pair 0 is read, and ignored)
            const uint32_t g = pc / nb, c = pc % nb;
            const uint64_t s0 = (uint64_t)g * plan.group;
            const uint32_t ns = (uint32_t)min((uint64_t)plan.group, (uint64_t)plan.n_slabs - s0);
            o_pair = pc;
            o_npair = s.gn[pc]; o_rowoff = s.gpre_row[pc]; o_entoff = s.gpre_ent[pc];
            o_binrow = P->binrow_first[c]; o_binent = P->binent_first[c];
            const uint32_t ta = 2 * tid < ns ? 2 * tid : 0u, tb = 2 * tid + 1 < ns ? 2 * tid + 1 : 0u;
            __builtin_memcpy(&o_raw, s.off + (s0 + ta) * (uint64_t)(nb + 1) + c, 4);    // off[c], off[c + 1] of its two slabs: split where they are used
            __builtin_memcpy(&o_raw2, s.off + (s0 + tb) * (uint64_t)(nb + 1) + c, 4);
        }
        // ---- (d) gather of piece it + 2 into the registers piece `it` has left: all loads of a lane in flight together
        {
            const uint32_t glen = m_len;
            uint32_t cr = 0, cnx = 0; unsigned long long cs = 0;
            if (wbase < glen) {
                const uint32_t e = wbase;
                uint32_t gu = (uint32_t)(((unsigned long long)e * r_ns) / r_npair);
                gu = gu < r_ns ? gu : r_ns - 1;
                while (rpre[gu] > e) --gu;
                while (rpre[gu + 1] <= e) ++gu;
                cr = gu; cnx = rpre[cr + 1]; cs = rsrc[cr];
            }
            const KbEnt2 *tmp2 = (const KbEnt2 *)s.tmp;
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const uint32_t e = wbase + 64 * q;
                unsigned long long src = 0;                               // (lanes past the piece read entry 0, and ignore it)
                if (e < glen) {
                    if (e >= cnx) {
                        do { ++cr; cnx = rpre[cr + 1]; } while (e >= cnx);
                        cs = rsrc[cr];
                    }
                    src = cs + e;
                }
                if constexpr (KW == 2) { const KbEnt2 v = tmp2[src]; klo[q] = v.lo; khi[q] = v.hi; }
                else klo[q] = s.tmp[src];
            }
        }
        // (the image is read by (c) and written by the next (a) after its two barriers; the run table is read by (d) and
        // written by the next (b) after the barriers of (a) or of its own scan)
        kb_lds_barrier();
    }
};

template <int KW>
__global__ __launch_bounds__(KB_THREADS) void kb_piecesort_pipe_kernel(KbPlan plan, KbScratch s, uint32_t pass_idx)
{
    constexpr int CHUNK = KbCfg<KW>::CHUNK, EPT = CHUNK / KB_THREADS;
    static_assert(EPT % 2 == 0 && CHUNK <= 65536, "ranks are kept two to a register");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    KbPipe<KW> p;
    p.slo = (uint64_t *)smem; p.s2 = (KbEnt2 *)smem;
    p.hist = (uint32_t *)(smem + (size_t)CHUNK * 8 * KW);               // [KB_F]
    p.offs = p.hist + KB_F;                                              // [KB_F]
    p.wsum = p.offs + KB_F;                                              // [32]
    p.rsrc = (unsigned long long *)(p.wsum + 32);                       // [KB_G_MAX]
    p.rpre = (uint32_t *)(p.rsrc + KB_G_MAX);                           // [KB_G_MAX + 4]
    p.P = s.pass + pass_idx;
    p.nb = 1u << plan.c1; p.n_pairs = plan.n_groups << plan.c1;
    p.eighth = (p.n_pairs + 7) >> 3;
    p.xcd = blockIdx.x & 7; p.j0 = blockIdx.x >> 3; p.nj = gridDim.x >> 3;                 // (the grid is a multiple of 8)
    p.n_it = p.j0 < p.eighth ? (int)((p.eighth - p.j0 + p.nj - 1) / p.nj) : 0;
    p.nf = 1 << plan.c2;
    p.row_base = p.P->row_base; p.ent_base = p.P->ent_base;
    for (int i = threadIdx.x; i < KB_F; i += KB_THREADS) p.hist[i] = 0;
    kb_lds_barrier();
    // two register sets: piece k lives in set k & 1 from the iteration that requests it (k - 2) to the one that sorts it (k)
    uint64_t alo[EPT], ahi[KW == 2 ? EPT : 1], blo[EPT], bhi[KW == 2 ? EPT : 1];
    uint32_t a_len = 0, b_len = 0; unsigned long long a_row = 0, a_dst0 = 0, b_row = 0, b_dst0 = 0;
#pragma unroll
    for (int q = 0; q < EPT; ++q) { alo[q] = 0; blo[q] = 0; if constexpr (KW == 2) { ahi[q] = 0; bhi[q] = 0; } }
    for (int it = -4; it < p.n_it; it += 2) {
        p.iter(plan, s, it, alo, ahi, a_len, a_row, a_dst0);
        p.iter(plan, s, it + 1, blo, bhi, b_len, b_row, b_dst0);
    }
}

// second launch: the pieces beyond a pair's first (skewed input only); a few persistent workgroups walk the list
template <int KW>
__global__ __launch_bounds__(KB_THREADS) void kb_piecesort_more_kernel(KbPlan plan, KbScratch s, uint32_t pass_idx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t n = min(s.ovf[0], s.ovf_cap);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        kb_sort_piece<KW>(plan, s, s.pass + pass_idx, smem, s.ovf[1 + 2 * i], s.ovf[2 + 2 * i]);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// C: one workgroup per TABLE bucket.
enum { KB_MODE_INSERT = 0, KB_MODE_FILTERED = 1 };

__device__ __forceinline__ void kb_lds_sat_add(uint32_t *p, uint32_t add) {
    uint32_t old = atomicAdd(p, add);
    if (old + add < old || old + add == 0xFFFFFFFFu) atomicMax(p, 0xFFFFFFFFu);
}

// The runs of partition bucket (c, f) over every pending pass, as ONE flat list r = 0 .. n_runs - 1 (run = the bucket's
// slice of one sorted chunk).  setup(): per pass the bin's first chunk row / first entry / end in LDS (one global latency
// for all passes); locate(): run r -> (first entry, length, wide keys: distance from the h words to the hi words).
#define KB_RI_LDS_BYTES ((KB_MAX_PASS + 2) * 4 + KB_MAX_PASS * 8)
struct KbRunIndex {
    uint32_t *ppref;                 // [KB_MAX_PASS + 1] runs (pieces) of this bin before pass m
    unsigned long long *prow;        // [KB_MAX_PASS] row of the bin's first piece in pass m
    __device__ __forceinline__ void bind(char *p) {
        prow = (unsigned long long *)p; ppref = (uint32_t *)(prow + KB_MAX_PASS);
    }
    // all threads of the workgroup call it (it ends with a barrier); returns the number of runs
    __device__ __forceinline__ uint32_t setup(const KbPlan &plan, const KbScratch &s, uint32_t c) {
        if (threadIdx.x < 64) {
            uint32_t n = 0;
            if (threadIdx.x < plan.n_pass) {
                const KbPass *P = s.pass + threadIdx.x;
                const uint32_t j0 = P->binrow_first[c], j1 = P->binrow_first[c + 1];
                n = j1 - j0;
                prow[threadIdx.x] = P->row_base + j0;
            }
            uint32_t inc = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if ((int)threadIdx.x >= o) inc += t; }
            if (threadIdx.x < KB_MAX_PASS) ppref[threadIdx.x + 1] = inc;
            if (threadIdx.x == 0) ppref[0] = 0;
        }
        __syncthreads();
        return ppref[plan.n_pass];
    }
    template <int KW>
    __device__ __forceinline__ void locate(const KbPlan &plan, const KbScratch &s, uint32_t f, uint32_t r,
                                           unsigned long long &first, uint32_t &len, uint32_t &hioff) const {
        uint32_t m = 0;
        if (plan.n_pass > 1) {                                  // largest m with ppref[m] <= r
            uint32_t lo = 0, hi = plan.n_pass;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ppref[mid] <= r) lo = mid; else hi = mid; }
            m = lo;
        }
        const unsigned long long row = prow[m] + (r - ppref[m]);      // the r-th piece of the bin
        const unsigned long long orow = row * plan.off_stride;
        const uint32_t r0 = s.chunk_off[orow + f], r1 = s.chunk_off[orow + f + 1];
        len = r1 - r0;
        const unsigned long long cs = s.row_ent[row];           // first entry of the piece
        if constexpr (KW == 2) {                                // wide: word index of the run's h words; the hi words follow the piece's h words
            hioff = s.row_len[row];
            first = 2 * cs + r0;
        } else { hioff = 0; first = cs + r0; }
    }
};

// count += 1 at LDS slot sl for the lanes with `hit` (the whole wave calls it together).  AGG: when every hit lane names
// the SAME slot -- a key of enormous multiplicity: a homopolymer k-mer took 1.4 % of all windows of a repeat-rich genome, all
// of them in one workgroup -- one lane adds the lot instead of 64 adds serialising on one LDS address (kernel C 14.6 -> 10.2
// ms there, pass 24.8 -> 20.4 ms).  The test costs every wave ~8 instructions per key (+3 % on the kernel for a uniform
// genome), so it lives in its own instantiation of the kernel (VAR 2), launched when a pending pass reported a skewed coarse
// histogram (totals[7], kb_scan1_kernel) -- known before kernel C runs, since C is deferred.
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

// narrow keys, one key: linear probing in the LDS slice from slot `sl`
template <int MODE>
__device__ __forceinline__ void kb_probe_narrow(uint64_t *tlo, uint32_t *tcnt, uint32_t bmask, uint64_t klo, uint32_t sl,
                                                uint32_t &claimed, bool &failed) {
    // FOUR slots per iteration, their reads in flight together: a wave runs as many iterations as its longest probe, and
    // every iteration costs scalar exec-mask bookkeeping on the CU's one scalar unit
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

// Wide keys in the LDS slice: a slot is CLAIMED BY ITS HASH WORD -- one 64-bit CAS EMPTY -> h on tlo, as for narrow keys;
// the winner then stores the key's high word into thi (which reads EMPTY until then).  A prober that finds its h compares
// thi: equal = the key; EMPTY = claimed a moment ago, not yet published (BLOCKED: retried under a wave-uniform loop, no
// lane ever waits inside a divergent loop); anything else = another key with the same 64-bit hash (2^-64 per pair): probe on.
// (Rounds 1-2 claimed by `hi | PENDING` on thi and published lo, final hi: every slot comparison took two words and a flag
// test, and the bucket kernel issued more scalar than vector instructions -- 3.9 G against 3.6 G per pass at k = 63.)
// A key whose h IS the empty marker cannot be stored this way: its bucket is handed to the replay path (direct kernels).
// One ATTEMPT from slot `sl`: 0 done, 1 bucket full, 2 blocked.
template <int MODE>
__device__ __forceinline__ int kb_probe_wide_once(uint64_t *tlo, uint64_t *thi, uint32_t *tcnt, uint32_t bmask,
                                                  uint64_t klo, uint64_t khi, uint32_t sl, uint32_t &claimed, uint32_t add = 1u) {
    for (uint32_t n = 0; n <= bmask; ++n) {
        uint64_t c = __hip_atomic_load(&tlo[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (c == KDF_EMPTY) {
            if (MODE != KB_MODE_INSERT) return 0;                 // FILTERED: absent
            c = atomicCAS((unsigned long long *)&tlo[sl], KDF_EMPTY, klo);
            if (c == KDF_EMPTY) {
                __hip_atomic_store(&thi[sl], khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                claimed++;
                if (add == 1u) atomicAdd(&tcnt[sl], 1u); else kb_lds_sat_add(&tcnt[sl], add);
                return 0;
            }
        }
        if (c == klo) {
            const uint64_t h2 = __hip_atomic_load(&thi[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (h2 == khi) { if (add == 1u) atomicAdd(&tcnt[sl], 1u); else kb_lds_sat_add(&tcnt[sl], add); return 0; }
            if (h2 == KDF_EMPTY) return 2;                        // claimed, not yet published
        }
        sl = (sl + 1) & bmask;
    }
    return 1;
}
// all lanes of the wave call this together (`todo`: this lane has a key); returns when every key is placed
template <int MODE>
__device__ __forceinline__ void kb_probe_wide_wave(uint64_t *tlo, uint64_t *thi, uint32_t *tcnt, uint32_t bmask, bool todo,
                                                   uint64_t klo, uint64_t khi, uint32_t sl, uint32_t &claimed, bool &failed, uint32_t add = 1u) {
    while (__any(todo)) {
        if (todo) {
            const int res = kb_probe_wide_once<MODE>(tlo, thi, tcnt, bmask, klo, khi, sl, claimed, add);
            if (res != 2) { todo = false; if (res == 1) failed = true; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The first KB_C_LA slots of every probe sequence are read at once and resolved in straight-line code (a wave pays the
// longest probe of its 64 lanes for every key it probes in a loop); the keys that need more go to a wave-private queue
// in LDS (ballot + mbcnt, no atomics, no barrier) and are probed densely, one per lane, after the batch.  Measured on
// the bench pass: kernel C 7.6 -> 6.5 ms at k = 31, 16.0 -> 12.7 ms at k = 63.
// VAR 1: the default.  VAR 2: + wave-aggregated adds and heavy-bucket listing (skewed input).
#ifndef KB_C_LA
#define KB_C_LA    2                   // slots of the probe sequence read up front
#endif
#define KB_C_QCAPT(KW, CT_) (((KW) == 2 ? KB_C_WQ_W : 128) * ((CT_) / 64))   // queue entries per workgroup
#define KB_C_LDS_T(KW, BB, CT_, RUNS_) (((size_t)8 * (KW) + 4) * ((size_t)1 << (BB)) + (2 + 32 + (RUNS_)) * 4 + (size_t)(RUNS_) * 8 + ((KW) == 2 ? (size_t)(RUNS_) * 4 : 0) \
                          + KB_C_QCAPT(KW, CT_) * ((KW) == 2 ? 18 : 10) + 16 + ((RUNS_) + 4) * 4 + KB_RI_LDS_BYTES)
#define KB_C_LDS(KW, BB) KB_C_LDS_T(KW, BB, KB_C_CTB(KW, (BB) > KB_BB_SMALL(KW)), KB_C_RUNS_T(KW, (BB) > KB_BB_SMALL(KW)))
template <int KW, int MODE, int VAR, bool BIG = false, bool DUMP = false>
__global__ __launch_bounds__(KB_C_CTB(KW, BIG)) __attribute__((amdgpu_waves_per_eu(BIG ? 4 : KB_C_WPE, BIG ? 4 : KB_C_WPE))) void kb_bucket_kernel(
    KbPlan plan, KbScratch s, KdfTable t, KdfCtl *ctl, int table_nonempty)
{
    constexpr uint32_t CT = KB_C_CTB(KW, BIG), QCAP = KB_C_QCAPT(KW, CT);      // threads and queue entries per workgroup
    constexpr uint32_t RUNS = KB_C_RUNS_T(KW, BIG);                                // runs staged per round
    static_assert(RUNS <= CT, "a thread per run");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t B = 1u << plan.bucket_bits;
    uint64_t *tlo = (uint64_t *)smem;                         // [B]
    uint64_t *thi = KW == 2 ? tlo + B : nullptr;              // [B] wide
    uint32_t *tcnt = (uint32_t *)(smem + (size_t)B * 8 * KW); // [B]
    uint32_t &sh_failed = tcnt[B], &sh_claimed = tcnt[B + 1];
    uint32_t *wsum = tcnt + B + 2;                            // [32]
    uint32_t *run_pref = wsum + 32;                           // [RUNS] exclusive prefix of run lengths
    unsigned long long *run_first = (unsigned long long *)(run_pref + RUNS);   // [RUNS] (B + 34 + RUNS is even: 8-aligned)
    uint32_t *run_hi = (uint32_t *)(run_first + RUNS);   // [RUNS] wide keys only: distance (words) from a run's h words to its hi words
    uint64_t *qk = (uint64_t *)(run_hi + (KW == 2 ? RUNS : 0));   // [QCAP] keys whose probe goes past the lookahead (per wave: QCAP / waves)
    uint16_t *qs = (uint16_t *)(qk + QCAP);              // [QCAP] slot to go on from
    uint64_t *qk2 = (uint64_t *)((uint32_t *)(qs + QCAP) + 2 + RUNS + 4);   // [QCAP] wide keys: hi words of the queued keys
    uint32_t *rpw = (uint32_t *)(qs + QCAP) + 2;                                  // [RUNS + 4] run_pref shifted by one, padded with `total`
    KbRunIndex ri;
    ri.bind((char *)(qk2 + (KW == 2 ? QCAP : 0)));

    // Workgroups are dealt to the 8 XCDs round robin by blockIdx, and each XCD has its own L2.  Neighbouring buckets
    // (f, f + 1 of one coarse bin) read neighbouring runs of the same chunks -- they share the cache line at every run
    // boundary and the lines of the offset table -- so an XCD takes a contiguous eighth of the buckets, in order.
    const uint32_t nbk = gridDim.x;
    const uint64_t bucket = (nbk & 7) ? blockIdx.x : (uint64_t)(blockIdx.x & 7) * (nbk >> 3) + (blockIdx.x >> 3);   // table bucket
    const uint64_t pb = bucket >> plan.sub_bits;              // partition bucket holding its entries
    const uint32_t c = (uint32_t)(pb >> plan.c2), f = (uint32_t)(pb & ((1u << plan.c2) - 1));
    const uint64_t slot0 = bucket << plan.bucket_bits;        // first slot of the bucket in HBM
    if (threadIdx.x == 0) { sh_failed = 0; sh_claimed = 0; }
    bool failed = false;
    KB_T_INIT;
    if (table_nonempty) {
        for (uint32_t i = threadIdx.x; i < B; i += CT) {
            tlo[i] = t.lo[slot0 + i];
            if constexpr (KW == 2) {
                thi[i] = t.hi[slot0 + i];
                // (a stored key whose hash word equals the empty marker -- put there by the direct kernels, 2^-64 per key --
                // cannot live in a slice whose emptiness is read off the hash word: the bucket goes to the replay path)
                if ((tlo[i] == KDF_EMPTY) != (thi[i] == KDF_EMPTY)) failed = true;
            }
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
    KB_T(s.trash, 30);                                        // slice initialised (or its loads issued)
    const uint32_t n_runs = ri.setup(plan, s, c);             // (barrier inside)
    KB_T(s.trash, 31);                                        // pass descriptors

    constexpr uint32_t WQ = QCAP / (CT / 64);
    uint64_t *wqk = qk + (threadIdx.x >> 6) * WQ;
    uint64_t *wqk2 = qk2 + (threadIdx.x >> 6) * WQ;
    uint16_t *wqs = qs + (threadIdx.x >> 6) * WQ;
    uint32_t wq_n = 0;
    const uint32_t bmask = B - 1;
    const uint32_t hsh_r = 64 - plan.log2cap;                 // home slot = h >> hsh_r (binned tables have no hash_shift)
    uint32_t claimed = 0;
    // Runs of this bucket: one per chunk of its coarse bin and pending pass.  Their bounds are
    // fetched by all threads at once (one global latency, not one per run) and
    // laid out in LDS as a flat work list; threads then take entries round
    // robin, so every lane is busy whatever the run lengths are.
    for (uint32_t rb = 0; rb < n_runs; rb += RUNS) {
        uint32_t len = 0, hioff = 0; unsigned long long first = 0;
        if (threadIdx.x < RUNS && rb + threadIdx.x < n_runs) ri.locate<KW>(plan, s, f, rb + threadIdx.x, first, len, hioff);
        uint32_t total = 0;
        KB_T(s.trash, 32);                                    // run bounds requested
        const uint32_t ex = kb_block_exscan(len, wsum, &total);
        KB_T(s.trash, 33);                                    // ... arrived, scanned
        if constexpr (VAR == 2) {
            // A heavy bucket (a few keys of enormous multiplicity) is not for ONE workgroup: it is left untouched here, as
            // a failed bucket would be, and KB_HV_SLICES workgroups share its runs afterwards (kb_heavy_slice_kernel).
            if (rb == 0 && total > KB_C_HEAVY && s.hv_ctr && plan.sub_bits == 0) {   // (the host launches the heavy kernels under the same conditions)
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
                    // (count --if: the keys stay where they are, kb_heavy_filtered_kernel adds to the counts in HBM)
                    if (MODE == KB_MODE_INSERT && !table_nonempty)
                        for (uint32_t i = threadIdx.x; i < B; i += CT) {
                            t.lo[slot0 + i] = KDF_EMPTY; t.cnt[slot0 + i] = 0;
                            if constexpr (KW == 2) t.hi[slot0 + i] = KDF_EMPTY;
                        }
                    return;
                }
                __syncthreads();
            }
        }
        if (threadIdx.x < RUNS) { run_pref[threadIdx.x] = ex; run_first[threadIdx.x] = first; if constexpr (KW == 2) run_hi[threadIdx.x] = hioff; }
        if (threadIdx.x < RUNS) rpw[threadIdx.x + 1] = ex;              // (threads past the last run hold ex == total)
        if (threadIdx.x < 3) rpw[RUNS + 1 + threadIdx.x] = total;
        if (threadIdx.x == 0) rpw[0] = 0;
        __syncthreads();
        const uint32_t nruns = n_runs - rb < (uint32_t)RUNS ? n_runs - rb : (uint32_t)RUNS;
        constexpr int EPB = KW == 2 ? KB_C_EPB_W : KB_C_EPB_N;    // entries per thread per batch: EPB (x KW) loads in flight per lane
        // (ei * inv_total) >> 32 ~= ei * nruns / total
        const unsigned long long inv_total = total ? (((unsigned long long)nruns << 32) / total) : 0;
        for (uint32_t e0 = 0; e0 < total; e0 += CT * EPB) {   // wave-uniform trip count
          uint64_t bklo[EPB], bkhi[KW == 2 ? EPB : 1];
          // Flat index of this thread's q-th entry of the batch.  Narrow keys: a WAVE takes 64 * EPB
          // consecutive entries, lane l's q-th entry is wbase + 64 q -- one load instruction still reads 64
          // consecutive entries, and a lane's consecutive entries are about one run (64 entries) further on,
          // so the run is searched once per batch and then only advanced.
          // Wide keys have 16-entry runs (four runs per step): they keep the per-entry windowed search.
          constexpr bool WAVE_SPANS = KW == 1;
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
                    bklo[q] = s.ent[KB_ABL(plan, 256) ? ((cf + (ei - cpf)) & 0x3FFFFull) : cf + (ei - cpf)];      // (ablation 256: entries from 2 MB -- timing only)
                }
            }
          } else {
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
                        bklo[q0 + g] = s.ent[src];
                        if constexpr (KW == 2) bkhi[q0 + g] = s.ent[src + run_hi[lo4[g]]];
                    }
                }
            }
          }
#ifdef KB_TIMING
          KB_T(s.trash, 34);                                  // run table in LDS, entry loads issued
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          KB_T(s.trash, 35);                                  // entries arrived
#endif
          if constexpr (KW == 1) {
            constexpr int G = 4;                                 // keys resolved together: G * KB_C_LA LDS reads in flight
#pragma unroll
            for (int q0 = 0; q0 < EPB; q0 += G) {
                if (wbase - (threadIdx.x & 63) + QSTEP * q0 >= total) break;      // wave-uniform: nothing left for this wave in this batch
                uint64_t cur[G][KB_C_LA]; uint32_t sl0[G]; bool td[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t ei = wbase + QSTEP * (q0 + g);
                    const uint64_t home = bklo[q0 + g] >> hsh_r;                  // the entry is the hash
                    td[g] = ei < total && !(plan.sub_bits && (home >> plan.bucket_bits) != bucket);
                    sl0[g] = (uint32_t)home & bmask;
#pragma unroll
                    // (Measured and dropped: a mirror of slot 0 at tlo[B], so that the two slots are always adjacent and
                    // go out as ONE ds_read2_b64 -- 4.89-4.92 against 4.80 ms.)
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
            KB_T(s.trash, 36);                                // lookahead resolve
            // drain this wave's queue: dense probing, one queued key per lane (wave-private: no barrier)
            {
                const uint32_t nq = wq_n < WQ ? wq_n : WQ;
                for (uint32_t i = threadIdx.x & 63; i < nq; i += 64)
                    kb_probe_narrow<MODE>(tlo, tcnt, bmask, wqk[i], wqs[i], claimed, failed);
                wq_n = 0;
            }
          } else {
            // Two-word keys, same scheme on the HASH words (claim protocol: kb_probe_wide_once above): two slots of tlo read
            // up front, the first that holds h or is empty resolved in straight-line code -- CAS where it read empty, the
            // winner publishes its high word, a lane that found h compares the high word -- and whatever needs more (both
            // slots taken by other keys, a lost CAS, a slot not yet published) goes to the wave queue, drained under the
            // wave-uniform retry loop.
            constexpr int G = 2;
#pragma unroll
            for (int q0 = 0; q0 < EPB; q0 += G) {
                if (e0 + q0 * CT >= total) break;
                uint64_t cur[G][KB_C_LA]; uint32_t sl0[G]; bool td[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t ei = wbase + QSTEP * (q0 + g);
                    const uint64_t home = bklo[q0 + g] >> hsh_r;                  // the entry's first word is the hash
                    td[g] = ei < total && !(plan.sub_bits && (home >> plan.bucket_bits) != bucket);
                    if (td[g] && bklo[q0 + g] == KDF_EMPTY) { failed = true; td[g] = false; }   // (2^-64: the replay path stores it)
                    sl0[g] = (uint32_t)home & bmask;
#pragma unroll
                    for (int i = 0; i < KB_C_LA; ++i) cur[g][i] = __hip_atomic_load(&tlo[(sl0[g] + i) & bmask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < KB_C_LA; ++i) asm volatile("" : "+v"(cur[g][i]) :: "memory");
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint64_t klo = bklo[q0 + g], khi = bkhi[q0 + g];
                    uint32_t r = KB_C_LA; bool found = false;                // found: the slot holds h
#pragma unroll
                    for (int i = KB_C_LA - 1; i >= 0; --i) {
                        const bool m = cur[g][i] == klo, e = cur[g][i] == KDF_EMPTY;
                        if (m || e) { r = (uint32_t)i; found = m; }
                    }
                    if (!__any(td[g])) continue;
                    uint32_t sl = (sl0[g] + r) & bmask;
                    bool hit = false;
                    bool more = td[g] && r == KB_C_LA;
                    if (td[g] && !more && !found) {                          // read empty
                        if constexpr (MODE == KB_MODE_INSERT) {
                            const uint64_t old = atomicCAS((unsigned long long *)&tlo[sl], KDF_EMPTY, klo);
                            if (old == KDF_EMPTY) {
                                __hip_atomic_store(&thi[sl], khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                claimed++; hit = true;
                            } else if (old == klo) found = true;             // taken meanwhile by this hash: compare the high word below
                            else { more = true; sl = (sl + 1) & bmask; }
                        }                                                    // FILTERED: absent
                    }
                    if (td[g] && found) {
                        const uint64_t h2 = __hip_atomic_load(&thi[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (h2 == khi) hit = true;
                        else { more = true; if (h2 != KDF_EMPTY) sl = (sl + 1) & bmask; }   // not yet published: the queue looks again; another key: probe on
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
          }
        }
        KB_T(s.trash, 37);                                    // queue drained
        __syncthreads();       // run_pref / run_first are rewritten by the next round
        KB_T(s.trash, 38);                                    // the other waves
    }
    if (failed) atomicOr(&sh_failed, 1u);
    if (claimed) atomicAdd(&sh_claimed, claimed);
    if (DUMP && threadIdx.x == 0) wsum[0] = 0;               // (the fused dump's counter; the scans are over)
    __syncthreads();
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
    // LDS counts were advanced with plain (wrapping, non-returning) adds.  A flush
    // adds fewer than 2^32 to a slot, so a slot wrapped iff its new value is below
    // the value it had in HBM: saturate those (Jellyfish's 4-byte counter).
    // two slots per lane and step: 16-byte LDS reads and HBM stores for the keys, 8-byte ones for the counts
    // (slot0 is a multiple of B, B is even: everything stays aligned)
    if (KB_ABL(plan, 512)) return;                                         // (ablation 512: no write-back -- timing only)
    for (uint32_t i = threadIdx.x; i < B / 2; i += CT) {
        if constexpr (MODE == KB_MODE_INSERT) {
            ((ulonglong2 *)(t.lo + slot0))[i] = ((const ulonglong2 *)tlo)[i];
            if constexpr (KW == 2) ((ulonglong2 *)(t.hi + slot0))[i] = ((const ulonglong2 *)thi)[i];
        }
        uint2 c2 = ((const uint2 *)tcnt)[i];
        if (table_nonempty) {
            const uint2 o = ((const uint2 *)(t.cnt + slot0))[i];
            if (c2.x < o.x || c2.y < o.y) {
                if (c2.x < o.x) c2.x = 0xFFFFFFFFu;
                if (c2.y < o.y) c2.y = 0xFFFFFFFFu;
                if constexpr (DUMP) ((uint2 *)tcnt)[i] = c2;     // (the dump below reads the counts from LDS)
            }
        }
        ((uint2 *)(t.cnt + slot0))[i] = c2;
    }
    if (threadIdx.x == 0 && sh_claimed)
        atomicAdd(&ctl->distinct[(bucket % KDF_SHARDS) * 16], (unsigned long long)sh_claimed);
    // DUMP: `dump -L dump_min` while the bucket is here.  The last flush before a dump sees every key's final count (every
    // flush rewrites every bucket), so the separate pass over the table -- 6.4 GB for the bench's 2^29 slots, 1.4 ms -- is
    // saved.  A thread counts what it keeps among the slot pairs it has just written back (its own LDS words: no barrier),
    // the waves add up in LDS, ONE global atomic per bucket reserves the range, and the kept slots go out as (key, count).
    // Buckets that failed or were left to the heavy-bucket kernels never get here: the host then dumps the usual way.
    // (Measured the same: the reservation issued BEFORE the write-back, to hide its round trip under those stores,
    // profiles/r03b_fused_dump.txt.)
    if constexpr (DUMP) {
        const uint32_t dm = plan.dump_min;
        uint32_t n = 0;
        for (uint32_t i = threadIdx.x; i < B / 2; i += CT) {
            const uint2 c2 = ((const uint2 *)tcnt)[i];
            n += (c2.x >= dm) + (c2.y >= dm);
        }
        uint32_t inc = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if ((int)(threadIdx.x & 63) >= o) inc += v; }
        uint32_t wb = 0;
        if ((threadIdx.x & 63) == 63 && inc) wb = atomicAdd(&wsum[0], inc);
        wb = __shfl(wb, 63);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tot = wsum[0];
            *(unsigned long long *)(wsum + 2) = tot ? atomicAdd(&ctl->cursor, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        if (n) {
            unsigned long long pos = *(const unsigned long long *)(wsum + 2) + wb + (inc - n);
            for (uint32_t i = threadIdx.x; i < B / 2; i += CT) {
                const uint2 c2 = ((const uint2 *)tcnt)[i];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const uint32_t cv = e ? c2.y : c2.x;
                    if (cv >= dm) {
                        if (pos < s.dump_cap) {
                            uint64_t hi = 0;
                            if constexpr (KW == 2) hi = thi[2 * i + e];
                            s.dump_lo[pos] = kdf_key_lo(tlo[2 * i + e], hi);
                            if constexpr (KW == 2) if (s.dump_hi) s.dump_hi[pos] = hi;
                            if (s.dump_cnt) s.dump_cnt[pos] = cv;
                        }
                        ++pos;
                    }
                }
            }
        }
    }
#ifdef KB_TIMING
    KB_T(s.trash, 39);                                        // write-back issued
    if (threadIdx.x == 0) atomicAdd((unsigned long long *)&s.trash[8 + 40], 1ull);
#endif
}

// Replay of the buckets kernel C flagged (s.failed): their entries go through the global-atomic path into table t,
// which the host has grown since.  `plan` is the geometry the failed flush ran with (bucket numbering of s.failed).
template <int KW>
__global__ __launch_bounds__(256) void kb_replay_kernel(KbPlan plan, KbScratch s, KdfTable t, KdfCtl *ctl) {
    __shared__ __attribute__((aligned(16))) char rmem[KB_RI_LDS_BYTES];
    const uint64_t bucket = blockIdx.x;
    if (!((s.failed[bucket >> 5] >> (bucket & 31)) & 1)) return;
    const uint64_t pb = bucket >> plan.sub_bits;
    const uint32_t c = (uint32_t)(pb >> plan.c2), f = (uint32_t)(pb & ((1u << plan.c2) - 1));
    KbRunIndex ri; ri.bind(rmem);
    const uint32_t n_runs = ri.setup(plan, s, c);
    const uint32_t hsh_r = 64 - plan.log2cap;
    uint32_t claimed = 0; bool failed = false;
    for (uint32_t r = 0; r < n_runs; ++r) {
        unsigned long long first; uint32_t len, hioff;
        ri.locate<KW>(plan, s, f, r, first, len, hioff);
        for (uint32_t i0 = 0; i0 < len; i0 += 256) {           // wave-uniform trip count (the wide claim protocol needs whole waves)
            const uint32_t i = i0 + threadIdx.x;
            bool todo = i < len;
            const uint64_t klo = todo ? s.ent[first + i] : 0, khi = (KW == 2 && todo) ? s.ent[first + i + hioff] : 0;
            if (plan.sub_bits && ((klo >> hsh_r) >> plan.bucket_bits) != bucket) todo = false;   // sibling bucket's entry
            const uint64_t slot = kdf_home(t, klo);
            bool ok = true;
            if constexpr (KW == 1) { if (todo) ok = kdf_add_narrow<true>(t, klo, 1u, slot, t.lo[slot], claimed); }
            else ok = kdf_add_wide<true>(t, todo, klo, khi, 1u, slot, claimed);
            if (!ok) failed = true;
        }
    }
    if (failed) atomicOr(&ctl->error, 1u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) claimed += __shfl_down(claimed, o);
    if ((threadIdx.x & 63) == 0 && claimed) atomicAdd(&ctl->distinct[(bucket % KDF_SHARDS) * 16], (unsigned long long)claimed);
}


// ---------------------------------------------------------------------------
// Heavy buckets of a skewed flush (narrow keys, insert mode; see kb_count_hits).  kb_bucket_kernel<.., VAR 2> lists the
// buckets whose first 256 runs alone hold more than KB_C_HEAVY entries and leaves them untouched.  Here KB_HV_SLICES
// workgroups share such a bucket's runs (run r to slice r % KB_HV_SLICES), each counting into a PRIVATE empty LDS table,
// and stage their distinct (key, count) pairs; kb_heavy_combine_kernel then folds the staged pairs into the bucket the way
// kernel C would have: slice in LDS, transactional (no room: the bucket is flagged for the replay pass and stays as it was).
// Measured on the repeat-rich 100 Mbp genome (38 heavy buckets, the heaviest 15.9 M entries): kernel C 10.0 -> 6.2 ms, the
// pass 20.1 -> 16.0 ms.
template <int KW>
__global__ __launch_bounds__(256) void kb_heavy_slice_kernel(KbPlan plan, KbScratch s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!s.hv_ctr) return;
    const uint32_t nh = min(s.hv_ctr[0], KB_HV_MAX), h = blockIdx.y, slice = blockIdx.x;
    if (h >= nh) return;
    const uint32_t B = 1u << plan.bucket_bits, bmask = B - 1;
    uint64_t *tlo = (uint64_t *)smem;
    uint64_t *thi = KW == 2 ? tlo + B : nullptr;
    uint32_t *tcnt = (uint32_t *)(smem + (size_t)B * 8 * KW);
    KbRunIndex ri; ri.bind(smem + (size_t)B * (8 * KW + 4));
    __shared__ uint32_t sh_fail, sh_n, sh_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t i = tid; i < B; i += 256) { tlo[i] = KDF_EMPTY; if constexpr (KW == 2) thi[i] = KDF_EMPTY; tcnt[i] = 0; }
    if (tid == 0) { sh_fail = 0; sh_n = 0; }
    const uint64_t bucket = s.hv_bucket[h];
    const uint32_t c = (uint32_t)(bucket >> plan.c2), f = (uint32_t)(bucket & ((1u << plan.c2) - 1));
    const uint32_t n_runs = ri.setup(plan, s, c);               // (barrier inside)
    const uint32_t hsh_r = 64 - plan.log2cap;
    uint32_t claimed = 0; bool failed = false;
    for (uint32_t r = slice; r < n_runs; r += KB_HV_SLICES) {
        unsigned long long first; uint32_t n, hioff;
        ri.locate<KW>(plan, s, f, r, first, n, hioff);
        const uint64_t *ent = s.ent + first;
        constexpr int U = KW == 2 ? 4 : 8;                         // entries per thread in flight: one workgroup has to cover the HBM latency alone
        for (uint32_t i0 = 0; i0 < n; i0 += 256 * U) {             // whole waves: the hit counting is a wave operation
            uint64_t keys[U], his[KW == 2 ? U : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = i0 + u * 256 + tid;
                keys[u] = i < n ? ent[i] : 0;
                if constexpr (KW == 2) his[u] = i < n ? ent[i + hioff] : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u * 256 >= n) break;                      // (uniform)
                bool todo = i0 + u * 256 + tid < n;
                const uint64_t key = keys[u];
                const uint32_t sl = (uint32_t)(key >> hsh_r) & bmask;
                if constexpr (KW == 2) {
                    const uint64_t khi = his[u];
                    if (todo && key == KDF_EMPTY) { failed = true; todo = false; }       // (2^-64: the replay path stores it)
                    // the heavy key sits in its home slot after its first insertion (published: its hi word is there)
                    const bool hit = todo && __hip_atomic_load(&tlo[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == key
                                          && __hip_atomic_load(&thi[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == khi;
                    kb_count_hits<true>(tcnt, sl, hit);
                    kb_probe_wide_wave<KB_MODE_INSERT>(tlo, thi, tcnt, bmask, todo && !hit, key, khi, sl, claimed, failed);
                } else {
                    const bool hit = todo && tlo[sl] == key;       // the heavy key sits in its home slot after its first insertion
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
    uint64_t *oh_ = KW == 2 ? s.hv_khi + (size_t)h * room : nullptr;
    for (uint32_t i = tid; i < B; i += 256)
        if (tlo[i] != KDF_EMPTY) { ok_[pos] = tlo[i]; if constexpr (KW == 2) oh_[pos] = thi[i]; oc_[pos] = tcnt[i]; ++pos; }
}

template <int KW>
__global__ __launch_bounds__(256) void kb_heavy_combine_kernel(KbPlan plan, KbScratch s, KdfTable t, KdfCtl *ctl, int table_nonempty) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!s.hv_ctr) return;
    const uint32_t nh = min(s.hv_ctr[0], KB_HV_MAX), h = blockIdx.x;
    if (h >= nh) return;
    if (h == 0 && threadIdx.x == 0) s.totals[4] = s.hv_ctr[0];     // (statistics: heavy buckets of this flush; the first KB_HV_MAX were split)
    const uint32_t B = 1u << plan.bucket_bits, bmask = B - 1;
    uint64_t *tlo = (uint64_t *)smem;
    uint64_t *thi = KW == 2 ? tlo + B : nullptr;
    uint32_t *tcnt = (uint32_t *)(smem + (size_t)B * 8 * KW);
    __shared__ uint32_t sh_fail, sh_claimed;
    const uint32_t tid = threadIdx.x;
    const uint64_t bucket = s.hv_bucket[h];
    const uint64_t slot0 = bucket << plan.bucket_bits;
    bool failed = false;
    for (uint32_t i = tid; i < B; i += 256) {
        tlo[i] = table_nonempty ? t.lo[slot0 + i] : KDF_EMPTY;
        if constexpr (KW == 2) {
            thi[i] = table_nonempty ? t.hi[slot0 + i] : KDF_EMPTY;
            if ((tlo[i] == KDF_EMPTY) != (thi[i] == KDF_EMPTY)) failed = true;      // (as kernel C: a stored hash word equal to the empty marker)
        }
        tcnt[i] = table_nonempty ? t.cnt[slot0 + i] : 0u;
    }
    if (tid == 0) { sh_fail = s.hv_failed[h]; sh_claimed = 0; }
    __syncthreads();
    const size_t room = (size_t)KB_HV_SLICES << plan.bucket_bits;
    const uint64_t *ik = s.hv_key + (size_t)h * room; const uint32_t *ic = s.hv_cnt + (size_t)h * room;
    const uint64_t *ih = KW == 2 ? s.hv_khi + (size_t)h * room : nullptr;
    const uint32_t n = s.hv_n[h];
    const uint32_t hsh_r = 64 - plan.log2cap;
    uint32_t claimed = 0;
    if (!sh_fail) {
        if constexpr (KW == 2) {
            for (uint32_t i0 = 0; i0 < n; i0 += 256) {          // whole waves: the wide claim protocol retries under a wave-uniform loop
                const uint32_t i = i0 + tid;
                const bool todo = i < n;
                const uint64_t key = todo ? ik[i] : 0, khi = todo ? ih[i] : 0; const uint32_t add = todo ? ic[i] : 0;
                kb_probe_wide_wave<KB_MODE_INSERT>(tlo, thi, tcnt, bmask, todo, key, khi, (uint32_t)(key >> hsh_r) & bmask, claimed, failed, add);
            }
        } else {
            for (uint32_t i = tid; i < n; i += 256) {
                const uint64_t key = ik[i]; const uint32_t add = ic[i];
                uint32_t at = (uint32_t)(key >> hsh_r) & bmask; bool done = false;
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
        }
    }
    if (failed) sh_fail = 1;
    if (claimed) atomicAdd(&sh_claimed, claimed);
    __syncthreads();
    if (sh_fail) {                                                 // as kernel C: untouched in HBM, flagged for the replay pass
        if (tid == 0) { atomicOr(&s.failed[bucket >> 5], 1u << (bucket & 31)); atomicAdd(&s.totals[2], 1ull); }
        return;                                                    // (kernel C already wrote an empty slice into a lazily cleared table)
    }
    for (uint32_t i = tid; i < B; i += 256) {
        t.lo[slot0 + i] = tlo[i]; t.cnt[slot0 + i] = tcnt[i];
        if constexpr (KW == 2) t.hi[slot0 + i] = thi[i];
    }
    if (tid == 0 && sh_claimed) atomicAdd(&ctl->distinct[(bucket % KDF_SHARDS) * 16], (unsigned long long)sh_claimed);
}

// Heavy buckets of a skewed `count --if` flush.  Every window of the parents' reads is partitioned, also the ones whose key
// is not in the filter: a homopolymer k-mer is 1.4 % of ALL windows of a repeat-rich genome, and all of them arrive at one
// bucket -- for one workgroup to look them up one after the other (a 30x human parent: ~10^9 entries).  The bucket's keys do
// not change in this mode, so KB_HV_SLICES workgroups simply share its runs, each with a read-only copy of the key slice and
// private counts in LDS, and add what they counted to the counts in HBM (saturating; a few slots per workgroup).
template <int KW>
__global__ __launch_bounds__(256) void kb_heavy_filtered_kernel(KbPlan plan, KbScratch s, KdfTable t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!s.hv_ctr) return;
    const uint32_t nh = min(s.hv_ctr[0], KB_HV_MAX), h = blockIdx.y, slice = blockIdx.x;
    if (h >= nh) return;
    if (h == 0 && slice == 0 && threadIdx.x == 0) s.totals[4] = s.hv_ctr[0];
    const uint32_t B = 1u << plan.bucket_bits, bmask = B - 1;
    uint64_t *tlo = (uint64_t *)smem;
    uint64_t *thi = KW == 2 ? tlo + B : nullptr;
    uint32_t *tcnt = (uint32_t *)(smem + (size_t)B * 8 * KW);
    KbRunIndex ri; ri.bind(smem + (size_t)B * (8 * KW + 4));
    const uint32_t tid = threadIdx.x;
    const uint64_t bucket = s.hv_bucket[h];
    const uint64_t slot0 = bucket << plan.bucket_bits;
    for (uint32_t i = tid; i < B; i += 256) { tlo[i] = t.lo[slot0 + i]; if constexpr (KW == 2) thi[i] = t.hi[slot0 + i]; tcnt[i] = 0; }
    const uint32_t c = (uint32_t)(bucket >> plan.c2), f = (uint32_t)(bucket & ((1u << plan.c2) - 1));
    const uint32_t n_runs = ri.setup(plan, s, c);               // (barrier inside)
    const uint32_t hsh_r = 64 - plan.log2cap;
    for (uint32_t r = slice; r < n_runs; r += KB_HV_SLICES) {
        unsigned long long first; uint32_t n, hioff;
        ri.locate<KW>(plan, s, f, r, first, n, hioff);
        const uint64_t *ent = s.ent + first;
        constexpr int U = KW == 2 ? 4 : 8;
        for (uint32_t i0 = 0; i0 < n; i0 += 256 * U) {             // whole waves: the hit counting is a wave operation
            uint64_t keys[U], his[KW == 2 ? U : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = i0 + u * 256 + tid;
                keys[u] = i < n ? ent[i] : 0;
                if constexpr (KW == 2) his[u] = i < n ? ent[i + hioff] : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u * 256 >= n) break;                      // (uniform)
                const bool todo = i0 + u * 256 + tid < n;
                const uint64_t key = keys[u];
                uint32_t sl = (uint32_t)(key >> hsh_r) & bmask;
                // read-only linear probe: the key, or the first empty slot (absent).  Almost every entry of a heavy bucket
                // is THE heavy key: found in its home slot, or absent at the first look.
                bool hit = false;
                if (todo) {
                    for (uint32_t p_ = 0; p_ <= bmask; ++p_) {
                        const uint64_t cur = tlo[sl];
                        if (KW == 1 ? cur == KDF_EMPTY : thi[sl] == KDF_EMPTY) break;
                        if (cur == key && (KW == 1 || thi[sl] == his[KW == 2 ? u : 0])) { hit = true; break; }
                        sl = (sl + 1) & bmask;
                    }
                }
                kb_count_hits<true>(tcnt, sl, hit);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < B; i += 256) { const uint32_t a = tcnt[i]; if (a) kdf_sat_add(&t.cnt[slot0 + i], a); }
}

// ---------------------------------------------------------------------------
// L1 of the deferral: small batches are first CONCATENATED, packed as they arrive (2.25 bits per position), in a pending
// stream; the partition kernels then run over ~2^30 positions at a time whatever the caller's batch size is.  A batch
// starts on a tile boundary of the pending stream; the bits of its last mask word past n_bases read "invalid" whatever the
// source holds there, and the padded tail kdf_stream_words() promises (2 mask words of ones, 4 packed words) follows it.
__global__ __launch_bounds__(256) void kb_append_kernel(uint64_t *__restrict__ dp, uint64_t *__restrict__ dm,
                                                        const uint64_t *__restrict__ sp, const uint64_t *__restrict__ sm,
                                                        uint64_t n_bases) {
    const uint64_t n_tiles = (n_bases + 63) >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * n_tiles) dp[i] = i < ((n_bases + 31) >> 5) ? sp[i] : 0ull;
    else if (i < 2 * n_tiles + 4) dp[i] = 0ull;
    if (i < n_tiles) {
        uint64_t m = sm[i];
        if (i == n_tiles - 1 && (n_bases & 63)) m |= ~0ull << (n_bases & 63);
        dm[i] = m;
    } else if (i < n_tiles + 2) dm[i] = ~0ull;
}
