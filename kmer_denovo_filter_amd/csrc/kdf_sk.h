// kdf_sk.h -- the minimizer-partitioned ("super-k-mer") count pipeline, narrow keys (16 <= k <= 32).
//
// The binned pipeline (kdf_binned.h) moves every k-mer INSTANCE (8 B) through two partition passes.  Here the unit
// that travels is a RECORD: a run of consecutive windows of a read that share their minimizer (the smallest canonical
// 12-mer under kdf_sk_order), stored as 2 bits per base in 16 bytes (~9.5 k-mers per record at k = 31).  The table
// is bucketed by the minimizer (KdfTable::sk, kdf_device.h), so every k-mer of a record lands in one bucket and
//
//   S1  sk_extract_kernel   per-window minimizers (sliding minimum), record boundaries as bit masks, records in
//                           canonical orientation, LDS counting sort by coarse bin (top c1 bits of the order value)
//                           and run-wise copy-out into 4 KB chunks taken from a global pool (no histogram pass, no
//                           fixed-capacity cells: a workgroup takes a new chunk for a bin when its current one is full)
//   K1  sk_binscan / sk_chunklist   bin -> its chunks, in groups of SK_GROUP chunks (two tiny kernels, no atomics)
//   S2  sk_finesort_kernel  one workgroup per group: the group's <= 8192 records sorted by the next c2 bits of the
//                           order value (recomputed from the minimizer offset stored in the record) + offset table
//   S3  sk_bucket_kernel    one workgroup per table bucket: slice in LDS, the bucket's records gathered from all
//                           groups of its bin, identical records merged (32-bit CAS per hash slot naming a
//                           representative + multiplicity), every DISTINCT record expanded once and its k-mers
//                           inserted with count += multiplicity; the slice is written back once.
//
// A key whose probe sequence is full (KDF_SK_MAXPROBE slots) goes to the table's overflow array through a spill list
// (sk_spill_insert_kernel); a bucket that cannot queue its spills is left untouched, flagged and replayed with a
// spill list sized for the worst case (MODE_REPLAY).  Nothing is ever dropped silently.
#pragma once
#include "kdf_device.h"
#include "kdf_binned.h"

#define SK_THREADS   1024
#define SK_WPT       16
#define SK_SLAB      (SK_THREADS * SK_WPT)
#define SK_CAP       3584                  // records one S1 round holds in LDS (a slab of random sequence yields ~1700)
#define SK_C1_MAX    9
#define SK_C2_MAX    10
#define SK_CHUNK     256                   // records per pool chunk (4 KB)
#define SK_GROUP     32                    // chunks per S2 group
#define SK_GREC      (SK_CHUNK * SK_GROUP) // 8192 records: 128 KB of LDS
#define SK_NONE      0xFFFFFFFFu
#define SK_WP        (SK_SLAB / 32 + 6)    // packed words one slab's threads read: one before the slab, up to 52 bases past it
#define SK_WM        (SK_SLAB / 64 + 4)
#define SK_MIN_K     16
// bucket kernel
#ifndef SK_C_THREADS
#define SK_C_THREADS 256
#endif
#ifndef SK_C_RC
#define SK_C_RC      512                   // records per dedupe round (a 2048-slot bucket of the bench holds ~470)
#endif
#define SK_C_DT      1024                  // dedupe hash slots
#ifndef SK_C_IC
#define SK_C_IC      3072                  // (record, k-mer) items of the flat expansion list per round
#endif
#ifndef SK_BUCKET_BITS
#define SK_BUCKET_BITS 11                  // slots per bucket of an SK-layout table (2^11: three 256-thread bucket workgroups per CU)
#endif
#ifndef SK_C_LA
#define SK_C_LA      2                     // slots of an item's probe sequence read up front
#endif
#ifndef SK_C_WQ
#define SK_C_WQ      96                    // per wave: keys whose probe goes past the two slots read up front
#endif
#define SK_C_RUNS    256                   // runs (groups of the bin) staged per round: one per thread (SK_C_THREADS >= 256)
#define SK_C_SQ      64                    // LDS spill queue entries
#define SK_C_DPROBE  32                    // dedupe probes before a record is expanded on its own

struct __attribute__((aligned(16))) SkRec { uint64_t lo, hi; };
// record: bases 0..31 in lo, bases 32..51 in hi bits 0..39 (base i in bits 2i: the stream's packing), the offset of
// the minimizer m-mer inside the record in hi bits 48..53, the number of k-mers (windows) in hi bits 56..61
#define SK_HI_BASES  ((1ull << 40) - 1)
#define SK_REC_OFF(hi) ((uint32_t)((hi) >> 48) & 63u)
#define SK_REC_NK(hi)  ((uint32_t)((hi) >> 56) & 63u)

template <int K> struct SkK {
    static constexpr int M = KDF_SK_M;
    static constexpr int W = K - M + 1;                       // m-mers per window
    static constexpr int NM = W + SK_WPT;                     // m-mers per thread (one look-back window)
    static constexpr int MAXNK = (W < 53 - K) ? W : 53 - K;   // windows per record (<= 52 bases)
    static constexpr int G = MAXNK / 2 > 0 ? MAXNK / 2 : 1;   // forced-cut grid (a record spans <= 2 G windows)
};

struct SkPlan {
    uint32_t c1, c2, sub_bits;            // coarse bins, fine bins, table buckets per partition bucket
    uint32_t log2cap, bucket_bits;
    uint32_t key_parts, key_part;
    uint32_t k;
    uint32_t goff_stride;                 // 2^c2 + 1
    uint32_t dbg;
    const uint16_t *assign;               // balanced bucket-in-bin table indexed by the spread order value, or null
};

enum { SKC_POOL = 0, SKC_EXHAUSTED = 1, SKC_GROUPS = 2, SKC_SPILL = 3, SKC_FAILED = 4, SKC_BADNK = 5, SKC_SPILL_LOST = 6, SKC_N = 16 };

struct SkScratch {
    SkRec *chunks;              // [max_chunks][SK_CHUNK]
    uint32_t *chunk_bin;        // [max_chunks]
    uint32_t *chunk_pos;        // [max_chunks] index of the chunk among the chunks of its bin
    uint32_t *chunk_fill;       // [max_chunks]
    uint32_t *bin_nchunks;      // [2^c1]
    uint32_t *bin_chunk_start;  // [2^c1 + 1]
    uint32_t *group_first;      // [2^c1 + 1]
    uint32_t *chunk_list;       // [max_chunks] chunk ids grouped by bin
    SkRec *sorted;              // [max_groups][SK_GREC]
    uint32_t *goff;             // [max_groups][2^c2 + 1]
    uint32_t *failed;           // bitmap over table buckets
    uint64_t *sp_key;           // spill list
    uint32_t *sp_cnt;
    uint32_t *ctrs;             // [SKC_N]
    uint32_t max_chunks, max_groups, sp_cap, pad;
};

// ---------------------------------------------------------------------------------------------------------------
// S1
// reverse the sixteen 2-bit groups of a 32-bit word
__device__ __forceinline__ uint32_t sk_rev2_32(uint32_t x) {
    x = __builtin_bitreverse32(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
__device__ __forceinline__ uint64_t sk_shr128(uint64_t a, uint64_t b, int sh) {      // (b:a) >> sh, low 64 bits, sh in 0..127
    return sh >= 64 ? (b >> (sh - 64)) : kdf_funnel(a, b, sh);
}

// Per thread: the 16 windows starting at stream position P (a multiple of 16) plus the window before them.
// Out: mv[i] = (order value << 8 | m-mer index in the span) of the minimizer of window P - 1 + i, i = 0..16, and
// bit i of v17 = window P - 1 + i is valid.  The span starts at S = P - 1; for P = 0 a virtual invalid base stands
// at position -1.
// `packed` / `invalid` are (LDS) copies of the stream words from word index pw0 / mw0 on.
template <int K>
__device__ __forceinline__ void sk_windows(const uint64_t *packed, const uint64_t *invalid, uint64_t pw0, uint64_t mw0,
                                           uint64_t P, uint32_t (&mv)[SK_WPT + 1], uint32_t &v17) {
    using C = SkK<K>;
    constexpr int NM = C::NM, W = C::W, M = C::M;
    uint64_t e0, e1, inv;
    if (P > 0) {
        const uint64_t S = P - 1;
        const uint64_t w0 = (S >> 5) - pw0; const int sh = (int)(S & 31) * 2;
        const uint64_t x0 = packed[w0], x1 = packed[w0 + 1], x2 = packed[w0 + 2];
        e0 = kdf_funnel(x0, x1, sh); e1 = kdf_funnel(x1, x2, sh);
        const uint64_t mw = (S >> 6) - mw0; const int msh = (int)(S & 63);
        inv = kdf_funnel(invalid[mw], invalid[mw + 1], msh);
    } else {
        const uint64_t x0 = packed[0], x1 = packed[1];
        e0 = x0 << 2; e1 = (x1 << 2) | (x0 >> 62);
        inv = (invalid[0] << 1) | 1ull;
    }
    // bit j of a: span positions j .. j + K - 1 are all valid bases (K + 16 <= 48 positions matter)
    uint64_t a = ~inv;
    {
        int r = 1;
#pragma unroll
        while (r < K) { const int s = (K - r) < r ? (K - r) : r; a &= a >> s; r += s; }
    }
    v17 = (uint32_t)a & 0x1FFFFu;
    // 32-bit words of the span, and of its 2-bit-group reversal: every m-mer is one v_alignbit + one v_and
    const uint32_t w[4] = {(uint32_t)e0, (uint32_t)(e0 >> 32), (uint32_t)e1, (uint32_t)(e1 >> 32)};
    const uint32_t f[4] = {sk_rev2_32(w[3]), sk_rev2_32(w[2]), sk_rev2_32(w[1]), sk_rev2_32(w[0])};
    constexpr uint32_t MM = (1u << (2 * M)) - 1;
    uint32_t p[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int ro = 2 * j, fo = 128 - 2 * M - 2 * j;
        // reverse complement of the m-mer (MSB-first code) = ~(its bits as they stand); forward code from the reversed span
        const uint32_t rc = ~__builtin_amdgcn_alignbit(w[(ro >> 5) + 1 > 3 ? 3 : (ro >> 5) + 1], w[ro >> 5], ro & 31) & MM;
        const uint32_t fw = __builtin_amdgcn_alignbit((fo >> 5) + 1 > 3 ? 0u : f[(fo >> 5) + 1 > 3 ? 3 : (fo >> 5) + 1], f[fo >> 5], fo & 31) & MM;
        p[j] = (kdf_sk_order(fw < rc ? fw : rc) << 8) | (uint32_t)j;
    }
    // sliding minimum over W consecutive m-mers (van Herk: block suffix / prefix minima)
    uint32_t sfx[NM], pfx[NM];
#pragma unroll
    for (int j = NM - 1; j >= 0; --j) sfx[j] = (j == NM - 1 || (j + 1) % W == 0) ? p[j] : min(p[j], sfx[j + 1]);
#pragma unroll
    for (int j = 0; j < NM; ++j) pfx[j] = (j % W == 0) ? p[j] : min(p[j], pfx[j - 1]);
#pragma unroll
    for (int i = 0; i <= SK_WPT; ++i) mv[i] = min(sfx[i], pfx[i + W - 1]);
}

// pattern with a bit at every multiple of G
template <int G> __device__ __forceinline__ constexpr uint32_t sk_grid_pattern() {
    uint32_t m = 0;
    for (int i = 0; i < 32; i += G) m |= 1u << i;
    return m;
}

template <int K>
__global__ __launch_bounds__(SK_THREADS) void sk_extract_kernel(
    const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid, uint64_t n_bases,
    SkPlan plan, SkScratch s, uint32_t slabs_per_wg)
{
    using C = SkK<K>;
    constexpr int NBMAX = 1 << SK_C1_MAX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *lds_mv = (uint32_t *)smem;                                       // [SK_WPT][SK_THREADS]: 64 KB ...
    SkRec *img = (SkRec *)smem;                                                // ... reused as the sorted image [SK_CAP]
    SkRec *U = (SkRec *)(smem + (size_t)SK_WPT * SK_THREADS * 4);              // [SK_CAP] records in emission order
    uint32_t *T = (uint32_t *)(U + SK_CAP);                                    // [SK_CAP] bin << 16 | rank
    uint32_t *hist = T + SK_CAP;                                               // [NBMAX + 1]
    uint32_t *offs = hist + NBMAX + 1;                                         // [NBMAX + 1]
    uint32_t *cur_chunk = offs + NBMAX + 1;                                    // [NBMAX]
    uint32_t *cur_fill = cur_chunk + NBMAX;                                    // [NBMAX]
    uint32_t *wsum = cur_fill + NBMAX;                                         // [40]
    uint32_t *nxt = wsum + 40;                                                 // [NBMAX] first of the chunks taken for the bin this round
    uint64_t *wp = (uint64_t *)(nxt + NBMAX);                                  // [SK_WP] the slab's packed words (+ look-back / look-ahead)
    uint64_t *wm = wp + SK_WP;                                                 // [SK_WM] the slab's mask words
    uint16_t *nat = (uint16_t *)(wm + SK_WM);                                  // [SK_THREADS] natural breaks
    uint16_t *brk = nat + SK_THREADS;                                          // [SK_THREADS + 4] final breaks
    const int nb = 1 << plan.c1;
    const int tid = threadIdx.x;
    for (int i = tid; i < nb; i += SK_THREADS) { hist[i] = 0; cur_chunk[i] = SK_NONE; cur_fill[i] = 0; }
    if (tid < 4) brk[SK_THREADS + tid] = 0xFFFFu;                              // the slab's end is a break
    __syncthreads();
    const uint64_t slab0 = (uint64_t)blockIdx.x * slabs_per_wg;
    const int grp16 = tid >> 4, lane16 = tid & 15;
    constexpr int NGRP = SK_THREADS / 16;
    const bool stampit = (plan.dbg & 32) && tid == 0;
    uint64_t st_t = stampit ? __builtin_amdgcn_s_memtime() : 0;
    uint32_t st_acc[6] = {0, 0, 0, 0, 0, 0};
#define SK1_STAMP(I) if (stampit) { const uint64_t n_ = __builtin_amdgcn_s_memtime(); st_acc[I] += (uint32_t)((n_ - st_t) >> 4); st_t = n_; }
    for (uint32_t sl = 0; sl < slabs_per_wg; ++sl) {
        const uint64_t P0 = (slab0 + sl) * (uint64_t)SK_SLAB;
        if (P0 >= n_bases) break;                                              // uniform
        const uint64_t P = P0 + (uint64_t)tid * SK_WPT;
        // the slab's stream words -> LDS (every later read of the stream, the record assembly's included, is an LDS read)
        const uint64_t pw0 = P0 ? (P0 - 1) >> 5 : 0, mw0 = P0 ? (P0 - 1) >> 6 : 0;
        {
            const uint64_t pw_end = ((n_bases + KDF_TILE - 1) / KDF_TILE) * 2 + 4, mw_end = (n_bases + KDF_TILE - 1) / KDF_TILE + 2;   // kdf_stream_words
            if (tid < SK_WP) wp[tid] = pw0 + tid < pw_end ? packed[pw0 + tid] : 0ull;
            else if (tid - SK_WP < SK_WM) wm[tid - SK_WP] = mw0 + (tid - SK_WP) < mw_end ? invalid[mw0 + tid - SK_WP] : ~0ull;
        }
        __syncthreads();
        uint32_t round = 0, rounds = 1;
        do {
            // ---- windows, minimizers, record boundaries (recomputed per round: rounds > 1 only for pathological slabs)
            uint32_t start16 = 0, brk16 = 0xFFFFu;
            {
                uint32_t mv[SK_WPT + 1], v17 = 0;
                if (P < n_bases) sk_windows<K>(wp, wm, pw0, mw0, P, mv, v17);
                else {
#pragma unroll
                    for (int i = 0; i <= SK_WPT; ++i) mv[i] = 0;
                }
                uint32_t neq = 0;
#pragma unroll
                for (int i = 0; i < SK_WPT; ++i) {
                    neq |= (((mv[i + 1] ^ mv[i]) >> 8) ? 1u : 0u) << i;
                    lds_mv[i * SK_THREADS + tid] = mv[i + 1];
                }
                const uint32_t v16 = (v17 >> 1) & 0xFFFFu;
                uint32_t pv16 = v17 & 0xFFFFu;
                if (tid == 0) pv16 &= ~1u;                                      // a slab starts a record
                const uint32_t nat_start = v16 & (~pv16 | neq);
                const uint32_t natbrk = (nat_start | ~v16) & 0xFFFFu;
                nat[tid] = (uint16_t)natbrk;
                kb_lds_barrier();
                // forced cuts: at grid positions that have no natural break in the G windows before them, so that a
                // run of one minimizer VALUE (tandem repeats) is cut into records of at most 2 G <= MAXNK windows
                const uint32_t prev = tid ? (uint32_t)nat[tid - 1] : 0xFFFFu;
                const uint32_t b32 = prev | (natbrk << 16);
                const uint32_t pm = (uint32_t)(P % (uint64_t)C::G);
                const uint32_t grid = (sk_grid_pattern<C::G>() << ((C::G - pm) % C::G)) & 0xFFFFu;
                uint32_t forced = 0;
#pragma unroll
                for (int i = 0; i < SK_WPT; ++i)
                    forced |= (((b32 >> (16 + i - C::G)) & ((1u << C::G) - 1)) == 0 ? 1u : 0u) << i;
                forced &= grid & v16 & ~nat_start;
                start16 = nat_start | forced;
                brk16 = (start16 | ~v16) & 0xFFFFu;
                brk[tid] = (uint16_t)brk16;
            }
            SK1_STAMP(0)
            const uint32_t nrec = __popc(start16);
            uint32_t total = 0;
            const uint32_t tbase = kb_block_exscan(nrec, wsum, &total);       // (its barriers publish brk[])
            SK1_STAMP(1)
            rounds = (total + SK_CAP - 1) / SK_CAP;
            if (rounds == 0) break;
            const uint32_t r_lo = round * SK_CAP;
            // ---- emission.  First the starts of this round as a dense list (T[r] = thread << 4 | window), then one
            // record per LIST ENTRY: the record assembly runs with every lane busy instead of inside a per-thread loop
            // over start bits (a thread has 1.7 starts on average, a wave's busiest lane 5-6).
            {
                uint32_t sb = start16, ord = tbase - r_lo;
                while (sb) {
                    const int i = __ffs(sb) - 1; sb &= sb - 1;
                    if (ord < SK_CAP) T[ord] = ((uint32_t)tid << 4) | (uint32_t)i;   // (unsigned wrap: earlier rounds' records too)
                    ++ord;
                }
            }
            kb_lds_barrier();
            {
                const uint32_t nr = min((uint32_t)SK_CAP, total - r_lo);
                for (uint32_t r = tid; r < nr; r += SK_THREADS) {
                    const uint32_t ti = T[r], t_ = ti >> 4; const int i = (int)(ti & 15u);
                    const uint64_t look = (uint64_t)brk[t_] | ((uint64_t)brk[t_ + 1] << 16) | ((uint64_t)brk[t_ + 2] << 32);
                    const uint64_t after = look >> (i + 1);
                    const int nk = after ? __ffsll((unsigned long long)after) : 48;
                    if (nk > C::MAXNK) {                                       // cannot happen (forced cuts); never silent: the host fails the pass
                        s.ctrs[SKC_BADNK] = 1; U[r] = SkRec{0, 0}; T[r] = atomicAdd(&hist[0], 1u); continue;
                    }
                    const uint32_t mvv = lds_mv[i * SK_THREADS + t_];
                    const uint32_t g = mvv >> 8;
                    const int off_fw = (int)(mvv & 0xFF) - 1 - i;              // minimizer m-mer offset inside the record
                    const uint64_t Q = P0 + (uint64_t)t_ * SK_WPT + i;
                    const int nbases = nk + K - 1;                             // <= 52
                    const uint64_t w0 = (Q >> 5) - pw0; const int sh = (int)(Q & 31) * 2;
                    const uint64_t x0 = wp[w0], x1 = wp[w0 + 1], x2 = wp[w0 + 2];
                    uint64_t lo = kdf_funnel(x0, x1, sh), hi = kdf_funnel(x1, x2, sh);
                    const int hb = 2 * nbases - 64;                            // bits used in hi
                    const uint64_t hmask = hb > 0 ? ((1ull << hb) - 1) : 0ull;
                    const uint64_t lmask = hb >= 0 ? ~0ull : ((1ull << (2 * nbases)) - 1);
                    hi &= hmask; lo &= lmask;
                    // reverse complement of the string: 2-bit-group reversal of ~(hi:lo), shifted down
                    const uint64_t r1 = kdf_rev2(~lo), r0 = kdf_rev2(~hi);
                    const int sft = 128 - 2 * nbases;
                    uint64_t clo = sk_shr128(r0, r1, sft), chi = sft >= 64 ? 0 : (r1 >> sft);
                    chi &= hmask; clo &= lmask;
                    int off = off_fw;
                    if (chi < hi || (chi == hi && clo < lo)) { lo = clo; hi = chi; off = nbases - KDF_SK_M - off_fw; }
                    hi |= ((uint64_t)(uint32_t)off << 48) | ((uint64_t)(uint32_t)nk << 56);
                    const uint32_t bin = plan.c1 ? (kdf_sk_spread(g) >> (24 - plan.c1)) : 0u;
                    const uint32_t rank = atomicAdd(&hist[bin], 1u);
                    U[r] = SkRec{lo, hi};
                    T[r] = (bin << 16) | rank;
                }
            }
            kb_lds_barrier();
            SK1_STAMP(2)
            if (tid < 64) {                                                    // exclusive scan of hist[0..nb) by one wave
                const int per = (nb + 63) >> 6;
                const int b0 = tid * per;
                uint32_t sum = 0;
                for (int i = 0; i < per; ++i) sum += (b0 + i < nb) ? hist[b0 + i] : 0;
                uint32_t inc = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { uint32_t t_ = __shfl_up(inc, o); if (tid >= o) inc += t_; }
                uint32_t run = inc - sum;
                for (int i = 0; i < per; ++i) if (b0 + i < nb) { offs[b0 + i] = run; run += hist[b0 + i]; }
            } else if (tid - 64 < nb) {
                // meanwhile a thread per bin takes the chunks its run needs beyond the bin's current chunk: all bins at
                // once, two independent atomics each, so their latency is paid once per round and not inside the copy-out
                const int bin = tid - 64;
                const uint32_t n = hist[bin], ch = cur_chunk[bin];
                const uint32_t room = ch == SK_NONE ? 0u : (uint32_t)SK_CHUNK - cur_fill[bin];
                uint32_t id0 = SK_NONE;
                if (n > room) {
                    const uint32_t need = (n - room + SK_CHUNK - 1) / SK_CHUNK;
                    id0 = atomicAdd(&s.ctrs[SKC_POOL], need);
                    const uint32_t pos0 = atomicAdd(&s.bin_nchunks[bin], need);
                    if (id0 + need > s.max_chunks || id0 + need < id0) { s.ctrs[SKC_EXHAUSTED] = 1; id0 = SK_NONE - 1; }
                    else for (uint32_t q = 0; q < need; ++q) { s.chunk_bin[id0 + q] = (uint32_t)bin; s.chunk_pos[id0 + q] = pos0 + q; }
                }
                nxt[bin] = id0;
            }
            kb_lds_barrier();
            SK1_STAMP(3)
            {
                const uint32_t nr = min((uint32_t)SK_CAP, total - r_lo);
                for (uint32_t r = tid; r < nr; r += SK_THREADS) {
                    const uint32_t tg = T[r];
                    img[offs[tg >> 16] + (tg & 0xFFFFu)] = U[r];
                }
            }
            kb_lds_barrier();
            SK1_STAMP(4)
            // ---- copy-out: 16 lanes per bin; the bin's run goes to this workgroup's current chunk of the bin and on
            // into the chunks taken above (consecutive ids)
            for (int bin = grp16; bin < nb; bin += NGRP) {
                const uint32_t n = hist[bin], o = offs[bin];
                if (n == 0) continue;
                uint32_t ch = cur_chunk[bin], fl = cur_fill[bin], nx = nxt[bin], done = 0;
                while (done < n) {
                    if (ch == SK_NONE || fl == SK_CHUNK) {
                        if (lane16 == 0 && ch != SK_NONE && ch < s.max_chunks) s.chunk_fill[ch] = SK_CHUNK;
                        ch = nx; fl = 0;
                        if (nx < SK_NONE - 1) ++nx;                            // (SK_NONE - 1: the pool ran out, nothing is written)
                    }
                    const uint32_t take = min(n - done, (uint32_t)SK_CHUNK - fl);
                    if (ch < s.max_chunks) {
                        SkRec *dst = s.chunks + (size_t)ch * SK_CHUNK + fl;
                        for (uint32_t i = lane16; i < take; i += 16) dst[i] = img[o + done + i];
                    }
                    done += take; fl += take;
                }
                if (lane16 == 0) { cur_chunk[bin] = ch; cur_fill[bin] = fl; hist[bin] = 0; }
            }
            kb_lds_barrier();
            SK1_STAMP(5)
            ++round;
        } while (round < rounds);
    }
    if (stampit) for (int i = 0; i < 6; ++i) atomicAdd(&s.ctrs[8 + i], st_acc[i] >> 6);
    for (int i = tid; i < nb; i += SK_THREADS) {
        const uint32_t ch = cur_chunk[i];
        if (ch != SK_NONE && ch < s.max_chunks) s.chunk_fill[ch] = cur_fill[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K1: chunks of each bin, in groups
__global__ __launch_bounds__(1024) void sk_binscan_kernel(SkPlan plan, SkScratch s) {
    __shared__ uint32_t a[(1 << SK_C1_MAX) + 1], g[(1 << SK_C1_MAX) + 1];
    const int nb = 1 << plan.c1;
    if (threadIdx.x == 0) {
        uint32_t acc = 0, gacc = 0;
        for (int i = 0; i < nb; ++i) {
            a[i] = acc; g[i] = gacc;
            const uint32_t n = s.ctrs[SKC_EXHAUSTED] ? 0u : s.bin_nchunks[i];
            acc += n; gacc += (n + SK_GROUP - 1) / SK_GROUP;
        }
        a[nb] = acc; g[nb] = gacc;
        s.ctrs[SKC_GROUPS] = gacc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= nb; i += blockDim.x) { s.bin_chunk_start[i] = a[i]; s.group_first[i] = g[i]; }
}
__global__ __launch_bounds__(256) void sk_chunklist_kernel(SkScratch s) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (s.ctrs[SKC_EXHAUSTED]) return;
    const uint32_t n = min(s.ctrs[SKC_POOL], s.max_chunks);
    if (c >= n) return;
    s.chunk_list[s.bin_chunk_start[s.chunk_bin[c]] + s.chunk_pos[c]] = c;
}

// order value of a record's minimizer, from the stored offset
__device__ __forceinline__ uint32_t sk_rec_order(uint64_t lo, uint64_t hi) {
    constexpr uint32_t MM = (1u << (2 * KDF_SK_M)) - 1;
    // (offsets up to 40: a record that was stored reverse-complemented keeps its minimizer near its END when the run of one
    // minimizer VALUE outlasted the first instance -- tandem repeats, homopolymers -- so the shift can reach 80 bits)
    const uint32_t e = (uint32_t)sk_shr128(lo, hi & SK_HI_BASES, 2 * (int)SK_REC_OFF(hi)) & MM;   // m-mer, base i in bits 2i
    const uint32_t rc = ~e & MM;
    const uint32_t fw = sk_rev2_32(e) >> (32 - 2 * KDF_SK_M);
    return kdf_sk_order(fw < rc ? fw : rc);
}

// ---------------------------------------------------------------------------------------------------------------
// S2: one workgroup per group of SK_GROUP chunks of one bin
__global__ __launch_bounds__(SK_THREADS) void sk_finesort_kernel(SkPlan plan, SkScratch s)
{
    constexpr int EPT = SK_GREC / SK_THREADS;                                  // 8
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkRec *img = (SkRec *)smem;                                                // [SK_GREC]
    uint32_t *hist = (uint32_t *)(img + SK_GREC);                              // [2^c2]
    uint32_t *offs = hist + (1 << SK_C2_MAX);                                  // [2^c2]
    uint32_t *wsum = offs + (1 << SK_C2_MAX);                                  // [40]
    uint32_t *cid = wsum + 40;                                                 // [SK_GROUP] chunk ids
    uint32_t *cfl = cid + SK_GROUP;                                            // [SK_GROUP] fills
    const uint32_t grp = blockIdx.x;
    if (s.ctrs[SKC_EXHAUSTED] || grp >= s.ctrs[SKC_GROUPS]) return;
    const int nf = 1 << plan.c2, tid = threadIdx.x;
    for (int i = tid; i < nf; i += SK_THREADS) hist[i] = 0;
    if (tid < SK_GROUP) {
        const int nbn = 1 << plan.c1;
        int lo_ = 0, hi_ = nbn;                                                // largest bin with group_first[bin] <= grp
        while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if (s.group_first[mid] <= grp) lo_ = mid; else hi_ = mid; }
        const uint32_t lc = s.bin_chunk_start[lo_] + (grp - s.group_first[lo_]) * SK_GROUP + tid;
        const bool ok = lc < s.bin_chunk_start[lo_ + 1];
        const uint32_t id = ok ? s.chunk_list[lc] : 0u;
        cid[tid] = id; cfl[tid] = ok ? s.chunk_fill[id] : 0u;
    }
    __syncthreads();
    uint64_t rl[EPT], rh[EPT]; uint32_t br[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = e * SK_THREADS + tid, c = i >> 8, o = i & (SK_CHUNK - 1);
        br[e] = SK_NONE;
        if (o < cfl[c]) { const SkRec v = s.chunks[(size_t)cid[c] * SK_CHUNK + o]; rl[e] = v.lo; rh[e] = v.hi; br[e] = 0; }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        if (br[e] != SK_NONE) {
            const uint32_t g = kdf_sk_spread(sk_rec_order(rl[e], rh[e]));
            const uint32_t f = plan.assign ? (uint32_t)plan.assign[g] : plan.c2 ? ((g >> (24 - plan.c1 - plan.c2)) & ((1u << plan.c2) - 1)) : 0u;
            br[e] = (f << 16) | atomicAdd(&hist[f], 1u);
        }
    }
    __syncthreads();
    {
        const uint32_t v = tid < nf ? hist[tid] : 0;
        uint32_t len = 0;
        const uint32_t ex = kb_block_exscan(v, wsum, &len);
        uint32_t *go = s.goff + (size_t)grp * plan.goff_stride;
        if (tid < nf) { offs[tid] = ex; go[tid] = ex; }
        if (tid == 0) go[nf] = len;
        wsum[39] = len;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (br[e] != SK_NONE) img[offs[br[e] >> 16] + (br[e] & 0xFFFFu)] = SkRec{rl[e], rh[e]};
    __syncthreads();
    const uint32_t len = wsum[39];
    SkRec *dst = s.sorted + (size_t)grp * SK_GREC;
    for (uint32_t i = tid; i < len; i += SK_THREADS) dst[i] = img[i];
}

// ---------------------------------------------------------------------------------------------------------------
// S3: one workgroup per table bucket
enum { SK_MODE_COUNT = 0, SK_MODE_REPLAY = 1 };

__device__ __forceinline__ uint32_t sk_rec_hash(uint64_t lo, uint64_t hi) {
    uint32_t h = (uint32_t)lo * 0x9E3779B1u;
    h ^= (uint32_t)(lo >> 32) * 0x85EBCA77u;
    h ^= (uint32_t)hi * 0xC2B2AE3Du;
    h ^= (uint32_t)(hi >> 32) * 0x27D4EB2Fu;
    return h ^ (h >> 15);
}

// LDS bytes of the bucket kernel for 2^bucket_bits slots
__host__ __device__ constexpr size_t sk_bucket_lds(uint32_t bucket_bits) {
    static_assert(SK_C_THREADS >= SK_C_RUNS && SK_C_THREADS % 64 == 0 && SK_C_RC <= 1024 && SK_C_IC % 4 == 0 && SK_C_WQ % 4 == 0, "bucket kernel geometry");
    return ((size_t)12 << bucket_bits) + (size_t)SK_C_RC * 16 + (size_t)SK_C_SQ * 12 + 16 + (size_t)SK_C_DT * 4
           + (size_t)(2 * SK_C_RUNS + 2 + 40 + 8) * 4 + (size_t)SK_C_IC * 2 + (size_t)(SK_C_THREADS / 64) * SK_C_WQ * 14 + (size_t)SK_C_RC * 2 + 16;
}

// One key into the LDS slice from slot `sl` on, `n` slots of its probe sequence already seen.  FOUR slots per
// iteration (their reads in flight together): the loop's exec-mask bookkeeping is scalar work, the scalar unit issues
// one instruction per cycle for the whole CU, and a wave runs as many iterations as its longest probe -- the one-slot
// loop made the bucket kernels scalar-issue bound.  false: the key's neighbourhood of `lim` slots holds neither the
// key nor room.  (A slot read as EMPTY may have been taken since: the CAS tells; a taken slot stays taken.)
__device__ __forceinline__ bool sk_slice_add(uint64_t *tlo, uint32_t *tcnt, uint32_t bmask, uint32_t lim, uint64_t key, uint32_t mult,
                                             uint32_t sl, uint64_t /*cur*/, uint32_t &claimed, uint32_t n = 0) {
    while (n < lim) {
        uint64_t c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = tlo[(sl + i) & bmask];
        // first of the four that holds the key or is empty (4 = none)
        uint32_t f = 4; bool isk = false;
#pragma unroll
        for (int i = 3; i >= 0; --i) { const bool k_ = c[i] == key, e_ = c[i] == KDF_EMPTY; if (k_ || e_) { f = (uint32_t)i; isk = k_; } }
        if (f < 4 && n + f >= lim) return false;
        if (f == 4) { sl = (sl + 4) & bmask; n += 4; continue; }
        const uint32_t at = (sl + f) & bmask;
        if (!isk) {
            const uint64_t old = atomicCAS((unsigned long long *)&tlo[at], KDF_EMPTY, key);
            if (old == KDF_EMPTY) { ++claimed; isk = true; }
            else if (old == key) isk = true;
        }
        if (isk) { atomicAdd(&tcnt[at], mult); return true; }
        sl = (at + 1) & bmask; n += f + 1;                      // another key took the slot: go on behind it
    }
    return false;
}

template <int MODE>
__global__ __launch_bounds__(SK_C_THREADS) void sk_bucket_kernel(SkPlan plan, SkScratch s, KdfTable t, KdfCtl *ctl, int table_nonempty)
{
    constexpr uint32_t CT = SK_C_THREADS, EMPTY32 = 0xFFFFFFFFu;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t B = 1u << plan.bucket_bits, bmask = B - 1;
    uint64_t *tlo = (uint64_t *)smem;                                          // [B]
    uint64_t *rlo = tlo + B, *rhi = rlo + SK_C_RC;                             // [RC] records of this round
    uint64_t *sqk = rhi + SK_C_RC;                                             // [SQ] spill queue keys
    unsigned long long *w64 = (unsigned long long *)(sqk + SK_C_SQ);           // [2] windows counted by this bucket
    uint32_t *tcnt = (uint32_t *)(w64 + 2);                                    // [B]
    uint32_t *own = tcnt + B;                                                  // [DT] multiplicity << 16 | representative
    uint32_t *sqc = own + SK_C_DT;                                             // [SQ]
    uint32_t *run_pref = sqc + SK_C_SQ;                                        // [RUNS + 2]
    uint32_t *run_first = run_pref + SK_C_RUNS + 2;                            // [RUNS]
    uint32_t *wsum = run_first + SK_C_RUNS;                                    // [40]
    uint32_t *sh = wsum + 40;                                                  // [8] n_items, n_sq, failed, claimed, sp_base
    uint16_t *items = (uint16_t *)(sh + 8);                                    // [IC] representative record << 5 | k-mer index (0xFFFF: none)
    uint16_t *qs = items + SK_C_IC;                                            // [waves][WQ] slot a queued key goes on from
    uint32_t *qm = (uint32_t *)(qs + (CT / 64) * SK_C_WQ);                     // [waves][WQ] its multiplicity
    uint64_t *qk = (uint64_t *)(qm + (CT / 64) * SK_C_WQ);                     // [waves][WQ] the key (IC, WQ even: 8-byte aligned)
    uint16_t *mr = (uint16_t *)(qk + (CT / 64) * SK_C_WQ);                     // [RC] multiplicity of the record (representatives only)

    const uint32_t nbk = gridDim.x, tid = threadIdx.x;
    const bool stampit = (plan.dbg & 16) && tid == 0 && (blockIdx.x & 63) == 0;
    uint64_t st_t = stampit ? __builtin_amdgcn_s_memtime() : 0;
    uint32_t st_acc[6] = {0, 0, 0, 0, 0, 0};
#define SK_STAMP(I) if (stampit) { const uint64_t n_ = __builtin_amdgcn_s_memtime(); st_acc[I] += (uint32_t)(n_ - st_t); st_t = n_; }
    // an XCD takes a contiguous eighth of the buckets (neighbouring buckets share the lines of the offset tables)
    const uint64_t bucket = (nbk & 7) ? blockIdx.x : (uint64_t)(blockIdx.x & 7) * (nbk >> 3) + (blockIdx.x >> 3);
    if (s.ctrs[SKC_EXHAUSTED]) return;                                         // S1 ran out of chunks: nothing may be inserted
    if constexpr (MODE == SK_MODE_REPLAY) {
        if (!((s.failed[bucket >> 5] >> (bucket & 31)) & 1)) return;
    }
    const uint64_t pb = bucket >> plan.sub_bits;
    const uint32_t bin = (uint32_t)(pb >> plan.c2), f = (uint32_t)(pb & ((1u << plan.c2) - 1));
    const uint64_t slot0 = bucket << plan.bucket_bits;
    const bool load = table_nonempty || MODE == SK_MODE_REPLAY;
    if (tid < 8) sh[tid] = 0;
    if (tid < 2) w64[tid] = 0;
    if (load) {
        for (uint32_t i = tid; i < B; i += CT) { tlo[i] = t.lo[slot0 + i]; tcnt[i] = t.cnt[slot0 + i]; }
    } else {
        const ulonglong2 e2 = {KDF_EMPTY, KDF_EMPTY};
        for (uint32_t i = tid; i < B / 2; i += CT) { ((ulonglong2 *)tlo)[i] = e2; ((uint2 *)tcnt)[i] = uint2{0u, 0u}; }
    }
    __syncthreads();
    SK_STAMP(0)

    const int k = (int)plan.k;
    const uint64_t kmask = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const bool sliced = plan.key_parts > 1;
    const uint32_t nb_bits = plan.log2cap - plan.bucket_bits;
    const uint32_t lim = B < KDF_SK_MAXPROBE ? B : KDF_SK_MAXPROBE;
    uint32_t claimed = 0; unsigned long long nwin = 0; bool failed = false;
    // one key that found no room: queue it for the overflow table
    auto spill = [&](uint64_t key, uint32_t mult) {
        if constexpr (MODE == SK_MODE_REPLAY) {
            const uint32_t p = atomicAdd(&s.ctrs[SKC_SPILL], 1u);
            if (p < s.sp_cap) { s.sp_key[p] = key; s.sp_cnt[p] = mult; }
            else s.ctrs[SKC_SPILL_LOST] = 1;                                   // host sized the list for the worst case: never taken
        } else {
            const uint32_t q = atomicAdd(&sh[1], 1u);
            if (q < SK_C_SQ) { sqk[q] = key; sqc[q] = mult; }
            else failed = true;
        }
    };
    // k-mer j of the record an item names -> (key, multiplicity, first slot); false: not this slice's / not this table bucket's
    auto item_key = [&](uint32_t it, uint64_t &key, uint32_t &mult, uint32_t &sl) -> bool {
        const uint32_t j = it & 31u, r = it >> 5;
        mult = mr[r];
        const uint64_t lo = rlo[r], hb = rhi[r] & SK_HI_BASES;
        key = kdf_canon_narrow(kdf_funnel(lo, hb, 2 * (int)j), k, kmask);
        sl = kdf_sk_slot(key, plan.bucket_bits);
        return !(sliced && kdf_slice(kdf_mix64(key), plan.key_parts) != plan.key_part);
    };

    const uint32_t g0 = s.group_first[bin], g1 = s.group_first[bin + 1];
    for (uint32_t gb = g0; gb < g1; gb += SK_C_RUNS) {
        const uint32_t nruns = min((uint32_t)SK_C_RUNS, g1 - gb);
        // run bounds of this bucket in nruns groups, fetched by all threads at once
        {
            uint32_t len = 0, first = 0;
            if (tid < nruns) {
                const uint32_t *go = s.goff + (size_t)(gb + tid) * plan.goff_stride;
                const uint32_t a = go[f], b = go[f + 1];
                len = b - a; first = (gb + tid) * (uint32_t)SK_GREC + a;
            }
            uint32_t tot = 0;
            const uint32_t ex = kb_block_exscan(len, wsum, &tot);
            if (tid < SK_C_RUNS) { run_pref[tid] = ex; run_first[tid] = first; }
            if (tid == 0) run_pref[SK_C_RUNS] = tot;
            __syncthreads();
        }
        SK_STAMP(1)
        const uint32_t total = run_pref[SK_C_RUNS];
        for (uint32_t rb = 0; rb < total; rb += SK_C_RC) {
            const uint32_t nrec = min((uint32_t)SK_C_RC, total - rb);
            // ---- this round's records -> LDS: eight lanes per run (a run holds ~8 records), no search
            for (uint32_t r0 = 0; r0 < nruns; r0 += CT / 8) {
                const uint32_t r = r0 + (tid >> 3);
                if (r < nruns) {
                    const uint32_t a = run_pref[r], b = run_pref[r + 1 < nruns ? r + 1 : SK_C_RUNS];
                    const uint32_t lo_ = a > rb ? a : rb, hi_ = b < rb + SK_C_RC ? b : rb + SK_C_RC;       // the part of the run in this round
                    const uint32_t first = run_first[r];
                    for (uint32_t i = lo_ + (tid & 7); i < hi_; i += 8) {
                        const SkRec v = s.sorted[(size_t)first + (i - a)];
                        rlo[i - rb] = v.lo; rhi[i - rb] = v.hi;
                    }
                }
            }
            for (uint32_t i = tid; i < SK_C_RC / 2; i += CT) ((uint32_t *)mr)[i] = 0u;
            for (uint32_t i = tid; i < SK_C_DT; i += CT) own[i] = EMPTY32;
            if (tid == 0) sh[0] = 0;
            __syncthreads();
            SK_STAMP(2)
            // ---- merge identical records: a CAS names the slot's representative, later copies add to its multiplicity.
            // The winner (and a record that found no dedupe slot) appends one ITEM per k-mer to the flat work list.
            uint32_t pend_ent[(SK_C_RC + CT - 1) / CT], pend_n = 0;             // records whose items did not fit the list
            for (uint32_t i = tid; i < nrec && !(plan.dbg & 128); i += CT) {
                const uint64_t ml = rlo[i], mh = rhi[i];
                if (plan.sub_bits && kdf_sk_bucket_of(sk_rec_order(ml, mh), nb_bits) != (uint32_t)bucket) continue;   // sibling bucket's record
                uint32_t hs = sk_rec_hash(ml, mh) & (SK_C_DT - 1);
                uint32_t rep = i;                                              // default (no dedupe slot found): expanded on its own
                for (uint32_t n = 0; n < SK_C_DPROBE; ++n) {
                    uint32_t o = own[hs];
                    if (o == EMPTY32) {
                        o = atomicCAS(&own[hs], EMPTY32, i);
                        if (o == EMPTY32) break;                               // this record is the slot's representative
                    }
                    if (rlo[o] == ml && rhi[o] == mh) { rep = o; break; }
                    hs = (hs + 1) & (SK_C_DT - 1);
                }
                // (16-bit halves of one 32-bit word: the add cannot carry, a round has <= 1024 records)
                atomicAdd((uint32_t *)mr + (rep >> 1), (rep & 1) ? 0x10000u : 1u);
                if (rep == i) {
                    const uint32_t nk = SK_REC_NK(mh);
                    const uint32_t base = atomicAdd(&sh[0], nk);
                    if (base + nk <= SK_C_IC) {
#pragma unroll 4
                        for (uint32_t j = 0; j < nk; ++j) items[base + j] = (uint16_t)((i << 5) | j);
                    } else {                                                   // no room: the reserved part of the list holds no items
                        pend_ent[pend_n++] = i;
                        for (uint32_t q = base; q < SK_C_IC; ++q) { if (q >= base + nk) break; items[q] = 0xFFFFu; }
                    }
                }
            }
            __syncthreads();
            SK_STAMP(3)
            // ---- expand: every distinct record's k-mers once, count += multiplicity.  Flat list: four items per
            // thread at a time, their first-slot reads in flight together.
            const uint32_t nit = (plan.dbg & 64) ? 0u : min(sh[0], (uint32_t)SK_C_IC);
            {
                // A wave of 64 lanes pays the LONGEST probe of its lanes for every key, so the common case -- the key or
                // an empty slot within the first two slots -- is straight-line code over four items at once (eight slot
                // reads in flight), and the keys that need more go to a wave-private queue (ballot + mbcnt: no atomics,
                // no barrier) that is drained densely, one key per lane (kernel C's scheme, kdf_binned.h).
                uint64_t *wqk = qk + (tid >> 6) * SK_C_WQ; uint32_t *wqm = qm + (tid >> 6) * SK_C_WQ; uint16_t *wqs = qs + (tid >> 6) * SK_C_WQ;
                uint32_t wq_n = 0;
                for (uint32_t i0 = tid & ~63u; i0 < nit; i0 += 4 * CT) {       // wave-uniform trip count
                    uint64_t key[4], c[4][SK_C_LA]; uint32_t mult[4], sl0[4]; bool td[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t i = i0 + (tid & 63u) + u * CT;
                        const uint32_t it = i < nit ? (uint32_t)items[i] : 0xFFFFu;
                        td[u] = it != 0xFFFFu;
                        key[u] = 0; mult[u] = 0; sl0[u] = 0;
                        if (td[u]) td[u] = item_key(it, key[u], mult[u], sl0[u]);
#pragma unroll
                        for (int a = 0; a < SK_C_LA; ++a) c[u][a] = tlo[(sl0[u] + a) & bmask];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int a = 0; a < SK_C_LA; ++a) asm volatile("" : "+v"(c[u][a]));    // all reads issued before the first key is resolved
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (!__any(td[u])) continue;
                        uint32_t r = SK_C_LA; bool hit = false;
#pragma unroll
                        for (int a = SK_C_LA - 1; a >= 0; --a)
                            if (c[u][a] == key[u] || c[u][a] == KDF_EMPTY) { r = (uint32_t)a; hit = c[u][a] == key[u]; }
                        uint32_t sl = (sl0[u] + r) & bmask;
                        bool more = td[u] && r == SK_C_LA;
                        hit = hit && td[u];
                        if (td[u]) nwin += mult[u];
                        if (td[u] && !more && !hit) {                            // read as empty: the CAS tells
                            const uint64_t old = atomicCAS((unsigned long long *)&tlo[sl], KDF_EMPTY, key[u]);
                            if (old == KDF_EMPTY) { ++claimed; hit = true; }
                            else if (old == key[u]) hit = true;
                            else { more = true; sl = (sl + 1) & bmask; }
                        }
                        if (hit) atomicAdd(&tcnt[sl], mult[u]);
                        const unsigned long long mk = __ballot(more);
                        if (mk) {
                            const uint32_t at = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                            if (more) {
                                if (at < SK_C_WQ) { wqk[at] = key[u]; wqm[at] = mult[u]; wqs[at] = (uint16_t)sl; }
                                else if (!sk_slice_add(tlo, tcnt, bmask, lim, key[u], mult[u], sl, tlo[sl], claimed, (sl - sl0[u]) & bmask)) spill(key[u], mult[u]);
                            }
                            wq_n += (uint32_t)__popcll(mk);
                        }
                    }
                    if (wq_n >= SK_C_WQ - 64 || i0 + 4 * CT >= nit) {            // drain (wave-uniform): dense probing, one queued key per lane
                        const uint32_t nq = wq_n < SK_C_WQ ? wq_n : SK_C_WQ;
                        for (uint32_t q = tid & 63u; q < nq; q += 64) {
                            const uint64_t kq = wqk[q]; const uint32_t sq_ = wqs[q], mq = wqm[q];
                            const uint32_t home = kdf_sk_slot(kq, plan.bucket_bits);
                            if (!sk_slice_add(tlo, tcnt, bmask, lim, kq, mq, sq_, tlo[sq_], claimed, (sq_ - home) & bmask)) spill(kq, mq);
                        }
                        wq_n = 0;
                    }
                }
            }
            // records whose items did not fit the list (a bucket far above the mean): one k-mer after the other
            for (uint32_t p_ = 0; p_ < pend_n; ++p_) {
                const uint32_t nk = SK_REC_NK(rhi[pend_ent[p_]]);
                for (uint32_t j = 0; j < nk; ++j) {
                    uint64_t key; uint32_t mult, sl;
                    if (!item_key((pend_ent[p_] << 5) | j, key, mult, sl)) continue;
                    nwin += mult;
                    if (!sk_slice_add(tlo, tcnt, bmask, lim, key, mult, sl, tlo[sl], claimed)) spill(key, mult);
                }
            }
            __syncthreads();
            SK_STAMP(4)
        }
    }
    if (failed) atomicOr(&sh[2], 1u);
    if (claimed) atomicAdd(&sh[3], claimed);
    if (nwin) atomicAdd(&w64[0], nwin);
    __syncthreads();
    if constexpr (MODE == SK_MODE_COUNT) {
        // reserve room for this bucket's spills; no room = the bucket fails as a whole
        if (tid == 0 && !sh[2] && sh[1]) {
            // one fetch-add (a CAS loop collapses when hundreds of workgroups spill at once); a reservation that
            // does not fit leaves a hole, which the zeroed count array marks (the host clears it before the pass)
            const uint32_t n = sh[1];
            const uint32_t base = atomicAdd(&s.ctrs[SKC_SPILL], n);
            if (base >= s.sp_cap || n > s.sp_cap - base) sh[2] = 1; else sh[4] = base;
        }
        if (sh[1]) __syncthreads();                                            // (sh[1] is uniform)
        if (sh[2]) {
            // transactional: leave the bucket as it was in HBM and flag it for the replay pass (a lazily
            // cleared table holds garbage there: write an empty slice instead)
            if (tid == 0) { atomicOr(&s.failed[bucket >> 5], 1u << (bucket & 31)); atomicAdd(&s.ctrs[SKC_FAILED], 1u); }
            if (!table_nonempty)
                for (uint32_t i = tid; i < B; i += CT) { t.lo[slot0 + i] = KDF_EMPTY; t.cnt[slot0 + i] = 0; }
            return;
        }
        const uint32_t nsq = sh[1], base = sh[4];
        for (uint32_t i = tid; i < nsq; i += CT) { s.sp_key[base + i] = sqk[i]; s.sp_cnt[base + i] = sqc[i]; }
    }
    // write-back.  LDS counts were advanced with wrapping adds; a pass adds fewer than 2^32 to a slot, so a slot
    // wrapped iff its new value is below the value it had in HBM: saturate those (Jellyfish's 4-byte counter).
    for (uint32_t i = tid; i < B / 2 && !(plan.dbg & 512); i += CT) {
        ((ulonglong2 *)(t.lo + slot0))[i] = ((const ulonglong2 *)tlo)[i];
        uint2 c = ((const uint2 *)tcnt)[i];
        if (load) {
            const uint2 o = ((const uint2 *)(t.cnt + slot0))[i];
            if (c.x < o.x) c.x = 0xFFFFFFFFu;
            if (c.y < o.y) c.y = 0xFFFFFFFFu;
        }
        ((uint2 *)(t.cnt + slot0))[i] = c;
    }
    SK_STAMP(5)
    if (stampit) for (int i = 0; i < 6; ++i) atomicAdd(&s.ctrs[8 + i], st_acc[i] >> 6);
    if (tid == 0) {
        const uint32_t shard = (uint32_t)(bucket % KDF_SHARDS) * 16;
        if (sh[3]) atomicAdd(&ctl->distinct[shard], (unsigned long long)sh[3]);
        if (w64[0]) atomicAdd(&ctl->windows[shard], w64[0]);
    }
}

// overflow-table insert of the spill list (thread per entry); n is read from the device counter
__global__ __launch_bounds__(256) void sk_spill_insert_kernel(SkScratch s, KdfTable t, KdfCtl *ctl) {
    const uint32_t n = min(s.ctrs[SKC_SPILL], s.sp_cap);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t claimed = 0; bool full = false;
    if (i < n && s.sp_cnt[i] && !kdf_sk_ovf_add<true>(t, s.sp_key[i], s.sp_cnt[i], claimed)) full = true;     // count 0: a hole
    if (full) atomicOr(&ctl->error, 1u);
    uint32_t c = claimed;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c)
        atomicAdd(&ctl->distinct[((blockIdx.x * 4 + (threadIdx.x >> 6)) % KDF_SHARDS) * 16], (unsigned long long)c);
}

// (key, count) pairs into an SK table through global memory (index loads, merges, rehash): the bucket first, what does
// not fit its neighbourhood is appended to the spill list (room for all n pairs) and inserted by sk_spill_insert_kernel
__global__ __launch_bounds__(256) void sk_insert_keys_kernel(const uint64_t *__restrict__ klo, const uint32_t *__restrict__ add,
                                                            uint64_t n, KdfTable t, KdfCtl *ctl, SkScratch s, int skip_empty) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t claimed = 0;
    if (i < n) {
        const uint64_t key = klo[i];
        if (!(skip_empty && key == KDF_EMPTY)) {
            const uint32_t a = add ? add[i] : 0u;
            if (!kdf_sk_main_add<true>(t, key, a, claimed)) {
                const uint32_t p = atomicAdd(&s.ctrs[SKC_SPILL], 1u);
                if (p < s.sp_cap) { s.sp_key[p] = key; s.sp_cnt[p] = a; }
                else s.ctrs[SKC_SPILL_LOST] = 1;
            }
        }
    }
    uint32_t c = claimed;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c)
        atomicAdd(&ctl->distinct[((blockIdx.x * 4 + (threadIdx.x >> 6)) % KDF_SHARDS) * 16], (unsigned long long)c);
}

// old overflow entries into a new (larger) overflow array; *ctr receives the keys placed
__global__ __launch_bounds__(256) void sk_ovf_rehash_kernel(const uint64_t *__restrict__ olo, const uint32_t *__restrict__ ocnt,
                                                           uint64_t n, KdfTable t, KdfCtl *ctl, uint32_t *ctr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t claimed = 0;
    if (i < n && olo[i] != KDF_EMPTY && !kdf_sk_ovf_add<true>(t, olo[i], ocnt[i], claimed)) atomicOr(&ctl->error, 1u);
    uint32_t c = claimed;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(ctr, c);
}

// ---------------------------------------------------------------------------------------------------------------
// Balanced minimizer -> bucket assignment.  With m = 12 only a few "active" minimizers land in each bucket and
// their sizes differ widely, so the fullest bucket of a plain hash holds about twice the mean.  Weights are sampled
// from the first batch's records (every SK_WSAMPLE-th chunk: k-mer instances per minimizer) or taken from the live
// table (one per stored key, when a table grows); per coarse bin the active minimizers are sorted by weight and
// dealt to the bin's buckets in snake order (heaviest first).
#define SK_WSAMPLE 8
#define SK_A_CAP   12288                  // active minimizers of one coarse bin the sort holds (8 B each)

__global__ __launch_bounds__(256) void sk_weight_records_kernel(SkScratch s, uint32_t *__restrict__ weights) {
    // every SK_WSAMPLE-th chunk OF EACH BIN (chunk ids are handed out bin after bin within a round, so a stride over
    // the ids would sample some bins only)
    const uint32_t n_chunks = min(s.ctrs[SKC_POOL], s.max_chunks);
    const uint32_t c = blockIdx.x;
    if (s.ctrs[SKC_EXHAUSTED] || c >= n_chunks || s.chunk_pos[c] % SK_WSAMPLE) return;
    const uint32_t fill = s.chunk_fill[c];
    if (threadIdx.x < fill) {
        const SkRec v = s.chunks[(size_t)c * SK_CHUNK + threadIdx.x];
        atomicAdd(&weights[kdf_sk_spread(sk_rec_order(v.lo, v.hi))], SK_REC_NK(v.hi));
    }
}
__global__ __launch_bounds__(256) void sk_weight_table_kernel(const uint64_t *__restrict__ klo, uint64_t n, int k, uint32_t *__restrict__ weights) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && klo[i] != KDF_EMPTY) atomicAdd(&weights[kdf_sk_spread(kdf_sk_min_of_key(klo[i], k))], 1u);
}
// one workgroup per coarse bin; assign[h] for every h of the bin
__global__ __launch_bounds__(1024) void sk_assign_kernel(const uint32_t *__restrict__ weights, uint16_t *__restrict__ assign, uint32_t c1, uint32_t c2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *ent = (unsigned long long *)smem;                      // [SK_A_CAP] weight << 32 | h offset
    __shared__ uint32_t n_act;
    const uint32_t H = 1u << (24 - c1), h0 = blockIdx.x << (24 - c1), nf = 1u << c2, tid = threadIdx.x;
    __shared__ uint32_t whist[33], wmin;
    if (tid == 0) n_act = 0;
    if (tid < 33) whist[tid] = 0;
    __syncthreads();
    // The sort holds SK_A_CAP minimizers; a bin may have more with a non-zero weight (most of them light).  Take the
    // heaviest weight classes (bit lengths) that fit; the light rest keeps the default bucket.
    for (uint32_t i = tid; i < H; i += 1024) { const uint32_t w = weights[h0 + i]; if (w) atomicAdd(&whist[32 - __clz(w)], 1u); }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0, c = 32;
        while (c >= 1 && acc + whist[c] <= SK_A_CAP) { acc += whist[c]; --c; }
        wmin = c >= 32 ? 0xFFFFFFFFu : (c == 0 ? 1u : (1u << c));            // weights >= wmin are sorted
    }
    __syncthreads();
    const uint32_t wthr = wmin;
    // default for every h: the next bits of h (what the table-less layout uses); the sorted ones are overwritten below
    for (uint32_t i = tid; i < H; i += 1024) {
        const uint32_t w = weights[h0 + i];
        assign[h0 + i] = (uint16_t)(c2 ? ((i >> (24 - c1 - c2)) & (nf - 1)) : 0u);
        if (w >= wthr) { const uint32_t p = atomicAdd(&n_act, 1u); if (p < SK_A_CAP) ent[p] = ((unsigned long long)w << 32) | i; }
    }
    __syncthreads();
    const uint32_t n = min(n_act, (uint32_t)SK_A_CAP);
    uint32_t np2 = 1; while (np2 < n) np2 <<= 1;
    for (uint32_t i = n + tid; i < np2; i += 1024) ent[i] = 0ull;              // (np2 <= 16384 entries = 128 KB)
    __syncthreads();
    // bitonic sort, descending (ties broken by h: deterministic whatever order the atomics appended in)
    for (uint32_t kk = 2; kk <= np2; kk <<= 1)
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < np2; i += 1024) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const unsigned long long a = ent[i], b = ent[l];
                    const bool desc = (i & kk) == 0;
                    if (desc ? a < b : a > b) { ent[i] = b; ent[l] = a; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = tid; i < n; i += 1024) {
        const uint32_t round = i >> c2, pos = i & (nf - 1);
        assign[h0 + (uint32_t)(ent[i] & 0xFFFFFFFFu)] = (uint16_t)((round & 1) ? nf - 1 - pos : pos);
    }
}
