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
#define SK_MIN_K     16
// bucket kernel
#ifndef SK_C_THREADS
#define SK_C_THREADS 256
#endif
#define SK_C_RC      1024                  // records per dedupe round
#define SK_C_DT      1024                  // dedupe hash slots
#define SK_C_RUNS    SK_C_THREADS          // runs (groups of the bin) staged per round: one per thread
#define SK_C_SQ      128                   // LDS spill queue entries
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
template <int K>
__device__ __forceinline__ void sk_windows(const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid,
                                           uint64_t P, uint32_t (&mv)[SK_WPT + 1], uint32_t &v17) {
    using C = SkK<K>;
    constexpr int NM = C::NM, W = C::W, M = C::M;
    uint64_t e0, e1, inv;
    if (P > 0) {
        const uint64_t S = P - 1;
        const uint64_t w0 = S >> 5; const int sh = (int)(S & 31) * 2;
        const uint64_t x0 = packed[w0], x1 = packed[w0 + 1], x2 = packed[w0 + 2];
        e0 = kdf_funnel(x0, x1, sh); e1 = kdf_funnel(x1, x2, sh);
        const uint64_t mw = S >> 6; const int msh = (int)(S & 63);
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
    uint16_t *nat = (uint16_t *)(wsum + 40);                                   // [SK_THREADS] natural breaks
    uint16_t *brk = nat + SK_THREADS;                                          // [SK_THREADS + 4] final breaks
    const int nb = 1 << plan.c1;
    const int tid = threadIdx.x;
    for (int i = tid; i < nb; i += SK_THREADS) { hist[i] = 0; cur_chunk[i] = SK_NONE; cur_fill[i] = 0; }
    if (tid < 4) brk[SK_THREADS + tid] = 0xFFFFu;                              // the slab's end is a break
    __syncthreads();
    const uint64_t slab0 = (uint64_t)blockIdx.x * slabs_per_wg;
    const int half = tid >> 5, lane32 = tid & 31;
    constexpr int NHALF = SK_THREADS / 32;
    for (uint32_t sl = 0; sl < slabs_per_wg; ++sl) {
        const uint64_t P0 = (slab0 + sl) * (uint64_t)SK_SLAB;
        if (P0 >= n_bases) break;                                              // uniform
        const uint64_t P = P0 + (uint64_t)tid * SK_WPT;
        uint32_t round = 0, rounds = 1;
        do {
            // ---- windows, minimizers, record boundaries (recomputed per round: rounds > 1 only for pathological slabs)
            uint32_t start16 = 0, brk16 = 0xFFFFu;
            {
                uint32_t mv[SK_WPT + 1], v17 = 0;
                if (P < n_bases) sk_windows<K>(packed, invalid, P, mv, v17);
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
            const uint32_t nrec = __popc(start16);
            uint32_t total = 0;
            const uint32_t tbase = kb_block_exscan(nrec, wsum, &total);       // (its barriers publish brk[])
            rounds = (total + SK_CAP - 1) / SK_CAP;
            if (rounds == 0) break;
            const uint32_t r_lo = round * SK_CAP;
            // ---- emission: one record per start bit
            {
                const uint64_t look = (uint64_t)brk16 | ((uint64_t)brk[tid + 1] << 16) | ((uint64_t)brk[tid + 2] << 32);
                uint32_t sb = start16, ord = tbase;
                while (sb) {
                    const int i = __ffs(sb) - 1; sb &= sb - 1;
                    const uint32_t r = ord - r_lo; ++ord;
                    if (r >= SK_CAP) continue;                                 // another round's record (unsigned wrap: earlier rounds too)
                    const uint64_t after = look >> (i + 1);
                    const int nk = after ? __ffsll((unsigned long long)after) : 48;
                    if (nk > C::MAXNK) { s.ctrs[SKC_BADNK] = 1; continue; }  // cannot happen (forced cuts); never silent
                    const uint32_t mvv = lds_mv[i * SK_THREADS + tid];
                    const uint32_t g = mvv >> 8;
                    const int off_fw = (int)(mvv & 0xFF) - 1 - i;              // minimizer m-mer offset inside the record
                    const uint64_t Q = P + i;
                    const int nbases = nk + K - 1;                             // <= 52
                    const uint64_t w0 = Q >> 5; const int sh = (int)(Q & 31) * 2;
                    const uint64_t x0 = packed[w0], x1 = packed[w0 + 1], x2 = packed[w0 + 2];
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
                    const uint32_t bin = plan.c1 ? (g >> (24 - plan.c1)) : 0u;
                    const uint32_t rank = atomicAdd(&hist[bin], 1u);
                    U[r] = SkRec{lo, hi};
                    T[r] = (bin << 16) | rank;
                }
            }
            kb_lds_barrier();
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
            }
            kb_lds_barrier();
            {
                const uint32_t nr = min((uint32_t)SK_CAP, total - r_lo);
                for (uint32_t r = tid; r < nr; r += SK_THREADS) {
                    const uint32_t tg = T[r];
                    img[offs[tg >> 16] + (tg & 0xFFFFu)] = U[r];
                }
            }
            kb_lds_barrier();
            // ---- copy-out: a half-wave per bin; the bin's run goes to this workgroup's current chunk of the bin
            for (int bin = half; bin < nb; bin += NHALF) {
                const uint32_t n = hist[bin], o = offs[bin];
                if (n == 0) continue;
                uint32_t ch = cur_chunk[bin], fl = cur_fill[bin], done = 0;
                while (done < n) {
                    if (ch == SK_NONE || fl == SK_CHUNK) {
                        uint32_t nid = 0;
                        if (lane32 == 0) {
                            if (ch != SK_NONE && ch < s.max_chunks) s.chunk_fill[ch] = SK_CHUNK;
                            nid = atomicAdd(&s.ctrs[SKC_POOL], 1u);
                            if (nid < s.max_chunks) {
                                s.chunk_bin[nid] = (uint32_t)bin;
                                s.chunk_pos[nid] = atomicAdd(&s.bin_nchunks[bin], 1u);
                            } else s.ctrs[SKC_EXHAUSTED] = 1;
                        }
                        ch = __shfl(nid, 0, 32); fl = 0;
                    }
                    const uint32_t take = min(n - done, (uint32_t)SK_CHUNK - fl);
                    if (ch < s.max_chunks) {
                        SkRec *dst = s.chunks + (size_t)ch * SK_CHUNK + fl;
                        for (uint32_t i = lane32; i < take; i += 32) dst[i] = img[o + done + i];
                    }
                    done += take; fl += take;
                }
                if (lane32 == 0) { cur_chunk[bin] = ch; cur_fill[bin] = fl; hist[bin] = 0; }
            }
            kb_lds_barrier();
            ++round;
        } while (round < rounds);
    }
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
    const uint32_t e = (uint32_t)kdf_funnel(lo, hi & SK_HI_BASES, 2 * (int)SK_REC_OFF(hi)) & MM;   // m-mer, base i in bits 2i
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
            const uint32_t g = sk_rec_order(rl[e], rh[e]);
            const uint32_t f = plan.c2 ? ((g >> (24 - plan.c1 - plan.c2)) & ((1u << plan.c2) - 1)) : 0u;
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
    uint32_t *run_pref = sqc + SK_C_SQ;                                        // [RUNS + 1]
    uint32_t *run_first = run_pref + SK_C_RUNS + 1;                            // [RUNS]
    uint32_t *wsum = run_first + SK_C_RUNS;                                    // [40]
    uint32_t *sh = wsum + 40;                                                  // [8] n_dl, n_sq, failed, claimed, sp_base
    uint16_t *dl = (uint16_t *)(sh + 8);                                       // [RC] distinct list

    const uint32_t nbk = gridDim.x, tid = threadIdx.x;
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

    const int k = (int)plan.k;
    const uint64_t kmask = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const bool sliced = plan.key_parts > 1;
    const uint32_t nb_bits = plan.log2cap - plan.bucket_bits;
    uint32_t claimed = 0; unsigned long long nwin = 0; bool failed = false;
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
        const uint32_t total = run_pref[SK_C_RUNS];
        for (uint32_t rb = 0; rb < total; rb += SK_C_RC) {
            const uint32_t nrec = min((uint32_t)SK_C_RC, total - rb);
            // ---- this round's records -> LDS
            for (uint32_t i = tid; i < nrec; i += CT) {
                const uint32_t flat = rb + i;
                uint32_t lo_ = 0, hi_ = nruns;                                 // largest run with run_pref[run] <= flat
                while (hi_ - lo_ > 1) { const uint32_t mid = (lo_ + hi_) >> 1; if (run_pref[mid] <= flat) lo_ = mid; else hi_ = mid; }
                const SkRec v = s.sorted[(size_t)run_first[lo_] + (flat - run_pref[lo_])];
                rlo[i] = v.lo; rhi[i] = v.hi;
            }
            for (uint32_t i = tid; i < SK_C_DT; i += CT) own[i] = EMPTY32;
            if (tid == 0) sh[0] = 0;
            __syncthreads();
            // ---- merge identical records: a CAS names the slot's representative, later copies add to its multiplicity
            for (uint32_t i = tid; i < nrec; i += CT) {
                const uint64_t ml = rlo[i], mh = rhi[i];
                if (plan.sub_bits && kdf_sk_bucket_of(sk_rec_order(ml, mh), nb_bits) != (uint32_t)bucket) continue;   // sibling bucket's record
                uint32_t hs = sk_rec_hash(ml, mh) & (SK_C_DT - 1);
                uint32_t entry = 0x8000u | i;                                  // default: expanded on its own
                bool append = true;
                for (uint32_t n = 0; n < SK_C_DPROBE; ++n) {
                    uint32_t o = own[hs];
                    if (o == EMPTY32) {
                        o = atomicCAS(&own[hs], EMPTY32, (1u << 16) | i);
                        if (o == EMPTY32) { entry = hs; break; }
                    }
                    const uint32_t rep = o & 0xFFFFu;
                    if (rlo[rep] == ml && rhi[rep] == mh) { atomicAdd(&own[hs], 1u << 16); append = false; break; }
                    hs = (hs + 1) & (SK_C_DT - 1);
                }
                if (append) dl[atomicAdd(&sh[0], 1u)] = (uint16_t)entry;
            }
            __syncthreads();
            // ---- expand every distinct record once
            const uint32_t ndl = sh[0];
            for (uint32_t e = tid; e < ndl; e += CT) {
                const uint32_t ent = dl[e];
                uint32_t r, mult;
                if (ent & 0x8000u) { r = ent & 0x7FFFu; mult = 1; }
                else { const uint32_t o = own[ent]; r = o & 0xFFFFu; mult = o >> 16; }
                const uint64_t lo = rlo[r], hw = rhi[r];
                const uint32_t nk = SK_REC_NK(hw);
                const uint64_t hb = hw & SK_HI_BASES;
                for (uint32_t j = 0; j < nk; ++j) {
                    const uint64_t key = kdf_canon_narrow(kdf_funnel(lo, hb, 2 * (int)j), k, kmask);
                    const uint64_t hsh = kdf_mix64(key);
                    if (sliced && kdf_slice(hsh, plan.key_parts) != plan.key_part) continue;
                    nwin += mult;
                    uint32_t sl = (uint32_t)(hsh >> (64 - plan.bucket_bits));
                    const uint32_t lim = B < KDF_SK_MAXPROBE ? B : KDF_SK_MAXPROBE;
                    bool placed = false;
                    for (uint32_t n = 0; n < lim; ++n) {
                        uint64_t cur = tlo[sl];
                        if (cur == KDF_EMPTY) {
                            cur = atomicCAS((unsigned long long *)&tlo[sl], KDF_EMPTY, key);
                            if (cur == KDF_EMPTY) { ++claimed; cur = key; }
                        }
                        if (cur == key) { atomicAdd(&tcnt[sl], mult); placed = true; break; }
                        sl = (sl + 1) & bmask;
                    }
                    if (!placed) {                                             // the key's neighbourhood is full: overflow table
                        if constexpr (MODE == SK_MODE_REPLAY) {
                            const uint32_t p = atomicAdd(&s.ctrs[SKC_SPILL], 1u);
                            if (p < s.sp_cap) { s.sp_key[p] = key; s.sp_cnt[p] = mult; }
                            else s.ctrs[SKC_SPILL_LOST] = 1;                   // host sized the list for the worst case: never taken
                        } else {
                            const uint32_t q = atomicAdd(&sh[1], 1u);
                            if (q < SK_C_SQ) { sqk[q] = key; sqc[q] = mult; }
                            else failed = true;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    if (failed) atomicOr(&sh[2], 1u);
    if (claimed) atomicAdd(&sh[3], claimed);
    if (nwin) atomicAdd(&w64[0], nwin);
    __syncthreads();
    if constexpr (MODE == SK_MODE_COUNT) {
        // reserve room for this bucket's spills; no room = the bucket fails as a whole
        if (tid == 0 && !sh[2] && sh[1]) {
            const uint32_t n = sh[1];
            uint32_t old = s.ctrs[SKC_SPILL];
            for (;;) {                                                         // CAS: a failed reservation leaves no hole in the list
                if (old + n > s.sp_cap || old + n < old) { sh[2] = 1; break; }
                const uint32_t seen = atomicCAS(&s.ctrs[SKC_SPILL], old, old + n);
                if (seen == old) { sh[4] = old; break; }
                old = seen;
            }
        }
        __syncthreads();
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
    for (uint32_t i = tid; i < B / 2; i += CT) {
        ((ulonglong2 *)(t.lo + slot0))[i] = ((const ulonglong2 *)tlo)[i];
        uint2 c = ((const uint2 *)tcnt)[i];
        if (load) {
            const uint2 o = ((const uint2 *)(t.cnt + slot0))[i];
            if (c.x < o.x) c.x = 0xFFFFFFFFu;
            if (c.y < o.y) c.y = 0xFFFFFFFFu;
        }
        ((uint2 *)(t.cnt + slot0))[i] = c;
    }
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
    if (i < n && !kdf_sk_ovf_add<true>(t, s.sp_key[i], s.sp_cnt[i], claimed)) full = true;
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
