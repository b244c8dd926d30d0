// kdf_merge.h -- the two ends of the multi-GPU merge of per-rank counts (`jellyfish merge` of partial indexes,
// reference kmer_denovo_filter/core/jellyfish_wrappers.py:335-366, done in HBM across ranks).
//
// SENDER  (kdf_export_parts_dev): the hash-layout table is dumped in HASH ORDER -- exactly grouped by the top
//   P = log2cap - 12 + KM_SUB_BITS bits of the key's hash -- into one contiguous range per owner rank.  Two reads of
//   the table (count per 4096-slot block, scan, write); inside a block the kept entries are counting-sorted in LDS by
//   the next KM_SUB_BITS hash bits, because a block of the sender's table covers several buckets of an owner's table
//   (the owner drops the log2(world) hash bits that name it, KdfTable::hshift).
// OWNER   (kdf_add_pairs_multi_dev): every source's segment arrives grouped by the owner table's buckets, in ascending
//   order.  km_bounds_kernel finds, per segment, the range of pairs of each bucket (and proves the grouping: one
//   descending step anywhere raises `flag` and everything falls back to the global-atomic insert, so any input is
//   merged correctly); km_merge_kernel then gives each table bucket to one workgroup: the bucket is staged in LDS
//   (or starts empty there: FRESH, the table's deferred clear is folded into the write), the bucket's pairs of ALL
//   segments are inserted with LDS atomics, and the bucket goes back to HBM once.  Algorithmic traffic: 12 (20) B per
//   pair read + 12 (20) B per table slot written (+ read when the table held keys), against one CAS and one atomic
//   add through L2 per pair in the atomic kernel.
#pragma once
#include "kdf_device.h"

#define KM_BLOCK_SLOTS 4096u              // table slots one workgroup of the dump owns (a narrow bucket, two wide ones)
#define KM_THREADS     256
#ifndef KM_MERGE_THREADS
#define KM_MERGE_THREADS 512              // km_merge_kernel: threads per table bucket (256 / 512 / 1024: a fresh owner table 5.3 / 4.6 / 5.3 ms)
#endif
#define KM_SPT         (KM_BLOCK_SLOTS / KM_THREADS)     // slots per thread
#define KM_SUB_BITS    6                  // hash bits below the block prefix that order the dump
#define KM_MAX_SEGS    64

struct KmSegs {                           // the segments of one merge call (device pointers), by value
    const uint64_t *lo[KM_MAX_SEGS];
    const uint64_t *hi[KM_MAX_SEGS];
    const uint32_t *cnt[KM_MAX_SEGS];
    uint32_t n[KM_MAX_SEGS];
    uint32_t nseg;
};

template <int KW>
__device__ __forceinline__ bool km_keep(uint64_t lo, uint64_t hi, uint32_t c, uint32_t min_count) {
    const bool occ = KW == 1 ? (lo != KDF_EMPTY) : (hi != KDF_EMPTY);
    return occ && c >= min_count;
}

// pass 1 of the ordered dump: kept entries per block of 4096 slots
template <int KW>
__global__ __launch_bounds__(KM_THREADS) void km_count_kernel(KdfTable t, uint32_t min_count, uint32_t *__restrict__ blk_cnt) {
    const uint64_t cap = 1ull << t.log2cap;
    const uint64_t first = (uint64_t)blockIdx.x * KM_BLOCK_SLOTS;
    __shared__ uint32_t wsum[KM_THREADS / 64];
    uint32_t mine = 0;
#pragma unroll 4
    for (uint32_t r = 0; r < KM_SPT; ++r) {
        const uint64_t i = first + r * KM_THREADS + threadIdx.x;
        if (i < cap) {
            if (min_count >= 1) mine += t.cnt[i] >= min_count ? 1u : 0u;      // count > 0 implies occupied
            else mine += (KW == 1 ? t.lo[i] != KDF_EMPTY : t.hi[i] != KDF_EMPTY) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t s = 0; for (int w = 0; w < KM_THREADS / 64; ++w) s += wsum[w]; blk_cnt[blockIdx.x] = s; }
}

// exclusive scan of the block counts (one workgroup) + the offsets at which the owner parts start
__global__ __launch_bounds__(1024) void km_scan_kernel(const uint32_t *__restrict__ blk_cnt, unsigned long long *__restrict__ blk_off,
                                                       uint64_t nblk, const uint64_t *__restrict__ part_first_blk, uint32_t parts,
                                                       unsigned long long *__restrict__ part_off /* [parts + 1] */) {
    __shared__ unsigned long long wtot[16];
    __shared__ unsigned long long carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < nblk; b0 += 1024) {
        const uint64_t b = b0 + tid;
        const unsigned long long v = b < nblk ? blk_cnt[b] : 0ull;
        unsigned long long x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long y = __shfl_up(x, o); if ((int)lane >= o) x += y; }
        if (lane == 63) wtot[w] = x;
        __syncthreads();
        unsigned long long pre = carry;
        for (uint32_t q = 0; q < w; ++q) pre += wtot[q];
        if (b < nblk) blk_off[b] = pre + x - v;
        __syncthreads();
        if (tid == 1023) carry = pre + x;
        __syncthreads();
    }
    if (tid == 0) blk_off[nblk] = carry;
    __syncthreads();
    if (tid <= parts && part_off) part_off[tid] = tid == parts ? carry : blk_off[part_first_blk[tid]];
}

// Packed layout of the dump (what the all-to-all sends): part p is ONE byte segment [lo x n_p | hi x n_p (wide) | counts x n_p],
// segments back to back, each start rounded up to 8 bytes.  part_base[p] = byte offset of segment p (part_base[parts] = total).
__global__ void km_packbase_kernel(const unsigned long long *__restrict__ part_off, uint32_t parts, uint32_t esz,
                                   unsigned long long *__restrict__ part_base) {
    if (threadIdx.x || blockIdx.x) return;
    unsigned long long b = 0;
    for (uint32_t p = 0; p < parts; ++p) {
        part_base[p] = b;
        b += ((part_off[p + 1] - part_off[p]) * esz + 7ull) & ~7ull;
    }
    part_base[parts] = b;
}

// pass 2: a block's kept entries, grouped by the KM_SUB_BITS hash bits below the block prefix, at blk_off[block]
// PACKED: olo is the byte buffer of the packed layout (km_packbase_kernel), out_cap its size in bytes; a block lies in one part.
template <int KW, bool PACKED>
__global__ __launch_bounds__(KM_THREADS) void km_write_kernel(KdfTable t, uint32_t min_count, const unsigned long long *__restrict__ blk_off,
                                                              uint64_t *__restrict__ olo, uint64_t *__restrict__ ohi,
                                                              uint32_t *__restrict__ ocnt, uint64_t out_cap,
                                                              const unsigned long long *__restrict__ part_off,
                                                              const unsigned long long *__restrict__ part_base, uint32_t parts) {
    constexpr uint32_t NB = 1u << KM_SUB_BITS;
    const uint64_t cap = 1ull << t.log2cap;
    const uint64_t first = (uint64_t)blockIdx.x * KM_BLOCK_SLOTS;
    const unsigned long long base = blk_off[blockIdx.x];
    if (blk_off[blockIdx.x + 1] == base) return;
    __shared__ uint32_t hist[NB], start[NB];
    if (threadIdx.x < NB) hist[threadIdx.x] = 0;
    // packed: this block's part (owner of the 16-bit hash prefix of its first slot), its pair range and byte segment
    unsigned long long p_first = 0, p_n = 0; char *seg = nullptr;
    if constexpr (PACKED) {
        const uint32_t part = (uint32_t)((((first >> (t.log2cap - 16)) & 0xFFFFu) * parts) >> 16);
        p_first = part_off[part]; p_n = part_off[part + 1] - p_first;
        const unsigned long long pb = part_base[part];
        if (pb + ((p_n * (8ull * KW + 4ull) + 7ull) & ~7ull) > out_cap) return;          // (the host checks the total and fails the call)
        seg = (char *)olo + pb;
    }
    __syncthreads();
    // the block prefix is the top (log2cap - 12) hash bits (fewer than 12 bits of table: one block, no prefix)
    const uint32_t pre_bits = t.log2cap > 12 ? t.log2cap - 12 : 0;
    uint64_t lo[KM_SPT], hi[KM_SPT]; uint32_t c[KM_SPT], sub[KM_SPT], rank[KM_SPT];
#pragma unroll
    for (uint32_t r = 0; r < KM_SPT; ++r) {
        const uint64_t i = first + r * KM_THREADS + threadIdx.x;
        lo[r] = KDF_EMPTY; hi[r] = KW == 2 ? KDF_EMPTY : 0; c[r] = 0;
        if (i < cap) { lo[r] = t.lo[i]; if (KW == 2) hi[r] = t.hi[i]; c[r] = t.cnt[i]; }
    }
#pragma unroll
    for (uint32_t r = 0; r < KM_SPT; ++r) {
        sub[r] = 0xFFFFFFFFu;
        if (km_keep<KW>(lo[r], hi[r], c[r], min_count)) {
            const uint64_t h = lo[r] << t.hshift;                   // the slot holds the hash (stored form)
            sub[r] = (uint32_t)((h << pre_bits) >> (64 - KM_SUB_BITS));
            rank[r] = atomicAdd(&hist[sub[r]], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                    // NB = 64: one wave scans the histogram
        uint32_t v = hist[threadIdx.x], x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if ((int)threadIdx.x >= o) x += y; }
        start[threadIdx.x] = x - v;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < KM_SPT; ++r) {
        if (sub[r] != 0xFFFFFFFFu) {
            const uint64_t pos = base + start[sub[r]] + rank[r];
            if constexpr (PACKED) {
                const uint64_t i = pos - p_first;
                ((uint64_t *)seg)[i] = kdf_key_lo(lo[r], KW == 2 ? hi[r] : 0);      // keys leave the engine as keys
                if (KW == 2) ((uint64_t *)seg)[p_n + i] = hi[r];
                ((uint32_t *)(seg + p_n * 8ull * KW))[i] = c[r];
            } else if (pos < out_cap) {
                olo[pos] = kdf_key_lo(lo[r], KW == 2 ? hi[r] : 0);
                if (KW == 2 && ohi) ohi[pos] = hi[r];
                if (ocnt) ocnt[pos] = c[r];
            }
        }
    }
}
static_assert(KM_SUB_BITS == 6, "km_write_kernel scans its histogram with one wave");

// ---- owner side -----------------------------------------------------------------------------------------------------

template <int KW>
__device__ __forceinline__ uint32_t km_bucket_of(const KdfTable &t, uint64_t lo, uint64_t hi) {
    return (uint32_t)(kdf_home(t, kdf_hash(lo, hi)) >> t.bucket_bits);
}

// per segment (blockIdx.y): first[seg * nb + b], last[seg * nb + b] = the range of the segment's pairs that belong to
// table bucket b (both 0: none); flag[0] |= 1 when a segment is not grouped by bucket in ascending order
template <int KW>
__global__ __launch_bounds__(256) void km_bounds_kernel(KdfTable t, KmSegs sg, uint32_t nb, uint32_t *__restrict__ first,
                                                        uint32_t *__restrict__ last, uint32_t *__restrict__ flag) {
    const uint32_t seg = blockIdx.y, n = sg.n[seg];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if ((uint64_t)blockIdx.x * 256 >= n) return;
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;     // already known: the fallback runs
    const uint64_t *klo = sg.lo[seg], *khi = sg.hi[seg];
    uint32_t b = 0xFFFFFFFFu;
    if (i < n) b = km_bucket_of<KW>(t, klo[i], KW == 2 ? khi[i] : 0);
    uint32_t pb = __shfl_up(b, 1);
    if ((threadIdx.x & 63) == 0) pb = (i > 0 && i < n) ? km_bucket_of<KW>(t, klo[i - 1], KW == 2 ? khi[i - 1] : 0) : 0xFFFFFFFFu;
    if (i >= n) return;
    uint32_t *f = first + (size_t)seg * nb, *l = last + (size_t)seg * nb;
    if (i == 0) f[b] = 0;
    else if (b != pb) {
        if (b < pb) atomicOr(flag, 1u);
        else { f[b] = (uint32_t)i; l[pb] = (uint32_t)i; }
    }
    if (i == n - 1) l[b] = n;
}

__device__ __forceinline__ void km_lds_sat_add(uint32_t *p, uint32_t add) {
    const uint32_t old = atomicAdd(p, add);
    if (old + add < old || old + add == 0xFFFFFFFFu) atomicMax(p, 0xFFFFFFFFu);
}

// one (key, add) into the LDS bucket, narrow keys: four slots per iteration (as kb_probe_narrow, kdf_binned.h)
__device__ __forceinline__ void km_probe_narrow(uint64_t *tlo, uint32_t *tcnt, uint32_t bmask, uint64_t klo, uint32_t add, uint32_t sl,
                                                uint32_t &claimed, bool &failed) {
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
            const uint64_t old = atomicCAS((unsigned long long *)&tlo[at], KDF_EMPTY, klo);
            if (old == KDF_EMPTY) { claimed++; isk = true; }
            else if (old == klo) isk = true;
        }
        if (isk) { if (add) km_lds_sat_add(&tcnt[at], add); return; }
        sl = (at + 1) & bmask; n += f + 1;
    }
    failed = true;
}

// wide keys, one ATTEMPT: 0 done, 1 bucket full, 2 blocked by a slot another lane is publishing (the caller retries
// under a wave-uniform loop, as kb_probe_wide_wave does)
__device__ __forceinline__ int km_probe_wide_once(uint64_t *tlo, uint64_t *thi, uint32_t *tcnt, uint32_t bmask, uint64_t klo, uint64_t khi,
                                                  uint32_t add, uint32_t sl, uint32_t &claimed) {
    for (uint32_t n = 0; n <= bmask; ++n) {
        uint64_t chi = __hip_atomic_load(&thi[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (chi == KDF_EMPTY) {
            chi = atomicCAS((unsigned long long *)&thi[sl], KDF_EMPTY, khi | KDF_PENDING);
            if (chi == KDF_EMPTY) {
                __hip_atomic_store(&tlo[sl], klo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&thi[sl], khi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                claimed++;
                if (add) km_lds_sat_add(&tcnt[sl], add);
                return 0;
            }
        }
        if ((chi & ~KDF_PENDING) == khi) {
            if (chi & KDF_PENDING) return 2;
            const uint64_t clo = __hip_atomic_load(&tlo[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (clo == klo) { if (add) km_lds_sat_add(&tcnt[sl], add); return 0; }
        }
        sl = (sl + 1) & bmask;
    }
    return 1;
}

// one workgroup per table bucket: all segments' pairs of the bucket go in through LDS
template <int KW, bool FRESH>
__global__ __launch_bounds__(KM_MERGE_THREADS) void km_merge_kernel(KdfTable t, KmSegs sg, uint32_t nb, const uint32_t *__restrict__ first,
                                                              const uint32_t *__restrict__ last, const uint32_t *__restrict__ flag, KdfCtl *ctl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t B = 1u << t.bucket_bits, bmask = B - 1;
    uint64_t *tlo = (uint64_t *)smem;
    uint64_t *thi = KW == 2 ? tlo + B : nullptr;
    uint32_t *tcnt = (uint32_t *)(smem + (size_t)B * 8 * KW);
    __shared__ uint32_t sh_tot, sh_claimed, sh_failed;
    __shared__ uint32_t sf[KM_MAX_SEGS], sl_[KM_MAX_SEGS], spre[KM_MAX_SEGS + 1];
    __shared__ const uint64_t *sp_lo[KM_MAX_SEGS], *sp_hi[KM_MAX_SEGS];
    __shared__ const uint32_t *sp_cnt[KM_MAX_SEGS];
    const uint32_t tid = threadIdx.x;
    // an XCD takes a contiguous eighth of the buckets (workgroups are dealt to the XCDs round robin): the segments are
    // in bucket order, so its L2 sees each segment as one forward stream
    const uint32_t nbk = gridDim.x;
    const uint32_t bucket = (nbk & 7) ? blockIdx.x : (blockIdx.x & 7) * (nbk >> 3) + (blockIdx.x >> 3);
    const bool bad = flag[0] != 0;                              // not grouped: the atomic kernel inserts, this one only clears
    if (tid == 0) { sh_tot = 0; sh_claimed = 0; sh_failed = 0; }
    __syncthreads();
    if (tid < sg.nseg && !bad) {
        const uint32_t a = first[(size_t)tid * nb + bucket], b = last[(size_t)tid * nb + bucket];
        sf[tid] = a; sl_[tid] = b;
        sp_lo[tid] = sg.lo[tid]; sp_hi[tid] = sg.hi[tid]; sp_cnt[tid] = sg.cnt[tid];
        if (b > a) atomicAdd(&sh_tot, b - a);
    }
    __syncthreads();
    if (tid == 0 && !bad) {                                     // where every segment's pairs start in the bucket's flat list
        uint32_t acc = 0;
        for (uint32_t s = 0; s < sg.nseg; ++s) { spre[s] = acc; acc += sl_[s] > sf[s] ? sl_[s] - sf[s] : 0u; }
        spre[sg.nseg] = acc;
    }
    const uint32_t tot = sh_tot;
    if (!FRESH && tot == 0) return;
    const uint64_t base = (uint64_t)bucket << t.bucket_bits;
    if (FRESH && tot == 0) {                                    // the deferred clear of an empty bucket
        for (uint32_t j = tid * 2; j < B; j += KM_MERGE_THREADS * 2) {
            *(ulonglong2 *)&t.lo[base + j] = make_ulonglong2(KDF_EMPTY, KDF_EMPTY);
            if (KW == 2) *(ulonglong2 *)&t.hi[base + j] = make_ulonglong2(KDF_EMPTY, KDF_EMPTY);
            *(uint2 *)&t.cnt[base + j] = make_uint2(0u, 0u);
        }
        return;
    }
    for (uint32_t j = tid * 2; j < B; j += KM_MERGE_THREADS * 2) {
        if (FRESH) {
            *(ulonglong2 *)&tlo[j] = make_ulonglong2(KDF_EMPTY, KDF_EMPTY);
            if (KW == 2) *(ulonglong2 *)&thi[j] = make_ulonglong2(KDF_EMPTY, KDF_EMPTY);
            *(uint2 *)&tcnt[j] = make_uint2(0u, 0u);
        } else {
            *(ulonglong2 *)&tlo[j] = *(const ulonglong2 *)&t.lo[base + j];
            if (KW == 2) *(ulonglong2 *)&thi[j] = *(const ulonglong2 *)&t.hi[base + j];
            *(uint2 *)&tcnt[j] = *(const uint2 *)&t.cnt[base + j];
        }
    }
    __syncthreads();
    uint32_t claimed = 0; bool failed = false;
    // The bucket's pairs of ALL segments as one flat list: a thread takes entries tid, tid + KM_MERGE_THREADS, ... and has KM_U of
    // them in flight.  (Segment after segment, as rounds 2-3 did it, is one global latency per segment for ~130 pairs
    // each at world 8: the owner merge ran at 37 G pairs/s against kernel C's 264 G entries/s.)
    constexpr int KM_U = 4;
    const uint32_t nseg = sg.nseg;
    for (uint32_t p0 = 0; p0 < tot; p0 += KM_MERGE_THREADS * KM_U) {  // (wave-uniform trip count: the wide retry loop needs whole waves)
        uint64_t lo[KM_U], hi[KW == 2 ? KM_U : 1]; uint32_t add[KM_U]; bool todo[KM_U];
#pragma unroll
        for (int u = 0; u < KM_U; ++u) {
            const uint32_t p = p0 + u * KM_MERGE_THREADS + tid;
            todo[u] = p < tot;
            lo[u] = KDF_EMPTY; add[u] = 0; if constexpr (KW == 2) hi[u] = 0;
            if (todo[u]) {
                uint32_t a = 0, b = nseg;                        // largest a with spre[a] <= p
                while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (spre[m] <= p) a = m; else b = m; }
                const uint32_t i = sf[a] + (p - spre[a]);
                lo[u] = sp_lo[a][i];
                if constexpr (KW == 2) hi[u] = sp_hi[a][i];
                const uint32_t *kc = sp_cnt[a];
                add[u] = kc ? kc[i] : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < KM_U; ++u) {
            if (p0 + u * KM_MERGE_THREADS >= tot) break;               // (uniform)
            const bool absent = KW == 1 ? lo[u] == KDF_EMPTY : hi[KW == 2 ? u : 0] == KDF_EMPTY;     // (tested on the key as it arrived)
            const uint64_t khi = KW == 2 ? hi[KW == 2 ? u : 0] & ~KDF_PENDING : 0;
            const uint64_t h = kdf_hash(lo[u], khi);             // the table holds stored forms
            const uint32_t sl = (uint32_t)kdf_home(t, h) & bmask;
            if (KW == 1) {
                if (todo[u] && !absent) km_probe_narrow(tlo, tcnt, bmask, h, add[u], sl, claimed, failed);
            } else {
                bool td = todo[u] && !absent;
                while (__any(td)) {
                    if (td) {
                        const int res = km_probe_wide_once(tlo, thi, tcnt, bmask, h, khi, add[u], sl, claimed);
                        if (res != 2) { td = false; if (res == 1) failed = true; }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) claimed += __shfl_xor(claimed, o);
    if ((tid & 63) == 0 && claimed) atomicAdd(&sh_claimed, claimed);
    if (failed) sh_failed = 1;
    __syncthreads();
    for (uint32_t j = tid * 2; j < B; j += KM_MERGE_THREADS * 2) {
        *(ulonglong2 *)&t.lo[base + j] = *(const ulonglong2 *)&tlo[j];
        if (KW == 2) *(ulonglong2 *)&t.hi[base + j] = *(const ulonglong2 *)&thi[j];
        *(uint2 *)&t.cnt[base + j] = *(const uint2 *)&tcnt[j];
    }
    if (tid == 0) {
        if (sh_claimed) atomicAdd(&ctl->distinct[(bucket % KDF_SHARDS) * 16], (unsigned long long)sh_claimed);
        if (sh_failed) atomicOr(&ctl->error, 1u);
    }
}

// the fallback: thread per pair through global atomics, only when km_bounds_kernel found a segment that is not grouped
template <int KW>
__global__ __launch_bounds__(256) void km_insert_guarded_kernel(KdfTable t, KmSegs sg, const uint32_t *__restrict__ flag, KdfCtl *ctl) {
    if (flag[0] == 0) return;
    const uint32_t seg = blockIdx.y, n = sg.n[seg];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if ((uint64_t)blockIdx.x * 256 >= n) return;
    uint32_t claimed = 0; bool full = false;
    {
        const uint64_t lo = i < n ? sg.lo[seg][i] : KDF_EMPTY, hi = (KW == 2 && i < n) ? sg.hi[seg][i] : (KW == 2 ? KDF_EMPTY : 0);
        const bool todo = i < n && (KW == 1 ? lo != KDF_EMPTY : hi != KDF_EMPTY);
        const uint32_t a = (todo && sg.cnt[seg]) ? sg.cnt[seg][i] : 0u;
        const uint64_t h = kdf_hash(lo, KW == 2 ? hi & ~KDF_PENDING : 0);
        const uint64_t slot = kdf_home(t, h);
        if constexpr (KW == 1) {
            if (todo && !kdf_add_narrow<true>(t, h, a, slot, t.lo[slot], claimed)) full = true;
        } else {
            if (!kdf_add_wide<true>(t, todo, h, hi & ~KDF_PENDING, a, slot, claimed)) full = true;
        }
    }
    if (full) atomicOr(&ctl->error, 1u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) claimed += __shfl_down(claimed, o);
    if ((threadIdx.x & 63) == 0 && claimed)
        atomicAdd(&ctl->distinct[((blockIdx.x * 4 + (threadIdx.x >> 6)) % KDF_SHARDS) * 16], (unsigned long long)claimed);
}
