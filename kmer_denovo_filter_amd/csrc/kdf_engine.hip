// kdf_engine.hip -- libkdf.so: HIP kernels (gfx950) + engine + the C ABI of
// include/kdf.h.  See DESIGN.md for the data layout and the roofline of each
// kernel.  No CPU fallback exists in this library: every entry point either
// runs on the GPU or returns an error.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "kdf.h"
#include "kdf_device.h"
#include "kdf_binned.h"
#include "kdf_merge.h"

// sorted export lives in kdf_sort.hip (rocPRIM radix sort)
int kdf_sort_pairs_device(uint64_t *d_lo, uint64_t *d_hi, uint32_t *d_cnt, uint64_t n,
                          hipStream_t stream, std::string &err);

// ===========================================================================
// kernels
// ===========================================================================

enum { MODE_INSERT = 0, MODE_FILTERED = 1, MODE_SCAN = 2 };

// One thread = one tile of 64 window starts.  Windows are processed in batches
// of 8: the 8 home-slot key loads are issued back to back before any of them is
// resolved, so a wave keeps 8 x 64 random HBM reads in flight.
template <int KW, int MODE>
__global__ __launch_bounds__(256) void kdf_stream_kernel(
    const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid,
    uint64_t tile0, uint64_t n_tiles, int k, KdfTable t, KdfCtl *ctl,
    uint64_t *__restrict__ hit_bits)
{
    const uint64_t tile = tile0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = tile < tile0 + n_tiles;
    uint32_t claimed = 0, nwin = 0;
    bool full = false;
    if (active) {
        const uint64_t m0 = invalid[tile], m1 = invalid[tile + 1];
        uint64_t valid = kdf_valid_windows(m0, m1, k);
        nwin = __popcll(valid);
        const bool sliced = MODE == MODE_INSERT && t.key_parts > 1;       // count only this key-space slice (KdfTable::key_parts)
        uint64_t hits = 0;
        if (valid) {
            constexpr int NW = KW == 1 ? 3 : 4;
            uint64_t w[NW];
#pragma unroll
            for (int i = 0; i < NW; ++i) w[i] = packed[tile * 2 + i];
            const uint64_t kmask = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1);
#pragma unroll
            for (int b = 0; b < KDF_TILE; b += 8) {
                if (((valid >> b) & 0xFF) == 0) continue;
                uint64_t klo[8], khi[8], slot[8], cur[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if constexpr (KW == 1) {
                        klo[u] = kdf_window_narrow((const uint64_t (&)[3])w, b + u, k, kmask);
                        khi[u] = 0;
                    } else {
                        kdf_window_wide((const uint64_t (&)[4])w, b + u, k, klo[u], khi[u]);
                    }
                    const uint64_t hsh = kdf_hash(klo[u], khi[u]);
                    klo[u] = hsh;                                  // from here on the key is its stored form (kdf_device.h)
                    slot[u] = kdf_home(t, hsh);
                    if (sliced && ((valid >> (b + u)) & 1) && kdf_slice(hsh, t.key_parts) != t.key_part) { valid &= ~(1ull << (b + u)); --nwin; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = (valid >> (b + u)) & 1;
                    if constexpr (KW == 1) cur[u] = ok ? t.lo[slot[u]] : 0;
                    else cur[u] = 0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = (valid >> (b + u)) & 1;
                    if constexpr (MODE == MODE_SCAN) {
                        if (!ok) continue;
                        uint64_t s = KW == 1 ? kdf_find_narrow(t, klo[u]) : kdf_find_wide(t, klo[u], khi[u]);
                        if (s != ~0ull && t.cnt[s] != 0) hits |= 1ull << (b + u);
                    } else if constexpr (KW == 1) {
                        if (!ok) continue;
                        if (!kdf_add_narrow<MODE == MODE_INSERT>(t, klo[u], 1u, slot[u], cur[u], claimed)) full = true;
                    } else {
                        // every lane that reached this batch calls in; idle lanes pass todo = false
                        if (!kdf_add_wide<MODE == MODE_INSERT>(t, ok, klo[u], khi[u], 1u, slot[u], claimed)) full = true;
                    }
                }
            }
        }
        if constexpr (MODE == MODE_SCAN) hit_bits[tile] = hits;
    }
    if (full) atomicOr(&ctl->error, 1u);
    // statistics: wave-reduce, one atomic per wave into a sharded counter
    uint32_t c = claimed, n = nwin;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_down(c, o); n += __shfl_down(n, o); }
    if ((threadIdx.x & 63) == 0) {
        const int shard = (blockIdx.x * 4 + (threadIdx.x >> 6)) % KDF_SHARDS;
        if (c) atomicAdd(&ctl->distinct[shard * 16], (unsigned long long)c);
        if (n) atomicAdd(&ctl->windows[shard * 16], (unsigned long long)n);
    }
}

// thread per key: insert with an explicit add (filter load: add = 0; rehash: add = count).
// stored != 0: klo[] already holds stored forms (the slots of a table that is being rehashed), else keys.
template <int KW>
__global__ __launch_bounds__(256) void kdf_insert_keys_kernel(
    const uint64_t *__restrict__ klo, const uint64_t *__restrict__ khi,
    const uint32_t *__restrict__ add, uint64_t n, KdfTable t, KdfCtl *ctl, int skip_empty, int stored)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t claimed = 0;
    bool full = false;
    {
        const uint64_t lo = i < n ? klo[i] : KDF_EMPTY, hi = (KW == 2 && i < n) ? khi[i] : (KW == 2 ? KDF_EMPTY : 0);
        const bool present = KW == 1 ? (lo != KDF_EMPTY) : (hi != KDF_EMPTY);
        const bool todo = i < n && (present || !skip_empty);
        const uint32_t a = (todo && add) ? add[i] : 0u;
        const uint64_t h = stored ? lo : kdf_hash(lo, (KW == 2 ? hi & ~KDF_PENDING : 0));
        const uint64_t slot = kdf_home(t, h);
        if constexpr (KW == 1) {
            if (todo && !kdf_add_narrow<true>(t, h, a, slot, t.lo[slot], claimed)) full = true;
        } else {
            if (!kdf_add_wide<true>(t, todo, h, hi & ~KDF_PENDING, a, slot, claimed)) full = true;
        }
    }
    if (full) atomicOr(&ctl->error, 1u);
    uint32_t c = claimed;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c)
        atomicAdd(&ctl->distinct[((blockIdx.x * 4 + (threadIdx.x >> 6)) % KDF_SHARDS) * 16],
                  (unsigned long long)c);
}

template <int KW>
__global__ __launch_bounds__(256) void kdf_query_kernel(
    const uint64_t *__restrict__ klo, const uint64_t *__restrict__ khi, uint64_t n,
    KdfTable t, uint32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = KW == 1 ? kdf_find_narrow(t, kdf_hash(klo[i], 0)) : kdf_find_wide(t, kdf_hash(klo[i], khi[i]), khi[i]);
    out[i] = (s == ~0ull) ? 0u : t.cnt[s];
}

// thread per key: the count of a STORED key becomes counts[i] (the merged counts of a sharded count --if go back into
// every rank's table); a key that is not stored raises the error flag
template <int KW>
__global__ __launch_bounds__(256) void kdf_set_counts_kernel(
    const uint64_t *__restrict__ klo, const uint64_t *__restrict__ khi, const uint32_t *__restrict__ counts, uint64_t n,
    KdfTable t, KdfCtl *ctl)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = KW == 1 ? kdf_find_narrow(t, kdf_hash(klo[i], 0)) : kdf_find_wide(t, kdf_hash(klo[i], khi[i]), khi[i]);
    if (s == ~0ull) atomicOr(&ctl->error, 2u);
    else t.cnt[s] = counts[i];
}

// dump -L: count / append entries with cnt >= min_count.  Each wave owns a
// chunk of EXPORT_ROWS x 64 consecutive slots: it counts its matches, reserves
// its output range with ONE atomic, then re-reads the (cache-resident) chunk and
// writes.  One same-address atomic per 2048 slots instead of one per 64.
#define KDF_EXPORT_ROWS 32
template <int KW, bool WRITE>
__global__ __launch_bounds__(256) void kdf_export_kernel(
    KdfTable t, uint32_t min_count, KdfCtl *ctl, uint64_t *__restrict__ olo,
    uint64_t *__restrict__ ohi, uint32_t *__restrict__ ocnt, uint64_t out_cap)
{
    const uint64_t cap = 1ull << t.log2cap;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t first = wave * (KDF_EXPORT_ROWS * 64);
    if (first >= cap) return;
    uint32_t mine = 0;
    if (!WRITE && min_count >= 1 && first + KDF_EXPORT_ROWS * 64 <= cap) {
        // count > 0 implies the slot is occupied: stream the counts array only, 16 B per lane
        const uint4 *c4 = (const uint4 *)(t.cnt + first);
#pragma unroll
        for (int r = 0; r < KDF_EXPORT_ROWS / 4; ++r) {
            const uint4 v = c4[r * 64 + lane];
            mine += (v.x >= min_count) + (v.y >= min_count) + (v.z >= min_count) + (v.w >= min_count);
        }
    } else
#pragma unroll 4
    for (int r = 0; r < KDF_EXPORT_ROWS; ++r) {
        const uint64_t i = first + (uint64_t)r * 64 + lane;
        if (i < cap) {
            if (min_count >= 1) {          // count > 0 implies the slot is occupied: counts array only
                mine += t.cnt[i] >= min_count ? 1u : 0u;
            } else {
                const bool occ = KW == 1 ? (t.lo[i] != KDF_EMPTY) : (t.hi[i] != KDF_EMPTY);
                mine += occ ? 1u : 0u;
            }
        }
    }
    uint32_t tot = mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (tot == 0) return;
    if (!WRITE) {      // counting pass: no positions needed, spread the adds over 64 lines
        if (lane == 0) atomicAdd(&ctl->tally[(uint32_t)(wave % KDF_SHARDS) * 16], (unsigned long long)tot);
        return;
    }
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&ctl->cursor, (unsigned long long)tot);
    base = __shfl(base, 0);
    for (int r = 0; r < KDF_EXPORT_ROWS; ++r) {
        const uint64_t i = first + (uint64_t)r * 64 + lane;
        bool keep = false; uint64_t lo = 0, hi = 0; uint32_t c = 0;
        if (i < cap) {
            lo = t.lo[i];
            if (KW == 2) hi = t.hi[i];
            const bool occ = KW == 1 ? (lo != KDF_EMPTY) : (hi != KDF_EMPTY);
            c = t.cnt[i];
            keep = occ && c >= min_count;
        }
        const unsigned long long b = __ballot(keep);
        if (keep) {
            const uint64_t pos = base + __popcll(b & ((1ull << lane) - 1));
            if (pos < out_cap) {
                olo[pos] = kdf_key_lo(lo, hi);                       // the key back from its stored form
                if (KW == 2 && ohi) ohi[pos] = hi;
                if (ocnt) ocnt[pos] = c;
            }
        }
        base += __popcll(b);
    }
}

// dump -L into device buffers, narrow keys, min_count >= 1: ONE read of the table.  A wave takes 16 rows of 64 slots;
// all 32 loads (keys + counts) are issued before anything is used, the kept entries are counted with ballots, the
// wave reserves its output range with one atomic and writes.  (The two-phase kernel above reads the counts twice and
// runs at half the memory rate; it stays for wide keys, min_count = 0 and the owner-grouped dump.)
#define KDF_EXPORT1_ROWS 16
#define KDF_EXPORT1_THREADS 1024
template <int KW>
__global__ __launch_bounds__(KDF_EXPORT1_THREADS) void kdf_export1_kernel(KdfTable t, uint32_t min_count, KdfCtl *ctl, uint64_t *__restrict__ olo,
                                                                         uint64_t *__restrict__ ohi, uint32_t *__restrict__ ocnt, uint64_t out_cap)
{
    // one atomic per WORKGROUP (16 K slots): a same-address returning atomic per wave was what bounded the dump
    __shared__ uint32_t wtot[KDF_EXPORT1_THREADS / 64];
    __shared__ unsigned long long wg_base;
    constexpr int ROWS = KW == 1 ? KDF_EXPORT1_ROWS : KDF_EXPORT1_ROWS / 2;     // wide keys: 5 registers per slot
    const uint64_t cap = 1ull << t.log2cap;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t wave = (uint64_t)blockIdx.x * (KDF_EXPORT1_THREADS / 64) + wv;
    const uint64_t first = wave * (ROWS * 64);
    uint64_t lo[ROWS], hi[KW == 2 ? ROWS : 1]; uint32_t c[ROWS];
    unsigned long long kb[ROWS]; uint32_t tot = 0;
    if (KW == 1 && min_count >= 1) {
        // a count > 0 implies an occupied slot: read the counts first and the keys only where they are kept (a `dump -L 3`
        // keeps a fifth of the slots: the memory system fetches the key sectors that hold at least one kept slot)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { const uint64_t i = first + (uint64_t)r * 64 + lane; c[r] = i < cap ? t.cnt[i] : 0u; }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint64_t i = first + (uint64_t)r * 64 + lane;
            const bool keep = c[r] >= min_count;
            lo[r] = keep ? t.lo[i] : KDF_EMPTY;
            kb[r] = __ballot(keep); tot += (uint32_t)__popcll(kb[r]);
        }
    } else {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint64_t i = first + (uint64_t)r * 64 + lane;
        const bool in = i < cap;
        c[r] = in ? t.cnt[i] : 0u;
        lo[r] = in ? t.lo[i] : KDF_EMPTY;
        if constexpr (KW == 2) hi[r] = in ? t.hi[i] : KDF_EMPTY;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const bool occ = KW == 1 ? (lo[r] != KDF_EMPTY) : (hi[r] != KDF_EMPTY);
        kb[r] = __ballot(c[r] >= min_count && occ); tot += (uint32_t)__popcll(kb[r]);
    }
    }
    if (lane == 0) wtot[wv] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int i = 0; i < KDF_EXPORT1_THREADS / 64; ++i) { const uint32_t v = wtot[i]; wtot[i] = acc; acc += v; }
        wg_base = acc ? atomicAdd(&ctl->cursor, (unsigned long long)acc) : 0ull;
    }
    __syncthreads();
    unsigned long long base = wg_base + wtot[wv];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if ((kb[r] >> lane) & 1) {
            const uint64_t pos = base + __popcll(kb[r] & ((1ull << lane) - 1));
            if (pos < out_cap) { olo[pos] = kdf_key_lo(lo[r], KW == 2 ? hi[r] : 0); if (KW == 2 && ohi) ohi[pos] = hi[r]; if (ocnt) ocnt[pos] = c[r]; }
        }
        base += __popcll(kb[r]);
    }
}

// ---------------------------------------------------------------------------
// count --if through a membership sieve.  In the parent-filter / VCF stages almost every window MISSES the filter
// (discovery/pipeline.py:377-443: a whole parent's reads against the child's candidate k-mers), so partitioning
// every window (binned path) or probing the hash table for every window (direct path) is wasted work.  The filter
// keys are folded into a blocked Bloom filter (two bits inside ONE 64-bit word per key; 16-32 bits per key, so it
// lives in L2 / the Infinity Cache); a window costs its canonical k-mer, one hash and ONE word load, and only the
// survivors (hits + a 0.4-1.4 % false-positive share) probe the table -- densely, 64 at a time from a wave-private
// LDS queue (ballot + mbcnt, no atomics, no barrier), so the rare slow path runs with every lane busy.
struct KdfSieve { const uint64_t *words; uint64_t wmask; };

__device__ __forceinline__ void kdf_sieve_bits(uint64_t hsh, uint64_t wmask, uint64_t &word, uint64_t &bits) {
    word = (hsh >> 12) & wmask;
    bits = (1ull << (hsh & 63)) | (1ull << ((hsh >> 6) & 63));
}

template <int KW>
__global__ __launch_bounds__(256) void kdf_sieve_build_kernel(const uint64_t *__restrict__ klo, const uint64_t *__restrict__ khi, uint64_t n,
                                                             uint64_t *__restrict__ words, uint64_t wmask) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t w, b;
    kdf_sieve_bits(kdf_hash(klo[i], KW == 2 ? khi[i] : 0), wmask, w, b);
    atomicOr((unsigned long long *)&words[w], (unsigned long long)b);
}


// the sieve over the keys the (hash-layout) table holds NOW: an index that was loaded with kdf_add_pairs or counted has
// none (kdf_load_filter builds one from the key list)
template <int KW>
__global__ __launch_bounds__(256) void kdf_sieve_from_table_kernel(KdfTable t, uint64_t *__restrict__ words, uint64_t wmask) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1ull << t.log2cap)) return;
    const uint64_t lo = t.lo[i], hi = KW == 2 ? t.hi[i] : 0;
    if ((KW == 1 ? lo : hi) == KDF_EMPTY) return;
    uint64_t w, b;
    kdf_sieve_bits(lo, wmask, w, b);                                // the slot holds the hash
    atomicOr((unsigned long long *)&words[w], (unsigned long long)b);
}

#define KDF_SV_WQ 128                     // queue entries per wave (drained 64 at a time)
#define KDF_SV_LDS_WORDS 8192             // a sieve of up to 64 KB is copied into LDS by every (persistent) workgroup:
                                          // filters of up to 64 K keys (VCF mode, Module 3) then cost no L2 request per window
// SCAN: the Module-3 probe (kdf_scan_reads_dev) through the same sieve: a survivor that is stored with count > 0 sets its
// window's bit in hit_bits (zeroed by the host first); nothing is counted.
template <int KW, bool IN_LDS = false, bool SCAN = false>
__global__ __launch_bounds__(KB_THREADS) void kdf_sieve_count_kernel(
    const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid, uint64_t n_tiles, int k,
    KdfTable t, KdfCtl *ctl, KdfSieve sv, uint32_t slabs_per_wg, unsigned long long *__restrict__ hit_bits = nullptr)
{
    constexpr int WPT = KbCfg<KW>::WPT, TPT = 64 / WPT;
    constexpr uint32_t TILES_PER_SLAB = KB_THREADS / TPT;
    __shared__ uint64_t qlo[(KB_THREADS / 64) * KDF_SV_WQ];
    __shared__ uint64_t qhi[KW == 2 ? (KB_THREADS / 64) * KDF_SV_WQ : 1];
    __shared__ uint64_t lsv[IN_LDS ? KDF_SV_LDS_WORDS : 1];
    __shared__ uint32_t qpos[SCAN ? (KB_THREADS / 64) * KDF_SV_WQ : 1];       // SCAN: stream position of the queued window (< 2^32: host-checked)
    uint32_t *wqpos = qpos + (SCAN ? (threadIdx.x >> 6) * KDF_SV_WQ : 0);
    if constexpr (IN_LDS) {
        for (uint32_t i = threadIdx.x; i <= (uint32_t)sv.wmask; i += KB_THREADS) lsv[i] = sv.words[i];
        __syncthreads();
    }
    const uint64_t *const svw = IN_LDS ? lsv : sv.words;
    uint64_t *wqlo = qlo + (threadIdx.x >> 6) * KDF_SV_WQ, *wqhi = qhi + (KW == 2 ? (threadIdx.x >> 6) * KDF_SV_WQ : 0);
    const int lane = threadIdx.x & 63;
    uint32_t wq_n = 0, nwin = 0, claimed = 0;
    bool full = false;
    // probe the table for 64 queued keys (or the rest): every lane has a key
    auto drain = [&](uint32_t from, uint32_t cnt) {
        const bool todo = (uint32_t)lane < cnt;
        const uint64_t klo = todo ? wqlo[from + lane] : 0, khi = (KW == 2 && todo) ? wqhi[from + lane] : 0;
        if constexpr (SCAN) {
            if (todo) {
                const uint64_t sl = KW == 1 ? kdf_find_narrow(t, klo) : kdf_find_wide(t, klo, khi);
                if (sl != ~0ull && t.cnt[sl] != 0) { const uint32_t p = wqpos[from + lane]; atomicOr(&hit_bits[p >> 6], 1ull << (p & 63)); }
            }
        } else {
            const uint64_t slot = kdf_home(t, klo);              // the queue holds stored forms
            if constexpr (KW == 1) { if (todo && !kdf_add_narrow<false>(t, klo, 1u, slot, t.lo[slot], claimed)) full = true; }
            else { if (!kdf_add_wide<false>(t, todo, klo, khi, 1u, slot, claimed)) full = true; }
        }
    };
    const uint64_t slab0 = (uint64_t)blockIdx.x * slabs_per_wg;
    for (uint32_t sl = 0; sl < slabs_per_wg; ++sl) {
        if ((slab0 + sl) * TILES_PER_SLAB >= n_tiles) break;
        const uint64_t tile = (slab0 + sl) * TILES_PER_SLAB + threadIdx.x / TPT;
        KbWindows<KW> win;
        win.load(packed, invalid, tile, n_tiles, threadIdx.x % TPT, k);
        nwin += __popc(win.valid);
        // eight sieve words in flight per lane; only the word and 12 hash bits are kept per window (the key of a
        // survivor is taken again from the registers that hold the stream), so eight waves fit a SIMD
        constexpr int HB = WPT < 8 ? WPT : 8;
#pragma unroll
        for (int u0 = 0; u0 < WPT; u0 += HB) {
            uint64_t w[HB]; uint32_t hb[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                uint64_t hsh, hi; win.stored(u0 + u, hsh, hi);
                hb[u] = (uint32_t)hsh & 0xFFFu;
                // only VALID windows ask for their sieve word: the kernel runs at the L2's request rate, and
                // 22 % of the window slots of 150 bp reads are invalid (N, read ends) -- 420 M of 2 031 M requests per parent
                w[u] = 0;
                if ((win.valid >> (u0 + u)) & 1) w[u] = svw[(hsh >> 12) & sv.wmask];
            }
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const bool ok = ((win.valid >> (u0 + u)) & 1) && (((w[u] >> (hb[u] & 63)) & (w[u] >> (hb[u] >> 6)) & 1ull) != 0);
                const unsigned long long mk = __ballot(ok);
                if (mk) {
                    const uint32_t at = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                    if (ok) {
                        uint64_t lo, hi; win.stored(u0 + u, lo, hi); wqlo[at] = lo; if constexpr (KW == 2) wqhi[at] = hi;
                        if constexpr (SCAN) wqpos[at] = (uint32_t)(tile * 64 + (threadIdx.x % TPT) * WPT + u0 + u);
                    }
                    wq_n += (uint32_t)__popcll(mk);
                    if (wq_n >= 64) { wq_n -= 64; drain(wq_n, 64); }        // (wave-uniform)
                }
            }
        }
    }
    if (wq_n) drain(0, wq_n);
    if (full) atomicOr(&ctl->error, 1u);
    uint32_t n = nwin;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if (!SCAN && lane == 0 && n) atomicAdd(&ctl->windows[((blockIdx.x * 16 + (threadIdx.x >> 6)) % KDF_SHARDS) * 16], (unsigned long long)n);
}

__global__ void kdf_ctl_reduce_kernel(KdfCtl *ctl, unsigned long long *out3) {
    // out3 = {distinct, windows, error}; single wave
    unsigned long long d = ctl->distinct[threadIdx.x * 16], w = ctl->windows[threadIdx.x * 16], y = ctl->tally[threadIdx.x * 16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { d += __shfl_down(d, o); w += __shfl_down(w, o); y += __shfl_down(y, o); }
    if (threadIdx.x == 0) { out3[0] = d; out3[1] = w; out3[2] = ctl->error; out3[3] = ctl->cursor + y; }
}

// ===========================================================================
// engine
// ===========================================================================

#define KDF_MERGE_MIN_PAIRS (1u << 16)
struct kdf_engine {
    int device = 0;
    int k = 0;
    int kw = 1;
    int n_cu = 256;               // compute units of the device (persistent-kernel grids)
    uint64_t dev_total_bytes = 0; // HBM of the device (sizes the entry ring's budget)
    KdfTable t{};                 // live table
    uint64_t cap = 0;
    KdfCtl *ctl = nullptr;        // device
    unsigned long long *d_out4 = nullptr;   // device scratch for ctl readback
    unsigned long long *h_out4 = nullptr;   // pinned host mirror
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    uint64_t distinct = 0;        // host mirror after the last sync
    uint64_t windows = 0;
    bool filter_mode = false;
    bool lazy_empty = false;      // logically empty, HBM slices not yet reset (see kdf_clear)
    // grow-only device staging for the host-buffer entry points
    void *stage[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t stage_bytes[4] = {0, 0, 0, 0};
    // ---- binned (LDS-bucket) path: scratch + options -------------------------------------------------------------------
    unsigned long long *kb_small = nullptr;   // totals[16]
    unsigned long long *kb_totals_host = nullptr;   // pinned [16]
    void *kb_buf[8] = {nullptr};                     // ring entries, tmp (slab-sorted pass), chunk_off, failed, off rows, pass planning, row tables
    size_t kb_bytes[8] = {0};
    KbPass *kb_pass = nullptr;                       // [KB_MAX_PASS] descriptors of the pending passes (device)
    // The entry ring: A0/A1/B append a partitioned pass per call; kernel C applies all pending passes at once when the
    // table is needed or the ring is full (kb_flush).  Reserved by upper bounds (one entry per stream position), so no
    // host round trip sits between the stages.
    uint64_t ring_entries = 0, ring_rows = 0;        // capacity: entries (8 B x kw each), rows (pieces)
    uint64_t ring_used = 0, rows_used = 0;           // reserved by the pending passes
    uint32_t n_pass = 0;
    uint64_t pend_positions = 0;                     // stream positions of the pending passes (upper bound of their entries)
    KbPlan pend_plan{};                              // geometry (c1, c2, key slice) the pending passes were partitioned with
    bool pend_filtered = false;                      // the pending passes are count --if passes
    uint64_t stat_flushes = 0;
    double grow_ratio = 0.0;                         // new distinct keys per counted window at the last flush (0: unknown)
    uint64_t dens_windows = 0, dens_positions = 0;   // valid windows / stream positions of every pass this engine has flushed: sizes the piece groups
    // L1: small insert batches are concatenated (packed) in a pending stream first; the partition runs over ~2^30 positions
    uint64_t *l1_packed = nullptr, *l1_mask = nullptr;
    uint64_t l1_cap_tiles = 0, l1_tiles = 0;
    uint32_t opt_key_parts = 0, opt_key_part = 0;    // count only one slice of the key space (KdfTable::key_parts)
    uint64_t opt_binned_min_positions = 1ull << 22;  // fewer pending positions at flush time use the direct global-table kernels
    uint64_t opt_binned_bytes_per_position = 70;     // ... and so does a flush of fewer than table_bytes / 70 positions (use_binned)
    uint64_t opt_binned_max_positions = 1ull << 31;  // longer streams are partitioned in several passes
    uint32_t opt_binned_filtered_min_log2cap = 23;   // count --if goes binned from 2^23 slots (measured crossover)
    uint32_t opt_big_bucket_log2cap = 32;            // tables from 2^32 slots on have buckets of twice the slots (KDF_BIG_BUCKET_LOG2CAP / option big_bucket_log2cap)
    uint64_t opt_merge_min_pairs = KDF_MERGE_MIN_PAIRS;   // below this many pairs a merge goes straight to the atomic insert (tests lower it)
    uint32_t opt_hash_shift = 0;                     // KdfTable::hshift of the tables this engine creates (owner tables)
    int opt_force_path = 0;                          // 0 auto, 1 direct, 2 binned, 4 sieve only (count --if)
    int opt_defer = 1;                               // 1: kernel C is deferred over the pending passes; 0: every count call ends with a flush
    uint64_t opt_defer_max_bytes = 0;                // budget of the entry ring (0: 40 % of the device's memory)
    // 1: a dump (min_count >= 1, caller-sized device buffers) asked for while passes are pending is written by the flush itself
    // -- kernel C holds every bucket anyway (kb_bucket_kernel<.., DUMP>).  Off by default: at bench size the step is 13.1 ->
    // 12.6 ms (the 1.4 ms table pass of the dump saved), but kernel C takes 0.9 ms longer for it (two barriers and a global
    // reservation per bucket); option "fused_dump" / env KDF_FUSED_DUMP=1
    int opt_fused_dump = [] { const char *e = getenv("KDF_FUSED_DUMP"); return e ? atoi(e) != 0 : 0; }();
    // the request a dump hands to the LAST flush before it (fuse_min > 0), and what came of it
    uint32_t fuse_min = 0; uint64_t *fuse_lo = nullptr, *fuse_hi = nullptr; uint32_t *fuse_cnt = nullptr; uint64_t fuse_cap = 0;
    bool fuse_done = false; uint64_t fuse_n = 0;
    uint64_t stat_fused_dumps = 0;
    uint64_t opt_l1_positions = 1ull << 30;          // pending-stream size from which it is partitioned
    uint64_t opt_l1_direct_positions = 1ull << 28;   // batches from this size on are partitioned where they lie (no copy)
    // double-buffered feeding (kdf_upload_reads_async / kdf_count_uploaded): two device staging slots filled on a copy
    // stream of their own, so the H2D copy of batch i + 1 runs under the count of batch i
    void *up_buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; size_t up_bytes[2][2] = {{0, 0}, {0, 0}};
    uint64_t up_n[2] = {0, 0}; bool up_valid[2] = {false, false};
    hipStream_t copy_stream = nullptr; hipEvent_t up_done[2] = {nullptr, nullptr}, use_done[2] = {nullptr, nullptr};
    uint64_t stat_heavy_buckets = 0;
    void *kb_heavy = nullptr;                        // heavy buckets of skewed flushes (kdf_binned.h kb_heavy_slice_kernel)
    void *merge_buf = nullptr; size_t merge_bytes = 0;   // kdf_merge.h: block counts / offsets of the ordered dump, bucket ranges of a merge
    uint32_t merge_flag_host = 0;
    int last_merge_path = 0;                         // 0 none yet, 1 LDS bucket merge launched, 2 plain atomic insert
    int opt_sieve_bits = 0;                          // sieve bits per filter key (0: 32 up to 2^20 keys, 16 beyond)
    bool merge_attrs_set[3] = {false, false, false};  // km_merge_kernel's LDS limit raised (big buckets)
    bool attrs_set[4] = {false, false, false, false};   // hipFuncSetAttribute done (per key width)
    int last_path = 0;                               // count path of the last count call: 0 direct, 1 binned, 3 sieve
    uint64_t *sieve = nullptr;                       // blocked Bloom filter over the filter keys (count --if)
    uint64_t sieve_words = 0, sieve_alloc = 0;
    bool sieve_valid = false;
    uint32_t opt_debug_flags = 0;                    // experiments only (KbPlan::dbg)
    uint64_t stat_binned_passes = 0, stat_replayed_buckets = 0;
    // optional HIP-event timing of the dominant (stream) kernel
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;   // pending start/stop pairs
    std::vector<uint64_t> prof_tiles;
    double prof_ms = 0.0;
    uint64_t prof_launches = 0, prof_positions = 0;
    // binned path: events around each stage (A0 hist, A1 scatter, B finesort | C bucket)
    std::vector<std::vector<hipEvent_t>> prof_stage_ev;   // 4 events: a partition (A0, A1, B); 2 events: a flush (C)
    double prof_stage_ms[4] = {0, 0, 0, 0};
    uint64_t prof_stage_passes = 0;
    std::string err;
};

static thread_local std::string g_err;

static int fail(kdf_engine *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_err = buf;
    return code;
}

#define HIPCHK(h, call)                                                                 \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(h, e_ == hipErrorOutOfMemory ? KDF_ERR_NOMEM : KDF_ERR_HIP,     \
                        "%s failed: %s", #call, hipGetErrorString(e_));                 \
    } while (0)

static uint32_t log2ceil(uint64_t x) { uint32_t l = 0; while ((1ull << l) < x) ++l; return l; }

// slots needed so that n keys sit at load <= 0.5 (narrow: 8192-slot buckets,
// wide: 4096-slot buckets = what one workgroup can hold in LDS)
static uint32_t cap_log2_for(uint64_t n_keys) {
    uint32_t l = log2ceil(std::max<uint64_t>(n_keys, 1) * 2);
    return std::max<uint32_t>(l, 10);
}

static int table_alloc(kdf_engine *h, uint32_t log2cap, KdfTable &t, bool clear = true) {
    const uint64_t cap = 1ull << log2cap;
    t = KdfTable{};
    t.log2cap = log2cap;
    // 48 / 40 KB of LDS per bucket; tables of 2^big_bucket_log2cap slots and more: twice that (kdf_binned.h: KB_C_CT_BIG)
    t.bucket_bits = std::min<uint32_t>(log2cap, KB_BB_SMALL(h->kw) + (log2cap >= h->opt_big_bucket_log2cap ? 1u : 0u));
    t.hshift = h->opt_hash_shift;
    {
        hipError_t e = hipMalloc((void **)&t.lo, cap * 8);
        if (e == hipSuccess && h->kw == 2) e = hipMalloc((void **)&t.hi, cap * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&t.cnt, cap * 4);
        if (e != hipSuccess) {
            if (t.lo) (void)hipFree(t.lo);
            if (t.hi) (void)hipFree(t.hi);
            t = KdfTable{};
            (void)hipGetLastError();
            return fail(h, e == hipErrorOutOfMemory ? KDF_ERR_NOMEM : KDF_ERR_HIP,
                        "a table of 2^%u slots (%.1f GB) does not fit the device (%s): count the sample in key-space "
                        "slices (option key_parts / key_part; KDF_KEY_PARTS for the child count)",
                        log2cap, (double)cap * (8.0 * h->kw + 4.0) / 1e9, hipGetErrorString(e));
        }
    }
    if (!clear) return KDF_OK;                       // the caller keeps the engine's deferred-clear flag set
    HIPCHK(h, hipMemsetAsync(t.lo, 0xFF, cap * 8, h->stream));
    if (h->kw == 2) HIPCHK(h, hipMemsetAsync(t.hi, 0xFF, cap * 8, h->stream));
    HIPCHK(h, hipMemsetAsync(t.cnt, 0, cap * 4, h->stream));
    return KDF_OK;
}
static void table_free(KdfTable &t) {
    if (t.lo) (void)hipFree(t.lo);
    if (t.hi) (void)hipFree(t.hi);
    if (t.cnt) (void)hipFree(t.cnt);
    t = KdfTable{};
}

// kdf_clear defers the 12 B/slot memset: a binned insert into an empty table
// rewrites every bucket anyway.  Everything else calls this first.
static int materialize(kdf_engine *h) {
    if (!h->lazy_empty) return KDF_OK;
    HIPCHK(h, hipMemsetAsync(h->t.lo, 0xFF, h->cap * 8, h->stream));
    if (h->kw == 2) HIPCHK(h, hipMemsetAsync(h->t.hi, 0xFF, h->cap * 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->t.cnt, 0, h->cap * 4, h->stream));
    h->lazy_empty = false;
    return KDF_OK;
}

static int stage_reserve(kdf_engine *h, int i, size_t bytes) {
    if (h->stage_bytes[i] >= bytes) return KDF_OK;
    if (h->stage[i]) { (void)hipStreamSynchronize(h->stream); (void)hipFree(h->stage[i]); h->stage[i] = nullptr; h->stage_bytes[i] = 0; }
    size_t want = bytes + bytes / 8 + 4096;
    HIPCHK(h, hipMalloc(&h->stage[i], want));
    h->stage_bytes[i] = want;
    return KDF_OK;
}

// read distinct / windows / error from the control block (synchronises)
static int ctl_sync(kdf_engine *h, bool *table_full, uint64_t *cursor = nullptr) {
    hipLaunchKernelGGL(kdf_ctl_reduce_kernel, dim3(1), dim3(64), 0, h->stream, h->ctl, h->d_out4);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->h_out4, h->d_out4, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->distinct = h->h_out4[0];
    h->windows = h->h_out4[1];
    if (table_full) *table_full = h->h_out4[2] != 0;
    if (cursor) *cursor = h->h_out4[3];
    return KDF_OK;
}

static int ctl_reset(kdf_engine *h, bool keep_windows) {
    if (keep_windows) {
        HIPCHK(h, hipMemsetAsync(h->ctl->distinct, 0, sizeof(h->ctl->distinct), h->stream));
        HIPCHK(h, hipMemsetAsync(h->ctl->tally, 0, sizeof(h->ctl->tally) + 16, h->stream));
    } else {
        HIPCHK(h, hipMemsetAsync(h->ctl, 0, sizeof(KdfCtl), h->stream));
    }
    return KDF_OK;
}

template <typename F>
static int by_width(kdf_engine *h, F &&f) { return h->kw == 1 ? f(std::integral_constant<int, 1>{}) : f(std::integral_constant<int, 2>{}); }

// rehash the live table into one with 2^new_log2 slots
static int table_rehash(kdf_engine *h, uint32_t new_log2) {
    // (the window counter lives on the device between synchronisations: pending partition passes have added to it)
    { int rc0 = ctl_sync(h, nullptr); if (rc0) return rc0; }
    if (h->lazy_empty || h->distinct == 0) {              // nothing to carry over: a new table (a lazily cleared one stays so)
        KdfTable nt0;
        int rc0 = table_alloc(h, new_log2, nt0, !h->lazy_empty);
        if (rc0) return rc0;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        table_free(h->t);
        h->t = nt0; h->t.key_parts = h->opt_key_parts; h->t.key_part = h->opt_key_part;
        h->cap = 1ull << new_log2;
        return KDF_OK;
    }
    KdfTable nt;
    int rc = table_alloc(h, new_log2, nt);
    if (rc) { table_free(nt); return rc; }
    const uint64_t old_cap = h->cap;
    const uint64_t windows = h->windows;
    rc = ctl_reset(h, false);
    if (rc) { table_free(nt); return rc; }
    // (a launch holds fewer than 2^32 threads: tables of 2^32 slots and more go in pieces of 2^30 slots)
    for (uint64_t off = 0; off < old_cap; off += 1ull << 30) {
        const uint64_t n = std::min<uint64_t>(1ull << 30, old_cap - off);
        const unsigned blocks = (unsigned)((n + 255) / 256);
        if (h->kw == 1)
            hipLaunchKernelGGL(kdf_insert_keys_kernel<1>, dim3(blocks), dim3(256), 0, h->stream,
                               (const uint64_t *)h->t.lo + off, (const uint64_t *)nullptr, (const uint32_t *)h->t.cnt + off, n, nt, h->ctl, 1, 1);
        else
            hipLaunchKernelGGL(kdf_insert_keys_kernel<2>, dim3(blocks), dim3(256), 0, h->stream,
                               (const uint64_t *)h->t.lo + off, (const uint64_t *)h->t.hi + off, (const uint32_t *)h->t.cnt + off, n, nt, h->ctl, 1, 1);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { table_free(nt); return fail(h, KDF_ERR_HIP, "rehash launch failed: %s", hipGetErrorString(e)); }
    bool full = false;
    rc = ctl_sync(h, &full);
    if (rc) { table_free(nt); return rc; }
    if (full) { table_free(nt); return fail(h, KDF_ERR_TABLE_FULL, "rehash: bucket overflow at 2^%u slots", new_log2); }
    table_free(h->t);
    h->t = nt;
    h->t.key_parts = h->opt_key_parts; h->t.key_part = h->opt_key_part;
    h->cap = 1ull << new_log2;
    // restore the window counter (host-side accumulation)
    h->windows = windows;
    HIPCHK(h, hipMemcpyAsync(&h->ctl->windows[0], &h->windows, 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return KDF_OK;
}

template <int MODE>
static void launch_stream(kdf_engine *h, const uint64_t *d_packed, const uint64_t *d_invalid,
                          uint64_t tile0, uint64_t n_tiles, uint64_t *d_hits) {
    const unsigned blocks = (unsigned)((n_tiles + 255) / 256);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, h->stream);
    }
    if (h->kw == 1)
        hipLaunchKernelGGL((kdf_stream_kernel<1, MODE>), dim3(blocks), dim3(256), 0, h->stream,
                           d_packed, d_invalid, tile0, n_tiles, h->k, h->t, h->ctl, d_hits);
    else
        hipLaunchKernelGGL((kdf_stream_kernel<2, MODE>), dim3(blocks), dim3(256), 0, h->stream,
                           d_packed, d_invalid, tile0, n_tiles, h->k, h->t, h->ctl, d_hits);
    if (h->prof) {
        (void)hipEventRecord(e1, h->stream);
        h->prof_ev.emplace_back(e0, e1);
        h->prof_tiles.push_back(n_tiles);
    }
}

// fold finished event pairs into the running totals (synchronises on them)
static void prof_collect(kdf_engine *h) {
    for (size_t i = 0; i < h->prof_ev.size(); ++i) {
        float ms = 0.f;
        (void)hipEventSynchronize(h->prof_ev[i].second);
        if (hipEventElapsedTime(&ms, h->prof_ev[i].first, h->prof_ev[i].second) == hipSuccess) {
            h->prof_ms += ms; h->prof_positions += h->prof_tiles[i] * KDF_TILE;
            if (h->prof_tiles[i]) h->prof_launches++;                // (a flush has no positions of its own: its time belongs to the passes it applies)
        }
        (void)hipEventDestroy(h->prof_ev[i].first); (void)hipEventDestroy(h->prof_ev[i].second);
    }
    h->prof_ev.clear(); h->prof_tiles.clear();
    for (auto &ev : h->prof_stage_ev) {
        if (ev.size() == 4) {                                     // a partition pass: A0, A1, B
            (void)hipEventSynchronize(ev[3]);
            bool ok = true; float ms[3];
            for (int i = 0; i < 3; ++i) ok = ok && hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]) == hipSuccess;
            if (ok) { for (int i = 0; i < 3; ++i) h->prof_stage_ms[i] += ms[i]; h->prof_stage_passes++; }
        } else if (ev.size() == 2) {                              // a flush: kernel C over the pending passes
            (void)hipEventSynchronize(ev[1]);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) h->prof_stage_ms[3] += ms;
        }
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
    h->prof_stage_ev.clear();
}

// ---------------------------------------------------------------------------
// binned path (kdf_binned.h): partition passes into the entry ring, deferred kernel C

static int kb_reserve(kdf_engine *h, int i, size_t bytes, bool exact = false) {
    if (h->kb_bytes[i] >= bytes) return KDF_OK;
    if (h->kb_buf[i]) { (void)hipStreamSynchronize(h->stream); (void)hipFree(h->kb_buf[i]); h->kb_buf[i] = nullptr; h->kb_bytes[i] = 0; }
    const size_t want = exact ? bytes + 4096 : bytes + bytes / 16 + 4096;
    HIPCHK(h, hipMalloc(&h->kb_buf[i], want));
    h->kb_bytes[i] = want;
    return KDF_OK;
}

// the partition geometry for a table: coarse / fine bits; sub_bits = what the bucket kernel resolves itself
static KbPlan kb_make_plan(const kdf_engine *h, const KdfTable &t) {
    KbPlan p{};
    p.log2cap = t.log2cap; p.bucket_bits = t.bucket_bits;
    const uint32_t nb_bits = t.log2cap - t.bucket_bits;
    // 8 + 8 bits while that resolves the table; then the FINE radix widens first (8 + 9: the piece sort gathers 512-byte
    // runs instead of 256-byte ones and ranks into 512 bins; the bucket kernel pays with twice the runs of half the
    // length -- measured at bench size, 9 + 8 / 8 + 9 / 10 + 7: piece sort 4.74 / 4.21 / 5.88 ms, bucket kernel 4.40 / 4.79 / 4.20,
    // pass 12.35 / 12.16 / 13.60), then the coarse one (9 + 9, 10 + 9)
    p.c2 = std::min<uint32_t>(KB_F_BITS, nb_bits);
    p.c1 = std::min<uint32_t>(8, nb_bits - p.c2);
    if (p.c1 + p.c2 < nb_bits) p.c2 = std::min<uint32_t>(9, nb_bits - p.c1);               // fine runs halve: still >= 240 B
    if (p.c1 + p.c2 < nb_bits) p.c1 = std::min<uint32_t>(KB_C1_MAX, nb_bits - p.c2);      // coarse runs halve
    // (beyond 2^19 buckets kernel C reads every run once per sub-bucket: 10 fine bits -- 16-entry runs -- measured worse,
    // 9.65 against 8.86 ms per 10 M reads into 2^32 slots)
    if (const char *ev = getenv("KDF_C1")) {                       // (experiments: another split of the same bits)
        const uint32_t c1 = (uint32_t)atoi(ev);
        if (c1 <= KB_C1_MAX && c1 <= nb_bits && nb_bits - c1 <= KB_F_BITS_MAX) { p.c1 = c1; p.c2 = nb_bits - c1; }
    }
    p.sub_bits = nb_bits - p.c1 - p.c2;
    p.off_stride = (1u << p.c2) + 1;
    // Slabs per group: a piece (bin x group) is ~0.93 CHUNK entries.  Windows per stream position: what this engine has
    // seen so far (150 bp reads at k = 31: 0.78), 1 before its first flush; a pass that turns out denser only gets some
    // pairs of two pieces (kb_piecesort_more_kernel).
    const uint64_t chunk = h->kw == 1 ? KbCfg<1>::CHUNK : KbCfg<2>::CHUNK, slab = h->kw == 1 ? KbCfg<1>::SLAB : KbCfg<2>::SLAB;
    double dens = 1.0;
    if (h->dens_positions >= (1u << 20)) dens = std::min(1.0, std::max(0.05, 1.03 * (double)h->dens_windows / (double)h->dens_positions));
    // (0.98 of a piece's capacity, on a density taken 3 % high: pieces come out 95 % full -- 6 sigma below 16 K entries on
    // uniform input; measured 0.93 / 0.97 / 1.0 / 1.03: pass 12.48 / 12.29 / 12.25 / 13.12 ms, the last with overflow pieces)
    static const double fill = [] { const char *e = getenv("KDF_PIECE_FILL"); const double v = e ? atof(e) : 0.0; return v > 0.1 && v <= 1.2 ? v : 0.98; }();
    p.group = (uint32_t)std::min<double>(KB_G_MAX, std::max<double>(1.0, fill * (double)chunk * (double)((uint64_t)1 << p.c1) / ((double)slab * dens)));
    return p;
}

template <int KW>
static int kb_set_lds_attrs(kdf_engine *h, size_t a, size_t b, size_t c, size_t hv) {
    HIPCHK(h, hipFuncSetAttribute((const void *)(kb_slabsort_kernel<KW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)a));
    HIPCHK(h, hipFuncSetAttribute((const void *)(kb_slabsort_kernel<KW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)a));
    HIPCHK(h, hipFuncSetAttribute((const void *)kb_piecesort_kernel<KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b));
    HIPCHK(h, hipFuncSetAttribute((const void *)kb_piecesort_more_kernel<KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b));
    HIPCHK(h, hipFuncSetAttribute((const void *)kb_piecesort_pipe_kernel<KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b));
    const size_t cbig = KB_C_LDS(KW, KB_BB_SMALL(KW) + 1);
#define KB_SETV(V) \
    HIPCHK(h, hipFuncSetAttribute((const void *)(kb_bucket_kernel<KW, KB_MODE_INSERT, V, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c)); \
    HIPCHK(h, hipFuncSetAttribute((const void *)(kb_bucket_kernel<KW, KB_MODE_FILTERED, V, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c)); \
    HIPCHK(h, hipFuncSetAttribute((const void *)(kb_bucket_kernel<KW, KB_MODE_INSERT, V, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cbig)); \
    HIPCHK(h, hipFuncSetAttribute((const void *)(kb_bucket_kernel<KW, KB_MODE_FILTERED, V, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cbig)); \
    HIPCHK(h, hipFuncSetAttribute((const void *)(kb_bucket_kernel<KW, KB_MODE_INSERT, V, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c)); \
    HIPCHK(h, hipFuncSetAttribute((const void *)(kb_bucket_kernel<KW, KB_MODE_INSERT, V, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cbig));
    KB_SETV(1)
    KB_SETV(2)
#undef KB_SETV
    HIPCHK(h, hipFuncSetAttribute((const void *)kb_heavy_slice_kernel<KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hv));
    HIPCHK(h, hipFuncSetAttribute((const void *)kb_heavy_combine_kernel<KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hv));
    HIPCHK(h, hipFuncSetAttribute((const void *)kb_heavy_filtered_kernel<KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hv));
    return KDF_OK;
}

static int table_rehash(kdf_engine *h, uint32_t new_log2);

// the small device arrays of the binned path, allocated once per engine; fills the pointers of `s` from the engine's buffers
static int kb_scratch(kdf_engine *h, KbScratch &s) {
    if (!h->kb_small) {
        HIPCHK(h, hipMalloc((void **)&h->kb_small, (16 + 64) * 8));
        HIPCHK(h, hipMemsetAsync(h->kb_small, 0, (16 + 64) * 8, h->stream));
        HIPCHK(h, hipHostMalloc((void **)&h->kb_totals_host, 16 * 8));
        HIPCHK(h, hipMalloc((void **)&h->kb_pass, sizeof(KbPass) * KB_MAX_PASS));
    }
    const size_t hv_pairs = (size_t)KB_HV_MAX * KB_HV_SLICES << (KB_BB_SMALL(h->kw) + 1);   // (room for the buckets of big tables)
    const size_t hv_pair_bytes = 8 * (size_t)h->kw + 4;
    if (!h->kb_heavy) {                                           // heavy buckets of skewed flushes (kb_heavy_slice_kernel): ~200 MB, once
        HIPCHK(h, hipMalloc((void **)&h->kb_heavy, hv_pairs * hv_pair_bytes + (4 + 3 * KB_HV_MAX) * 4));
        HIPCHK(h, hipMemsetAsync((char *)h->kb_heavy + hv_pairs * hv_pair_bytes, 0, (4 + 3 * KB_HV_MAX) * 4, h->stream));
    }
    s = KbScratch{};
    s.totals = h->kb_small; s.trash = (uint64_t *)(h->kb_small + 16);
    s.pass = h->kb_pass;
    if (h->kb_heavy) {
        s.hv_key = (uint64_t *)h->kb_heavy;
        s.hv_khi = h->kw == 2 ? s.hv_key + hv_pairs : nullptr;
        s.hv_cnt = (uint32_t *)(s.hv_key + hv_pairs * h->kw);
        s.hv_ctr = s.hv_cnt + hv_pairs; s.hv_bucket = s.hv_ctr + 4; s.hv_n = s.hv_bucket + KB_HV_MAX; s.hv_failed = s.hv_n + KB_HV_MAX;
    }
    s.ent = (uint64_t *)h->kb_buf[0]; s.tmp = (uint64_t *)h->kb_buf[1];
    s.chunk_off = (uint32_t *)h->kb_buf[2]; s.failed = (uint32_t *)h->kb_buf[3];
    s.off = (uint16_t *)h->kb_buf[4];
    // row tables of the ring: row_ent u64 | row_len u32, ring_rows of each
    s.row_ent = (unsigned long long *)h->kb_buf[6];
    s.row_len = (uint32_t *)(s.row_ent + h->ring_rows);
    return KDF_OK;
}

// the ring is empty again: nothing pending, the flush-wide counters zeroed
static int kb_ring_reset(kdf_engine *h) {
    h->n_pass = 0; h->ring_used = 0; h->rows_used = 0; h->pend_positions = 0;
    if (h->kb_small) HIPCHK(h, hipMemsetAsync(h->kb_small, 0, 16 * 8, h->stream));
    return KDF_OK;
}

static int kb_flush_ring(kdf_engine *h);

// Room for a pass of need_e entries (upper bound: its stream positions) in need_r pieces.  A full ring is applied to the
// table first; a ring that was too small for the pending passes plus this one grows (doubling, up to the budget) while
// it is empty.
static int kb_ring_make_room(kdf_engine *h, uint64_t need_e, uint64_t need_r, uint32_t off_stride) {
    int rc;
    const uint64_t rows_cap = std::min<uint64_t>(h->ring_rows, h->kb_bytes[2] / ((uint64_t)off_stride * 4));
    bool forced = false;
    if (h->n_pass >= KB_MAX_PASS || h->ring_used + need_e > h->ring_entries || h->rows_used + need_r > rows_cap) {
        forced = h->n_pass > 0;
        if (h->n_pass && (rc = kb_flush_ring(h))) return rc;
    }
    const uint64_t esz = 8ull * h->kw;
    const uint64_t budget = h->opt_defer_max_bytes ? h->opt_defer_max_bytes : h->dev_total_bytes / 5 * 2;
    uint64_t want_e = h->ring_entries;
    // (at least 128 / 256 MB: small batches never force a flush; with deferral on, room for one more pass like this one)
    if (need_e > want_e) want_e = std::max<uint64_t>(h->opt_defer ? std::max<uint64_t>(need_e, std::min<uint64_t>(2 * need_e, budget / esz)) : need_e, 1ull << 24);
    if (forced && h->opt_defer && h->ring_entries * esz < budget)
        want_e = std::max<uint64_t>(want_e, std::min<uint64_t>(2 * h->ring_entries, std::max<uint64_t>(budget / esz, need_e)));
    // pieces: in proportion to the entries (passes of one sample look alike), and never fewer than this pass needs
    const uint64_t want_r = std::max<uint64_t>(need_r, (uint64_t)((double)need_r * ((double)want_e / (double)need_e))) + 4096;
    if (want_e > h->ring_entries || need_r > rows_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if ((rc = kb_reserve(h, 0, want_e * esz, true))) return rc;
        if ((rc = kb_reserve(h, 2, want_r * (uint64_t)off_stride * 4, true))) return rc;
        if ((rc = kb_reserve(h, 6, want_r * 12, true))) return rc;
        h->ring_entries = std::max(h->ring_entries, want_e); h->ring_rows = std::max(h->ring_rows, want_r);
        // (kb_reserve only ever grows a buffer: the row tables are laid out for ring_rows rows)
        if (h->kb_bytes[6] < h->ring_rows * 12) return fail(h, KDF_ERR_STATE, "entry ring: row tables out of step");
    }
    return KDF_OK;
}

// ONE partition pass (A, P, B) of a device-resident stream of at most opt_binned_max_positions positions into the ring
template <int KW>
static int kb_partition(kdf_engine *h, const uint64_t *d_packed, const uint64_t *d_invalid, uint64_t n_bases, bool filtered) {
    constexpr int WPT = KbCfg<KW>::WPT, TPT = 64 / WPT, CHUNK = KbCfg<KW>::CHUNK, SLAB = KbCfg<KW>::SLAB;
    constexpr uint32_t TILES_PER_SLAB = KB_A_THREADS / TPT;
    const uint64_t n_tiles = (n_bases + KDF_TILE - 1) / KDF_TILE;
    if (n_tiles == 0) return KDF_OK;
    int rc;
    // pending passes must share their geometry, key slice and mode (kdf_set_option / a mode change flush first)
    if (h->n_pass && h->pend_filtered != filtered && (rc = kb_flush_ring(h))) return rc;
    KbPlan plan = h->n_pass ? h->pend_plan : kb_make_plan(h, h->t);
    if (h->n_pass == 0) { plan.key_parts = filtered ? 0 : h->t.key_parts; plan.key_part = h->t.key_part; }
    const int nbins = 1 << plan.c1;
    const uint64_t n_slabs = (n_tiles + TILES_PER_SLAB - 1) / TILES_PER_SLAB;
    const uint64_t n_groups = (n_slabs + plan.group - 1) / plan.group;
    const uint64_t n_entries_max = n_tiles * KDF_TILE;
    const uint64_t n_rows_max = n_groups * nbins + n_entries_max / CHUNK + 1;
    if (n_rows_max >= (1ull << 32) || n_slabs >= (1ull << 31)) return fail(h, KDF_ERR_INVALID, "binned pass: too many positions for one pass");
    if ((rc = kb_ring_make_room(h, n_entries_max, n_rows_max, plan.off_stride))) return rc;
    if (h->n_pass == 0) { h->pend_plan = plan; h->pend_filtered = filtered; }
    plan.dbg = h->opt_debug_flags;
    plan.n_slabs = (uint32_t)n_slabs; plan.n_groups = (uint32_t)n_groups;
    const int nb1 = 1 << KB_C1_MAX;
    const size_t lds_a = (size_t)(SLAB + 64) * 8 * KW + (size_t)(2 * (nb1 + 96)) * 4;
    const size_t lds_b = (size_t)CHUNK * 8 * KW + (size_t)(2 * KB_F + 32) * 4 + (size_t)KB_G_MAX * 12 + 32;
    if (!h->attrs_set[KW]) {                                   // once per engine
        const size_t lds_c = KB_C_LDS(KW, KB_BB_SMALL(KW));
        if ((rc = kb_set_lds_attrs<KW>(h, lds_a, lds_b, lds_c, ((size_t)(8 * KW + 4) << (KB_BB_SMALL(KW) + 1)) + KB_RI_LDS_BYTES))) return rc;
        h->attrs_set[KW] = true;
    }
    // the pass's own buffers: slab-sorted entries, offset rows, planning arrays (reused by the next pass: stream order)
    if ((rc = kb_reserve(h, 1, n_slabs * (uint64_t)SLAB * 8 * KW))) return rc;
    if ((rc = kb_reserve(h, 4, n_slabs * (uint64_t)(nbins + 1) * 2 + 64))) return rc;
    const uint64_t n_pairs = n_groups * nbins;
    const uint64_t ovf_cap = n_entries_max / CHUNK + 1;       // pieces beyond the first of their pair: sum (np - 1) <= entries / CHUNK
    if ((rc = kb_reserve(h, 5, n_pairs * 16 + (size_t)nb1 * 12 + (2 * ovf_cap + 2) * 4 + 64))) return rc;
    KbScratch s;
    if ((rc = kb_scratch(h, s))) return rc;
    s.gpre_ent = (unsigned long long *)h->kb_buf[5];
    s.bin_ent = s.gpre_ent + n_pairs;
    s.gn = (uint32_t *)(s.bin_ent + nb1); s.gpre_row = s.gn + n_pairs; s.bin_rows = s.gpre_row + n_pairs; s.ovf = s.bin_rows + nb1;
    s.ovf_cap = (uint32_t)ovf_cap;

    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<hipEvent_t> sev;
    auto stamp = [&]() { if (h->prof) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, h->stream); sev.push_back(e); } };
    if (h->prof) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, h->stream); }
    stamp();                                                   // start of A

    const uint32_t pass_idx = h->n_pass;
    const bool sliced = plan.key_parts > 1;
    // A: a workgroup takes a few consecutive slabs (the next slab's words are prefetched under the current one)
    const uint32_t slabs_per_wg = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(8, n_slabs / ((uint64_t)h->n_cu * 8)));
    const unsigned grid_a = (unsigned)((n_slabs + slabs_per_wg - 1) / slabs_per_wg);
    if (sliced) hipLaunchKernelGGL((kb_slabsort_kernel<KW, true>), dim3(grid_a), dim3(KB_A_THREADS), lds_a, h->stream, d_packed, d_invalid, n_tiles, h->k, plan, s, slabs_per_wg);
    else hipLaunchKernelGGL((kb_slabsort_kernel<KW, false>), dim3(grid_a), dim3(KB_A_THREADS), lds_a, h->stream, d_packed, d_invalid, n_tiles, h->k, plan, s, slabs_per_wg);
    stamp();                                                   // end of A
    hipLaunchKernelGGL(kb_groupsum_kernel, dim3((unsigned)n_groups, (unsigned)((nbins + 63) / 64)), dim3(256), 0, h->stream, plan, s);
    hipLaunchKernelGGL(kb_binscan_kernel<CHUNK>, dim3((unsigned)nbins), dim3(256), 0, h->stream, plan, s);
    hipLaunchKernelGGL(kb_binfirst_kernel, dim3(1), dim3(KB_THREADS), 0, h->stream, plan, s, pass_idx, (unsigned long long)h->ring_used,
                       (unsigned long long)h->rows_used, h->ctl);
    HIPCHK(h, hipGetLastError());
    stamp();                                                   // end of the planning kernels
    // No host round trip: the pass's share of the ring is sized for one entry per position; B's first launch has one
    // workgroup per (group, bin) pair, its second walks the (usually empty) list of further pieces.
    // (debug flag 64: the one-piece-per-workgroup kernel of round 3's first half, for same-box comparisons)
    if (plan.dbg & 64) hipLaunchKernelGGL(kb_piecesort_kernel<KW>, dim3((unsigned)((n_pairs + 7) / 8 * 8)), dim3(KB_THREADS), lds_b, h->stream, plan, s, pass_idx);
    else hipLaunchKernelGGL(kb_piecesort_pipe_kernel<KW>, dim3((unsigned)std::max(8, h->n_cu / 8 * 8)), dim3(KB_THREADS), lds_b, h->stream, plan, s, pass_idx);
    hipLaunchKernelGGL(kb_piecesort_more_kernel<KW>, dim3((unsigned)std::min<uint64_t>(ovf_cap, (uint64_t)h->n_cu)), dim3(KB_THREADS), lds_b, h->stream, plan, s, pass_idx);
    stamp();                                                   // end of B
    HIPCHK(h, hipGetLastError());
    if (h->prof) {
        (void)hipEventRecord(e1, h->stream);
        h->prof_ev.emplace_back(e0, e1);
        h->prof_tiles.push_back(n_tiles);
        h->prof_stage_ev.push_back(sev);
    }
    h->n_pass++;
    h->ring_used += n_entries_max; h->rows_used += n_rows_max;
    h->pend_positions += n_entries_max;
    h->stat_binned_passes++;
    h->last_path = 1;
    return KDF_OK;
}

// partition a stream of any length: passes of at most opt_binned_max_positions
// positions, each starting on a tile boundary (windows that start in a pass may read on
// into the next tiles: the stream is one buffer)
static int kb_partition_stream(kdf_engine *h, const uint64_t *d_packed, const uint64_t *d_invalid, uint64_t n_bases, bool filtered) {
    const uint64_t step = h->opt_binned_max_positions;
    for (uint64_t off = 0; off < n_bases; off += step) {
        const uint64_t len = std::min<uint64_t>(step, n_bases - off);
        const uint64_t *p = d_packed + off / 32, *m = d_invalid + off / 64;
        int rc = h->kw == 1 ? kb_partition<1>(h, p, m, len, filtered) : kb_partition<2>(h, p, m, len, filtered);
        if (rc) return rc;
    }
    return KDF_OK;
}

// Kernel C over every pending pass: the ring is applied to the table and emptied.
static int kb_flush_ring(kdf_engine *h) {
    if (h->n_pass == 0) return KDF_OK;
    int rc;
    KbScratch s;
    if ((rc = kb_scratch(h, s))) return rc;
    const bool filtered = h->pend_filtered;
    // what the pending passes hold (entries, skew) and what the table holds now
    HIPCHK(h, hipMemcpyAsync(h->kb_totals_host, s.totals, 16 * 8, hipMemcpyDeviceToHost, h->stream));
    if ((rc = ctl_sync(h, nullptr))) return rc;
    const uint64_t n_entries = h->kb_totals_host[0];
    h->dens_windows += n_entries; h->dens_positions += h->pend_positions;
    const bool skewed = h->kb_totals_host[7] != 0 || (h->opt_debug_flags & 4096);          // (debug flag 4096 forces VAR 2: fuzzing)
    const uint64_t distinct_before = h->distinct;
    // Grow BEFORE the flush when the last flush's rate of new keys says the pending entries will not fit: growing now
    // rehashes the smaller table, and kernel C resolves the extra bucket bits itself (sub_bits) -- no failed buckets, no
    // replay through the global-atomic path.  (First flush of a table: the caller's capacity hint is trusted.)
    if (!filtered && h->grow_ratio > 0.0) {
        const double est = (double)h->distinct + h->grow_ratio * (double)n_entries;
        while (est > 0.6 * (double)h->cap && h->t.log2cap < 40) {
            if (h->lazy_empty) {                                  // nothing to carry over: a new table, still to be cleared
                KdfTable nt;
                if ((rc = table_alloc(h, h->t.log2cap + 1, nt, false))) break;       // (no room: kernel C will tell what really overflows)
                HIPCHK(h, hipStreamSynchronize(h->stream));
                table_free(h->t);
                h->t = nt; h->t.key_parts = h->opt_key_parts; h->t.key_part = h->opt_key_part;
                h->cap = 1ull << h->t.log2cap;
            } else if ((rc = table_rehash(h, h->t.log2cap + 1))) { (void)hipGetLastError(); h->err.clear(); break; }
        }
    }
    if (filtered && (rc = materialize(h))) return rc;
    KbPlan plan = h->pend_plan;
    plan.n_pass = h->n_pass; plan.dbg = h->opt_debug_flags;
    plan.log2cap = h->t.log2cap; plan.bucket_bits = h->t.bucket_bits;
    // a dump is waiting for this flush: kernel C writes it out of the buckets it holds (the request is taken: a second
    // flush of the same call must not dump again)
    const uint32_t fuse_min = filtered ? 0u : h->fuse_min;
    h->fuse_min = 0; h->fuse_done = false;
    if (fuse_min) {
        plan.dump_min = fuse_min;
        s.dump_lo = h->fuse_lo; s.dump_hi = h->fuse_hi; s.dump_cnt = h->fuse_cnt; s.dump_cap = h->fuse_cap;
        HIPCHK(h, hipMemsetAsync(h->ctl->tally, 0, sizeof(h->ctl->tally) + 8, h->stream));   // tally[] + cursor: ctl_sync reports their sum (a kdf_count_ge before this leaves its tally behind)
    }
    plan.sub_bits = (h->t.log2cap - h->t.bucket_bits) - plan.c1 - plan.c2;       // a table that grew since the partition: more sub-buckets
    const uint64_t nb_table = 1ull << (plan.c1 + plan.c2 + plan.sub_bits);
    const size_t failed_bytes = (size_t)((nb_table + 31) / 32) * 4;
    if ((rc = kb_reserve(h, 3, failed_bytes))) return rc;
    s.failed = (uint32_t *)h->kb_buf[3];
    HIPCHK(h, hipMemsetAsync(s.failed, 0, failed_bytes, h->stream));
    const int nonempty = h->lazy_empty ? 0 : 1;   // 0: kernel C rewrites every bucket (this IS the clear)
    std::vector<hipEvent_t> sev;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, h->stream);
        hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, h->stream); sev.push_back(e);
    }
    if (plan.dbg & 2048) return kb_ring_reset(h);                 // (ablation: the partition passes are timed alone, what they wrote is dropped)
    const bool heavy = skewed && s.hv_ctr && plan.sub_bits == 0;
    if (heavy) {
        HIPCHK(h, hipMemsetAsync(s.hv_ctr, 0, (4 + 3 * KB_HV_MAX) * 4, h->stream));
    }
    by_width(h, [&](auto KWc) {
        constexpr int KW = decltype(KWc)::value;
        const size_t lds_c = KB_C_LDS(KW, plan.bucket_bits);
        const bool big = plan.bucket_bits > KB_BB_SMALL(KW);
#define KB_LV(M, V, D) do { if (big) hipLaunchKernelGGL((kb_bucket_kernel<KW, M, V, true, D>), dim3((unsigned)nb_table), dim3(KB_C_CT_BIG), lds_c, h->stream, plan, s, h->t, h->ctl, nonempty); \
                         else hipLaunchKernelGGL((kb_bucket_kernel<KW, M, V, false, D>), dim3((unsigned)nb_table), dim3(KB_C_CT(KW)), lds_c, h->stream, plan, s, h->t, h->ctl, nonempty); } while (0)
        if (filtered) { if (skewed) KB_LV(KB_MODE_FILTERED, 2, false); else KB_LV(KB_MODE_FILTERED, 1, false); }
        else if (fuse_min) { if (skewed) KB_LV(KB_MODE_INSERT, 2, true); else KB_LV(KB_MODE_INSERT, 1, true); }
        else { if (skewed) KB_LV(KB_MODE_INSERT, 2, false); else KB_LV(KB_MODE_INSERT, 1, false); }
#undef KB_LV
        return 0;
    });
    if (heavy) {
        // the buckets the skewed instantiation left aside
        const size_t lds_h = ((size_t)(8 * h->kw + 4) << plan.bucket_bits) + KB_RI_LDS_BYTES;
        if (filtered) {                                            // the keys stay put: the slices add to the counts in HBM
            if (h->kw == 1) hipLaunchKernelGGL(kb_heavy_filtered_kernel<1>, dim3(KB_HV_SLICES, KB_HV_MAX), dim3(256), lds_h, h->stream, plan, s, h->t);
            else hipLaunchKernelGGL(kb_heavy_filtered_kernel<2>, dim3(KB_HV_SLICES, KB_HV_MAX), dim3(256), lds_h, h->stream, plan, s, h->t);
        } else if (h->kw == 1) {
            hipLaunchKernelGGL(kb_heavy_slice_kernel<1>, dim3(KB_HV_SLICES, KB_HV_MAX), dim3(256), lds_h, h->stream, plan, s);
            hipLaunchKernelGGL(kb_heavy_combine_kernel<1>, dim3(KB_HV_MAX), dim3(256), lds_h, h->stream, plan, s, h->t, h->ctl, nonempty);
        } else {
            hipLaunchKernelGGL(kb_heavy_slice_kernel<2>, dim3(KB_HV_SLICES, KB_HV_MAX), dim3(256), lds_h, h->stream, plan, s);
            hipLaunchKernelGGL(kb_heavy_combine_kernel<2>, dim3(KB_HV_MAX), dim3(256), lds_h, h->stream, plan, s, h->t, h->ctl, nonempty);
        }
    }
    HIPCHK(h, hipGetLastError());
    if (h->prof) {
        hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, h->stream); sev.push_back(e);
        (void)hipEventRecord(e1, h->stream);
        h->prof_ev.emplace_back(e0, e1);
        h->prof_tiles.push_back(0);                            // (its positions were counted with the partition passes)
        h->prof_stage_ev.push_back(sev);
    }
    HIPCHK(h, hipMemcpyAsync(h->kb_totals_host, s.totals, 16 * 8, hipMemcpyDeviceToHost, h->stream));
    bool full = false;
    uint64_t cursor = 0;
    if ((rc = ctl_sync(h, &full, &cursor))) return rc;
    // the fused dump is whole only if every bucket went through kernel C's write-back (none failed, none was left to the
    // heavy-bucket kernels); otherwise the caller dumps from the table as usual
    if (fuse_min && h->kb_totals_host[2] == 0 && h->kb_totals_host[4] == 0) { h->fuse_done = true; h->fuse_n = cursor; h->stat_fused_dumps++; }
    h->stat_flushes++;
    h->lazy_empty = false;
    h->stat_heavy_buckets += h->kb_totals_host[4];
    const uint64_t n_failed = h->kb_totals_host[2];
    if (n_failed) {
        if (filtered) { (void)kb_ring_reset(h); return fail(h, KDF_ERR_STATE, "binned count --if: a bucket failed (corrupt table?)"); }
        // some buckets overflowed: they are untouched in HBM.  Grow the table so
        // that even if every entry of the failed buckets were new the load stays
        // <= 0.5, then replay exactly those buckets through the global-atomic path.
        h->stat_replayed_buckets += n_failed;
        const uint64_t worst = h->distinct + std::min<uint64_t>(n_entries, n_failed * ((n_entries / std::max<uint64_t>(nb_table, 1)) * 4 + 4096));
        const uint32_t want = std::max<uint32_t>(h->t.log2cap + 1, cap_log2_for(worst));
        if ((rc = table_rehash(h, want))) { (void)kb_ring_reset(h); return rc; }
        if (h->kw == 1) hipLaunchKernelGGL(kb_replay_kernel<1>, dim3((unsigned)nb_table), dim3(256), 0, h->stream, plan, s, h->t, h->ctl);
        else hipLaunchKernelGGL(kb_replay_kernel<2>, dim3((unsigned)nb_table), dim3(256), 0, h->stream, plan, s, h->t, h->ctl);
        HIPCHK(h, hipGetLastError());
        if ((rc = ctl_sync(h, &full))) { (void)kb_ring_reset(h); return rc; }
        if (full) { (void)kb_ring_reset(h); return fail(h, KDF_ERR_TABLE_FULL, "binned count: bucket overflow during replay (capacity 2^%u)", h->t.log2cap); }
    }
    if (!filtered && n_entries >= 100000) h->grow_ratio = (double)(h->distinct - distinct_before) / (double)n_entries;
    if ((rc = kb_ring_reset(h))) return rc;
    // keep the load <= 0.7 for what comes next (a 2048-slot bucket then holds
    // 1434 +- 38 keys: overflow, which is handled anyway, stays a rare event)
    if (!filtered)
        while (h->distinct * 10 > h->cap * 7)
            if ((rc = table_rehash(h, h->t.log2cap + 1))) return rc;
    return KDF_OK;
}

// can the binned pipeline work on this engine's table at all?
static bool kb_eligible(const kdf_engine *h) {
    if (h->opt_hash_shift) return false;                       // an owner table: the bins assume home = top hash bits
    if (h->t.log2cap <= h->t.bucket_bits) return false;        // a single bucket: nothing to partition
    return h->opt_force_path != 1;
}

// Is a flush of n_bases pending positions worth the binned pipeline?  Kernel C reads and rewrites EVERY bucket of the
// table (0.5 ms per GB), whatever the pending passes hold; the direct kernels cost 0.056 ms per million positions whatever
// the table.  Crossover (scratch/bigtable_probe.py): ~14 M positions per GB of table.
static bool use_binned(const kdf_engine *h, uint64_t n_bases, bool filtered) {
    if (!kb_eligible(h)) return false;
    if (h->opt_force_path == 2) return true;
    if (n_bases < h->opt_binned_min_positions) return false;
    if (filtered && h->t.log2cap < h->opt_binned_filtered_min_log2cap) return false;
    if (filtered && h->opt_defer) return true;                 // (more batches will follow before the counts are read: the rewrite is shared)
    // (a table that was only `clear`ed: the binned flush is also its clear, the direct path pays a memset first --
    // 0.2 ms per GB -- which moves the crossover to ~8.6 M positions per GB)
    const uint64_t table_bytes = h->cap * (uint64_t)(8 * h->kw + 4);
    const uint64_t per = h->lazy_empty ? h->opt_binned_bytes_per_position * 5 / 3 : h->opt_binned_bytes_per_position;
    return n_bases * per >= table_bytes;
}

// insert-mode count through the global-atomic kernels.  The stream is walked in
// chunks sized so that even if every position were a new key the table stays
// at load <= 0.8; the table doubles when fewer than cap/8 positions fit.
static int direct_insert(kdf_engine *h, const uint64_t *d_packed, const uint64_t *d_invalid, uint64_t n_bases) {
    const uint64_t n_tiles = (n_bases + KDF_TILE - 1) / KDF_TILE;
    { int rc0 = materialize(h); if (rc0) return rc0; }
    h->last_path = 0;
    uint64_t tile = 0;
    while (tile < n_tiles) {
        uint64_t room = (h->cap / 10) * 8 > h->distinct ? (h->cap / 10) * 8 - h->distinct : 0;
        if (room < h->cap / 8) {
            int rc = table_rehash(h, h->t.log2cap + 1);
            if (rc) return rc;
            continue;
        }
        uint64_t chunk = std::min<uint64_t>(n_tiles - tile, std::max<uint64_t>(room / KDF_TILE, 1));
        launch_stream<MODE_INSERT>(h, d_packed, d_invalid, tile, chunk, nullptr);
        HIPCHK(h, hipGetLastError());
        bool full = false;
        int rc = ctl_sync(h, &full);
        if (rc) return rc;
        if (full) return fail(h, KDF_ERR_TABLE_FULL, "count: bucket overflow (capacity 2^%u, %llu distinct)",
                              h->t.log2cap, (unsigned long long)h->distinct);
        tile += chunk;
    }
    return KDF_OK;
}

// L1: append a batch to the pending stream (tile aligned; the copy forces the mask bits past n_bases to "invalid")
static int l1_append(kdf_engine *h, const uint64_t *d_packed, const uint64_t *d_invalid, uint64_t n_bases) {
    const uint64_t n_tiles = (n_bases + KDF_TILE - 1) / KDF_TILE;
    if (h->l1_tiles + n_tiles > h->l1_cap_tiles) {
        // grow (the pending stream moves): up to the size at which it is partitioned anyway
        const uint64_t target = std::max<uint64_t>((h->opt_l1_positions + h->opt_l1_direct_positions) / KDF_TILE + 1, h->l1_tiles + n_tiles);
        const uint64_t cap = std::min<uint64_t>(target, std::max<uint64_t>({2 * h->l1_cap_tiles, 4 * (h->l1_tiles + n_tiles), (uint64_t)1 << 18}));
        uint64_t *np = nullptr, *nm = nullptr;
        HIPCHK(h, hipMalloc((void **)&np, (cap * 2 + 4) * 8));
        hipError_t e = hipMalloc((void **)&nm, (cap + 2) * 8);
        if (e != hipSuccess) { (void)hipFree(np); (void)hipGetLastError(); return fail(h, KDF_ERR_NOMEM, "pending stream: %s", hipGetErrorString(e)); }
        if (h->l1_tiles) {
            HIPCHK(h, hipMemcpyAsync(np, h->l1_packed, (h->l1_tiles * 2 + 4) * 8, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(nm, h->l1_mask, (h->l1_tiles + 2) * 8, hipMemcpyDeviceToDevice, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->l1_packed) (void)hipFree(h->l1_packed);
        if (h->l1_mask) (void)hipFree(h->l1_mask);
        h->l1_packed = np; h->l1_mask = nm; h->l1_cap_tiles = cap;
    }
    const unsigned blocks = (unsigned)((2 * n_tiles + 4 + 255) / 256);
    hipLaunchKernelGGL(kb_append_kernel, dim3(blocks), dim3(256), 0, h->stream, h->l1_packed + 2 * h->l1_tiles, h->l1_mask + h->l1_tiles,
                       d_packed, d_invalid, n_bases);
    HIPCHK(h, hipGetLastError());
    h->l1_tiles += n_tiles;
    return KDF_OK;
}

// everything pending (the concatenated small batches, the partitioned passes) goes into the table
static int pending_flush(kdf_engine *h, bool fuse = false) {
    int rc;
    const uint32_t want_fuse = fuse ? h->fuse_min : 0u;
    h->fuse_min = 0; h->fuse_done = false;                     // (only the LAST flush below may dump: earlier ones see counts that are not final)
    if (h->l1_tiles) {
        const uint64_t n = h->l1_tiles * KDF_TILE;
        h->l1_tiles = 0;
        if (h->n_pass > 0 ? kb_eligible(h) : use_binned(h, n, false)) rc = kb_partition_stream(h, h->l1_packed, h->l1_mask, n, false);
        else rc = direct_insert(h, h->l1_packed, h->l1_mask, n);
        if (rc) return rc;
    }
    h->fuse_min = want_fuse;
    rc = kb_flush_ring(h);
    h->fuse_min = 0;
    return rc;
}
// ... or is forgotten (kdf_clear)
static int pending_drop(kdf_engine *h) {
    h->l1_tiles = 0;
    return kb_ring_reset(h);
}

// insert-mode count over a device-resident stream
static int count_insert_dev(kdf_engine *h, const uint64_t *d_packed, const uint64_t *d_invalid, uint64_t n_bases) {
    if (h->filter_mode) return fail(h, KDF_ERR_STATE, "kdf_count_reads: a filter is loaded; call kdf_clear first");
    h->sieve_valid = false;                          // new keys join the table: a sieve built from it earlier (scan) is stale
    h->t.key_parts = h->opt_key_parts; h->t.key_part = h->opt_key_part;       // (tables are re-created by reserve / rehash: set per call)
    if (n_bases == 0) return KDF_OK;
    int rc;
    if (h->opt_force_path == 2 && kb_eligible(h)) rc = kb_partition_stream(h, d_packed, d_invalid, n_bases, false);
    else if (!kb_eligible(h) || h->opt_force_path == 1) {
        if ((rc = pending_flush(h))) return rc;
        rc = direct_insert(h, d_packed, d_invalid, n_bases);
    } else if (n_bases >= h->opt_l1_direct_positions) rc = kb_partition_stream(h, d_packed, d_invalid, n_bases, false);   // big enough by itself
    else {
        rc = l1_append(h, d_packed, d_invalid, n_bases);
        if (!rc && h->l1_tiles * KDF_TILE >= h->opt_l1_positions) {
            const uint64_t n = h->l1_tiles * KDF_TILE;
            h->l1_tiles = 0;
            rc = kb_partition_stream(h, h->l1_packed, h->l1_mask, n, false);
        }
    }
    if (rc) return rc;
    if (!h->opt_defer) return pending_flush(h);
    return KDF_OK;
}

static int count_filtered_dev(kdf_engine *h, const uint64_t *d_packed, const uint64_t *d_invalid, uint64_t n_bases) {
    if (!h->filter_mode) return fail(h, KDF_ERR_STATE, "kdf_count_reads_filtered: no filter loaded (kdf_load_filter)");
    const uint64_t n_tiles = (n_bases + KDF_TILE - 1) / KDF_TILE;
    if (n_tiles == 0) return KDF_OK;
    { int rc0 = materialize(h); if (rc0) return rc0; }
    if (h->sieve_valid && (h->opt_force_path == 0 || h->opt_force_path == 4)) {
        // persistent workgroups over slabs of 1024 x WPT positions
        const int WPT = h->kw == 1 ? KbCfg<1>::WPT : KbCfg<2>::WPT;
        const uint64_t tiles_per_slab = KB_THREADS / (64 / WPT);
        const uint64_t n_slabs = (n_tiles + tiles_per_slab - 1) / tiles_per_slab;
        const uint32_t n_wg = (uint32_t)std::min<uint64_t>(n_slabs, (uint64_t)h->n_cu * 8);
        const uint32_t spw = (uint32_t)((n_slabs + n_wg - 1) / n_wg);
        const unsigned grid = (unsigned)((n_slabs + spw - 1) / spw);
        KdfSieve sv{h->sieve, h->sieve_words - 1};
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (h->prof) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, h->stream); }
        const bool in_lds = h->sieve_words <= KDF_SV_LDS_WORDS && !(h->opt_debug_flags & 2048);
        if (h->kw == 1 && in_lds) hipLaunchKernelGGL((kdf_sieve_count_kernel<1, true>), dim3(grid), dim3(KB_THREADS), 0, h->stream, d_packed, d_invalid, n_tiles, h->k, h->t, h->ctl, sv, spw);
        else if (h->kw == 1) hipLaunchKernelGGL((kdf_sieve_count_kernel<1, false>), dim3(grid), dim3(KB_THREADS), 0, h->stream, d_packed, d_invalid, n_tiles, h->k, h->t, h->ctl, sv, spw);
        else if (in_lds) hipLaunchKernelGGL((kdf_sieve_count_kernel<2, true>), dim3(grid), dim3(KB_THREADS), 0, h->stream, d_packed, d_invalid, n_tiles, h->k, h->t, h->ctl, sv, spw);
        else hipLaunchKernelGGL((kdf_sieve_count_kernel<2, false>), dim3(grid), dim3(KB_THREADS), 0, h->stream, d_packed, d_invalid, n_tiles, h->k, h->t, h->ctl, sv, spw);
        if (h->prof) { (void)hipEventRecord(e1, h->stream); h->prof_ev.emplace_back(e0, e1); h->prof_tiles.push_back(n_tiles); }
        HIPCHK(h, hipGetLastError());
        h->last_path = 3;
        return KDF_OK;
    }
    if (h->opt_force_path == 4) return fail(h, KDF_ERR_STATE, "force_path 4 (sieve): no sieve for this filter (it would not fit the caches, or keys were added after kdf_load_filter)");
    if (use_binned(h, n_bases, true)) {
        int rc = kb_partition_stream(h, d_packed, d_invalid, n_bases, true);
        if (!rc && !h->opt_defer) rc = kb_flush_ring(h);
        return rc;
    }
    { int rc0 = kb_flush_ring(h); if (rc0) return rc0; }           // (binned --if passes pending from earlier batches)
    h->last_path = 0;
    launch_stream<MODE_FILTERED>(h, d_packed, d_invalid, 0, n_tiles, nullptr);
    HIPCHK(h, hipGetLastError());
    return KDF_OK;
}

__global__ void kdf_mask_tail_kernel(uint64_t *word, uint64_t bits) { *word |= bits; }

static int upload_stream(kdf_engine *h, const uint64_t *packed, const uint64_t *invalid, uint64_t n_bases,
                         uint64_t **d_packed, uint64_t **d_invalid) {
    uint64_t pw, mw;
    kdf_stream_words(n_bases, &pw, &mw);
    int rc;
    if ((rc = stage_reserve(h, 0, pw * 8))) return rc;
    if ((rc = stage_reserve(h, 1, mw * 8))) return rc;
    // the caller's arrays hold ceil(n/32) / ceil(n/64) meaningful words; pad on device
    const uint64_t pw_in = (n_bases + 31) / 32, mw_in = (n_bases + 63) / 64;
    HIPCHK(h, hipMemsetAsync(h->stage[0], 0, pw * 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->stage[1], 0xFF, mw * 8, h->stream));
    if (pw_in) HIPCHK(h, hipMemcpyAsync(h->stage[0], packed, pw_in * 8, hipMemcpyHostToDevice, h->stream));
    if (mw_in) HIPCHK(h, hipMemcpyAsync(h->stage[1], invalid, mw_in * 8, hipMemcpyHostToDevice, h->stream));
    // bits past n_bases in the last mask word must read "invalid"
    if (n_bases % 64) {
        uint64_t last = invalid[mw_in - 1] | (~0ull << (n_bases % 64));
        HIPCHK(h, hipMemcpyAsync((uint64_t *)h->stage[1] + (mw_in - 1), &last, 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));     // `last` is a stack temporary
    }
    *d_packed = (uint64_t *)h->stage[0];
    *d_invalid = (uint64_t *)h->stage[1];
    return KDF_OK;
}

// ===========================================================================
// C ABI
// ===========================================================================

extern "C" {

const char *kdf_last_error(const kdf_engine *h) { return h ? h->err.c_str() : g_err.c_str(); }

int kdf_create(int device, int k, uint64_t capacity_hint, kdf_engine **out) {
    if (!out) return fail(nullptr, KDF_ERR_INVALID, "kdf_create: out is NULL");
    *out = nullptr;
    if (k < 1 || k > 63) return fail(nullptr, KDF_ERR_INVALID, "kdf_create: k=%d out of range 1..63", k);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, KDF_ERR_HIP, "kdf_create: no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, KDF_ERR_INVALID, "kdf_create: device %d of %d", device, ndev);
    kdf_engine *h = new kdf_engine();
    h->device = device; h->k = k; h->kw = k <= 32 ? 1 : 2;
    auto bail = [&](int rc) { g_err = h->err; kdf_destroy(h); return rc; };
    if ((e = hipSetDevice(device)) != hipSuccess) { h->err = hipGetErrorString(e); return bail(KDF_ERR_HIP); }
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) h->n_cu = ncu; }
    { size_t f = 0, tt = 0; if (hipMemGetInfo(&f, &tt) == hipSuccess) h->dev_total_bytes = tt; else (void)hipGetLastError(); }
    if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) { h->err = hipGetErrorString(e); return bail(KDF_ERR_HIP); }
    h->stream = h->own_stream;
    if ((e = hipMalloc((void **)&h->ctl, sizeof(KdfCtl))) != hipSuccess) { h->err = hipGetErrorString(e); return bail(KDF_ERR_NOMEM); }
    if ((e = hipMalloc((void **)&h->d_out4, 32)) != hipSuccess) { h->err = hipGetErrorString(e); return bail(KDF_ERR_NOMEM); }
    if ((e = hipHostMalloc((void **)&h->h_out4, 32)) != hipSuccess) { h->err = hipGetErrorString(e); return bail(KDF_ERR_NOMEM); }
    int rc = ctl_reset(h, false);
    if (rc) return bail(rc);
    if (const char *ev = getenv("KDF_BIG_BUCKET_LOG2CAP")) { const int v = atoi(ev); if (v >= 10 && v <= 64) h->opt_big_bucket_log2cap = (uint32_t)v; }
    rc = table_alloc(h, cap_log2_for(capacity_hint), h->t);
    if (rc) return bail(rc);
    h->cap = 1ull << h->t.log2cap;
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) { h->err = hipGetErrorString(e); return bail(KDF_ERR_HIP); }
    *out = h;
    return KDF_OK;
}

void kdf_destroy(kdf_engine *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    table_free(h->t);
    prof_collect(h);
    for (int i = 0; i < 4; ++i) if (h->stage[i]) (void)hipFree(h->stage[i]);
    for (int i = 0; i < 8; ++i) if (h->kb_buf[i]) (void)hipFree(h->kb_buf[i]);
    if (h->l1_packed) (void)hipFree(h->l1_packed);
    if (h->l1_mask) (void)hipFree(h->l1_mask);
    if (h->kb_pass) (void)hipFree(h->kb_pass);
    if (h->merge_buf) (void)hipFree(h->merge_buf);
    if (h->kb_heavy) (void)hipFree(h->kb_heavy);
    for (int sl = 0; sl < 2; ++sl) {
        for (int j = 0; j < 2; ++j) if (h->up_buf[sl][j]) (void)hipFree(h->up_buf[sl][j]);
        if (h->up_done[sl]) (void)hipEventDestroy(h->up_done[sl]);
        if (h->use_done[sl]) (void)hipEventDestroy(h->use_done[sl]);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->sieve) (void)hipFree(h->sieve);
    if (h->kb_small) (void)hipFree(h->kb_small);
    if (h->kb_totals_host) (void)hipHostFree(h->kb_totals_host);
    if (h->ctl) (void)hipFree(h->ctl);
    if (h->d_out4) (void)hipFree(h->d_out4);
    if (h->h_out4) (void)hipHostFree(h->h_out4);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int kdf_set_stream(kdf_engine *h, void *hip_stream) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    { int rcf = pending_flush(h); if (rcf) return rcf; }          // (pending passes were enqueued on the old stream)
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return KDF_OK;
}

int kdf_synchronize(kdf_engine *h) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->copy_stream) HIPCHK(h, hipStreamSynchronize(h->copy_stream));      // uploads still in flight read host buffers
    return KDF_OK;
}

int kdf_clear(kdf_engine *h) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ctl_reset(h, false);
    if (rc) return rc;
    h->distinct = 0; h->windows = 0; h->filter_mode = false;
    h->lazy_empty = true; h->sieve_valid = false;
    h->grow_ratio = 0.0;
    if ((rc = pending_drop(h))) return rc;     // what was counted but not yet applied is dropped with the rest
    return KDF_OK;
}

int kdf_reserve(kdf_engine *h, uint64_t n_keys) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    HIPCHK(h, hipSetDevice(h->device));
    { int rcf = pending_flush(h); if (rcf) return rcf; }
    const uint32_t want = cap_log2_for(std::max(n_keys, h->distinct));
    if (want <= h->t.log2cap) return KDF_OK;
    int rc = ctl_sync(h, nullptr);
    if (rc) return rc;
    return table_rehash(h, want);
}

int kdf_stats(kdf_engine *h, uint64_t *capacity, uint64_t *distinct, uint64_t *windows) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    HIPCHK(h, hipSetDevice(h->device));
    { int rcf = pending_flush(h); if (rcf) return rcf; }
    bool full = false;
    int rc = ctl_sync(h, &full);
    if (rc) return rc;
    if (capacity) *capacity = h->cap;
    if (distinct) *distinct = h->distinct;
    if (windows) *windows = h->windows;
    if (full) return fail(h, KDF_ERR_TABLE_FULL, "a bucket overflowed during an earlier call");
    return KDF_OK;
}

int kdf_flush(kdf_engine *h) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    HIPCHK(h, hipSetDevice(h->device));
    return pending_flush(h);
}

int kdf_count_reads_dev(kdf_engine *h, const void *d_packed, const void *d_invalid, uint64_t n_bases) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n_bases && (!d_packed || !d_invalid)) return fail(h, KDF_ERR_INVALID, "kdf_count_reads_dev: NULL stream");
    HIPCHK(h, hipSetDevice(h->device));
    return count_insert_dev(h, (const uint64_t *)d_packed, (const uint64_t *)d_invalid, n_bases);
}

int kdf_count_reads(kdf_engine *h, const uint64_t *packed, const uint64_t *invalid, uint64_t n_bases) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n_bases == 0) return KDF_OK;
    if (!packed || !invalid) return fail(h, KDF_ERR_INVALID, "kdf_count_reads: NULL stream");
    HIPCHK(h, hipSetDevice(h->device));
    uint64_t *dp, *dm;
    int rc = upload_stream(h, packed, invalid, n_bases, &dp, &dm);
    if (rc) return rc;
    return count_insert_dev(h, dp, dm, n_bases);
}

int kdf_host_alloc(uint64_t bytes, void **out) {
    if (!out) return fail(nullptr, KDF_ERR_INVALID, "kdf_host_alloc: NULL pointer");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, KDF_ERR_NOMEM, "kdf_host_alloc: %s", hipGetErrorString(e)); }
    return KDF_OK;
}
int kdf_host_free(void *p) {
    if (p) (void)hipHostFree(p);
    return KDF_OK;
}

int kdf_upload_reads_async(kdf_engine *h, int slot, const uint64_t *packed, const uint64_t *invalid, uint64_t n_bases) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (slot < 0 || slot > 1) return fail(h, KDF_ERR_INVALID, "kdf_upload_reads_async: slot must be 0 or 1");
    if (n_bases && (!packed || !invalid)) return fail(h, KDF_ERR_INVALID, "kdf_upload_reads_async: NULL stream");
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->copy_stream) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (int sl = 0; sl < 2; ++sl) {
            HIPCHK(h, hipEventCreateWithFlags(&h->up_done[sl], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->use_done[sl], hipEventDisableTiming));
        }
    }
    uint64_t pw, mw;
    kdf_stream_words(n_bases, &pw, &mw);
    const size_t want[2] = {(size_t)pw * 8, (size_t)mw * 8};
    for (int j = 0; j < 2; ++j) {
        if (h->up_bytes[slot][j] >= want[j]) continue;
        HIPCHK(h, hipStreamSynchronize(h->stream));            // (the slot's last count may still read the old buffer)
        HIPCHK(h, hipStreamSynchronize(h->copy_stream));
        if (h->up_buf[slot][j]) (void)hipFree(h->up_buf[slot][j]);
        h->up_buf[slot][j] = nullptr; h->up_bytes[slot][j] = 0;
        const size_t sz = want[j] + want[j] / 8 + 4096;
        HIPCHK(h, hipMalloc(&h->up_buf[slot][j], sz));
        h->up_bytes[slot][j] = sz;
    }
    h->up_valid[slot] = false;
    h->up_n[slot] = n_bases;
    if (n_bases == 0) { h->up_valid[slot] = true; return KDF_OK; }
    hipStream_t cs = h->copy_stream;
    if (h->use_done[slot]) HIPCHK(h, hipStreamWaitEvent(cs, h->use_done[slot], 0));   // the count that last read this slot
    const uint64_t pw_in = (n_bases + 31) / 32, mw_in = (n_bases + 63) / 64;
    // the caller's arrays hold ceil(n/32) / ceil(n/64) meaningful words: pad the rest on the device
    if (pw > pw_in) HIPCHK(h, hipMemsetAsync((uint64_t *)h->up_buf[slot][0] + pw_in, 0, (pw - pw_in) * 8, cs));
    if (mw > mw_in) HIPCHK(h, hipMemsetAsync((uint64_t *)h->up_buf[slot][1] + mw_in, 0xFF, (mw - mw_in) * 8, cs));
    HIPCHK(h, hipMemcpyAsync(h->up_buf[slot][0], packed, pw_in * 8, hipMemcpyHostToDevice, cs));
    HIPCHK(h, hipMemcpyAsync(h->up_buf[slot][1], invalid, mw_in * 8, hipMemcpyHostToDevice, cs));
    if (n_bases % 64)                                           // bits past n_bases in the last mask word must read "invalid"
        hipLaunchKernelGGL(kdf_mask_tail_kernel, dim3(1), dim3(1), 0, cs, (uint64_t *)h->up_buf[slot][1] + (mw_in - 1), ~0ull << (n_bases % 64));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->up_done[slot], cs));
    h->up_valid[slot] = true;
    return KDF_OK;
}

int kdf_count_uploaded(kdf_engine *h, int slot, int filtered) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (slot < 0 || slot > 1 || !h->up_valid[slot]) return fail(h, KDF_ERR_STATE, "kdf_count_uploaded: nothing was uploaded into slot %d", slot);
    HIPCHK(h, hipSetDevice(h->device));
    h->up_valid[slot] = false;
    const uint64_t n = h->up_n[slot];
    if (n == 0) return KDF_OK;
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->up_done[slot], 0));
    // The HOST waits for the copy as well: the caller recycles its (pinned) source buffer as soon as this call returns,
    // and a filtered count returns without any host synchronisation.  The copy was issued a whole batch ago and has
    // normally long finished.
    HIPCHK(h, hipEventSynchronize(h->up_done[slot]));
    const uint64_t *dp = (const uint64_t *)h->up_buf[slot][0], *dm = (const uint64_t *)h->up_buf[slot][1];
    const int rc = filtered ? count_filtered_dev(h, dp, dm, n) : count_insert_dev(h, dp, dm, n);
    (void)hipEventRecord(h->use_done[slot], h->stream);
    return rc;
}

// Size, allocate and zero the membership sieve for n keys (sieve_valid says whether there is one).  Every window costs
// one random 8-byte read of it, i.e. one L2 request: measured on the parent-filter workload (1.49 G windows, 1.9 M keys)
// 8.3 ms with a 2 MB sieve, 9.1 ms with 4 MB, 16 ms with 8 MB, 24 ms with 16 MB -- it must sit in the 4 MB L2 of every XCD
// beside the streamed reads.  So: the most bits per key out of 32 / 16 / 8 that keep it within 2 MB, 8 bits per key beyond
// that, and no sieve (the binned path) once even that passes 16 MB.  Up to 64 K keys it is shrunk to the 64 KB that the
// kernel copies into LDS (no L2 request per window at all: ~700 Gk-mer/s for the filters of VCF mode and Module 3).
static int sieve_prepare(kdf_engine *h, uint64_t n) {
    h->sieve_valid = false;
    uint64_t bpk = h->opt_sieve_bits ? (uint64_t)h->opt_sieve_bits : 32;
    if (!h->opt_sieve_bits) {
        while (bpk > 8 && n * bpk > (16ull << 20)) bpk >>= 1;
        if (n * 8 <= (uint64_t)KDF_SV_LDS_WORDS * 64) while (bpk > 8 && n * bpk > (uint64_t)KDF_SV_LDS_WORDS * 64) bpk >>= 1;
    }
    if (!(h->opt_sieve_bits || n * bpk <= (128ull << 20))) return KDF_OK;
    const uint64_t bits = n * bpk;
    const uint64_t words = std::max<uint64_t>(1024, 1ull << log2ceil((bits + 63) / 64));
    if (h->sieve_alloc < words) {
        if (h->sieve) (void)hipFree(h->sieve);
        h->sieve = nullptr; h->sieve_alloc = 0;
        HIPCHK(h, hipMalloc((void **)&h->sieve, words * 8));
        h->sieve_alloc = words;
    }
    h->sieve_words = words;
    HIPCHK(h, hipMemsetAsync(h->sieve, 0, words * 8, h->stream));
    h->sieve_valid = true;
    return KDF_OK;
}

// the table becomes exactly the n keys at d_lo / d_hi (device arrays) with count 0
static int load_filter_core(kdf_engine *h, const uint64_t *d_lo, const uint64_t *d_hi, uint64_t n) {
    int rc;
    h->sieve_valid = false;
    if ((rc = pending_drop(h))) return rc;            // the table becomes the filter: whatever was pending goes with the old contents
    // size the table for n keys at load <= 0.5, then start from empty
    const uint32_t want = cap_log2_for(n);
    if (want != h->t.log2cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        table_free(h->t);
        if ((rc = table_alloc(h, want, h->t))) return rc;
        h->cap = 1ull << want;
        if ((rc = ctl_reset(h, false))) return rc;
        h->distinct = 0; h->windows = 0; h->lazy_empty = false; h->filter_mode = false;
    } else if ((rc = kdf_clear(h))) return rc;
    if ((rc = materialize(h))) return rc;
    h->filter_mode = true;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (n) {
        if (h->kw == 1)
            hipLaunchKernelGGL(kdf_insert_keys_kernel<1>, dim3(blocks), dim3(256), 0, h->stream,
                               d_lo, (const uint64_t *)nullptr, (const uint32_t *)nullptr, n, h->t, h->ctl, 0, 0);
        else
            hipLaunchKernelGGL(kdf_insert_keys_kernel<2>, dim3(blocks), dim3(256), 0, h->stream,
                               d_lo, d_hi, (const uint32_t *)nullptr, n, h->t, h->ctl, 0, 0);
        HIPCHK(h, hipGetLastError());
        bool full = false;
        if ((rc = ctl_sync(h, &full))) return rc;
        if (full) return fail(h, KDF_ERR_TABLE_FULL, "kdf_load_filter: bucket overflow");
    }
    if ((rc = sieve_prepare(h, n)) != KDF_OK) return rc;
    if (h->sieve_valid && n) {
        if (h->kw == 1) hipLaunchKernelGGL(kdf_sieve_build_kernel<1>, dim3(blocks), dim3(256), 0, h->stream, d_lo, (const uint64_t *)nullptr, n, h->sieve, h->sieve_words - 1);
        else hipLaunchKernelGGL(kdf_sieve_build_kernel<2>, dim3(blocks), dim3(256), 0, h->stream, d_lo, d_hi, n, h->sieve, h->sieve_words - 1);
        HIPCHK(h, hipGetLastError());
    }
    return KDF_OK;
}

int kdf_load_filter(kdf_engine *h, const uint64_t *keys_lo, const uint64_t *keys_hi, uint64_t n) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n && (!keys_lo || (h->kw == 2 && !keys_hi))) return fail(h, KDF_ERR_INVALID, "kdf_load_filter: NULL keys");
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (n) {
        if ((rc = stage_reserve(h, 2, n * 8))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->stage[2], keys_lo, n * 8, hipMemcpyHostToDevice, h->stream));
        if (h->kw == 2) {
            if ((rc = stage_reserve(h, 3, n * 8))) return rc;
            HIPCHK(h, hipMemcpyAsync(h->stage[3], keys_hi, n * 8, hipMemcpyHostToDevice, h->stream));
        }
    }
    return load_filter_core(h, (const uint64_t *)h->stage[2], (const uint64_t *)h->stage[3], n);
}

int kdf_load_filter_dev(kdf_engine *h, const void *d_keys_lo, const void *d_keys_hi, uint64_t n) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n && (!d_keys_lo || (h->kw == 2 && !d_keys_hi))) return fail(h, KDF_ERR_INVALID, "kdf_load_filter_dev: NULL keys");
    HIPCHK(h, hipSetDevice(h->device));
    return load_filter_core(h, (const uint64_t *)d_keys_lo, (const uint64_t *)d_keys_hi, n);
}

int kdf_reset_counts(kdf_engine *h) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = pending_drop(h);                         // counts that were never applied need not be
    if (rc) return rc;
    if ((rc = materialize(h))) return rc;
    HIPCHK(h, hipMemsetAsync(h->t.cnt, 0, h->cap * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(h->ctl->windows, 0, sizeof(h->ctl->windows), h->stream));
    h->windows = 0;
    return KDF_OK;
}

static int merge_reserve(kdf_engine *h, size_t bytes) {
    if (h->merge_bytes >= bytes) return KDF_OK;
    if (h->merge_buf) { (void)hipStreamSynchronize(h->stream); (void)hipFree(h->merge_buf); h->merge_buf = nullptr; h->merge_bytes = 0; }
    HIPCHK(h, hipMalloc(&h->merge_buf, bytes));
    h->merge_bytes = bytes;
    return KDF_OK;
}

// insert-or-add (key, count) pairs resident in HBM, in `nseg` segments (one per source rank of a merge); grows the
// table first so the pairs fit at load <= 0.5 even if all of them are new.  Hash-layout tables take segments of any
// order: km_bounds_kernel tests on the device whether every segment is grouped by table bucket (a dump made by
// kdf_export_parts_dev is) and the LDS bucket merge or the atomic insert runs accordingly -- no host decision.
static int add_pairs_multi(kdf_engine *h, uint32_t nseg, const uint64_t *const *d_lo, const uint64_t *const *d_hi,
                           const uint32_t *const *d_cnt, const uint64_t *n) {
    uint64_t total = 0, nmax = 0;
    bool counts = true;
    for (uint32_t s = 0; s < nseg; ++s) { total += n[s]; nmax = std::max(nmax, n[s]); if (n[s] && !d_cnt[s]) counts = false; }
    if (total == 0) return KDF_OK;
    h->sieve_valid = false;                          // keys may join the table that the sieve has not seen
    int rc;
    if ((rc = pending_flush(h))) return rc;
    if ((rc = ctl_sync(h, nullptr))) return rc;
    const uint32_t want = cap_log2_for(h->distinct + total);
    if (want > h->t.log2cap) {
        if (h->lazy_empty) {                         // nothing to carry over: a new table, still to be cleared
            KdfTable nt;
            if ((rc = table_alloc(h, want, nt, false))) return rc;
            HIPCHK(h, hipStreamSynchronize(h->stream));          // (nothing in flight may still touch the old arrays)
            table_free(h->t);
            h->t = nt; h->t.key_parts = h->opt_key_parts; h->t.key_part = h->opt_key_part;
            h->cap = 1ull << want;
        } else if ((rc = table_rehash(h, want))) return rc;
    }
    const bool lds = counts && total >= h->opt_merge_min_pairs && nseg <= KM_MAX_SEGS && nmax < 0xFFFFFFFFull;
    if (!lds) {
        if ((rc = materialize(h))) return rc;
        for (uint32_t s = 0; s < nseg; ++s) {
            if (!n[s]) continue;
            const unsigned blocks = (unsigned)((n[s] + 255) / 256);
            if (h->kw == 1)
                hipLaunchKernelGGL(kdf_insert_keys_kernel<1>, dim3(blocks), dim3(256), 0, h->stream, d_lo[s], (const uint64_t *)nullptr, d_cnt[s], n[s], h->t, h->ctl, 0, 0);
            else
                hipLaunchKernelGGL(kdf_insert_keys_kernel<2>, dim3(blocks), dim3(256), 0, h->stream, d_lo[s], d_hi[s], d_cnt[s], n[s], h->t, h->ctl, 0, 0);
        }
        h->last_merge_path = 2;
    } else {
        KmSegs sg{};
        uint32_t m = 0;
        for (uint32_t s = 0; s < nseg; ++s) {
            if (!n[s]) continue;
            sg.lo[m] = d_lo[s]; sg.hi[m] = h->kw == 2 ? d_hi[s] : nullptr; sg.cnt[m] = d_cnt[s]; sg.n[m] = (uint32_t)n[s]; ++m;
        }
        sg.nseg = m;
        const uint32_t nb = (uint32_t)(h->cap >> h->t.bucket_bits);
        const size_t words = (size_t)m * nb * 2 + 16;
        if ((rc = merge_reserve(h, words * 4))) return rc;
        uint32_t *first = (uint32_t *)h->merge_buf, *last = first + (size_t)m * nb, *flag = last + (size_t)m * nb;
        HIPCHK(h, hipMemsetAsync(h->merge_buf, 0, words * 4, h->stream));
        const dim3 pg((unsigned)((nmax + 255) / 256), m);
        const size_t lds_bytes = ((size_t)8 * h->kw + 4) << h->t.bucket_bits;
        const bool fresh = h->lazy_empty;            // the kernel writes every bucket: it IS the deferred clear
        if (lds_bytes > 65536 && !h->merge_attrs_set[h->kw]) {       // big buckets: past the default limit of dynamic LDS
            int rca = by_width(h, [&](auto KWc) {
                constexpr int KW = decltype(KWc)::value;
                HIPCHK(h, hipFuncSetAttribute((const void *)(km_merge_kernel<KW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                HIPCHK(h, hipFuncSetAttribute((const void *)(km_merge_kernel<KW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                return KDF_OK;
            });
            if (rca) return rca;
            h->merge_attrs_set[h->kw] = true;
        }
        by_width(h, [&](auto KWc) {
            constexpr int KW = decltype(KWc)::value;
            hipLaunchKernelGGL(km_bounds_kernel<KW>, pg, dim3(256), 0, h->stream, h->t, sg, nb, first, last, flag);
            if (fresh) hipLaunchKernelGGL((km_merge_kernel<KW, true>), dim3(nb), dim3(KM_MERGE_THREADS), lds_bytes, h->stream, h->t, sg, nb, first, last, flag, h->ctl);
            else hipLaunchKernelGGL((km_merge_kernel<KW, false>), dim3(nb), dim3(KM_MERGE_THREADS), lds_bytes, h->stream, h->t, sg, nb, first, last, flag, h->ctl);
            hipLaunchKernelGGL(km_insert_guarded_kernel<KW>, pg, dim3(256), 0, h->stream, h->t, sg, flag, h->ctl);
            return 0;
        });
        h->lazy_empty = false;
        h->last_merge_path = 1;
        HIPCHK(h, hipMemcpyAsync(&h->merge_flag_host, flag, 4, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipGetLastError());
    bool full = false;
    if ((rc = ctl_sync(h, &full))) return rc;
    if (h->last_merge_path == 1 && h->merge_flag_host) h->last_merge_path = 2;     // a segment was not grouped: the atomic kernel did the work
    if (full) return fail(h, KDF_ERR_TABLE_FULL, "kdf_add_pairs: bucket overflow");
    return KDF_OK;
}

static int add_pairs_dev(kdf_engine *h, const uint64_t *d_lo, const uint64_t *d_hi, const uint32_t *d_cnt, uint64_t n) {
    return add_pairs_multi(h, 1, &d_lo, &d_hi, &d_cnt, &n);
}

int kdf_add_pairs_multi_dev(kdf_engine *h, uint32_t nseg, const void *const *d_keys_lo, const void *const *d_keys_hi,
                            const void *const *d_counts, const uint64_t *n) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (nseg == 0) return KDF_OK;
    if (!d_keys_lo || !n || !d_counts) return fail(h, KDF_ERR_INVALID, "kdf_add_pairs_multi_dev: NULL pointer");
    if (h->kw == 2 && !d_keys_hi) return fail(h, KDF_ERR_INVALID, "kdf_add_pairs_multi_dev: wide keys need the hi words");
    std::vector<const uint64_t *> lo(nseg), hi(nseg, nullptr);
    std::vector<const uint32_t *> cnt(nseg);
    for (uint32_t s = 0; s < nseg; ++s) {
        lo[s] = (const uint64_t *)d_keys_lo[s]; cnt[s] = (const uint32_t *)d_counts[s];
        if (h->kw == 2) hi[s] = (const uint64_t *)d_keys_hi[s];
        if (n[s] && (!lo[s] || (h->kw == 2 && !hi[s]))) return fail(h, KDF_ERR_INVALID, "kdf_add_pairs_multi_dev: segment %u has NULL keys", s);
    }
    HIPCHK(h, hipSetDevice(h->device));
    return add_pairs_multi(h, nseg, lo.data(), hi.data(), cnt.data(), n);
}

int kdf_add_pairs_dev(kdf_engine *h, const void *d_keys_lo, const void *d_keys_hi, const void *d_counts, uint64_t n) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n && (!d_keys_lo || (h->kw == 2 && !d_keys_hi))) return fail(h, KDF_ERR_INVALID, "kdf_add_pairs_dev: NULL keys");
    HIPCHK(h, hipSetDevice(h->device));
    return add_pairs_dev(h, (const uint64_t *)d_keys_lo, (const uint64_t *)d_keys_hi, (const uint32_t *)d_counts, n);
}

int kdf_add_pairs(kdf_engine *h, const uint64_t *keys_lo, const uint64_t *keys_hi, const uint32_t *counts, uint64_t n) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n == 0) return KDF_OK;
    if (!keys_lo || (h->kw == 2 && !keys_hi)) return fail(h, KDF_ERR_INVALID, "kdf_add_pairs: NULL keys");
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = stage_reserve(h, 2, n * 8))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->stage[2], keys_lo, n * 8, hipMemcpyHostToDevice, h->stream));
    if (h->kw == 2) {
        if ((rc = stage_reserve(h, 3, n * 8))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->stage[3], keys_hi, n * 8, hipMemcpyHostToDevice, h->stream));
    }
    if (counts) {
        if ((rc = stage_reserve(h, 0, n * 4))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->stage[0], counts, n * 4, hipMemcpyHostToDevice, h->stream));
    }
    return add_pairs_dev(h, (const uint64_t *)h->stage[2], (const uint64_t *)h->stage[3],
                         counts ? (const uint32_t *)h->stage[0] : nullptr, n);
}

int kdf_set_counts_dev(kdf_engine *h, const void *d_keys_lo, const void *d_keys_hi, const void *d_counts, uint64_t n) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n == 0) return KDF_OK;
    if (!d_keys_lo || !d_counts || (h->kw == 2 && !d_keys_hi)) return fail(h, KDF_ERR_INVALID, "kdf_set_counts_dev: NULL pointer");
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = pending_flush(h))) return rc;
    if ((rc = materialize(h))) return rc;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (h->kw == 1) hipLaunchKernelGGL(kdf_set_counts_kernel<1>, dim3(blocks), dim3(256), 0, h->stream, (const uint64_t *)d_keys_lo, (const uint64_t *)nullptr, (const uint32_t *)d_counts, n, h->t, h->ctl);
    else hipLaunchKernelGGL(kdf_set_counts_kernel<2>, dim3(blocks), dim3(256), 0, h->stream, (const uint64_t *)d_keys_lo, (const uint64_t *)d_keys_hi, (const uint32_t *)d_counts, n, h->t, h->ctl);
    HIPCHK(h, hipGetLastError());
    bool bad = false;
    if ((rc = ctl_sync(h, &bad))) return rc;
    if (bad) {
        HIPCHK(h, hipMemsetAsync(&h->ctl->error, 0, 4, h->stream));
        return fail(h, KDF_ERR_INVALID, "kdf_set_counts_dev: a key is not stored in the table");
    }
    return KDF_OK;
}

int kdf_count_reads_filtered_dev(kdf_engine *h, const void *d_packed, const void *d_invalid, uint64_t n_bases) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n_bases && (!d_packed || !d_invalid)) return fail(h, KDF_ERR_INVALID, "kdf_count_reads_filtered_dev: NULL stream");
    HIPCHK(h, hipSetDevice(h->device));
    return count_filtered_dev(h, (const uint64_t *)d_packed, (const uint64_t *)d_invalid, n_bases);
}

int kdf_count_reads_filtered(kdf_engine *h, const uint64_t *packed, const uint64_t *invalid, uint64_t n_bases) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (!h->filter_mode) return fail(h, KDF_ERR_STATE, "kdf_count_reads_filtered: no filter loaded (kdf_load_filter)");
    if (n_bases == 0) return KDF_OK;
    if (!packed || !invalid) return fail(h, KDF_ERR_INVALID, "kdf_count_reads_filtered: NULL stream");
    HIPCHK(h, hipSetDevice(h->device));
    uint64_t *dp, *dm;
    int rc = upload_stream(h, packed, invalid, n_bases, &dp, &dm);
    if (rc) return rc;
    if ((rc = count_filtered_dev(h, dp, dm, n_bases))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));   // staging buffers are reused by the next call
    return KDF_OK;
}

int kdf_query_dev(kdf_engine *h, const void *d_keys_lo, const void *d_keys_hi, uint64_t n, void *d_counts_out) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n == 0) return KDF_OK;
    if (!d_keys_lo || !d_counts_out || (h->kw == 2 && !d_keys_hi)) return fail(h, KDF_ERR_INVALID, "kdf_query_dev: NULL pointer");
    HIPCHK(h, hipSetDevice(h->device));
    { int rcf = pending_flush(h); if (rcf) return rcf; }
    { int rc0 = materialize(h); if (rc0) return rc0; }
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (h->kw == 1)
        hipLaunchKernelGGL(kdf_query_kernel<1>, dim3(blocks), dim3(256), 0, h->stream,
                           (const uint64_t *)d_keys_lo, (const uint64_t *)nullptr, n, h->t, (uint32_t *)d_counts_out);
    else
        hipLaunchKernelGGL(kdf_query_kernel<2>, dim3(blocks), dim3(256), 0, h->stream,
                           (const uint64_t *)d_keys_lo, (const uint64_t *)d_keys_hi, n, h->t, (uint32_t *)d_counts_out);
    HIPCHK(h, hipGetLastError());
    return KDF_OK;
}

int kdf_query(kdf_engine *h, const uint64_t *keys_lo, const uint64_t *keys_hi, uint64_t n, uint32_t *counts_out) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n == 0) return KDF_OK;
    if (!keys_lo || !counts_out || (h->kw == 2 && !keys_hi)) return fail(h, KDF_ERR_INVALID, "kdf_query: NULL pointer");
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = stage_reserve(h, 2, n * 8))) return rc;
    if ((rc = stage_reserve(h, 0, n * 4))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->stage[2], keys_lo, n * 8, hipMemcpyHostToDevice, h->stream));
    if (h->kw == 2) {
        if ((rc = stage_reserve(h, 3, n * 8))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->stage[3], keys_hi, n * 8, hipMemcpyHostToDevice, h->stream));
    }
    if ((rc = kdf_query_dev(h, h->stage[2], h->stage[3], n, h->stage[0]))) return rc;
    HIPCHK(h, hipMemcpyAsync(counts_out, h->stage[0], n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return KDF_OK;
}

static int export_pass(kdf_engine *h, uint32_t min_count, bool write, uint64_t *olo, uint64_t *ohi,
                       uint32_t *ocnt, uint64_t out_cap, uint64_t *n_out) {
    // passes pending: the flush that applies them holds every bucket of the table in LDS once -- it writes the dump too
    // (kb_bucket_kernel, KbPlan::dump_min) unless a bucket took another way (overflow replay, heavy-bucket split)
    const bool fuse = write && min_count >= 1 && h->opt_fused_dump && !h->filter_mode && olo && (h->kw == 1 || ohi) && (h->n_pass > 0 || h->l1_tiles > 0);
    if (fuse) { h->fuse_min = min_count; h->fuse_lo = olo; h->fuse_hi = ohi; h->fuse_cnt = ocnt; h->fuse_cap = out_cap; }
    { int rcf = pending_flush(h, fuse); if (rcf) return rcf; }
    if (fuse && h->fuse_done) { h->fuse_done = false; *n_out = h->fuse_n; return KDF_OK; }
    { int rc0 = materialize(h); if (rc0) return rc0; }
    HIPCHK(h, hipMemsetAsync(h->ctl->tally, 0, sizeof(h->ctl->tally) + 8, h->stream));   // tally[] + cursor
    const uint64_t waves = (h->cap + KDF_EXPORT_ROWS * 64 - 1) / (KDF_EXPORT_ROWS * 64);
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    if (write) {                                            // one read of the table (any min_count: occupancy is tested on the key)
        const unsigned rows = h->kw == 1 ? KDF_EXPORT1_ROWS : KDF_EXPORT1_ROWS / 2;
        const uint64_t w1 = (h->cap + rows * 64 - 1) / (rows * 64);
        const unsigned wpb = KDF_EXPORT1_THREADS / 64;
        if (h->kw == 1) hipLaunchKernelGGL(kdf_export1_kernel<1>, dim3((unsigned)((w1 + wpb - 1) / wpb)), dim3(KDF_EXPORT1_THREADS), 0, h->stream, h->t, min_count, h->ctl, olo, ohi, ocnt, out_cap);
        else hipLaunchKernelGGL(kdf_export1_kernel<2>, dim3((unsigned)((w1 + wpb - 1) / wpb)), dim3(KDF_EXPORT1_THREADS), 0, h->stream, h->t, min_count, h->ctl, olo, ohi, ocnt, out_cap);
    } else if (h->kw == 1) {
        hipLaunchKernelGGL((kdf_export_kernel<1, false>), dim3(blocks), dim3(256), 0, h->stream, h->t, min_count, h->ctl, olo, ohi, ocnt, out_cap);
    } else {
        hipLaunchKernelGGL((kdf_export_kernel<2, false>), dim3(blocks), dim3(256), 0, h->stream, h->t, min_count, h->ctl, olo, ohi, ocnt, out_cap);
    }
    HIPCHK(h, hipGetLastError());
    uint64_t cursor = 0;
    int rc = ctl_sync(h, nullptr, &cursor);
    if (rc) return rc;
    *n_out = cursor;
    return KDF_OK;
}

// The sender's half of the multi-GPU merge (kdf_merge.h): the table in hash order, one contiguous range per owner.
// packed: d_lo is the byte buffer of the packed layout, cap its size in bytes, part_bytes_out[parts + 1] the segment offsets.
static int export_parts(kdf_engine *h, uint32_t min_count, uint32_t parts, bool packed, void *d_lo, void *d_hi, void *d_cnt,
                        uint64_t cap, uint64_t *part_counts_out, uint64_t *part_bytes_out, uint64_t *n_out, const char *who) {
    if (parts < 1 || parts > KDF_SHARDS) return fail(h, KDF_ERR_INVALID, "%s: parts must be 1..%d", who, KDF_SHARDS);
    HIPCHK(h, hipSetDevice(h->device));
    { int rcf = pending_flush(h); if (rcf) return rcf; }
    { int rc0 = materialize(h); if (rc0) return rc0; }
    if (h->t.hshift) return fail(h, KDF_ERR_STATE, "%s: an owner table (hash_shift) is not dumped by owner again", who);
    if (h->t.log2cap < 28)                                  // an owner boundary (a 16-bit hash prefix) must be a block boundary
        return fail(h, KDF_ERR_STATE, "%s: table of 2^%u slots is too small for an owner-ordered dump (needs 2^28)", who, h->t.log2cap);
    const bool have_out = d_lo && (packed || h->kw == 1 || d_hi);
    const uint64_t nblk = h->cap / KM_BLOCK_SLOTS;
    // scratch: blk_off u64[nblk + 1] | part_first u64[S] | part_off u64[S + 1] | part_base u64[S + 1] | blk_cnt u32[nblk]
    const size_t off_words = nblk + 1 + KDF_SHARDS + 2 * (KDF_SHARDS + 1);
    int rc = merge_reserve(h, off_words * 8 + nblk * 4);
    if (rc) return rc;
    unsigned long long *blk_off = (unsigned long long *)h->merge_buf;
    uint64_t *part_first = (uint64_t *)(blk_off + nblk + 1);
    unsigned long long *part_off = (unsigned long long *)(part_first + KDF_SHARDS);
    unsigned long long *part_base = part_off + KDF_SHARDS + 1;
    uint32_t *blk_cnt = (uint32_t *)(part_base + KDF_SHARDS + 1);
    std::vector<uint64_t> pf(KDF_SHARDS, 0);
    for (uint32_t p = 0; p < parts; ++p) {                  // first 16-bit hash prefix t with ((t * parts) >> 16) == p
        const uint64_t t16 = ((uint64_t)p * 65536 + parts - 1) / parts;
        pf[p] = (t16 << (h->t.log2cap - 16)) / KM_BLOCK_SLOTS;
    }
    HIPCHK(h, hipMemcpyAsync(part_first, pf.data(), KDF_SHARDS * 8, hipMemcpyHostToDevice, h->stream));
    std::vector<unsigned long long> po(2 * (KDF_SHARDS + 1), 0);
    const uint32_t esz = 8u * h->kw + 4u;
    by_width(h, [&](auto KWc) {
        constexpr int KW = decltype(KWc)::value;
        hipLaunchKernelGGL(km_count_kernel<KW>, dim3((unsigned)nblk), dim3(KM_THREADS), 0, h->stream, h->t, min_count, blk_cnt);
        hipLaunchKernelGGL(km_scan_kernel, dim3(1), dim3(1024), 0, h->stream, blk_cnt, blk_off, nblk, part_first, parts, part_off);
        hipLaunchKernelGGL(km_packbase_kernel, dim3(1), dim3(64), 0, h->stream, part_off, parts, esz, part_base);
        if (have_out && packed)
            hipLaunchKernelGGL((km_write_kernel<KW, true>), dim3((unsigned)nblk), dim3(KM_THREADS), 0, h->stream, h->t, min_count, blk_off,
                               (uint64_t *)d_lo, (uint64_t *)nullptr, (uint32_t *)nullptr, cap, part_off, part_base, parts);
        else if (have_out)
            hipLaunchKernelGGL((km_write_kernel<KW, false>), dim3((unsigned)nblk), dim3(KM_THREADS), 0, h->stream, h->t, min_count, blk_off,
                               (uint64_t *)d_lo, h->kw == 2 ? (uint64_t *)d_hi : nullptr, (uint32_t *)d_cnt, cap, part_off, part_base, parts);
        return 0;
    });
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(po.data(), part_off, 2 * (KDF_SHARDS + 1) * 8, hipMemcpyDeviceToHost, h->stream));   // part_off | part_base
    HIPCHK(h, hipStreamSynchronize(h->stream));             // the one synchronisation of the dump (pf / po are pageable)
    const uint64_t n = po[parts];
    for (uint32_t p = 0; p < parts; ++p) part_counts_out[p] = po[p + 1] - po[p];
    if (part_bytes_out) for (uint32_t p = 0; p <= parts; ++p) part_bytes_out[p] = po[KDF_SHARDS + 1 + p];
    *n_out = n;
    if (n == 0) return KDF_OK;
    const uint64_t need = packed ? po[KDF_SHARDS + 1 + parts] : n;
    if (need > cap) return fail(h, KDF_ERR_INVALID, "%s: %llu %s, room for %llu", who, (unsigned long long)need, packed ? "bytes" : "entries", (unsigned long long)cap);
    if (!have_out) return fail(h, KDF_ERR_INVALID, "%s: NULL output", who);
    return KDF_OK;
}

int kdf_export_parts_dev(kdf_engine *h, uint32_t min_count, uint32_t parts, void *d_keys_lo_out, void *d_keys_hi_out,
                         void *d_counts_out, uint64_t cap, uint64_t *part_counts_out, uint64_t *n_out) {
    if (!h || !n_out || !part_counts_out) return fail(h, KDF_ERR_INVALID, "kdf_export_parts_dev: NULL pointer");
    return export_parts(h, min_count, parts, false, d_keys_lo_out, d_keys_hi_out, d_counts_out, cap, part_counts_out, nullptr, n_out,
                        "kdf_export_parts_dev");
}

int kdf_export_parts_packed_dev(kdf_engine *h, uint32_t min_count, uint32_t parts, void *d_buf, uint64_t cap_bytes,
                                uint64_t *part_counts_out, uint64_t *part_bytes_out, uint64_t *n_out) {
    if (!h || !n_out || !part_counts_out || !part_bytes_out) return fail(h, KDF_ERR_INVALID, "kdf_export_parts_packed_dev: NULL pointer");
    return export_parts(h, min_count, parts, true, d_buf, nullptr, nullptr, cap_bytes, part_counts_out, part_bytes_out, n_out,
                        "kdf_export_parts_packed_dev");
}

int kdf_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!free_bytes || !total_bytes) return fail(nullptr, KDF_ERR_INVALID, "kdf_device_memory: NULL pointer");
    size_t f = 0, t = 0;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMemGetInfo(&f, &t);
    if (e != hipSuccess) return fail(nullptr, KDF_ERR_HIP, "kdf_device_memory: %s", hipGetErrorString(e));
    *free_bytes = f; *total_bytes = t;
    return KDF_OK;
}

int kdf_count_ge(kdf_engine *h, uint32_t min_count, uint64_t *n_out) {
    if (!h || !n_out) return fail(h, KDF_ERR_INVALID, "kdf_count_ge: NULL pointer");
    HIPCHK(h, hipSetDevice(h->device));
    return export_pass(h, min_count, false, nullptr, nullptr, nullptr, 0, n_out);
}

int kdf_export_ge(kdf_engine *h, uint32_t min_count, uint64_t *keys_lo_out, uint64_t *keys_hi_out,
                  uint32_t *counts_out, uint64_t cap, uint64_t *n_out) {
    if (!h || !n_out) return fail(h, KDF_ERR_INVALID, "kdf_export_ge: NULL pointer");
    HIPCHK(h, hipSetDevice(h->device));
    uint64_t n = 0;
    int rc = export_pass(h, min_count, false, nullptr, nullptr, nullptr, 0, &n);
    if (rc) return rc;
    *n_out = n;
    if (n == 0) return KDF_OK;
    if (n > cap) return fail(h, KDF_ERR_INVALID, "kdf_export_ge: %llu entries, room for %llu",
                             (unsigned long long)n, (unsigned long long)cap);
    if (!keys_lo_out || (h->kw == 2 && !keys_hi_out)) return fail(h, KDF_ERR_INVALID, "kdf_export_ge: NULL key output");
    if ((rc = stage_reserve(h, 2, n * 8))) return rc;
    if ((rc = stage_reserve(h, 0, n * 4))) return rc;
    if (h->kw == 2 && (rc = stage_reserve(h, 3, n * 8))) return rc;
    uint64_t n2 = 0;
    rc = export_pass(h, min_count, true, (uint64_t *)h->stage[2], h->kw == 2 ? (uint64_t *)h->stage[3] : nullptr,
                     (uint32_t *)h->stage[0], n, &n2);
    if (rc) return rc;
    if (n2 != n) return fail(h, KDF_ERR_STATE, "kdf_export_ge: table changed between passes");
    std::string serr;
    if (kdf_sort_pairs_device((uint64_t *)h->stage[2], h->kw == 2 ? (uint64_t *)h->stage[3] : nullptr,
                              (uint32_t *)h->stage[0], n, h->stream, serr))
        return fail(h, KDF_ERR_HIP, "kdf_export_ge: sort failed: %s", serr.c_str());
    HIPCHK(h, hipMemcpyAsync(keys_lo_out, h->stage[2], n * 8, hipMemcpyDeviceToHost, h->stream));
    if (h->kw == 2) HIPCHK(h, hipMemcpyAsync(keys_hi_out, h->stage[3], n * 8, hipMemcpyDeviceToHost, h->stream));
    else if (keys_hi_out) memset(keys_hi_out, 0, n * 8);
    if (counts_out) HIPCHK(h, hipMemcpyAsync(counts_out, h->stage[0], n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return KDF_OK;
}

int kdf_export_ge_dev(kdf_engine *h, uint32_t min_count, void *d_keys_lo_out, void *d_keys_hi_out,
                      void *d_counts_out, uint64_t cap, int sorted, uint64_t *n_out) {
    if (!h || !n_out) return fail(h, KDF_ERR_INVALID, "kdf_export_ge_dev: NULL pointer");
    HIPCHK(h, hipSetDevice(h->device));
    if (cap && (!d_keys_lo_out || (h->kw == 2 && !d_keys_hi_out))) return fail(h, KDF_ERR_INVALID, "kdf_export_ge_dev: NULL key output");
    // ONE pass over the table: entries are appended through the cursor, nothing is written past `cap`, and the
    // cursor's final value is the number of entries the dump holds (kdf_count_ge gives it beforehand)
    uint64_t n = 0;
    int rc = export_pass(h, min_count, true, (uint64_t *)d_keys_lo_out, h->kw == 2 ? (uint64_t *)d_keys_hi_out : nullptr,
                         (uint32_t *)d_counts_out, cap, &n);
    if (rc) return rc;
    *n_out = n;
    if (n > cap) return fail(h, KDF_ERR_INVALID, "kdf_export_ge_dev: %llu entries, room for %llu",
                             (unsigned long long)n, (unsigned long long)cap);
    if (sorted && n) {
        if (!d_counts_out) return fail(h, KDF_ERR_INVALID, "kdf_export_ge_dev: sorted export needs the counts array");
        std::string serr;
        if (kdf_sort_pairs_device((uint64_t *)d_keys_lo_out, h->kw == 2 ? (uint64_t *)d_keys_hi_out : nullptr,
                                  (uint32_t *)d_counts_out, n, h->stream, serr))
            return fail(h, KDF_ERR_HIP, "kdf_export_ge_dev: sort failed: %s", serr.c_str());
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return KDF_OK;
}

int kdf_scan_reads_dev(kdf_engine *h, const void *d_packed, const void *d_invalid, uint64_t n_bases, void *d_hit_bits) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (n_bases == 0) return KDF_OK;
    if (!d_packed || !d_invalid || !d_hit_bits) return fail(h, KDF_ERR_INVALID, "kdf_scan_reads_dev: NULL pointer");
    HIPCHK(h, hipSetDevice(h->device));
    { int rcf = pending_flush(h); if (rcf) return rcf; }
    { int rc0 = materialize(h); if (rc0) return rc0; }
    const uint64_t n_tiles = (n_bases + KDF_TILE - 1) / KDF_TILE;
    if (h->opt_force_path != 1 && n_tiles * KDF_TILE < (1ull << 32)) {
        // through the membership sieve (section 3.5 of DESIGN.md): an index that was loaded with kdf_add_pairs has none yet
        int rc;
        if (!h->sieve_valid) {
            if ((rc = ctl_sync(h, nullptr))) return rc;
            if ((rc = sieve_prepare(h, h->distinct))) return rc;
            if (h->sieve_valid) {
                const unsigned tb = (unsigned)((h->cap + 255) / 256);
                if (h->kw == 1) hipLaunchKernelGGL(kdf_sieve_from_table_kernel<1>, dim3(tb), dim3(256), 0, h->stream, h->t, h->sieve, h->sieve_words - 1);
                else hipLaunchKernelGGL(kdf_sieve_from_table_kernel<2>, dim3(tb), dim3(256), 0, h->stream, h->t, h->sieve, h->sieve_words - 1);
                HIPCHK(h, hipGetLastError());
            }
        }
        if (h->sieve_valid) {
            HIPCHK(h, hipMemsetAsync(d_hit_bits, 0, n_tiles * 8, h->stream));
            const int WPT = h->kw == 1 ? KbCfg<1>::WPT : KbCfg<2>::WPT;
            const uint64_t tiles_per_slab = KB_THREADS / (64 / WPT);
            const uint64_t n_slabs = (n_tiles + tiles_per_slab - 1) / tiles_per_slab;
            const uint32_t n_wg = (uint32_t)std::min<uint64_t>(n_slabs, (uint64_t)h->n_cu * 8);
            const uint32_t spw = (uint32_t)((n_slabs + n_wg - 1) / n_wg);
            const unsigned grid = (unsigned)((n_slabs + spw - 1) / spw);
            KdfSieve sv{h->sieve, h->sieve_words - 1};
            unsigned long long *hb = (unsigned long long *)d_hit_bits;
            const bool in_lds = h->sieve_words <= KDF_SV_LDS_WORDS;
#define SV_SCAN(KWV, L) hipLaunchKernelGGL((kdf_sieve_count_kernel<KWV, L, true>), dim3(grid), dim3(KB_THREADS), 0, h->stream, (const uint64_t *)d_packed, (const uint64_t *)d_invalid, n_tiles, h->k, h->t, h->ctl, sv, spw, hb)
            if (h->kw == 1) { if (in_lds) SV_SCAN(1, true); else SV_SCAN(1, false); }
            else { if (in_lds) SV_SCAN(2, true); else SV_SCAN(2, false); }
#undef SV_SCAN
            HIPCHK(h, hipGetLastError());
            return KDF_OK;
        }
    }
    launch_stream<MODE_SCAN>(h, (const uint64_t *)d_packed, (const uint64_t *)d_invalid, 0, n_tiles, (uint64_t *)d_hit_bits);
    HIPCHK(h, hipGetLastError());
    return KDF_OK;
}

// host-side canonical key of the window at stream position p (used to count
// the DISTINCT hit k-mers of the few reads that carry hits)
static inline void host_window_key(const uint64_t *packed, uint64_t p, int k, uint64_t &klo, uint64_t &khi) {
    unsigned __int128 e = 0;
    for (int j = 0; j < k; ++j) {
        const uint64_t q = p + j;
        const unsigned __int128 b = (packed[q >> 5] >> ((q & 31) * 2)) & 3;
        e |= b << (2 * j);
    }
    const unsigned __int128 mask = (((unsigned __int128)1) << (2 * k)) - 1;
    unsigned __int128 rc = ~e & mask, fwd = 0;
    for (int j = 0; j < k; ++j) fwd |= ((e >> (2 * j)) & 3) << (2 * (k - 1 - j));
    const unsigned __int128 c = fwd < rc ? fwd : rc;
    klo = (uint64_t)c; khi = (uint64_t)(c >> 64);
}

int kdf_scan_reads(kdf_engine *h, const uint64_t *packed, const uint64_t *invalid, uint64_t n_bases,
                   const int64_t *read_offsets, int64_t n_reads, uint64_t *hit_bits, uint32_t *distinct_out) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    if (!hit_bits) return fail(h, KDF_ERR_INVALID, "kdf_scan_reads: hit_bits is NULL");
    uint64_t pw, mw;
    kdf_stream_words(n_bases, &pw, &mw);
    memset(hit_bits, 0, mw * 8);
    if (distinct_out && n_reads > 0) memset(distinct_out, 0, (size_t)n_reads * 4);
    if (n_bases == 0) return KDF_OK;
    if (!packed || !invalid) return fail(h, KDF_ERR_INVALID, "kdf_scan_reads: NULL stream");
    HIPCHK(h, hipSetDevice(h->device));
    uint64_t *dp, *dm;
    int rc = upload_stream(h, packed, invalid, n_bases, &dp, &dm);
    if (rc) return rc;
    if ((rc = stage_reserve(h, 2, mw * 8))) return rc;
    if ((rc = kdf_scan_reads_dev(h, dp, dm, n_bases, h->stage[2]))) return rc;
    const uint64_t n_tiles = (n_bases + KDF_TILE - 1) / KDF_TILE;
    HIPCHK(h, hipMemcpyAsync(hit_bits, h->stage[2], n_tiles * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!read_offsets || !distinct_out) return KDF_OK;
    // distinct hit k-mers per read: only reads with hits are touched
    std::vector<std::pair<uint64_t, uint64_t>> keys;
    for (int64_t r = 0; r < n_reads; ++r) {
        const uint64_t b = (uint64_t)read_offsets[r], e = (uint64_t)read_offsets[r + 1];
        keys.clear();
        for (uint64_t wd = b >> 6; wd <= (e ? (e - 1) >> 6 : 0) && wd < n_tiles; ++wd) {
            uint64_t bits = hit_bits[wd];
            while (bits) {
                const int bit = __builtin_ctzll(bits);
                bits &= bits - 1;
                const uint64_t p = (wd << 6) + bit;
                if (p < b || p >= e) continue;
                uint64_t lo, hi;
                host_window_key(packed, p, h->k, lo, hi);
                keys.emplace_back(hi, lo);
            }
        }
        if (keys.empty()) continue;
        std::sort(keys.begin(), keys.end());
        distinct_out[r] = (uint32_t)(std::unique(keys.begin(), keys.end()) - keys.begin());
    }
    return KDF_OK;
}

int kdf_profile(kdf_engine *h, int enable) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    prof_collect(h);
    h->prof = enable != 0;
    h->prof_ms = 0.0; h->prof_launches = 0; h->prof_positions = 0;
    for (double &m : h->prof_stage_ms) m = 0.0;
    h->prof_stage_passes = 0;
    return KDF_OK;
}

int kdf_profile_stages(kdf_engine *h, double *stage_ms4, uint64_t *passes) {
    if (!h || !stage_ms4) return fail(h, KDF_ERR_INVALID, "kdf_profile_stages: NULL pointer");
    prof_collect(h);
    for (int i = 0; i < 4; ++i) stage_ms4[i] = h->prof_stage_ms[i];
    if (passes) *passes = h->prof_stage_passes;
    return KDF_OK;
}

int kdf_profile_read(kdf_engine *h, double *kernel_ms, uint64_t *launches, uint64_t *positions) {
    if (!h) return fail(nullptr, KDF_ERR_INVALID, "NULL engine");
    prof_collect(h);
    if (kernel_ms) *kernel_ms = h->prof_ms;
    if (launches) *launches = h->prof_launches;
    if (positions) *positions = h->prof_positions;
    return KDF_OK;
}

int kdf_set_option(kdf_engine *h, const char *name, int64_t value) {
    if (!h || !name) return fail(h, KDF_ERR_INVALID, "kdf_set_option: NULL argument");
    const std::string n(name);
    // options change how the NEXT windows are counted: what is pending was counted under the old ones
    if (n != "debug_flags" && n != "defer_max_bytes") { HIPCHK(h, hipSetDevice(h->device)); int rcf = pending_flush(h); if (rcf) return rcf; }
    if (n == "key_parts" || n == "key_part") {
        const uint32_t parts = n == "key_parts" ? (uint32_t)value : h->opt_key_parts, part = n == "key_part" ? (uint32_t)value : h->opt_key_part;
        if (value < 0 || parts > 65536 || (n == "key_part" && part >= std::max<uint32_t>(parts, 1)))
            return fail(h, KDF_ERR_INVALID, "key_parts must be 0..65536 and key_part below it");
        h->opt_key_parts = parts; h->opt_key_part = n == "key_parts" ? 0 : part;
    }
    else if (n == "binned_min_positions") h->opt_binned_min_positions = (uint64_t)value;
    else if (n == "binned_bytes_per_position") h->opt_binned_bytes_per_position = (uint64_t)value;
    else if (n == "binned_max_positions") {
        if (value < KDF_TILE || value > (1ll << 31)) return fail(h, KDF_ERR_INVALID, "binned_max_positions must be in [64, 2^31]");
        h->opt_binned_max_positions = (uint64_t)value / KDF_TILE * KDF_TILE;      // passes start on tile boundaries
    }
    else if (n == "binned_filtered_min_log2cap") h->opt_binned_filtered_min_log2cap = (uint32_t)value;
    else if (n == "big_bucket_log2cap") {
        if (value < 10 || value > 64) return fail(h, KDF_ERR_INVALID, "big_bucket_log2cap must be 10..64");
        h->opt_big_bucket_log2cap = (uint32_t)value;
        const uint32_t bb = std::min<uint32_t>(h->t.log2cap, KB_BB_SMALL(h->kw) + (h->t.log2cap >= h->opt_big_bucket_log2cap ? 1u : 0u));
        if (bb != h->t.bucket_bits) { int rc = table_rehash(h, h->t.log2cap); if (rc) return rc; }     // same slots, other buckets
    }
    else if (n == "force_path") h->opt_force_path = (int)value;
    else if (n == "merge_min_pairs") h->opt_merge_min_pairs = (uint64_t)value;
    else if (n == "hash_shift") {
        if (value > 8) return fail(h, KDF_ERR_INVALID, "hash_shift must be 0..8");
        if ((uint32_t)value != h->opt_hash_shift) {
            int rc = ctl_sync(h, nullptr);
            if (rc) return rc;
            if (h->distinct) return fail(h, KDF_ERR_STATE, "hash_shift can only change on an empty table (kdf_clear first)");
        }
        h->opt_hash_shift = (uint32_t)value; h->t.hshift = (uint32_t)value;
    }
    else if (n == "defer") h->opt_defer = value != 0;
    else if (n == "defer_max_bytes") h->opt_defer_max_bytes = (uint64_t)value;
    else if (n == "fused_dump") h->opt_fused_dump = value != 0;
    else if (n == "l1_positions") h->opt_l1_positions = (uint64_t)std::max<int64_t>(value, KDF_TILE);
    else if (n == "l1_direct_positions") h->opt_l1_direct_positions = (uint64_t)std::max<int64_t>(value, 0);
    else if (n == "sieve_bits") h->opt_sieve_bits = (int)value;
    else if (n == "debug_flags") h->opt_debug_flags = (uint32_t)value;
    else return fail(h, KDF_ERR_INVALID, "kdf_set_option: unknown option %s", name);
    return KDF_OK;
}

int kdf_get_stat(kdf_engine *h, const char *name, int64_t *value) {
    if (!h || !name || !value) return fail(h, KDF_ERR_INVALID, "kdf_get_stat: NULL argument");
    const std::string n(name);
    if (n == "binned_passes") *value = (int64_t)h->stat_binned_passes;
    else if (n == "replayed_buckets") *value = (int64_t)h->stat_replayed_buckets;
    else if (n == "flushes") *value = (int64_t)h->stat_flushes;
    else if (n == "pending_passes") *value = (int64_t)h->n_pass;
    else if (n == "pending_positions") *value = (int64_t)(h->pend_positions + h->l1_tiles * KDF_TILE);
    else if (n == "ring_bytes") *value = (int64_t)(h->ring_entries * 8 * h->kw);
    else if (n == "defer") *value = h->opt_defer;
    else if (n == "last_count_path") *value = h->last_path;
    else if (n == "last_merge_path") *value = h->last_merge_path;
    else if (n == "heavy_buckets") *value = (int64_t)h->stat_heavy_buckets;
    else if (n == "fused_dumps") *value = (int64_t)h->stat_fused_dumps;
    else if (n == "fused_dump") *value = h->opt_fused_dump;
    else if (n == "hash_shift") *value = h->opt_hash_shift;
    else if (n == "log2cap") *value = h->t.log2cap;
    else if (n == "bucket_bits") *value = h->t.bucket_bits;
    else if (n.rfind("trash", 0) == 0 && n.size() > 5 && h->kb_small) {          // (variant builds with -DKB_TIMING: phase cycle sums)
        const int i = atoi(n.c_str() + 5);
        if (i < 0 || i >= 64) return fail(h, KDF_ERR_INVALID, "kdf_get_stat: trash0..trash63");
        unsigned long long v = 0;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpy(&v, h->kb_small + 16 + i, 8, hipMemcpyDeviceToHost));
        *value = (int64_t)v;
    }
    else return fail(h, KDF_ERR_INVALID, "kdf_get_stat: unknown stat %s", name);
    return KDF_OK;
}

}  // extern "C"
