// kdf_device.h -- device-side primitives shared by the kernels of libkdf.so:
// key type, hash, window extraction from the 2-bit stream, table views.
// gfx950 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KDF_EMPTY   0xFFFFFFFFFFFFFFFFull
#define KDF_PENDING 0x8000000000000000ull   // wide keys: hi word claimed, lo not yet published
#define KDF_TILE    64                      // window starts per thread = one mask word
#define KDF_SHARDS  64                      // sharded statistics counters

// ---------------------------------------------------------------------------
// hash: fold the high half into the low half, then ONE 64-bit multiply by an odd
// constant (Fibonacci hashing).  Only the TOP bits of the result are ever used for
// addressing (bucket = top bits, home slot = the bits below), and those depend on
// every input bit.  Bijective on 64 bits (xor-shift and odd multiply are both
// invertible: kdf_unmix64).
//
// STORED FORM.  Because the hash is a bijection, the engine never carries a key
// next to its hash: partition entries and table slots hold h = kdf_mix64(key) in
// place of the key (wide keys: (h, hi) with h = kdf_hash(lo, hi); lo comes back as
// kdf_unmix64(h) ^ rot(hi)), every stage takes its bits straight from h, and the
// key is recovered only where it leaves the engine (dump / export).  A 64-bit
// multiply is ~8 quarter-rate VALU instructions on CDNA4 and rounds 1-2 evaluated
// it four times per window along the binned pipeline.
//
// KDF_EMPTY (all ones) must not be the stored form of a key: with this multiplier
// kdf_unmix64(~0) = 0xfd54638d0fbbb8de, which is >= 2^62 (not a key for k <= 31)
// and, read as a 32-mer, starts with T and ends with G -- its reverse complement
// starts with C and is smaller, so it is not canonical either (k = 32).
// (0x9E3779B97F4A7C15, used in rounds 1-2, maps a valid canonical 31-/32-mer there.)
#define KDF_MIX_MUL     0x9FB21C651E98DF25ull
#define KDF_MIX_MUL_INV 0x02ab9c720d1024adull   /* KDF_MIX_MUL * KDF_MIX_MUL_INV == 1 (mod 2^64) */
__host__ __device__ __forceinline__ uint64_t kdf_mix64(uint64_t x) {
    x ^= x >> 32;
    x *= KDF_MIX_MUL;
    return x;
}
__host__ __device__ __forceinline__ uint64_t kdf_unmix64(uint64_t h) {
    h *= KDF_MIX_MUL_INV;
    return h ^ (h >> 32);
}
// wide keys: rotate hi so that its used (low) bits land on lo's upper half
__host__ __device__ __forceinline__ uint64_t kdf_rot_hi(uint64_t hi) { return (hi << 37) | (hi >> 27); }
__host__ __device__ __forceinline__ uint64_t kdf_hash(uint64_t lo, uint64_t hi) {
    return kdf_mix64(lo ^ kdf_rot_hi(hi));
}
// the low key word back from the stored form (hi = 0 for narrow keys)
__host__ __device__ __forceinline__ uint64_t kdf_key_lo(uint64_t h, uint64_t hi) {
    return kdf_unmix64(h) ^ kdf_rot_hi(hi);
}

// reverse the 32 two-bit groups of a 64-bit word
__host__ __device__ __forceinline__ uint64_t kdf_rev2(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x = __builtin_bitreverse64(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = __builtin_bswap64(x);
#endif
    return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

// ---------------------------------------------------------------------------
// Table view.  Open addressing, SoA: lo[cap] (+ hi[cap] for wide keys) and
// cnt[cap].  lo[] holds the STORED FORM h of the key (above), hi[] the key's high
// word as it is.  A key's home slot is the TOP log2cap bits of h; probing is
// linear and wraps inside the key's bucket of 2^bucket_bits slots, so a bucket
// (keys + counts) is a self-contained unit that an LDS-staged kernel can own.
struct KdfTable {
    uint64_t *lo;          // stored form h (narrow: kdf_mix64(key); wide: kdf_hash(lo, hi))
    uint64_t *hi;          // nullptr for k <= 32
    uint32_t *cnt;
    uint32_t log2cap;
    uint32_t bucket_bits;  // <= log2cap
    // Counting in key-space slices ("key_parts" option): an INSERT-mode count only takes the windows whose key
    // belongs to slice key_part of key_parts (kdf_slice below), so a sample whose distinct k-mers do not fit
    // one table is counted in several passes over the same stream.
    uint32_t key_parts;    // 0 or 1: everything
    uint32_t key_part;
    // Owner tables of the multi-GPU merge (option "hash_shift"): rank r of 2^w only ever sees keys whose top w hash
    // bits are r, so the home slot drops them -- the table is used over its whole length, and a dump that arrives in
    // the sender's slot order is still in this table's slot order.
    uint32_t hshift;
};

// Slice of a key from the LOW 16 bits of its hash.  (The multi-GPU owner function uses the TOP bits, which are
// the top bits of the home slot: owners are contiguous slot ranges, ideal for the owner-ordered dump and exactly
// wrong here -- a slice must spread over the whole table, or the table would be 1/parts full when it overflows.)
__host__ __device__ __forceinline__ uint32_t kdf_slice(uint64_t hash, uint32_t parts) {
    return (uint32_t)(((hash & 0xFFFFu) * parts) >> 16);
}

struct KdfCtl {            // device-resident control block (one per engine)
    unsigned long long distinct[KDF_SHARDS * 16];  // one 128-B line per shard
    unsigned long long windows[KDF_SHARDS * 16];
    unsigned long long tally[KDF_SHARDS * 16];     // sharded count-only tally (dump -L counting pass)
    unsigned long long cursor;                     // export append cursor
    unsigned int error;                            // != 0: a bucket overflowed
    unsigned int pad;
};

__device__ __forceinline__ uint64_t kdf_home(const KdfTable &t, uint64_t h) {
    return (h << t.hshift) >> (64 - t.log2cap);
}

__device__ __forceinline__ void kdf_sat_add(uint32_t *p, uint32_t add) {
    // saturating uint32 add (Jellyfish's output counter is 4 bytes): a wrapping
    // add is always followed by this thread's atomicMax, so the last operation
    // on a saturated counter leaves UINT32_MAX.
    uint32_t old = atomicAdd(p, add);
    if (old + add < old || old + add == 0xFFFFFFFFu) atomicMax(p, 0xFFFFFFFFu);
}

// ---- narrow keys (k <= 32) -------------------------------------------------
// (`key` / `klo` below are STORED FORMS h: what the table's lo[] words are compared with)

// returns false when the bucket is full (caller raises the error flag)
template <bool INSERT>
__device__ __forceinline__ bool kdf_add_narrow(const KdfTable &t, uint64_t key, uint32_t add,
                                               uint64_t slot, uint64_t cur, uint32_t &claimed) {
    const uint64_t bmask = (1ull << t.bucket_bits) - 1;
    const uint64_t base = slot & ~bmask;
    for (uint64_t i = 0;;) {
        if (cur == key) { if (add) kdf_sat_add(&t.cnt[slot], add); return true; }
        if (cur == KDF_EMPTY) {
            if (!INSERT) return true;                       // --if: absent keys are not counted
            uint64_t old = atomicCAS((unsigned long long *)&t.lo[slot], KDF_EMPTY, key);
            if (old == KDF_EMPTY) { claimed++; if (add) kdf_sat_add(&t.cnt[slot], add); return true; }
            if (old == key) { if (add) kdf_sat_add(&t.cnt[slot], add); return true; }
        }
        if (++i > bmask) return false;
        slot = base | ((slot + 1) & bmask);
        cur = t.lo[slot];
    }
}

// returns the slot of the key with stored form `key` or ~0 when absent
__device__ __forceinline__ uint64_t kdf_find_narrow(const KdfTable &t, uint64_t key) {
    const uint64_t bmask = (1ull << t.bucket_bits) - 1;
    uint64_t slot = kdf_home(t, key);
    const uint64_t base = slot & ~bmask;
    for (uint64_t i = 0; i <= bmask; ++i) {
        uint64_t cur = t.lo[slot];
        if (cur == key) return slot;
        if (cur == KDF_EMPTY) return ~0ull;
        slot = base | ((slot + 1) & bmask);
    }
    return ~0ull;
}

// ---- wide keys (33 <= k <= 63): hi holds 2k-64 <= 62 bits -------------------
// Claim protocol without a 128-bit CAS: CAS hi EMPTY -> (hi | PENDING), publish
// lo with a returning atomic (complete at memory before the next instruction
// issues), then store the final hi.  All shared words are accessed with
// device-scope atomics.
//
// NO LANE EVER WAITS for another lane inside a divergent loop: a prober that
// meets a PENDING slot whose hi matches its own returns KDF_BLOCKED, and the
// caller retries under a wave-uniform `while (__any(todo))` loop
// (kdf_add_wide).  A spin inside the probe loop deadlocks when the claimer is a
// lane of the same wave: its publish block leaves the loop, so the compiler may
// run it only after every lane has left the loop.

#define KDF_OK_ADD  0
#define KDF_FULL    1
#define KDF_BLOCKED 2

__device__ __forceinline__ uint64_t kdf_ld(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool INSERT>
__device__ __forceinline__ int kdf_try_add_wide(const KdfTable &t, uint64_t klo, uint64_t khi,
                                                uint32_t add, uint64_t slot, uint32_t &claimed) {
    const uint64_t bmask = (1ull << t.bucket_bits) - 1;
    const uint64_t base = slot & ~bmask;
    for (uint64_t i = 0;;) {
        uint64_t chi = INSERT ? kdf_ld(&t.hi[slot]) : t.hi[slot];
        if (chi == KDF_EMPTY) {
            if (!INSERT) return KDF_OK_ADD;
            uint64_t old = atomicCAS((unsigned long long *)&t.hi[slot], KDF_EMPTY, khi | KDF_PENDING);
            if (old == KDF_EMPTY) {
                uint64_t prev = atomicExch((unsigned long long *)&t.lo[slot], klo);
                asm volatile("s_waitcnt vmcnt(0)" :: "v"(prev) : "memory");
                __hip_atomic_store(&t.hi[slot], khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                claimed++;
                if (add) kdf_sat_add(&t.cnt[slot], add);
                return KDF_OK_ADD;
            }
            chi = old;
        }
        if ((chi & ~KDF_PENDING) == khi) {
            if (chi & KDF_PENDING) return KDF_BLOCKED;       // lo not published yet: retry later
            uint64_t clo = INSERT ? kdf_ld(&t.lo[slot]) : t.lo[slot];
            if (clo == klo) { if (add) kdf_sat_add(&t.cnt[slot], add); return KDF_OK_ADD; }
        }
        if (++i > bmask) return KDF_FULL;
        slot = base | ((slot + 1) & bmask);
    }
}

// Must be reached by the lanes of a wave together with `todo` telling which of
// them have a key (divergent callers pass todo = false for the idle lanes).
template <bool INSERT>
__device__ __forceinline__ bool kdf_add_wide(const KdfTable &t, bool todo, uint64_t klo, uint64_t khi,
                                             uint32_t add, uint64_t slot, uint32_t &claimed) {
    bool ok = true;
    while (__any(todo)) {
        if (todo) {
            const int r = kdf_try_add_wide<INSERT>(t, klo, khi, add, slot, claimed);
            if (r != KDF_BLOCKED) { todo = false; ok = (r == KDF_OK_ADD); }
        }
        __builtin_amdgcn_wave_barrier();
    }
    return ok;
}

__device__ __forceinline__ uint64_t kdf_find_wide(const KdfTable &t, uint64_t klo, uint64_t khi) {
    const uint64_t bmask = (1ull << t.bucket_bits) - 1;
    uint64_t slot = kdf_home(t, klo);
    const uint64_t base = slot & ~bmask;
    for (uint64_t i = 0; i <= bmask; ++i) {
        uint64_t chi = t.hi[slot];
        if (chi == KDF_EMPTY) return ~0ull;
        if (chi == khi && t.lo[slot] == klo) return slot;
        slot = base | ((slot + 1) & bmask);
    }
    return ~0ull;
}

// ---------------------------------------------------------------------------
// Window extraction.  A thread owns KDF_TILE = 64 consecutive window starts
// (stream positions tile*64 .. tile*64+63) and reads the 64 + k - 1 bases they
// cover: 3 packed words for k <= 32, 4 for k <= 63, plus 2 mask words.
//
// With base i in bits 2i of the stream (LSB first), the k-base window at p is
//   E   = (stream >> 2p) & kmask            (base j of the window in bits 2j)
//   rc  = ~E & kmask                         (reverse complement, MSB-first code)
//   fwd = rev2(E) >> (2*32 - 2k)             (forward k-mer, MSB-first code)
// so neither orientation needs a rolling loop or a warm-up.

// 64-bit validity bitmap of the tile: bit p set iff no invalid position in
// [p, p+k).  m0/m1 = mask words of this tile and the next.
__device__ __forceinline__ uint64_t kdf_valid_windows(uint64_t m0, uint64_t m1, int k) {
    // ok = positions that are valid bases; run-AND over k consecutive positions
    // by doubling (128-bit shifts done on the two halves).
    uint64_t a0 = ~m0, a1 = ~m1;
    int r = 1;
    while (r < k) {
        int s = (k - r) < r ? (k - r) : r;       // 1 <= s <= 32
        uint64_t b0 = (a0 >> s) | (a1 << (64 - s));
        uint64_t b1 = a1 >> s;
        a0 &= b0; a1 &= b1;
        r += s;
    }
    return a0;
}

template <int KW> struct KdfKey;
template <> struct KdfKey<1> { uint64_t lo; };
template <> struct KdfKey<2> { uint64_t lo, hi; };

// canonical key from an extracted window E (base j of the window in bits 2j)
__device__ __forceinline__ uint64_t kdf_canon_narrow(uint64_t e, int k, uint64_t kmask) {
    e &= kmask;
    const uint64_t rc = ~e & kmask;
    const uint64_t fwd = kdf_rev2(e) >> (64 - 2 * k);
    return fwd < rc ? fwd : rc;
}
__device__ __forceinline__ void kdf_canon_wide(uint64_t e0, uint64_t e1, int k, uint64_t &klo, uint64_t &khi) {
    const int hb = 2 * k - 64;                         // bits used in the high word, 2..62
    const uint64_t hmask = (1ull << hb) - 1;
    e1 &= hmask;
    const uint64_t rlo = ~e0, rhi = ~e1 & hmask;
    const uint64_t f1 = kdf_rev2(e0), f0 = kdf_rev2(e1);
    const int s = 128 - 2 * k;                          // 2..62
    const uint64_t flo = (f0 >> s) | (f1 << (64 - s));
    const uint64_t fhi = f1 >> s;
    const bool fw = (fhi < rhi) || (fhi == rhi && flo < rlo);
    klo = fw ? flo : rlo;
    khi = fw ? fhi : rhi;
}
// (hi:lo) >> sh, low 64 bits; sh in 0..63
__device__ __forceinline__ uint64_t kdf_funnel(uint64_t lo, uint64_t hi, int sh) {
    return sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
}

// the same for sh in 0..31, on the 32-bit halves: two v_alignbit_b32 (full rate) where the 64-bit form above compiles to a
// 64-bit shift (half rate), a 32-bit shift and an OR per result
__device__ __forceinline__ uint64_t kdf_funnel32(uint64_t lo, uint64_t hi, int sh) {
#ifdef KDF_FUNNEL64                                   // (variant builds: the 64-bit form, for same-box comparisons)
    return kdf_funnel(lo, hi, sh);
#endif
    const uint32_t l0 = (uint32_t)lo, l1 = (uint32_t)(lo >> 32), h0 = (uint32_t)hi;
    return ((uint64_t)__builtin_amdgcn_alignbit(h0, l1, (uint32_t)sh) << 32) | __builtin_amdgcn_alignbit(l1, l0, (uint32_t)sh);
}

// canonical key of the window starting at local position p (0..63), narrow.
__device__ __forceinline__ uint64_t kdf_window_narrow(const uint64_t (&w)[3], int p, int k, uint64_t kmask) {
    const int word = p >> 5, sh = (p & 31) * 2;
    uint64_t lo = w[word], hi = w[word + 1];
    uint64_t e = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    e &= kmask;
    uint64_t rc = ~e & kmask;
    uint64_t fwd = kdf_rev2(e) >> (64 - 2 * k);
    return fwd < rc ? fwd : rc;
}

// wide: window E is 2k <= 126 bits taken from 3 consecutive words.
__device__ __forceinline__ void kdf_window_wide(const uint64_t (&w)[4], int p, int k,
                                                uint64_t &klo, uint64_t &khi) {
    const int word = p >> 5, sh = (p & 31) * 2;
    uint64_t x0 = w[word], x1 = w[word + 1], x2 = (word + 2 < 4) ? w[word + 2] : 0;
    uint64_t e0 = sh ? ((x0 >> sh) | (x1 << (64 - sh))) : x0;
    uint64_t e1 = sh ? ((x1 >> sh) | (x2 << (64 - sh))) : x1;
    const int hb = 2 * k - 64;                         // bits used in the high word, 2..62
    const uint64_t hmask = (1ull << hb) - 1;
    e1 &= hmask;
    uint64_t rlo = ~e0, rhi = ~e1 & hmask;              // reverse complement
    // forward: reverse the 2-bit groups of the 128-bit value, shift right by 128-2k
    uint64_t f1 = kdf_rev2(e0), f0 = kdf_rev2(e1);     // (f1:f0) = rev2 over 128 bits
    const int s = 128 - 2 * k;                          // 2..62
    uint64_t flo = (f0 >> s) | (f1 << (64 - s));
    uint64_t fhi = f1 >> s;
    bool fw = (fhi < rhi) || (fhi == rhi && flo < rlo);
    klo = fw ? flo : rlo;
    khi = fw ? fhi : rhi;
}
