// EXPERIMENT, not part of the product path (see prototypes/superkmer/README.md).
//
// Question: what would a super-k-mer pipeline cost on MI355X?  Instead of moving every
// k-mer instance (8 B) through two partition passes, move minimizer-delimited read
// substrings ("records", ~2 bits per k-mer), partition those by a hash of their
// minimizer -- all k-mers of a record land in the same table bucket when the bucket is
// chosen by the minimizer -- dedupe identical records inside the bucket workgroup and
// insert each distinct record's k-mers once with its multiplicity.
//
// Stage 1 (this file, sk_extract_*): per-window minimizers, record boundaries, record
// emission.  Stage 2 (sk_bucket_*): per-bucket record dedupe + expansion + LDS insert.
// The stream layout and canonical k-mer code are the engine's (csrc/kdf_device.h).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "kdf_device.h"

#define SK_THREADS 1024
#define SK_WPT 16
#define SK_SLAB (SK_THREADS * SK_WPT)
#define SK_K 31
#define SK_M 12                                // 24-bit canonical m-mers: the order hash is one full-rate v_mul_u32_u24
#define SK_W (SK_K - SK_M + 1)                 // 20 m-mers per window
#define SK_NM (SK_WPT + 1 + SK_W - 1)          // 36 m-mers per thread (one look-back window)

struct SkRecord { uint64_t lo, hi; };          // bases 0..31 | bases 32..49 (<= 36 bits), k-mers of the record in bits 58..63

// injective 24-bit scramble of a canonical m-mer (odd multiply mod 2^24, xor-shift): the ORDER of the minimizer scheme
__device__ __forceinline__ uint32_t sk_order(uint32_t x) {
    uint32_t g = __umul24(x, 0x9E3779u) & 0xFFFFFFu;
    return g ^ (g >> 11);
}
// bucket hash of a minimizer (order value): any fixed mix
__device__ __forceinline__ uint32_t sk_bucket_hash(uint32_t g) { return (g * 0x9E3779B1u) ^ (g >> 9); }

// (b:a) >> sh for 128-bit value held in two 64-bit halves, low 64 bits; sh in 0..127
__device__ __forceinline__ uint64_t sk_shr128(uint64_t a, uint64_t b, int sh) {
    return sh >= 64 ? (b >> (sh - 64)) : kdf_funnel(a, b, sh);
}
// reverse the sixteen 2-bit groups of a 32-bit word
__device__ __forceinline__ uint32_t sk_rev2_32(uint32_t x) {
    x = __builtin_bitreverse32(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

struct SkThread {
    uint32_t start_bits;     // bit i: window i of this thread starts a record
    uint32_t brk_bits;       // bit i: window i starts a record or is invalid
};

// per-thread: minimizers of 16 windows (+1 look-back), start / break flags.  The minimizer order value of
// window i goes to lds_mv[i * SK_THREADS + tid] (read back by record: no dynamic register indexing).
template <bool BACK>
__device__ __forceinline__ void sk_windows_t(const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid,
                                             uint64_t P, uint64_t n_bases, SkThread &o, bool cut_before, uint32_t *lds_mv) {
    const uint64_t S = BACK ? P - 1 : P;                                // span start (47 bases)
    const uint64_t w0 = S >> 5; const int sh = (int)(S & 31) * 2;
    const uint64_t x0 = packed[w0], x1 = packed[w0 + 1], x2 = packed[w0 + 2];
    const uint64_t e0 = kdf_funnel(x0, x1, sh), e1 = kdf_funnel(x1, x2, sh);
    const uint64_t mw = S >> 6; const int msh = (int)(S & 63);
    const uint64_t inv = kdf_funnel(invalid[mw], invalid[mw + 1], msh);
    uint64_t a = ~inv;
    a &= a >> 1; a &= a >> 2; a &= a >> 4; a &= a >> 8; a &= a >> 15;   // bit j: window S + j valid
    const uint32_t v17 = BACK ? ((uint32_t)a & 0x1FFFFu) : (((uint32_t)a << 1) & 0x1FFFFu);   // bit i + 1: window P + i
    // 32-bit words of the span, of its complement (rc m-mer = ~e, MSB-first code) and of its 2-bit-group
    // reversal (fwd m-mer, MSB-first): every m-mer is one v_alignbit + one v_and
    const uint32_t w[4] = {(uint32_t)e0, (uint32_t)(e0 >> 32), (uint32_t)e1, (uint32_t)(e1 >> 32)};
    const uint32_t f[4] = {sk_rev2_32(w[3]), sk_rev2_32(w[2]), sk_rev2_32(w[1]), sk_rev2_32(w[0])};
    constexpr uint32_t MM = (1u << (2 * SK_M)) - 1;
    constexpr int NM = BACK ? SK_NM : SK_NM - 1;
    uint32_t g[SK_NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int ro = 2 * j, fo = 128 - 2 * SK_M - 2 * j;
        const uint32_t rc = ~__builtin_amdgcn_alignbit(w[(ro >> 5) + 1 > 3 ? 3 : (ro >> 5) + 1], w[ro >> 5], ro & 31) & MM;
        const uint32_t fw = __builtin_amdgcn_alignbit((fo >> 5) + 1 > 3 ? 0u : f[(fo >> 5) + 1 > 3 ? 3 : (fo >> 5) + 1], f[fo >> 5], fo & 31) & MM;
        g[j] = sk_order(fw < rc ? fw : rc);
    }
    // sliding minimum over SK_W m-mers (van Herk): span window t covers g[t .. t + SK_W - 1]
    uint32_t sfx[SK_W], pfx[SK_NM];
    sfx[SK_W - 1] = g[SK_W - 1];
#pragma unroll
    for (int j = SK_W - 2; j >= 0; --j) sfx[j] = min(g[j], sfx[j + 1]);
    pfx[SK_W] = g[SK_W];
#pragma unroll
    for (int j = SK_W + 1; j < NM; ++j) pfx[j] = min(g[j], pfx[j - 1]);
    // mv[i] = minimizer of window P + i - 1 (mv[0] = look-back)
    uint32_t mv[SK_WPT + 1];
    if constexpr (BACK) {
        mv[0] = sfx[0];
#pragma unroll
        for (int i = 1; i <= SK_WPT; ++i) mv[i] = min(sfx[i], pfx[i + SK_W - 1]);
    } else {
        mv[0] = 0; mv[1] = sfx[0];
#pragma unroll
        for (int i = 2; i <= SK_WPT; ++i) mv[i] = min(sfx[i - 1], pfx[i + SK_W - 2]);
    }
    uint32_t neq = 0;
#pragma unroll
    for (int i = 0; i < SK_WPT; ++i) {
        neq |= min(mv[i + 1] ^ mv[i], 1u) << i;
        lds_mv[i * SK_THREADS + threadIdx.x] = mv[i + 1];
    }
    // windows past the end of the stream are invalid by the stream's padding (all-ones mask words)
    const uint32_t v16 = (v17 >> 1) & 0xFFFFu;
    uint32_t pv16 = v17 & 0xFFFFu;
    if (cut_before) pv16 &= ~1u;
    o.start_bits = v16 & (~pv16 | neq);
    o.brk_bits = (o.start_bits | ~v16) & 0xFFFFu;
    (void)n_bases;
}
__device__ __forceinline__ void sk_windows(const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid,
                                           uint64_t P, uint64_t n_bases, SkThread &o, bool cut_before, uint32_t *lds_mv) {
    if (P > 0) sk_windows_t<true>(packed, invalid, P, n_bases, o, cut_before, lds_mv);
    else sk_windows_t<false>(packed, invalid, P, n_bases, o, cut_before, lds_mv);
}

// Stage 1a: count records per coarse bin (what an A0' histogram pass would do) and totals.
__global__ __launch_bounds__(SK_THREADS) void sk_count_kernel(const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid,
                                                            uint64_t n_bases, uint32_t coarse_bits, unsigned long long *totals,
                                                            uint32_t *hist /* [gridDim][2^coarse_bits] */)
{
    __shared__ uint32_t h[1024];
    __shared__ uint32_t lds_mv[SK_WPT * SK_THREADS];
    const uint32_t nb = 1u << coarse_bits;
    for (uint32_t i = threadIdx.x; i < nb; i += SK_THREADS) h[i] = 0;
    __syncthreads();
    const uint64_t P = ((uint64_t)blockIdx.x * SK_THREADS + threadIdx.x) * SK_WPT;
    uint32_t nrec = 0, nwin = 0;
    if (P < n_bases) {
        SkThread t; sk_windows(packed, invalid, P, n_bases, t, threadIdx.x == 0, lds_mv);
        nrec = __popc(t.start_bits); nwin = SK_WPT - __popc(t.brk_bits & ~t.start_bits);
        uint32_t sb = t.start_bits;
        while (sb) {
            const int i = __ffs(sb) - 1; sb &= sb - 1;
            atomicAdd(&h[sk_bucket_hash(lds_mv[i * SK_THREADS + threadIdx.x]) >> (32 - coarse_bits)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += SK_THREADS) hist[(size_t)blockIdx.x * nb + i] = h[i];
    // totals: records, windows
    for (int o = 32; o > 0; o >>= 1) { nrec += __shfl_xor(nrec, o); nwin += __shfl_xor(nwin, o); }
    __shared__ uint32_t tot2[2];
    if (threadIdx.x == 0) { tot2[0] = 0; tot2[1] = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicAdd(&tot2[0], nrec); atomicAdd(&tot2[1], nwin); }
    __syncthreads();
    if (threadIdx.x == 0) { atomicAdd(&totals[0], (unsigned long long)tot2[0]); atomicAdd(&totals[1], (unsigned long long)tot2[1]); }
}

// Stage 1b: emit records (canonical orientation) + their bucket hash to a flat array; slab-local compaction in LDS,
// one global cursor add per workgroup.  Record = 49 bases max: lo = bases 0..31, hi = bases 32..48 | nk << 58.
__global__ __launch_bounds__(SK_THREADS) void sk_emit_kernel(const uint64_t *__restrict__ packed, const uint64_t *__restrict__ invalid,
                                                           uint64_t n_bases, unsigned long long *cursor, SkRecord *out,
                                                           uint32_t *out_bucket, uint64_t out_cap)
{
    __shared__ uint32_t brk[SK_THREADS + 4];
    __shared__ uint32_t wsum[20];
    __shared__ uint32_t lds_mv[SK_WPT * SK_THREADS];
    __shared__ unsigned long long sh_base;
    const uint64_t P = ((uint64_t)blockIdx.x * SK_THREADS + threadIdx.x) * SK_WPT;
    SkThread t; t.start_bits = 0; t.brk_bits = 0xFFFFu;
    if (P < n_bases) sk_windows(packed, invalid, P, n_bases, t, threadIdx.x == 0, lds_mv);
    brk[threadIdx.x] = t.brk_bits;
    if (threadIdx.x < 4) brk[SK_THREADS + threadIdx.x] = 0xFFFFu;    // slab end = break (prototype simplification)
    __syncthreads();
    const uint32_t nrec = __popc(t.start_bits);
    // block exclusive scan of nrec
    uint32_t inc = nrec;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int i = 0; i < SK_THREADS / 64; ++i) { uint32_t v = wsum[i]; wsum[i] = acc; acc += v; }
        sh_base = atomicAdd(cursor, (unsigned long long)acc);
    }
    __syncthreads();
    unsigned long long pos = sh_base + wsum[wave] + inc - nrec;
    if (!nrec) return;
    // break bits of my 16 windows and the next 32 (a record is at most SK_W = 19 windows)
    const uint64_t look = (uint64_t)(brk[threadIdx.x] & 0xFFFFu) | ((uint64_t)(brk[threadIdx.x + 1] & 0xFFFFu) << 16) |
                          ((uint64_t)(brk[threadIdx.x + 2] & 0xFFFFu) << 32);
    uint32_t sb = t.start_bits;
    while (sb) {
        const int i = __ffs(sb) - 1; sb &= sb - 1;
        const uint64_t after = look >> (i + 1);
        int nk = after ? __ffsll((unsigned long long)after) : 48;     // windows until the next break
        // (a repeated minimizer VALUE can keep a record alive past SK_W windows -- tandem repeats; the prototype
        // clamps and loses those few windows, a product version starts a new record there)
        if (nk > SK_W) nk = SK_W;                                     // (cannot happen: a minimizer lives <= SK_W windows)
        const uint64_t Q = P + i;                                     // first base of the record
        const int nbases = nk + SK_K - 1;                             // <= 49
        const uint64_t w0 = Q >> 5; const int sh = (int)(Q & 31) * 2;
        const uint64_t x0 = packed[w0], x1 = packed[w0 + 1], x2 = packed[w0 + 2];
        uint64_t lo = kdf_funnel(x0, x1, sh), hi = kdf_funnel(x1, x2, sh);
        const int hb = 2 * nbases - 64;                               // bits used in hi: -2..34
        const uint64_t hmask = hb > 0 ? ((1ull << hb) - 1) : 0ull;
        const uint64_t lmask = hb >= 0 ? ~0ull : ((1ull << (2 * nbases)) - 1);
        hi &= hmask; lo &= lmask;
        // reverse complement of the nbases-base string (LSB-first): rev2 of ~(hi:lo), shifted down
        const uint64_t r1 = kdf_rev2(~lo), r0 = kdf_rev2(~hi);        // (r1:r0) reversed 128-bit; wanted bits start at 128 - 2*nbases
        const int s = 128 - 2 * nbases;                               // 30..66
        uint64_t clo = sk_shr128(r0, r1, s), chi = s >= 64 ? 0 : (r1 >> s);
        chi &= hmask; clo &= lmask;
        if (chi < hi || (chi == hi && clo < lo)) { lo = clo; hi = chi; }
        if (pos < out_cap) {
            out[pos].lo = lo; out[pos].hi = hi | ((uint64_t)nk << 58);
            out_bucket[pos] = lds_mv[i * SK_THREADS + threadIdx.x];     // the minimizer's order value (24 bits): the host picks the bucket function
        }
        ++pos;
    }
}


// ---------------------------------------------------------------------------
// Stage 2: one workgroup per table bucket.  Records of the bucket (contiguous) ->
// dedupe (a 32-bit "representative" CAS per hash slot; equal records add to its
// multiplicity, the record bytes themselves stay in global memory / L2) -> expand each
// distinct record once: k-mer j of a record goes to lane j of a 16-lane group and is
// added to the bucket's LDS slice with the record's multiplicity -> write the slice.
#define SKB_THREADS 512
#define SKB_BB 12                      // 4096-slot buckets, as the engine's narrow table
#define SKB_RT 2048                    // record dedupe slots
__global__ __launch_bounds__(SKB_THREADS) void sk_bucket_kernel(const SkRecord *__restrict__ recs, const uint32_t *__restrict__ boff,
                                                               uint64_t *__restrict__ tab_lo, uint32_t *__restrict__ tab_cnt,
                                                               unsigned long long *stats /* distinct, sum, ge3, overflow, expansions, max_fill */)
{
    constexpr uint32_t B = 1u << SKB_BB, bmask = B - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *tlo = (uint64_t *)smem;                  // [B]
    uint32_t *tcnt = (uint32_t *)(tlo + B);            // [B]
    uint32_t *own = tcnt + B, *mult = own + SKB_RT, *list = mult + SKB_RT;
    __shared__ uint32_t sh_n, sh_claimed, sh_over, sh_exp, sh_ge3;
    __shared__ unsigned long long sh_sum;
    const uint32_t bucket = blockIdx.x;
    const uint32_t r0 = boff[bucket], r1 = boff[bucket + 1], nrec = r1 - r0;
    for (uint32_t i = threadIdx.x; i < B; i += SKB_THREADS) { tlo[i] = KDF_EMPTY; tcnt[i] = 0; }
    for (uint32_t i = threadIdx.x; i < SKB_RT; i += SKB_THREADS) { own[i] = 0xFFFFFFFFu; mult[i] = 0; }
    if (threadIdx.x == 0) { sh_n = 0; sh_claimed = 0; sh_over = 0; sh_exp = 0; sh_ge3 = 0; sh_sum = 0; }
    __syncthreads();
    // ---- dedupe
    uint32_t over = 0;
    for (uint32_t r = threadIdx.x; r < nrec; r += SKB_THREADS) {
        const SkRecord me = recs[r0 + r];
        uint32_t sl = (uint32_t)(kdf_mix64(me.lo ^ (me.hi * 0xC2B2AE3D27D4EB4Full)) >> 40) & (SKB_RT - 1);
        bool done = false;
        for (uint32_t n = 0; n < SKB_RT && !done; ++n) {
            uint32_t o = own[sl];
            if (o == 0xFFFFFFFFu) {
                o = atomicCAS(&own[sl], 0xFFFFFFFFu, r);
                if (o == 0xFFFFFFFFu) { atomicAdd(&mult[sl], 1u); done = true; break; }
            }
            const SkRecord rep = recs[r0 + o];                  // the slot's representative (immutable input: no publish protocol)
            if (rep.lo == me.lo && rep.hi == me.hi) { atomicAdd(&mult[sl], 1u); done = true; break; }
            sl = (sl + 1) & (SKB_RT - 1);
        }
        if (!done) ++over;                                       // dedupe table full: would be expanded directly (not in the prototype)
    }
    if (over) atomicAdd(&sh_over, over);
    __syncthreads();
    // ---- compact the occupied dedupe slots
    for (uint32_t i = threadIdx.x; i < SKB_RT; i += SKB_THREADS)
        if (own[i] != 0xFFFFFFFFu) list[atomicAdd(&sh_n, 1u)] = i;
    __syncthreads();
    const uint32_t ndist = sh_n;
    // ---- expand: 16 lanes per distinct record
    const uint32_t grp = threadIdx.x >> 4, j0 = threadIdx.x & 15;
    uint32_t claimed = 0, expn = 0;
    for (uint32_t e = grp; e < ndist; e += SKB_THREADS / 16) {
        const uint32_t slot = list[e];
        const SkRecord rec = recs[r0 + own[slot]];
        const uint32_t c = mult[slot];
        const uint32_t nk = (uint32_t)(rec.hi >> 58);
        const uint64_t hi = rec.hi & ((1ull << 58) - 1);
        for (uint32_t j = j0; j < nk; j += 16) {
            // k-mer j: bases j .. j+30 of the record (LSB-first) -> canonical key (MSB-first code)
            const int sh = 2 * (int)j;                           // 0..36
            const uint64_t ewin = kdf_funnel(rec.lo, hi, sh);
            const uint64_t key = kdf_canon_narrow(ewin, SK_K, (1ull << (2 * SK_K)) - 1);
            uint32_t sl = (uint32_t)(kdf_mix64(key) >> 20) & bmask;     // slot inside the bucket: k-mer hash bits
            for (uint32_t n = 0; n <= bmask; ++n) {
                uint64_t cur = tlo[sl];
                if (cur == KDF_EMPTY) {
                    cur = atomicCAS((unsigned long long *)&tlo[sl], KDF_EMPTY, key);
                    if (cur == KDF_EMPTY) { ++claimed; cur = key; }
                }
                if (cur == key) { atomicAdd(&tcnt[sl], c); break; }
                sl = (sl + 1) & bmask;
            }
            ++expn;
        }
    }
    if (claimed) atomicAdd(&sh_claimed, claimed);
    if (expn) atomicAdd(&sh_exp, expn);
    __syncthreads();
    // ---- write back + statistics
    uint32_t ge3 = 0; unsigned long long sum = 0;
    const uint64_t slot0 = (uint64_t)bucket << SKB_BB;
    for (uint32_t i = threadIdx.x; i < B; i += SKB_THREADS) {
        const uint32_t c = tcnt[i];
        tab_lo[slot0 + i] = tlo[i]; tab_cnt[slot0 + i] = c;
        ge3 += c >= 3; sum += c;
    }
    for (int o = 32; o > 0; o >>= 1) { ge3 += __shfl_xor(ge3, o); sum += __shfl_xor(sum, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sh_sum, sum); atomicAdd(&sh_ge3, ge3); }
    __syncthreads();
    if (threadIdx.x == 0) {                             // per-bucket statistics row (reduced by the host: no contended global atomics)
        unsigned long long *row = stats + (size_t)bucket * 8;
        row[0] = sh_claimed; row[1] = sh_sum; row[2] = sh_ge3; row[3] = sh_over; row[4] = sh_exp; row[5] = ndist; row[6] = nrec; row[7] = 0;
    }
}

// Stage 2, second version: the representatives live in an LDS queue (speculative write, then a CAS on the
// hash slot publishes the queue index: a reader never sees a half-written record), so neither the dedupe
// compare nor the expansion goes back to global memory.
#define SKB_QCAP 1024
__global__ __launch_bounds__(SKB_THREADS) void sk_bucket2_kernel(const SkRecord *__restrict__ recs, const uint32_t *__restrict__ boff,
                                                                uint64_t *__restrict__ tab_lo, uint32_t *__restrict__ tab_cnt,
                                                                unsigned long long *stats)
{
    constexpr uint32_t B = 1u << SKB_BB, bmask = B - 1, RT = 1024, NONE = 0xFFFFFFFFu;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *tlo = (uint64_t *)smem;                  // [B]
    uint64_t *rq_lo = tlo + B, *rq_hi = rq_lo + SKB_QCAP;   // [QCAP] representative records
    uint32_t *tcnt = (uint32_t *)(rq_hi + SKB_QCAP);   // [B]
    uint32_t *own = tcnt + B;                          // [RT] hash slot -> queue index
    uint32_t *multq = own + RT;                        // [QCAP] multiplicity of the representative
    __shared__ uint32_t sh_qn, sh_claimed, sh_over, sh_exp, sh_ge3, sh_nd;
    __shared__ unsigned long long sh_sum;
    const uint32_t bucket = blockIdx.x;
    const uint32_t r0 = boff[bucket], r1 = boff[bucket + 1], nrec = r1 - r0;
    for (uint32_t i = threadIdx.x; i < B; i += SKB_THREADS) { tlo[i] = KDF_EMPTY; tcnt[i] = 0; }
    for (uint32_t i = threadIdx.x; i < RT; i += SKB_THREADS) { own[i] = NONE; multq[i] = 0; }
    if (threadIdx.x == 0) { sh_qn = 0; sh_claimed = 0; sh_over = 0; sh_exp = 0; sh_ge3 = 0; sh_sum = 0; sh_nd = 0; }
    __syncthreads();
    uint32_t over = 0;
    for (uint32_t r = threadIdx.x; r < nrec; r += SKB_THREADS) {
        const SkRecord me = recs[r0 + r];
        uint32_t sl = (uint32_t)(kdf_mix64(me.lo ^ (me.hi * 0xC2B2AE3D27D4EB4Full)) >> 40) & (RT - 1);
        uint32_t myq = NONE; bool done = false;
        for (uint32_t n = 0; n < RT; ++n) {
            uint32_t o = own[sl];
            if (o == NONE) {
                if (myq == NONE) {
                    myq = atomicAdd(&sh_qn, 1u);
                    if (myq >= SKB_QCAP) { myq = NONE; break; }
                    rq_lo[myq] = me.lo; rq_hi[myq] = me.hi;
                }
                o = atomicCAS(&own[sl], NONE, myq);
                if (o == NONE) { atomicAdd(&multq[myq], 1u); myq = NONE; done = true; break; }
            }
            if (rq_lo[o] == me.lo && rq_hi[o] == me.hi) { atomicAdd(&multq[o], 1u); done = true; break; }
            sl = (sl + 1) & (RT - 1);
        }
        if (myq != NONE) rq_hi[myq] = 0;                         // speculative entry not used: nk = 0, skipped below
        if (!done) ++over;
    }
    if (over) atomicAdd(&sh_over, over);
    __syncthreads();
    const uint32_t nq = sh_qn < SKB_QCAP ? sh_qn : SKB_QCAP;
    const uint32_t grp = threadIdx.x >> 4, j0 = threadIdx.x & 15;
    uint32_t claimed = 0, expn = 0, nd = 0;
    for (uint32_t q = grp; q < nq; q += SKB_THREADS / 16) {
        const uint64_t rlo = rq_lo[q], rhi = rq_hi[q];
        const uint32_t c = multq[q];
        const uint32_t nk = (uint32_t)(rhi >> 58);
        const uint64_t hi = rhi & ((1ull << 58) - 1);
        if (j0 == 0 && nk) ++nd;
        for (uint32_t j = j0; j < nk; j += 16) {
            const uint64_t ewin = kdf_funnel(rlo, hi, 2 * (int)j);
            const uint64_t key = kdf_canon_narrow(ewin, SK_K, (1ull << (2 * SK_K)) - 1);
            uint32_t sl = (uint32_t)(kdf_mix64(key) >> 20) & bmask;
            for (uint32_t n = 0; n <= bmask; ++n) {
                uint64_t cur = tlo[sl];
                if (cur == KDF_EMPTY) {
                    cur = atomicCAS((unsigned long long *)&tlo[sl], KDF_EMPTY, key);
                    if (cur == KDF_EMPTY) { ++claimed; cur = key; }
                }
                if (cur == key) { atomicAdd(&tcnt[sl], c); break; }
                sl = (sl + 1) & bmask;
            }
            ++expn;
        }
    }
    if (claimed) atomicAdd(&sh_claimed, claimed);
    if (expn) atomicAdd(&sh_exp, expn);
    if (nd) atomicAdd(&sh_nd, nd);
    __syncthreads();
    uint32_t ge3 = 0; unsigned long long sum = 0;
    const uint64_t slot0 = (uint64_t)bucket << SKB_BB;
    for (uint32_t i = threadIdx.x; i < B; i += SKB_THREADS) {
        const uint32_t c = tcnt[i];
        tab_lo[slot0 + i] = tlo[i]; tab_cnt[slot0 + i] = c;
        ge3 += c >= 3; sum += c;
    }
    for (int o = 32; o > 0; o >>= 1) { ge3 += __shfl_xor(ge3, o); sum += __shfl_xor(sum, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sh_sum, sum); atomicAdd(&sh_ge3, ge3); }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *row = stats + (size_t)bucket * 8;
        row[0] = sh_claimed; row[1] = sh_sum; row[2] = sh_ge3; row[3] = sh_over; row[4] = sh_exp; row[5] = sh_nd; row[6] = nrec; row[7] = sh_qn;
    }
}

// Stage 2, third version: like v1 (representatives stay in global memory) but the expansion runs over a FLAT
// list of (distinct record, k-mer index) items, so every lane has work whatever the record lengths are, and
// the insert reads two slots of the probe sequence up front (kernel C's scheme; the tail is probed in a loop).
__global__ __launch_bounds__(SKB_THREADS) void sk_bucket3_kernel(const SkRecord *__restrict__ recs, const uint32_t *__restrict__ boff,
                                                                uint64_t *__restrict__ tab_lo, uint32_t *__restrict__ tab_cnt,
                                                                unsigned long long *stats)
{
    constexpr uint32_t B = 1u << SKB_BB, bmask = B - 1, NONE = 0xFFFFFFFFu;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *tlo = (uint64_t *)smem;                  // [B]
    uint32_t *tcnt = (uint32_t *)(tlo + B);            // [B]
    uint32_t *own = tcnt + B, *mult = own + SKB_RT;    // [RT] representative record / multiplicity per dedupe slot
    uint32_t *lq = mult + SKB_RT;                      // [RT + 1] list: dedupe slot of entry e
    uint32_t *lp = lq + SKB_RT + 4;                    // [RT + 4] list: first flat item of entry e (padded with the total)
    __shared__ uint32_t wsum[20];
    __shared__ uint32_t sh_claimed, sh_over, sh_exp, sh_ge3, sh_ne, sh_items;
    __shared__ unsigned long long sh_sum;
    const uint32_t bucket = blockIdx.x;
    const uint32_t r0 = boff[bucket], r1 = boff[bucket + 1], nrec = r1 - r0;
    for (uint32_t i = threadIdx.x; i < B; i += SKB_THREADS) { tlo[i] = KDF_EMPTY; tcnt[i] = 0; }
    for (uint32_t i = threadIdx.x; i < SKB_RT; i += SKB_THREADS) { own[i] = NONE; mult[i] = 0; }
    if (threadIdx.x == 0) { sh_claimed = 0; sh_over = 0; sh_exp = 0; sh_ge3 = 0; sh_sum = 0; }
    __syncthreads();
    // ---- dedupe (as v1)
    uint32_t over = 0;
    for (uint32_t r = threadIdx.x; r < nrec; r += SKB_THREADS) {
        const SkRecord me = recs[r0 + r];
        uint32_t sl = (uint32_t)(kdf_mix64(me.lo ^ (me.hi * 0xC2B2AE3D27D4EB4Full)) >> 40) & (SKB_RT - 1);
        bool done = false;
        for (uint32_t n = 0; n < SKB_RT && !done; ++n) {
            uint32_t o = own[sl];
            if (o == NONE) {
                o = atomicCAS(&own[sl], NONE, r);
                if (o == NONE) { atomicAdd(&mult[sl], 1u); done = true; break; }
            }
            const SkRecord rep = recs[r0 + o];
            if (rep.lo == me.lo && rep.hi == me.hi) { atomicAdd(&mult[sl], 1u); done = true; break; }
            sl = (sl + 1) & (SKB_RT - 1);
        }
        if (!done) ++over;
    }
    if (over) atomicAdd(&sh_over, over);
    __syncthreads();
    // ---- list of distinct records with the prefix of their k-mer counts: thread t owns dedupe slots [t*PER, +PER)
    constexpr uint32_t PER = SKB_RT / SKB_THREADS;
    uint32_t nk_[PER], cnt = 0, items = 0;
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t o = own[threadIdx.x * PER + u];
        nk_[u] = o == NONE ? 0u : (uint32_t)(recs[r0 + o].hi >> 58);
        cnt += nk_[u] ? 1u : 0u; items += nk_[u];
    }
    {
        uint32_t v = (cnt << 16) | items, inc = v;                        // both sums stay below 65536
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t acc = 0;
            for (int i = 0; i < SKB_THREADS / 64; ++i) { const uint32_t t = wsum[i]; wsum[i] = acc; acc += t; }
            sh_ne = acc >> 16; sh_items = acc & 0xFFFFu;
        }
        __syncthreads();
        uint32_t ex = wsum[wave] + inc - v;
        uint32_t e = ex >> 16, pf = ex & 0xFFFFu;
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) if (nk_[u]) { lq[e] = threadIdx.x * PER + u; lp[e] = pf; ++e; pf += nk_[u]; }
    }
    __syncthreads();
    const uint32_t ne = sh_ne, total = sh_items;
    if (threadIdx.x < 4) lp[ne + threadIdx.x] = total;                    // padding for the windowed search
    __syncthreads();
    // ---- flat expansion
    uint32_t claimed = 0, expn = 0;
    const unsigned long long inv_total = total ? (((unsigned long long)ne << 32) / total) : 0;
    for (uint32_t i = threadIdx.x; i < total; i += SKB_THREADS) {
        uint32_t e = (uint32_t)(((unsigned long long)i * inv_total) >> 32);
        if (e >= ne) e = ne - 1;
        while (lp[e] > i) --e;
        while (lp[e + 1] <= i) ++e;
        const uint32_t slot = lq[e], j = i - lp[e];
        const SkRecord rec = recs[r0 + own[slot]];
        const uint32_t c = mult[slot];
        const uint64_t hi = rec.hi & ((1ull << 58) - 1);
        const uint64_t key = kdf_canon_narrow(kdf_funnel(rec.lo, hi, 2 * (int)j), SK_K, (1ull << (2 * SK_K)) - 1);
        uint32_t sl = (uint32_t)(kdf_mix64(key) >> 20) & bmask;
        // two slots up front, resolved in straight-line code; the rest in a loop
        const uint64_t c0 = tlo[sl], c1 = tlo[(sl + 1) & bmask];
        bool done = false;
        if (c0 == key) { atomicAdd(&tcnt[sl], c); done = true; }
        else if (c0 != KDF_EMPTY && c1 == key) { atomicAdd(&tcnt[(sl + 1) & bmask], c); done = true; }
        if (!done) {
            for (uint32_t n = 0; n <= bmask; ++n) {
                uint64_t cur = tlo[sl];
                if (cur == KDF_EMPTY) {
                    cur = atomicCAS((unsigned long long *)&tlo[sl], KDF_EMPTY, key);
                    if (cur == KDF_EMPTY) { ++claimed; cur = key; }
                }
                if (cur == key) { atomicAdd(&tcnt[sl], c); break; }
                sl = (sl + 1) & bmask;
            }
        }
        ++expn;
    }
    if (claimed) atomicAdd(&sh_claimed, claimed);
    if (expn) atomicAdd(&sh_exp, expn);
    __syncthreads();
    uint32_t ge3 = 0; unsigned long long sum = 0;
    const uint64_t slot0 = (uint64_t)bucket << SKB_BB;
    for (uint32_t i = threadIdx.x; i < B; i += SKB_THREADS) {
        const uint32_t c = tcnt[i];
        tab_lo[slot0 + i] = tlo[i]; tab_cnt[slot0 + i] = c;
        ge3 += c >= 3; sum += c;
    }
    for (int o = 32; o > 0; o >>= 1) { ge3 += __shfl_xor(ge3, o); sum += __shfl_xor(sum, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sh_sum, sum); atomicAdd(&sh_ge3, ge3); }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *row = stats + (size_t)bucket * 8;
        row[0] = sh_claimed; row[1] = sh_sum; row[2] = sh_ge3; row[3] = sh_over; row[4] = sh_exp; row[5] = ne; row[6] = nrec; row[7] = 0;
    }
}

extern "C" {

// returns 0 or a HIP error code; ms[0] = count kernel, ms[1] = emit kernel (HIP events); totals[0..2] = records, windows, emitted
int sk_extract(const void *d_packed, const void *d_invalid, uint64_t n_bases, uint32_t coarse_bits,
               void *d_records, void *d_buckets, uint64_t out_cap, uint64_t *totals_out, float *ms, int reps)
{
    const unsigned grid = (unsigned)((n_bases + SK_SLAB - 1) / SK_SLAB);
    unsigned long long *d_tot = nullptr; uint32_t *d_hist = nullptr;
    hipError_t e;
    if ((e = hipMalloc((void **)&d_tot, 64)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&d_hist, (size_t)grid * (1u << coarse_bits) * 4)) != hipSuccess) return (int)e;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int phase = 0; phase < 2; ++phase) {
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            (void)hipMemsetAsync(d_tot, 0, 64, 0);
            (void)hipEventRecord(e0, 0);
            if (phase == 0)
                hipLaunchKernelGGL(sk_count_kernel, dim3(grid), dim3(SK_THREADS), 0, 0, (const uint64_t *)d_packed, (const uint64_t *)d_invalid,
                                   n_bases, coarse_bits, d_tot, d_hist);
            else
                hipLaunchKernelGGL(sk_emit_kernel, dim3(grid), dim3(SK_THREADS), 0, 0, (const uint64_t *)d_packed, (const uint64_t *)d_invalid,
                                   n_bases, d_tot + 2, (SkRecord *)d_records, (uint32_t *)d_buckets, out_cap);
            (void)hipEventRecord(e1, 0);
            if ((e = hipEventSynchronize(e1)) != hipSuccess) return (int)e;
            float t; (void)hipEventElapsedTime(&t, e0, e1); if (t < best) best = t;
            unsigned long long h[4];
            (void)hipMemcpy(h, d_tot, 32, hipMemcpyDeviceToHost);
            if (phase == 0) { totals_out[0] = h[0]; totals_out[1] = h[1]; } else totals_out[2] = h[2];
        }
        ms[phase] = best;
    }
    (void)hipFree(d_tot); (void)hipFree(d_hist);
    return (int)hipGetLastError();
}


// stage 2 on records already grouped by bucket (boff[n_buckets + 1]); table arrays of n_buckets * 4096 slots
int sk_buckets(const void *d_recs, const void *d_boff, uint32_t n_buckets, void *d_tab_lo, void *d_tab_cnt,
               uint64_t *stats_out /* DEVICE [n_buckets][8] */, float *ms, int reps, int version)
{
    unsigned long long *d_st = (unsigned long long *)stats_out;      // DEVICE array [n_buckets][8]
    hipError_t e;
    const size_t lds = version == 2 ? (size_t)(1u << SKB_BB) * 12 + (size_t)SKB_QCAP * 20 + 1024 * 4
                     : version == 3 ? (size_t)(1u << SKB_BB) * 12 + (size_t)SKB_RT * 16 + 64
                                    : (size_t)(1u << SKB_BB) * 12 + (size_t)SKB_RT * 12;
    if ((e = hipFuncSetAttribute((const void *)sk_bucket_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return (int)e;
    if ((e = hipFuncSetAttribute((const void *)sk_bucket2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return (int)e;
    if ((e = hipFuncSetAttribute((const void *)sk_bucket3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return (int)e;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0, 0);
        if (version == 3)
            hipLaunchKernelGGL(sk_bucket3_kernel, dim3(n_buckets), dim3(SKB_THREADS), lds, 0, (const SkRecord *)d_recs, (const uint32_t *)d_boff,
                               (uint64_t *)d_tab_lo, (uint32_t *)d_tab_cnt, d_st);
        else if (version == 2)
            hipLaunchKernelGGL(sk_bucket2_kernel, dim3(n_buckets), dim3(SKB_THREADS), lds, 0, (const SkRecord *)d_recs, (const uint32_t *)d_boff,
                               (uint64_t *)d_tab_lo, (uint32_t *)d_tab_cnt, d_st);
        else
            hipLaunchKernelGGL(sk_bucket_kernel, dim3(n_buckets), dim3(SKB_THREADS), lds, 0, (const SkRecord *)d_recs, (const uint32_t *)d_boff,
                               (uint64_t *)d_tab_lo, (uint32_t *)d_tab_cnt, d_st);
        (void)hipEventRecord(e1, 0);
        if ((e = hipEventSynchronize(e1)) != hipSuccess) return (int)e;
        float t; (void)hipEventElapsedTime(&t, e0, e1); if (t < best) best = t;
    }
    *ms = best;
    return (int)hipGetLastError();
}

}
