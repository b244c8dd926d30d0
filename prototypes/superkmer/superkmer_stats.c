// What would a super-k-mer pipeline with record dedupe save?  CPU model on the
// bench's read model (uniform genome, uniform starts, strand flip, 0.5 % subs).
// Prints: windows, super-k-mer records (cap CAPK k-mers per record), distinct
// records, k-mer expansions after dedupe (sum over distinct records of their k-mers).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static uint64_t rng = 88172645463325252ull;
static inline uint64_t xr(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
static inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
typedef struct { uint64_t a, b; uint32_t cnt; } Ent;
int main(int argc, char **argv) {
    const int K = 31, M = argc > 1 ? atoi(argv[1]) : 13, L = 150, CAPK = argc > 2 ? atoi(argv[2]) : 30;
    const long G = argc > 3 ? atol(argv[3]) : 2000000, NR = (long)(G * 15.0 / L);   // 15x coverage like the bench
    uint8_t *g = malloc(G);
    for (long i = 0; i < G; ++i) g[i] = xr() & 3;
    size_t cap = 1; while (cap < (size_t)NR * 40) cap <<= 1;
    Ent *tab = calloc(cap, sizeof(Ent));
    long windows = 0, records = 0, distinct = 0, expansions = 0;
    uint8_t rd[256];
    for (long r = 0; r < NR; ++r) {
        long st = xr() % (G - L + 1);
        int flip = xr() & 1;
        for (int i = 0; i < L; ++i) rd[i] = flip ? 3 - g[st + L - 1 - i] : g[st + i];
        for (int i = 0; i < L; ++i) if (xr() % 1000 < 5) rd[i] = (rd[i] + 1 + xr() % 3) & 3;
        // canonical m-mer order value per position
        uint32_t ord[256];
        for (int j = 0; j + M <= L; ++j) {
            uint32_t f = 0, c = 0;
            for (int t = 0; t < M; ++t) { f = (f << 2) | rd[j + t]; c |= (uint32_t)(3 - rd[j + t]) << (2 * t); }
            ord[j] = mix32(f < c ? f : c);
        }
        // minimizer VALUE per window; a record = maximal run of windows with the same value, capped
        int w0 = 0; uint32_t cur = 0;
        for (int w = 0; w + K <= L; ++w) {
            uint32_t mn = 0xFFFFFFFFu;
            for (int j = w; j <= w + K - M; ++j) if (ord[j] < mn) mn = ord[j];
            ++windows;
            if (w == 0) { cur = mn; w0 = 0; continue; }
            if (mn != cur || w - w0 >= CAPK) {
                // emit [w0, w): bases w0 .. w-1+K-1 ; canonical string form: min(fwd, rc) so both strands dedupe together
                int nb = (w - w0) + K - 1; uint64_t fa = 0, fb = 0, ca = 0, cb = 0;
                for (int t = 0; t < nb; ++t) { uint64_t b = rd[w0 + t]; if (t < 32) fa |= b << (2 * t); else fb |= b << (2 * (t - 32)); }
                for (int t = 0; t < nb; ++t) { uint64_t b = 3 - rd[w0 + nb - 1 - t]; if (t < 32) ca |= b << (2 * t); else cb |= b << (2 * (t - 32)); }
                if (cb < fb || (cb == fb && ca < fa)) { fa = ca; fb = cb; }
                fb |= (uint64_t)nb << 58;
                size_t h = (size_t)((fa * 0x9E3779B97F4A7C15ull) ^ (fb * 0xC2B2AE3D27D4EB4Full)) & (cap - 1);
                while (tab[h].cnt && (tab[h].a != fa || tab[h].b != fb)) h = (h + 1) & (cap - 1);
                if (!tab[h].cnt) { tab[h].a = fa; tab[h].b = fb; ++distinct; expansions += w - w0; }
                tab[h].cnt++; ++records;
                w0 = w; cur = mn;
            }
        }
        { int w = L - K + 1; int nb = (w - w0) + K - 1; uint64_t fa = 0, fb = 0, ca = 0, cb = 0;
          for (int t = 0; t < nb; ++t) { uint64_t b = rd[w0 + t]; if (t < 32) fa |= b << (2 * t); else fb |= b << (2 * (t - 32)); }
          for (int t = 0; t < nb; ++t) { uint64_t b = 3 - rd[w0 + nb - 1 - t]; if (t < 32) ca |= b << (2 * t); else cb |= b << (2 * (t - 32)); }
          if (cb < fb || (cb == fb && ca < fa)) { fa = ca; fb = cb; }
          fb |= (uint64_t)nb << 58;
          size_t h = (size_t)((fa * 0x9E3779B97F4A7C15ull) ^ (fb * 0xC2B2AE3D27D4EB4Full)) & (cap - 1);
          while (tab[h].cnt && (tab[h].a != fa || tab[h].b != fb)) h = (h + 1) & (cap - 1);
          if (!tab[h].cnt) { tab[h].a = fa; tab[h].b = fb; ++distinct; expansions += w - w0; }
          tab[h].cnt++; ++records; }
    }
    printf("m=%d cap=%d: windows %ld  records %ld (%.2f k-mers/record)  distinct records %ld (%.1f%%)  k-mer inserts after dedupe %ld (%.1f%% of windows)\n",
           M, CAPK, windows, records, (double)windows / records, distinct, 100.0 * distinct / records, expansions, 100.0 * expansions / windows);
    return 0;
}
