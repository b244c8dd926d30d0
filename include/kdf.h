/*
 * kdf.h -- C ABI of the MI355X-native canonical k-mer count / filter / probe
 * engine (libkdf.so).  This is the drop-in boundary for the one hot path of
 * jlanej/kmer_denovo_filter; every entry point names the reference interface
 * it replaces (file:line relative to the reference repository root).
 *
 * The reference has no FFI: the path sits behind Python helper functions that
 * spawn `samtools fasta | jellyfish count [-C --if] ; jellyfish dump/query`.
 * A maintainer binds this header with ctypes (INTEGRATION.md shows the stub);
 * kmer_denovo_filter_amd/_native.py is that binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross the boundary.
 *   - every function returns 0 on success or a KDF_ERR_* code; the message is
 *     kdf_last_error(h) (or kdf_last_error(NULL) for create/IO failures).
 *     The Python mirror raises RuntimeError, matching the reference's
 *     RuntimeError("jellyfish ... failed: ...") convention
 *     (core/jellyfish_wrappers.py:239-242, discovery/pipeline.py:168-172).
 *   - the caller owns host buffers; the engine owns device memory.
 *   - one engine = one GPU = one table; a handle is not thread-safe.
 *   - keys are canonical k-mers in the Jellyfish encoding: 2 bits per base,
 *     A=0 C=1 G=2 T=3, leftmost base most significant, canonical = numeric min
 *     of the k-mer and its reverse complement (== kmer_utils.canonicalize,
 *     src/kmer_denovo_filter/kmer_utils.py:35-38).  k <= 32: one uint64 (lo);
 *     33 <= k <= 63: (lo, hi) pair.  `hi` arrays may be NULL when k <= 32.
 *
 * Read streams
 *   Reads are handed over as ONE 2-bit-packed base stream plus a 1-bit
 *   "invalid" mask.  Base i lives in bits 2*(i%32) of packed[i/32]; bit (i%64)
 *   of invalid[i/64] is set when position i is not A/C/G/T (N, IUPAC) or is the
 *   single separator position that kdf_pack_reads() inserts after every read,
 *   so that no window spans two records (Jellyfish: windows never span FASTA
 *   records).  A window [i, i+k) is counted iff none of its k positions is
 *   invalid.  This keeps the kernels free of per-read bookkeeping and load
 *   balanced for ragged reads.
 */
#ifndef KDF_H
#define KDF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KDF_OK               0
#define KDF_ERR_INVALID      1   /* bad argument (k out of range, NULL pointer ...) */
#define KDF_ERR_HIP          2   /* HIP runtime error; message carries hipGetErrorString */
#define KDF_ERR_NOMEM        3   /* host or device allocation failed */
#define KDF_ERR_TABLE_FULL   4   /* a bucket overflowed: never silent, grow and redo */
#define KDF_ERR_IO           5   /* file open/read/format error (BAM/FASTA/.jf) */
#define KDF_ERR_STATE        6   /* call not valid in the engine's current mode */

typedef struct kdf_engine kdf_engine;
typedef struct kdf_reader kdf_reader;

/* ---------------------------------------------------------------- engine -- */

/* Create an engine on HIP device `device` for k-mers of length k (1..63) with
 * room for at least capacity_hint distinct keys before the first grow.
 * Replaces the process launch + `-m k -s SIZE` of `jellyfish count`
 * (core/jellyfish_wrappers.py:167-176,313-321; discovery/pipeline.py:114-122). */
int kdf_create(int device, int k, uint64_t capacity_hint, kdf_engine **out);
void kdf_destroy(kdf_engine *h);
const char *kdf_last_error(const kdf_engine *h);

/* Use an externally created hipStream_t (e.g. torch's current stream) for all
 * subsequent launches; NULL restores the engine's own stream. */
int kdf_set_stream(kdf_engine *h, void *hip_stream);
int kdf_synchronize(kdf_engine *h);

/* Empty the table (all keys and counts dropped). */
int kdf_clear(kdf_engine *h);
/* Make room for at least n_keys distinct keys (rehashes the live entries). */
int kdf_reserve(kdf_engine *h, uint64_t n_keys);
/* capacity in slots, distinct keys currently stored, valid windows processed
 * by the count calls since the last clear.  Any pointer may be NULL. */
int kdf_stats(kdf_engine *h, uint64_t *capacity, uint64_t *distinct, uint64_t *windows);

/* Count calls in insert mode are DEFERRED: a call partitions its batch (or, for small batches, only appends it to a
 * pending stream) and returns; the table itself is updated when something reads it -- kdf_stats, kdf_query*,
 * kdf_count_ge, kdf_export_*, kdf_scan_*, kdf_add_pairs*, kdf_reserve, kdf_set_option -- or when the pending work
 * fills its budget, so that a sample streamed in hundreds of batches pays the table rewrite of the binned pipeline
 * once per flush, not once per batch (`jellyfish count` over the whole `samtools fasta` pipe,
 * discovery/pipeline.py:106-172).  kdf_flush applies everything pending now; errors of deferred work (a table that
 * cannot grow ...) surface there or in the call that triggered the flush.  kdf_clear drops pending work. */
int kdf_flush(kdf_engine *h);

/* Tuning knobs and counters (tests force either kernel path through these):
 *   options  "force_path" 0 auto / 1 direct global-table kernels / 2 binned LDS-bucket pipeline (every count call
 *            partitions its batch at once) / 4 count --if through the sieve only; "binned_min_positions" (pending
 *            positions below which a flush uses the direct kernels); "key_parts" / "key_part" (count only
 *            the windows whose key lies in slice key_part of key_parts of the key space --
 *            ranges of the LOW 16 hash bits, so a slice spreads over the whole table -- so that
 *            a sample whose distinct k-mers exceed one table is counted slice by slice over
 *            the same stream; insert mode only; 0/1 = everything); "binned_max_positions" (positions per
 *            partition pass, <= 2^31: longer streams take several passes); "binned_filtered_min_log2cap";
 *            "sieve_bits" (bits per filter key; 0 = by size); "binned_bytes_per_position" (a flush goes
 *            binned only when pending positions x this >= table bytes: kernel C rewrites the whole table; default 70);
 *            "defer" (1 default; 0: every count call ends with a flush), "defer_max_bytes" (budget of the ring of
 *            partitioned entries, 0 = 40 % of the device's memory), "l1_positions" (size from which the pending
 *            stream of small batches is partitioned, default 2^30), "l1_direct_positions" (batches from this size on
 *            are partitioned where they lie, default 2^28); "fused_dump" (0 default; 1: kdf_export_ge_dev with min_count >= 1
 *            called while partition passes are pending is written by the flush that applies them -- kernel C dumps every
 *            bucket it holds -- instead of by a pass over the table afterwards; falls back to that pass when a bucket
 *            overflowed or was split as heavy; env KDF_FUSED_DUMP=1 sets the default);
 *            "hash_shift" (0..8, empty table only: the home slot ignores that many top hash bits -- the table of
 *            an OWNER rank of the multi-GPU merge, see kdf_add_pairs_multi_dev; such an engine counts through the
 *            direct kernels only); "merge_min_pairs" (below this many pairs kdf_add_pairs* skips the bucket merge);
 *            "big_bucket_log2cap" (default 32; KDF_BIG_BUCKET_LOG2CAP: tables of 2^that slots and more have buckets of
 *            twice the slots -- an internal layout: dumps, queries and index files do not depend on it; a live table
 *            is re-bucketed when the option changes its bucket size); "debug_flags" (experiments: 64 one piece per
 *            workgroup in the piece sort, 2048 partition without the bucket kernel, 4096 force the skew instantiation).
 *            Environment, read when a partition is planned (experiments: DESIGN.md section 3.2 has what they measured):
 *            KDF_C1 (coarse bits of the partition), KDF_PIECE_FILL (how full the pieces are planned, default 0.98)
 *   stats    "binned_passes" (partition passes), "flushes" (kernel C launches), "pending_passes",
 *            "pending_positions", "ring_bytes", "replayed_buckets", "heavy_buckets" (buckets of skewed flushes that
 *            were shared by several workgroups), "log2cap", "bucket_bits", "hash_shift", "defer", "fused_dump", "fused_dumps" (dumps written by a flush),
 *            "last_count_path" (0 direct / 1 binned / 3 sieve), "last_merge_path" (1 LDS bucket
 *            merge, 2 global atomics); "trash0" .. "trash63" (phase cycle sums of -DKB_TIMING variant builds) */
int kdf_set_option(kdf_engine *h, const char *name, int64_t value);
/* Free / total HBM of a device (hipMemGetInfo): the child-count mirror sizes "key_parts" with it. */
int kdf_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);
int kdf_get_stat(kdf_engine *h, const char *name, int64_t *value);

/* Measurement hook (bench.py): when enabled, every launch of the dominant
 * stream kernel is bracketed by HIP events on the launch stream.
 * kdf_profile_read returns the summed kernel milliseconds, the number of
 * launches and the stream positions they covered since kdf_profile(h, 1). */
int kdf_profile(kdf_engine *h, int enable);
int kdf_profile_read(kdf_engine *h, double *kernel_ms, uint64_t *launches, uint64_t *positions);
/* Binned passes only: summed milliseconds of the four stages (A slab sort, the
 * planning kernels, B piece sort, C bucket kernel) and the number of passes. */
int kdf_profile_stages(kdf_engine *h, double *stage_ms4, uint64_t *passes);

/* ------------------------------------------------- count (insert) stage -- */

/* `jellyfish count -m k -C` over a read stream: every valid window's canonical
 * k-mer is inserted / incremented (saturating uint32, like Jellyfish's 4-byte
 * output counter).  Replaces the counting half of
 * _extract_child_kmers_discovery (discovery/pipeline.py:114-172) and of
 * _ensure_ref_jf (core/jellyfish_wrappers.py:313-326).  Host buffers. */
int kdf_count_reads(kdf_engine *h, const uint64_t *packed, const uint64_t *invalid,
                    uint64_t n_bases);
/* Same with the stream already resident in HBM (device pointers, padded as
 * kdf_stream_words() says). */
int kdf_count_reads_dev(kdf_engine *h, const void *d_packed, const void *d_invalid,
                        uint64_t n_bases);

/* Double-buffered feeding of a streamed sample (the `samtools fasta | jellyfish count` pipe,
 * core/jellyfish_wrappers.py:166-199, as two overlapping stages): kdf_upload_reads_async copies a host batch into
 * device staging slot 0 or 1 on a copy stream of the engine's own and returns at once when the host arrays are pinned
 * (kdf_host_alloc; pageable arrays work too, the copy is then synchronous); kdf_count_uploaded counts the batch a
 * slot holds (filtered != 0: count --if) on the engine's stream.  Upload batch i + 1, then count batch i: the copy
 * runs under the count.  The host arrays of a batch may be rewritten once kdf_count_uploaded for THAT batch returned. */
int kdf_host_alloc(uint64_t bytes, void **out);
int kdf_host_free(void *p);
int kdf_upload_reads_async(kdf_engine *h, int slot, const uint64_t *packed, const uint64_t *invalid,
                           uint64_t n_bases);
int kdf_count_uploaded(kdf_engine *h, int slot, int filtered);

/* Insert-or-add explicit (key, count) pairs: key i gains counts[i] (counts ==
 * NULL adds 0, i.e. plain insertion).  Loads an on-disk index into the table
 * (`jellyfish query` mmaps the .jf; discovery/pipeline.py:286-288) and merges
 * per-GPU partial counts after the owner-partitioned exchange (`jellyfish
 * merge`, core/jellyfish_wrappers.py:335-366). */
int kdf_add_pairs(kdf_engine *h, const uint64_t *keys_lo, const uint64_t *keys_hi,
                  const uint32_t *counts, uint64_t n);
int kdf_add_pairs_dev(kdf_engine *h, const void *d_keys_lo, const void *d_keys_hi,
                      const void *d_counts, uint64_t n);
/* The owner's half of the multi-GPU merge: `nseg` (<= 64 for the fast path) segments of pairs in HBM, one per source
 * rank, summed into the table in one call (d_keys_lo / d_keys_hi / d_counts / n are HOST arrays of nseg device
 * pointers / lengths; d_keys_hi may be NULL for k <= 32).  When every segment is grouped by this table's buckets in
 * ascending order -- what kdf_export_parts_dev writes -- each bucket is merged in LDS by one workgroup and written
 * once (the test runs on the device; any other order is merged through global atomics, same result).  An owner table
 * should be created with option "hash_shift" = floor(log2(world)): its home slots then ignore the hash bits that name
 * the owner, so the table is used over its whole length and the senders' order is its own. */
int kdf_add_pairs_multi_dev(kdf_engine *h, uint32_t nseg, const void *const *d_keys_lo,
                            const void *const *d_keys_hi, const void *const *d_counts, const uint64_t *n);

/* ------------------------------------------------ count --if (filter) ---- */

/* Load the `--if` filter: the table becomes exactly these canonical keys with
 * count 0 (core/jellyfish_wrappers.py:173, discovery/pipeline.py:383).  Keys
 * must already be canonical (the reference's filter files are). */
int kdf_load_filter(kdf_engine *h, const uint64_t *keys_lo, const uint64_t *keys_hi,
                    uint64_t n);
/* The same with the keys already resident in HBM (device pointers): the hand-off between the discovery stages
 * (child candidates -> reference subtraction -> parent filter, discovery/pipeline.py:207-226,286-304,515-532)
 * without a host round trip. */
int kdf_load_filter_dev(kdf_engine *h, const void *d_keys_lo, const void *d_keys_hi, uint64_t n);
/* Zero every count, keep the keys (and the filter mode): the same filter counted against the next parent
 * (discovery/pipeline.py:462-612 builds a fresh --if table per parent). */
int kdf_reset_counts(kdf_engine *h);
/* `jellyfish count -C --if`: only windows whose canonical k-mer is in the table
 * are counted; nothing is inserted.  Replaces _scan_parent_jellyfish
 * (core/jellyfish_wrappers.py:115-283) and _count_parent_jellyfish
 * (discovery/pipeline.py:322-459). */
int kdf_count_reads_filtered(kdf_engine *h, const uint64_t *packed,
                             const uint64_t *invalid, uint64_t n_bases);
int kdf_count_reads_filtered_dev(kdf_engine *h, const void *d_packed,
                                 const void *d_invalid, uint64_t n_bases);

/* ------------------------------------------------------- query / dump ---- */

/* `jellyfish query idx -s kmers.fa`: counts_out[i] = count of key i, 0 when
 * absent, INPUT ORDER (discovery/pipeline.py:286-304,515-532,565-583;
 * kmer_utils.py:152-183). */
int kdf_query(kdf_engine *h, const uint64_t *keys_lo, const uint64_t *keys_hi,
              uint64_t n, uint32_t *counts_out);
/* device-pointer form: d_counts_out can be a torch tensor that is then
 * all-reduced over RCCL (multi-GPU merge of per-rank filter counts). */
int kdf_query_dev(kdf_engine *h, const void *d_keys_lo, const void *d_keys_hi,
                  uint64_t n, void *d_counts_out);

/* `jellyfish dump -c -L min_count`: number of entries with count >= min_count
 * (min_count = 0 returns every stored key, counts 0 included). */
int kdf_count_ge(kdf_engine *h, uint32_t min_count, uint64_t *n_out);
/* ... and the entries themselves, ASCENDING key order (deterministic; the
 * reference does not rely on Jellyfish's hash order).  cap = room in the out
 * arrays; *n_out = entries written.  keys_hi_out / counts_out may be NULL.
 * (discovery/pipeline.py:207-226; core/jellyfish_wrappers.py:262-272) */
int kdf_export_ge(kdf_engine *h, uint32_t min_count, uint64_t *keys_lo_out,
                  uint64_t *keys_hi_out, uint32_t *counts_out, uint64_t cap,
                  uint64_t *n_out);

/* Device-to-device dump: the entries go to caller-owned HBM buffers (e.g. torch
 * tensors that are then exchanged over RCCL, or the next stage's filter); unsorted
 * unless sorted != 0.  ONE pass over the table: at most `cap` entries are written,
 * *n_out is the number the dump holds; KDF_ERR_INVALID when that exceeds cap
 * (kdf_count_ge sizes the buffers).  d_keys_hi_out may be NULL for k <= 32,
 * d_counts_out may be NULL when sorted == 0.  Synchronises the engine's stream. */
int kdf_export_ge_dev(kdf_engine *h, uint32_t min_count, void *d_keys_lo_out,
                      void *d_keys_hi_out, void *d_counts_out, uint64_t cap,
                      int sorted, uint64_t *n_out);
/* The multi-GPU exchange of the full count stage (Jellyfish `merge`,
 * core/jellyfish_wrappers.py:335-366, done over xGMI): dump every (key, count >= min_count)
 * pair to DEVICE arrays grouped by owner rank, owner(key) = ((hash(key) >> 48) * parts) >> 16
 * with the table's own hash -- so the owners are contiguous slot ranges and no sort or
 * partition pass is needed.  part_counts_out[parts] (host) receives the pairs per owner;
 * part p occupies [sum(counts[0..p)), +counts[p]) of the outputs.  The whole dump is in HASH
 * ORDER (grouped by the top log2cap - 6 hash bits, ascending), which is bucket order in every
 * owner table of up to 64x this table's slots per owner: kdf_add_pairs_multi_dev merges it
 * bucket by bucket in LDS.  One stream synchronisation.
 * parts <= 64; KDF_ERR_STATE when the table is smaller than 2^28 slots. */
int kdf_export_parts_dev(kdf_engine *h, uint32_t min_count, uint32_t parts, void *d_keys_lo_out,
                         void *d_keys_hi_out, void *d_counts_out, uint64_t cap,
                         uint64_t *part_counts_out, uint64_t *n_out);
/* The same dump written straight into the buffer the all-to-all sends: part p is ONE byte segment
 * [lo x n_p | hi x n_p (k > 32) | counts x n_p] starting at part_bytes_out[p] (multiples of 8; part_bytes_out[parts] =
 * total bytes; cap_bytes >= n * (12 or 20) + 8 * parts always suffices).  A receiver hands the segments it got to
 * kdf_add_pairs_multi_dev as they lie. */
int kdf_export_parts_packed_dev(kdf_engine *h, uint32_t min_count, uint32_t parts, void *d_buf, uint64_t cap_bytes,
                                uint64_t *part_counts_out, uint64_t *part_bytes_out, uint64_t *n_out);

/* The count of every listed key becomes counts[i] (device pointers; every key must be stored: KDF_ERR_INVALID
 * otherwise).  How the sum of the ranks' `count --if` tallies goes back into each rank's table after the all-reduce
 * (kmer_denovo_filter_amd/distributed.py; the reference's parent scan is one process, discovery/pipeline.py:377-443). */
int kdf_set_counts_dev(kdf_engine *h, const void *d_keys_lo, const void *d_keys_hi, const void *d_counts, uint64_t n);

/* ------------------------------------------------------ Module-3 scan ---- */

/* Probe every window of a read stream against the table: bit i of hit_bits is
 * set iff window i is valid and its canonical k-mer is stored with count > 0
 * (JellyfishKmerQuery: parts[1] != "0", kmer_utils.py:181).  hit_bits has
 * kdf_stream_words()' invalid-word count of uint64 words.
 * When read_offsets != NULL (n_reads+1 stream offsets of the read starts, as
 * kdf_pack_reads() returns them) distinct_out[r] = number of DISTINCT canonical
 * k-mers hit in read r (len(unique_in_read), core/bam_scanner.py:435-442).
 * Replaces the inner loop of _scan_contig_for_hits (core/bam_scanner.py:396-474)
 * and JellyfishKmerQuery.scan_read (kmer_utils.py:209-238). */
int kdf_scan_reads(kdf_engine *h, const uint64_t *packed, const uint64_t *invalid,
                   uint64_t n_bases, const int64_t *read_offsets, int64_t n_reads,
                   uint64_t *hit_bits, uint32_t *distinct_out);
int kdf_scan_reads_dev(kdf_engine *h, const void *d_packed, const void *d_invalid,
                       uint64_t n_bases, void *d_hit_bits);

/* ------------------------------------------------------- host utilities -- */

/* Words a stream of n_bases needs (padding included): the packed array must
 * hold *packed_words uint64, the invalid / hit arrays *mask_words uint64. */
void kdf_stream_words(uint64_t n_bases, uint64_t *packed_words, uint64_t *mask_words);

/* Pack ASCII records (A/C/G/T any case; everything else invalid) into a stream
 * with one separator after each record.  offsets[n_reads+1] delimit the records
 * in `ascii`.  stream_offsets_out[n_reads+1] (may be NULL) receives each
 * record's start in the stream (record r occupies [so[r], so[r]+len_r); the
 * last entry is n_bases).  Stream length = sum(len) + n_reads.
 * Restates what `samtools fasta` hands to Jellyfish (one FASTA record per read). */
int kdf_pack_reads(const char *ascii, const int64_t *offsets, int64_t n_reads,
                   uint64_t *packed_out, uint64_t *invalid_out,
                   int64_t *stream_offsets_out, uint64_t *n_bases_out);

/* Canonical key of one ASCII k-mer; returns KDF_ERR_INVALID on a non-ACGT byte
 * (kmer_utils.py:35-38). */
int kdf_canonical(const char *kmer, int k, uint64_t *lo, uint64_t *hi);

/* ------------------------------------------------- BAM / FASTA feeders ---- */

/* Streaming BAM reader with `samtools fasta -F flag_off` semantics
 * (core/jellyfish_wrappers.py:159-165, discovery/pipeline.py:106-112): records
 * with any flag_off bit are dropped; when collapse != 0 each run of consecutive
 * same-QNAME records yields at most one record per read part (READ1 / READ2 /
 * other), a record with qualities beating one without, first wins.
 * collapse = 0, flag_off = 0x500 gives Module 3's pysam iteration
 * (core/bam_scanner.py:405-409: skip SECONDARY and DUPLICATE, keep the rest). */
int kdf_bam_open(const char *path, uint32_t flag_off, int collapse, int threads,
                 kdf_reader **out);
/* The same reader over ONE RANGE of the file: the BGZF blocks that hold the records are cut at parts - 1 byte offsets
 * and the record stream at the first record boundary behind each cut -- the first QNAME-run boundary when runs are
 * collapsed, so no run is split: the parts 0 .. parts - 1 together yield exactly the records kdf_bam_open yields, each
 * once, whatever `parts` is, and a part only reads and inflates its own bytes.  This is how the read stream of one
 * sample is sharded over the GPUs of a node (reads are independent units) and over several reader
 * pipelines inside one process.  Record ordinals (kdf_reader_last_ordinals) count from the part's first record. */
int kdf_bam_open_range(const char *path, uint32_t flag_off, int collapse, int threads,
                       int part, int parts, kdf_reader **out);
/* Multi-record FASTA (reference genome, plain or gzip) as a reader: one stream
 * record per sequence, any case, non-ACGT invalid.  A sequence longer than a
 * batch is continued in the next batch k-1 bases back, so no window is lost or
 * counted twice.  Replaces Jellyfish's own FASTA parsing in _ensure_ref_jf
 * (core/jellyfish_wrappers.py:313-321). */
int kdf_fasta_open(const char *path, int k, kdf_reader **out);
/* Fill up to max_bases stream positions / max_reads records.  Returns the batch
 * through the out pointers; *n_reads_out = 0 at end of file.  packed_out /
 * invalid_out must hold kdf_stream_words(max_bases) words;
 * stream_offsets_out max_reads+1 entries.  A record longer than max_bases is an
 * error (KDF_ERR_INVALID). */
int kdf_reader_next(kdf_reader *r, uint64_t max_bases, int64_t max_reads,
                    uint64_t *packed_out, uint64_t *invalid_out,
                    int64_t *stream_offsets_out, int64_t *n_reads_out,
                    uint64_t *n_bases_out);
/* Per-record metadata of the LAST batch (BAM readers; arrays of n_reads):
 * flag, ref_id, pos, and the offsets of NUL-terminated names in name_buf. */
int kdf_reader_last_meta(kdf_reader *r, const uint16_t **flags, const int32_t **ref_ids,
                         const int32_t **positions, const char **name_buf,
                         const int64_t **name_offsets);
/* Alignment details for Module 3's post-processing of informative reads
 * (core/bam_scanner.py:97-117,284-337: CIGAR -> reference coordinates, SA tag,
 * soft clips).  Call kdf_reader_want_aux(r, 1) after kdf_bam_open; then, for the
 * last batch: cigar[cigar_offsets[i] .. cigar_offsets[i+1]) are record i's CIGAR
 * operations in BAM encoding (len << 4 | op), sa_offsets[i] is the offset of its
 * NUL-terminated SA:Z value in sa_buf or -1. */
int kdf_reader_want_aux(kdf_reader *r, int enable);
int kdf_reader_last_aux(kdf_reader *r, const uint32_t **cigar, const int64_t **cigar_offsets,
                        const char **sa_buf, const int64_t **sa_offsets);
/* Base qualities (qual[qual_offsets[i] .. qual_offsets[i+1]), 0xFF when the record
 * has none) and MAPQ of the last batch; needs kdf_reader_want_aux.  Used by the
 * VCF-mode producer (kmer_utils.py:1037-1172: --min-baseq, vcf/pipeline.py:673: --min-mapq). */
int kdf_reader_last_quals(kdf_reader *r, const uint8_t **qual, const int64_t **qual_offsets,
                          const uint8_t **mapq);
/* Record number in the file (0-based, counted before the flag filter) of every read
 * of the last batch: the handle kdf_bam_write_subset takes. */
int kdf_reader_last_ordinals(kdf_reader *r, const uint64_t **ordinals);
/* Reference sequence names of a BAM reader (header order = ref_id). */
int kdf_reader_ref_count(kdf_reader *r);
const char *kdf_reader_ref_name(kdf_reader *r, int i);
void kdf_reader_close(kdf_reader *r);
const char *kdf_reader_error(const kdf_reader *r);

/* N4: the informative-reads BAM (discovery/pipeline.py:1979-2079 `_write_informative_reads_discovery`,
 * vcf/pipeline.py:1307-1357 `_write_informative_reads`: pysam write + `samtools sort` + `samtools index`).
 * Copies the records ordinals[0..n) (strictly ascending file record numbers, see
 * kdf_reader_last_ordinals) of src_bam to dst_bam byte for byte, appending
 * aux[aux_offsets[i] .. aux_offsets[i+1]) -- optional fields in BAM encoding, e.g.
 * "dkC\x01" or "DVZchr1:5:A:T\0" -- to record i (aux may be NULL).  With sort_and_index
 * the records are coordinate sorted (samtools order: tid unsigned, pos, strand; stable),
 * the header gets @HD SO:coordinate and dst_bam + ".bai" is written (SAM spec 5.2). */
int kdf_bam_write_subset(const char *src_bam, const char *dst_bam, const uint64_t *ordinals, uint64_t n,
                         const uint8_t *aux, const uint64_t *aux_offsets, int sort_and_index, int threads,
                         uint64_t *n_written);

#ifdef __cplusplus
}
#endif
#endif /* KDF_H */
