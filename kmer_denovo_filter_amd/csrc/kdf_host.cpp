// kdf_host.cpp -- host side of libkdf.so: ASCII -> 2-bit stream packer and the
// streaming read feeders (BGZF/BAM with `samtools fasta -F` semantics, FASTA).
// Replaces the `samtools fasta` half of the reference's pipes
// (core/jellyfish_wrappers.py:159-165, discovery/pipeline.py:106-112,369-375)
// and Jellyfish's FASTA parsing of the reference genome
// (core/jellyfish_wrappers.py:313-321).  zlib only (no htslib in the image);
// CRAM is not supported.
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kdf.h"

namespace {

thread_local std::string g_host_err;

struct CodeTable {
    uint8_t t[256];
    CodeTable() {
        memset(t, 4, sizeof t);
        t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
    }
};
const CodeTable kCode;

// BAM 4-bit base code -> 2-bit code (A=1,C=2,G=4,T=8), 4 = invalid
const uint8_t kNt16[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};

// BAM packs two bases per byte (first base in the high nibble).  kSeq4[b] =
// 2-bit codes of the two bases in bits 0-3 (first base in bits 0-1) and their
// invalid flags in bits 4-5.
struct Seq4Table {
    uint8_t t[256];
    Seq4Table() {
        for (int b = 0; b < 256; ++b) {
            const uint8_t c0 = kNt16[b >> 4], c1 = kNt16[b & 15];
            t[b] = (uint8_t)((c0 < 4 ? c0 : 0) | ((c1 < 4 ? c1 : 0) << 2) | ((c0 > 3) << 4) | ((c1 > 3) << 5));
        }
    }
};
const Seq4Table kSeq4;

// Appends bases to a (packed, invalid) stream under construction.  The arrays
// must be zero on entry (begin() does that); finish() marks everything from the
// end of the stream to the end of the arrays invalid.
struct StreamWriter {
    uint64_t *packed;
    uint64_t *invalid;
    uint64_t n = 0;
    uint64_t pw = 0, mw = 0;
    StreamWriter(uint64_t *p, uint64_t *m) : packed(p), invalid(m) {}
    void begin(uint64_t capacity_bases) {
        kdf_stream_words(capacity_bases, &pw, &mw);
        memset(packed, 0, pw * 8);
        memset(invalid, 0, mw * 8);
    }
    inline void put_bits(uint64_t codes, uint64_t inv, int nb) {   // nb bases (<= 16), LSB first
        const uint64_t w = n >> 5; const int sh = (int)(n & 31) * 2;
        packed[w] |= codes << sh;
        if (sh + 2 * nb > 64) packed[w + 1] |= codes >> (64 - sh);
        const uint64_t m = n >> 6; const int ms = (int)(n & 63);
        invalid[m] |= inv << ms;
        if (ms + nb > 64) invalid[m + 1] |= inv >> (64 - ms);
        n += nb;
    }
    inline void put(uint8_t code) { put_bits(code < 4 ? code : 0, code > 3, 1); }
    inline void put_sep() { put_bits(0, 1, 1); }
    // BAM 4-bit sequence, l_seq bases
    void put_seq4(const uint8_t *sq, int64_t l_seq) {
        int64_t i = 0;
        for (; i + 16 <= l_seq; i += 16) {                       // 8 bytes -> 16 bases per append
            uint64_t codes = 0, inv = 0;
            for (int j = 0; j < 8; ++j) {
                const uint8_t v = kSeq4.t[sq[(i >> 1) + j]];
                codes |= (uint64_t)(v & 15) << (4 * j);
                inv |= (uint64_t)(v >> 4) << (2 * j);
            }
            put_bits(codes, inv, 16);
        }
        for (; i + 2 <= l_seq; i += 2) {
            const uint8_t v = kSeq4.t[sq[i >> 1]];
            put_bits(v & 15, v >> 4, 2);
        }
        if (i < l_seq) {
            const uint8_t v = kSeq4.t[sq[i >> 1]];
            put_bits(v & 3, (v >> 4) & 1, 1);
        }
    }
    // ASCII bases
    void put_ascii(const char *a, int64_t len) {
        int64_t i = 0;
        for (; i + 16 <= len; i += 16) {
            uint64_t codes = 0, inv = 0;
            for (int j = 0; j < 16; ++j) {
                const uint8_t c = kCode.t[(uint8_t)a[i + j]];
                codes |= (uint64_t)(c & 3) << (2 * j);
                inv |= (uint64_t)(c > 3) << j;
                if (c > 3) codes &= ~(3ull << (2 * j));
            }
            put_bits(codes, inv, 16);
        }
        for (; i < len; ++i) put(kCode.t[(uint8_t)a[i]]);
    }
    // npos positions of another stream (arrays with >= 1 word of slack after the last used word) from its position sp
    void append_stream(const uint64_t *sp_packed, const uint64_t *sp_invalid, uint64_t sp, uint64_t npos) {
        auto copy = [](uint64_t *dst, uint64_t db, const uint64_t *src, uint64_t sb, uint64_t nbits) {
            while (nbits) {
                const int take = nbits < 64 ? (int)nbits : 64;
                const uint64_t sw = sb >> 6; const int ss = (int)(sb & 63);
                uint64_t v = src[sw] >> ss;
                if (ss) v |= src[sw + 1] << (64 - ss);
                if (take < 64) v &= (1ull << take) - 1;
                const uint64_t dw = db >> 6; const int ds = (int)(db & 63);
                dst[dw] |= v << ds;
                if (ds && ds + take > 64) dst[dw + 1] |= v >> (64 - ds);
                sb += take; db += take; nbits -= take;
            }
        };
        copy(packed, 2 * n, sp_packed, 2 * sp, 2 * npos);
        copy(invalid, n, sp_invalid, sp, npos);
        n += npos;
    }
    void finish() {
        if (n & 63) invalid[n >> 6] |= ~0ull << (n & 63);
        const uint64_t first = (n + 63) >> 6;
        if (mw > first) memset(invalid + first, 0xFF, (mw - first) * 8);
    }
};

// ----------------------------------------------------------------- records --

struct Record {
    std::vector<uint8_t> seq4;    // BAM 4-bit packed sequence (FASTA readers: unused)
    int32_t l_seq = 0;
    std::string name;
    uint16_t flag = 0;
    int32_t ref_id = -1, pos = -1;
    // only filled when the reader was asked for alignment details (kdf_reader_want_aux)
    std::vector<uint32_t> cigar;  // BAM encoding: len << 4 | op
    std::string sa;               // SA:Z tag value, empty when absent
    bool has_sa = false;
    std::vector<uint8_t> qual;    // base qualities (0xFF... when absent)
    uint8_t mapq = 0;
    uint64_t ordinal = 0;         // 0-based record number in the file (before any flag filter)
};

}  // namespace

namespace {

struct BgzfBlock {
    std::vector<uint8_t> comp;     // deflate payload + 8-byte trailer
    std::vector<uint8_t> data;     // inflated
    int cdata = 0;
    uint32_t isize = 0;
    uint64_t file_off = 0;         // where the block starts in the file
    int state = 0;                 // 0 free, 1 queued, 2 done, 3 error
    std::string err;
};

// read one raw BGZF block.  returns 0 ok, 1 eof, -1 error (msg in err)
int bgzf_read_raw(FILE *fp, BgzfBlock &blk, std::string &err) {
    uint8_t hdr[18];
    blk.file_off = (uint64_t)ftello(fp);
    size_t got = fread(hdr, 1, 18, fp);
    if (got == 0) return 1;
    if (got != 18 || hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) { err = "not a BGZF block (bad gzip header)"; return -1; }
    const unsigned xlen = hdr[10] | (hdr[11] << 8);
    // the BC subfield is normally first (xlen == 6); handle the general case
    std::vector<uint8_t> extra(xlen);
    memcpy(extra.data(), hdr + 12, std::min<unsigned>(6, xlen));
    if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp) != xlen - 6) { err = "truncated BGZF extra field"; return -1; }
    int bsize = -1;
    for (unsigned o = 0; o + 4 <= xlen;) {
        const unsigned slen = extra[o + 2] | (extra[o + 3] << 8);
        if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = extra[o + 4] | (extra[o + 5] << 8);
        o += 4 + slen;
    }
    if (bsize < 0) { err = "BGZF block without BC subfield"; return -1; }
    const int cdata = bsize - (int)xlen - 19;
    if (cdata < 0) { err = "corrupt BGZF block size"; return -1; }
    blk.comp.resize((size_t)cdata + 8);
    if (fread(blk.comp.data(), 1, blk.comp.size(), fp) != blk.comp.size()) { err = "truncated BGZF block"; return -1; }
    blk.cdata = cdata;
    const uint8_t *t = blk.comp.data() + cdata;
    blk.isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
    return 0;
}

// inflate blk.comp -> blk.data.  returns false on error (msg in blk.err)
bool bgzf_inflate(BgzfBlock &blk) {
    blk.data.resize(blk.isize);
    if (blk.isize == 0) return true;
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) { blk.err = "inflateInit2 failed"; return false; }
    zs.next_in = blk.comp.data(); zs.avail_in = (uInt)blk.cdata;
    zs.next_out = blk.data.data(); zs.avail_out = blk.isize;
    const int zr = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (zr != Z_STREAM_END || zs.avail_out != 0) { blk.err = "BGZF inflate failed"; return false; }
    return true;
}

}  // namespace

// Ordered, threaded BGZF decompression: one I/O thread reads raw blocks into a
// ring, `nthreads` workers inflate them, the parser consumes them in file order
// (what `samtools -@ n` does for the reference, core/jellyfish_wrappers.py:158-163).
struct BgzfPool {
    FILE *fp = nullptr;
    std::vector<BgzfBlock> ring;
    size_t head = 0, tail = 0;            // consume / fill sequence numbers
    std::deque<size_t> work;
    bool eof_read = false, stop = false, io_error = false;
    std::string io_err;
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_free;
    std::thread io;
    std::vector<std::thread> workers;

    BgzfPool(FILE *f, int nthreads) : fp(f), ring((size_t)nthreads * 8) {
        io = std::thread([this] { io_loop(); });
        for (int i = 0; i < nthreads; ++i) workers.emplace_back([this] { work_loop(); });
    }
    ~BgzfPool() {
        shutdown();
        if (io.joinable()) io.join();
        for (auto &t : workers) if (t.joinable()) t.join();
    }
    void shutdown() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_work.notify_all(); cv_free.notify_all(); cv_done.notify_all();
    }
    void io_loop() {
        for (;;) {
            size_t slot;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_free.wait(lk, [this] { return stop || tail - head < ring.size(); });
                if (stop) return;
                slot = tail % ring.size();
            }
            std::string err;
            const int rc = bgzf_read_raw(fp, ring[slot], err);      // slot is free: no one else touches it
            std::lock_guard<std::mutex> g(mu);
            if (rc == 1) { eof_read = true; cv_done.notify_all(); return; }
            if (rc < 0) { io_error = true; io_err = err; eof_read = true; cv_done.notify_all(); return; }
            ring[slot].state = 1; ring[slot].err.clear();
            work.push_back(tail);
            ++tail;
            cv_work.notify_one();
        }
    }
    void work_loop() {
        for (;;) {
            size_t seq;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [this] { return stop || !work.empty(); });
                if (stop) return;
                seq = work.front(); work.pop_front();
            }
            BgzfBlock &blk = ring[seq % ring.size()];
            const bool ok = bgzf_inflate(blk);
            std::lock_guard<std::mutex> g(mu);
            blk.state = ok ? 2 : 3;
            cv_done.notify_all();
        }
    }
    // append the next block's bytes to `out`.  0 ok, 1 eof, -1 error (msg in err)
    int next(std::vector<uint8_t> &out, std::string &err, uint64_t *file_off = nullptr) {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [this] { return stop || (head < tail && ring[head % ring.size()].state >= 2) || (eof_read && head == tail); });
        if (stop) return 1;
        if (head == tail) {
            if (io_error) { err = io_err; return -1; }
            return 1;
        }
        BgzfBlock &blk = ring[head % ring.size()];
        if (blk.state == 3) { err = blk.err; return -1; }
        lk.unlock();
        if (file_off) *file_off = blk.file_off;
        out.insert(out.end(), blk.data.begin(), blk.data.end());
        lk.lock();
        blk.state = 0;
        ++head;
        cv_free.notify_one();
        return 0;
    }
};


// ---- parallel record parsing ---------------------------------------------------
// With threads > 1 (and no alignment details wanted) the single parser thread is the
// bottleneck once BGZF inflation is spread over a pool (~0.25 us per record).  A chunker
// thread cuts the inflated byte stream into chunks of whole records (on QNAME-run
// boundaries when runs are collapsed, so every chunk is self-contained), worker threads
// turn chunks into chunk-local packed streams + metadata, and kdf_reader_next appends
// those, in order, to the caller's batch with bit-shifted word copies.
struct RawChunk { std::vector<uint8_t> bytes; uint64_t first_ordinal = 0; };
struct ParsedChunk {
    std::vector<uint64_t> packed, invalid;        // chunk-local stream, bit 0 = first base
    std::vector<int64_t> off;                     // n + 1 positions
    std::vector<uint16_t> flags; std::vector<int32_t> ref, pos;
    std::string names; std::vector<int64_t> name_off; std::vector<uint64_t> ordinal;
    // alignment details (kdf_reader_want_aux)
    std::vector<uint32_t> cigar; std::vector<int64_t> cigar_off;      // n + 1
    std::string sa; std::vector<int64_t> sa_off;                      // n; -1 = no SA tag
    std::vector<uint8_t> qual; std::vector<int64_t> qual_off;         // n + 1
    std::vector<uint8_t> mapq;
    size_t next = 0;                              // consumer cursor (records)
    std::string err;
    size_t n() const { return flags.size(); }
};
struct kdf_reader;
struct ParsePipe {
    kdf_reader *r;
    std::vector<RawChunk> raw; std::vector<ParsedChunk> out; std::vector<int> state;   // 0 free, 1 raw, 2 parsed
    size_t head = 0, tail = 0;                    // consumer / chunker sequence numbers
    std::deque<size_t> work;
    bool stop = false, eof = false;
    std::string err;                              // chunker error (reported after the chunks before it)
    std::mutex mu;
    std::condition_variable cv_free, cv_work, cv_done;
    std::thread chunker;
    std::vector<std::thread> workers;
    ParsePipe(kdf_reader *reader, int nthreads);
    ~ParsePipe();
    void chunk_loop();
    void work_loop();
    void parse(const RawChunk &in, ParsedChunk &o) const;
    ParsedChunk *front(std::string &e);           // next parsed chunk in order; nullptr at end of file or on error (e set)
    void pop();
};

struct kdf_reader {
    enum Kind { BAM, FASTA } kind = BAM;
    std::string err;
    // ---- BAM
    FILE *fp = nullptr;
    std::unique_ptr<BgzfPool> pool;    // threaded inflate (threads > 1)
    std::unique_ptr<ParsePipe> pipe;   // threaded record parsing (threads > 1, no alignment details)
    int threads = 1;
    bool started = false;              // first kdf_reader_next seen (the parsing mode is fixed then)
    BgzfBlock blk;                     // synchronous inflate (threads <= 1)
    std::vector<uint8_t> inbuf;        // decompressed, not yet consumed
    size_t inpos = 0;
    bool eof = false;
    uint32_t flag_off = 0;
    bool collapse = false;
    // ---- a reader of ONE RANGE of the file (kdf_bam_open_range): it ends at the first record boundary (QNAME-run
    // boundary when runs are collapsed) at or after the first BGZF block whose file offset is >= range_hi
    uint64_t range_hi = UINT64_MAX;    // none
    uint64_t stop_pos = UINT64_MAX;    // position in the inflated stream (since this reader's start) where that block begins
    uint64_t consumed_base = 0;        // inflated bytes dropped from inbuf by compaction: position of inbuf[0]
    bool range_done = false;
    // collapse state (samtools bam2fq: best record per read part of a QNAME run)
    bool have_run = false;
    std::string run_name;
    Record best[3];
    int score[3] = {-1, -1, -1};
    std::deque<Record> ready;          // records ready to be emitted
    std::vector<Record> spare;         // recycled records (their buffers keep their capacity: no mallocs in steady state)
    // ---- FASTA
    gzFile gz = nullptr;
    int fasta_k = 0;
    std::string fa_name;
    std::vector<uint8_t> fa_codes;     // current record's not-yet-emitted bases
    bool fa_in_record = false;     // a record is open (some of its bases not yet emitted)
    bool fa_loaded_all = false;    // every base of the open record is in fa_codes
    bool fa_eof = false, fa_continued = false;
    bool fa_have_header = false;   // the header line of the NEXT record has been consumed
    std::string fa_pending_header;
    // ---- last batch metadata
    std::vector<uint16_t> m_flags;
    std::vector<int32_t> m_ref, m_pos;
    std::string m_names;
    std::vector<int64_t> m_name_off;
    bool want_aux = false;
    std::vector<uint32_t> m_cigar;
    std::vector<int64_t> m_cigar_off;   // n_reads + 1
    std::string m_sa;                   // NUL-terminated SA strings, back to back
    std::vector<int64_t> m_sa_off;      // n_reads; -1 = no SA tag
    std::vector<uint8_t> m_qual;        // base qualities, all records back to back (stream offsets index it minus separators)
    std::vector<int64_t> m_qual_off;    // n_reads + 1
    std::vector<uint8_t> m_mapq;        // n_reads
    std::vector<uint64_t> m_ordinal;    // n_reads: record number in the file
    uint64_t n_records = 0;             // raw records parsed so far
    std::vector<uint8_t> header_raw;    // magic .. end of the reference list, as in the file
    std::vector<int32_t> ref_lens;
    bool no_compact = false;            // while the header is being read (header_raw is cut from inbuf)
    const uint8_t *last_raw = nullptr;  // the record bam_next_raw just parsed (valid until the next call)
    size_t last_raw_len = 0;
    std::vector<std::string> ref_names;
};

namespace {

int rfail(kdf_reader *r, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (r) r->err = buf; else g_host_err = buf;
    return code;
}

// ---- BGZF ------------------------------------------------------------------

// read + inflate the next BGZF block, append to r->inbuf.  returns 0 ok, 1 eof, <0 error
int bgzf_read_block(kdf_reader *r) {
    // compact the consumed prefix now and then
    if (!r->no_compact && r->inpos > (1u << 20) && r->inpos * 2 > r->inbuf.size()) {
        r->inbuf.erase(r->inbuf.begin(), r->inbuf.begin() + r->inpos);
        r->consumed_base += r->inpos;
        r->inpos = 0;
    }
    std::string err;
    int rc;
    const uint64_t before = r->consumed_base + r->inbuf.size();
    uint64_t foff = 0;
    if (r->pool) {
        rc = r->pool->next(r->inbuf, err, &foff);
    } else {
        rc = bgzf_read_raw(r->fp, r->blk, err);
        foff = r->blk.file_off;
        if (rc == 0) {
            if (!bgzf_inflate(r->blk)) { rc = -1; err = r->blk.err; }
            else r->inbuf.insert(r->inbuf.end(), r->blk.data.begin(), r->blk.data.end());
        }
    }
    if (rc < 0) return rfail(r, -1, "%s", err.c_str());
    if (rc == 0 && r->stop_pos == UINT64_MAX && foff >= r->range_hi) r->stop_pos = before;    // the range ends inside / after this block
    return rc;
}

// make at least n bytes available at inpos; false at clean EOF / error
bool bam_need(kdf_reader *r, size_t n, bool *error) {
    *error = false;
    while (r->inbuf.size() - r->inpos < n) {
        if (r->eof) return false;
        int rc = bgzf_read_block(r);
        if (rc == 1) { r->eof = true; }
        else if (rc < 0) { *error = true; return false; }
    }
    return true;
}

inline int32_t le32(const uint8_t *p) { return (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)); }

int bam_read_header(kdf_reader *r) {
    bool e;
    r->no_compact = true;
    if (!bam_need(r, 12, &e)) { std::string m = e ? r->err : std::string("empty or truncated BAM"); return rfail(r, KDF_ERR_IO, "%s", m.c_str()); }
    const uint8_t *p = r->inbuf.data() + r->inpos;
    if (memcmp(p, "BAM\1", 4) != 0) return rfail(r, KDF_ERR_IO, "not a BAM file (bad magic)");
    const int32_t l_text = le32(p + 4);
    if (!bam_need(r, 12 + (size_t)l_text, &e)) return rfail(r, KDF_ERR_IO, "truncated BAM header");
    const size_t h0 = r->inpos;
    r->inpos += 8 + (size_t)l_text;
    p = r->inbuf.data() + r->inpos;
    const int32_t n_ref = le32(p);
    r->inpos += 4;
    for (int32_t i = 0; i < n_ref; ++i) {
        if (!bam_need(r, 4, &e)) return rfail(r, KDF_ERR_IO, "truncated BAM reference list");
        const int32_t l_name = le32(r->inbuf.data() + r->inpos);
        if (!bam_need(r, 8 + (size_t)l_name, &e)) return rfail(r, KDF_ERR_IO, "truncated BAM reference list");
        r->ref_names.emplace_back((const char *)r->inbuf.data() + r->inpos + 4, l_name > 0 ? (size_t)l_name - 1 : 0);
        r->ref_lens.push_back(le32(r->inbuf.data() + r->inpos + 4 + (size_t)l_name));
        r->inpos += 8 + (size_t)l_name;
    }
    r->header_raw.assign(r->inbuf.begin() + (long)h0, r->inbuf.begin() + (long)r->inpos);
    r->no_compact = false;
    return KDF_OK;
}

// walk the optional fields [a, end) for SA:Z; true + value (not NUL-terminated) when present
bool bam_find_sa(const uint8_t *a, const uint8_t *end, const char **val, size_t *len) {
    while (a + 3 <= end) {
        const char t0 = (char)a[0], t1 = (char)a[1], ty = (char)a[2];
        a += 3;
        size_t n = 0;
        if (ty == 'A' || ty == 'c' || ty == 'C') n = 1;
        else if (ty == 's' || ty == 'S') n = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') n = 4;
        else if (ty == 'Z' || ty == 'H') {
            const uint8_t *z = a;
            while (z < end && *z) ++z;
            if (t0 == 'S' && t1 == 'A' && ty == 'Z') { *val = (const char *)a; *len = (size_t)(z - a); return true; }
            n = (size_t)(z - a) + 1;
        } else if (ty == 'B') {
            if (a + 5 > end) break;
            const char sub = (char)a[0];
            const uint32_t cnt = (uint32_t)le32(a + 1);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            n = 5 + es * (size_t)cnt;
        } else break;                                       // unknown type: stop parsing
        if (a + n > end) break;
        a += n;
    }
    return false;
}

// parse the next raw alignment; returns 1 at EOF, 0 ok, <0 error.
int bam_next_raw(kdf_reader *r, Record &rec, bool &has_qual, uint64_t *start_pos = nullptr) {
    bool e;
    if (!bam_need(r, 4, &e)) return e ? -1 : 1;
    if (start_pos) *start_pos = r->consumed_base + r->inpos;
    const int32_t bs = le32(r->inbuf.data() + r->inpos);
    if (bs < 32) { rfail(r, KDF_ERR_IO, "corrupt BAM record"); return -1; }
    if (!bam_need(r, 4 + (size_t)bs, &e)) { if (!e) rfail(r, KDF_ERR_IO, "truncated BAM record"); return -1; }
    const uint8_t *p = r->inbuf.data() + r->inpos + 4;
    rec.ref_id = le32(p);
    rec.pos = le32(p + 4);
    const unsigned l_rn = p[8];
    const unsigned n_cig = p[12] | (p[13] << 8);
    rec.flag = (uint16_t)(p[14] | (p[15] << 8));
    const int32_t l_seq = le32(p + 16);
    const size_t need = 32 + l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (l_seq < 0 || need > (size_t)bs) { rfail(r, KDF_ERR_IO, "corrupt BAM record (field sizes)"); return -1; }
    rec.name.assign((const char *)p + 32, l_rn ? l_rn - 1 : 0);
    const uint8_t *sq = p + 32 + l_rn + 4 * n_cig;
    rec.l_seq = l_seq;
    rec.seq4.assign(sq, sq + ((size_t)l_seq + 1) / 2);
    has_qual = l_seq > 0 && sq[((size_t)l_seq + 1) / 2] != 0xFF;
    rec.cigar.clear(); rec.sa.clear(); rec.has_sa = false; rec.qual.clear();
    rec.mapq = p[9];
    rec.ordinal = r->n_records++;
    r->last_raw = p; r->last_raw_len = (size_t)bs;
    if (r->want_aux) {
        const uint8_t *ql = sq + ((size_t)l_seq + 1) / 2;
        rec.qual.assign(ql, ql + (size_t)l_seq);
        const uint8_t *cg = p + 32 + l_rn;
        rec.cigar.resize(n_cig);
        for (unsigned i = 0; i < n_cig; ++i) rec.cigar[i] = (uint32_t)le32(cg + 4 * i);
        const char *sa; size_t sa_len;
        if (bam_find_sa(p + need, p + bs, &sa, &sa_len)) { rec.sa.assign(sa, sa_len); rec.has_sa = true; }
    }
    r->inpos += 4 + (size_t)bs;
    return 0;
}

inline Record grab(kdf_reader *r) {
    if (r->spare.empty()) return Record();
    Record x = std::move(r->spare.back());
    r->spare.pop_back();
    return x;
}
inline void recycle(kdf_reader *r, Record &&x) { if (r->spare.size() < 64) r->spare.push_back(std::move(x)); }

void flush_run(kdf_reader *r) {
    for (int i = 0; i < 3; ++i)
        if (r->score[i] >= 0) { r->ready.push_back(std::move(r->best[i])); r->best[i] = grab(r); r->score[i] = -1; }
    r->have_run = false;
}

// fill r->ready with at least one record if any remain.  returns <0 on error.
int bam_pump(kdf_reader *r) {
    while (r->ready.empty()) {
        if (r->range_done) return 0;
        Record rec = grab(r); bool hq = false; uint64_t at = 0;
        const int rc = bam_next_raw(r, rec, hq, &at);
        if (rc < 0) return -1;
        if (rc == 1) { if (r->have_run) flush_run(r); return 0; }
        const bool zone = at >= r->stop_pos;                     // past the range's end: only the run that straddles it is finished
        if (zone && !r->collapse) { r->range_done = true; recycle(r, std::move(rec)); return 0; }
        if (rec.flag & r->flag_off) { recycle(r, std::move(rec)); continue; }
        if (!r->collapse) { r->ready.push_back(std::move(rec)); return 0; }
        if (!r->have_run || rec.name != r->run_name) {
            if (r->have_run) flush_run(r);
            if (zone) { r->range_done = true; recycle(r, std::move(rec)); return 0; }      // the next range's first run
            r->run_name = rec.name; r->have_run = true;
        }
        const bool r1 = rec.flag & 0x40, r2 = rec.flag & 0x80;
        const int part = (r1 && !r2) ? 1 : (r2 && !r1) ? 2 : 0;
        const int sc = hq ? 2 : 1;
        if (sc > r->score[part]) { std::swap(r->best[part], rec); r->score[part] = sc; }
        recycle(r, std::move(rec));
    }
    return 0;
}

// ---- FASTA -----------------------------------------------------------------

// Loads bases of the open FASTA record into fa_codes until it holds >= want
// bases or the record ends (next header / EOF -> fa_loaded_all).
void fasta_fill(kdf_reader *r, size_t want) {
    static thread_local char line[1 << 16];
    while (!r->fa_loaded_all && r->fa_codes.size() < want) {
        if (!gzgets(r->gz, line, sizeof line)) { r->fa_eof = true; r->fa_loaded_all = true; break; }
        size_t len = strlen(line);
        const bool full_line = len && line[len - 1] == '\n';
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) --len;
        if (len && line[0] == '>') {
            r->fa_pending_header.assign(line + 1, len - 1);
            r->fa_have_header = true;
            if (!full_line)   // header longer than the buffer: swallow the rest
                while (gzgets(r->gz, line, sizeof line)) { size_t l2 = strlen(line); if (l2 && line[l2 - 1] == '\n') break; }
            r->fa_loaded_all = true;
            break;
        }
        for (size_t i = 0; i < len; ++i) r->fa_codes.push_back(kCode.t[(uint8_t)line[i]]);
    }
}

}  // namespace


// ---- parallel record parsing: implementation ----------------------------------

ParsePipe::ParsePipe(kdf_reader *reader, int nthreads) : r(reader) {
    const size_t ring = (size_t)nthreads * 2 + 2;
    raw.resize(ring); out.resize(ring); state.assign(ring, 0);
    chunker = std::thread([this] { chunk_loop(); });
    for (int i = 0; i < nthreads; ++i) workers.emplace_back([this] { work_loop(); });
}

ParsePipe::~ParsePipe() {
    { std::lock_guard<std::mutex> g(mu); stop = true; }
    cv_free.notify_all(); cv_work.notify_all(); cv_done.notify_all();
    if (chunker.joinable()) chunker.join();
    for (auto &t : workers) if (t.joinable()) t.join();
}

// Cuts the inflated stream into chunks of whole records.  With QNAME collapsing a chunk
// ends only where the next KEPT record starts a new name run (records dropped by the flag
// filter do not break a run: bam_pump skips them before the run logic).
void ParsePipe::chunk_loop() {
    constexpr size_t TARGET_RECORDS = 2048;
    std::string prev_name; bool have_prev = false;
    for (;;) {
        size_t slot;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_free.wait(lk, [this] { return stop || tail - head < raw.size(); });
            if (stop) return;
            slot = tail % raw.size();
        }
        RawChunk &c = raw[slot];
        c.first_ordinal = r->n_records;
        // the chunk is scanned in place (r->inpos stays at its first byte, so the buffer's
        // compaction never drops unread chunk bytes) and copied out with ONE memcpy
        size_t nrec = 0, rel = 0; bool end = false; std::string e;        // rel: scan offset from r->inpos (which compaction may reset)
        for (;;) {
            bool er;
            if (!bam_need(r, rel + 4, &er)) { end = true; if (er) e = r->err; break; }
            const int32_t bs = le32(r->inbuf.data() + r->inpos + rel);
            if (bs < 32) { end = true; e = "corrupt BAM record"; break; }
            if (!bam_need(r, rel + 4 + (size_t)bs, &er)) { end = true; e = er ? r->err : std::string("truncated BAM record"); break; }
            const uint8_t *p = r->inbuf.data() + r->inpos + rel + 4;      // (bam_need may have moved or compacted the buffer)
            const unsigned l_rn = p[8];
            const uint16_t flag = (uint16_t)(p[14] | (p[15] << 8));
            const bool kept = !(flag & r->flag_off);
            const bool zone = r->consumed_base + r->inpos + rel >= r->stop_pos;   // past the end of this reader's range
            if (zone && !r->collapse) { end = true; break; }
            bool boundary = true;                               // may the chunk end before this record?
            if (r->collapse && kept) {
                const size_t nl = l_rn ? l_rn - 1 : 0;
                if (32 + (size_t)l_rn > (size_t)bs) { end = true; e = "corrupt BAM record (field sizes)"; break; }
                boundary = !have_prev || prev_name.size() != nl || memcmp(prev_name.data(), p + 32, nl) != 0;
            } else if (r->collapse) {
                boundary = false;                               // a dropped record inside a run does not end it
            }
            if (zone && r->collapse && kept && boundary) { end = true; break; }   // the next range's first run
            if (nrec >= TARGET_RECORDS && boundary) break;      // this record opens the next chunk
            if (r->collapse && kept && boundary) { prev_name.assign((const char *)p + 32, l_rn ? l_rn - 1 : 0); have_prev = true; }
            rel += 4 + (size_t)bs;
            ++nrec;
        }
        c.bytes.assign(r->inbuf.begin() + (long)r->inpos, r->inbuf.begin() + (long)(r->inpos + rel));
        r->inpos += rel;
        r->n_records += nrec;
        std::lock_guard<std::mutex> g(mu);
        if (nrec) { state[slot] = 1; work.push_back(tail); ++tail; cv_work.notify_one(); }
        if (end) { eof = true; err = e; cv_done.notify_all(); return; }
    }
}

void ParsePipe::work_loop() {
    for (;;) {
        size_t seq;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_work.wait(lk, [this] { return stop || !work.empty(); });
            if (stop) return;
            seq = work.front(); work.pop_front();
        }
        const size_t slot = seq % raw.size();
        parse(raw[slot], out[slot]);
        std::lock_guard<std::mutex> g(mu);
        state[slot] = 2;
        cv_done.notify_all();
    }
}

// bam_next_raw + bam_pump + the emit part of kdf_reader_next, on one chunk, without copying records
void ParsePipe::parse(const RawChunk &in, ParsedChunk &o) const {
    o.off.clear(); o.flags.clear(); o.ref.clear(); o.pos.clear(); o.names.clear(); o.name_off.clear(); o.ordinal.clear();
    o.cigar.clear(); o.cigar_off.clear(); o.sa.clear(); o.sa_off.clear(); o.qual.clear(); o.qual_off.clear(); o.mapq.clear();
    o.next = 0; o.err.clear();
    const bool aux = r->want_aux;
    // every record carries >= (l_seq + 1) / 2 sequence bytes: positions <= 2 * bytes + records <= 2.1 * bytes
    const size_t max_pos = in.bytes.size() * 2 + in.bytes.size() / 16 + 64;
    o.packed.assign(max_pos / 32 + 4, 0); o.invalid.assign(max_pos / 64 + 4, 0);
    StreamWriter w(o.packed.data(), o.invalid.data());
    struct Ref { const uint8_t *p; int32_t bs; uint64_t ordinal; };
    auto emit = [&](const Ref &x) {
        const uint8_t *p = x.p;
        const unsigned l_rn = p[8], n_cig = p[12] | (p[13] << 8);
        const int32_t l_seq = le32(p + 16);
        o.off.push_back((int64_t)w.n);
        w.put_seq4(p + 32 + l_rn + 4 * n_cig, l_seq);
        w.put_sep();
        o.flags.push_back((uint16_t)(p[14] | (p[15] << 8))); o.ref.push_back(le32(p)); o.pos.push_back(le32(p + 4));
        o.name_off.push_back((int64_t)o.names.size());
        o.names.append((const char *)p + 32, l_rn ? l_rn - 1 : 0); o.names.push_back('\0');
        o.ordinal.push_back(x.ordinal);
        if (aux) {
            const uint8_t *cg = p + 32 + l_rn, *sq = cg + 4 * n_cig, *ql = sq + ((size_t)l_seq + 1) / 2;
            o.cigar_off.push_back((int64_t)o.cigar.size());
            for (unsigned i = 0; i < n_cig; ++i) o.cigar.push_back((uint32_t)le32(cg + 4 * i));
            o.qual_off.push_back((int64_t)o.qual.size());
            o.qual.insert(o.qual.end(), ql, ql + (size_t)l_seq);
            o.mapq.push_back(p[9]);
            const char *sa; size_t sa_len;
            if (bam_find_sa(ql + (size_t)l_seq, p + x.bs, &sa, &sa_len)) {
                o.sa_off.push_back((int64_t)o.sa.size()); o.sa.append(sa, sa_len); o.sa.push_back('\0');
            } else o.sa_off.push_back(-1);
        }
    };
    Ref best[3]; int score[3] = {-1, -1, -1};
    const uint8_t *run_name = nullptr; size_t run_len = 0; bool have_run = false;
    auto flush = [&]() {
        for (int i = 0; i < 3; ++i) if (score[i] >= 0) { emit(best[i]); score[i] = -1; }
        have_run = false;
    };
    const uint8_t *q = in.bytes.data(), *end = q + in.bytes.size();
    uint64_t ordinal = in.first_ordinal;
    while (q < end) {
        const int32_t bs = le32(q);
        const uint8_t *p = q + 4;
        q += 4 + (size_t)bs;
        const uint64_t my = ordinal++;
        const unsigned l_rn = p[8], n_cig = p[12] | (p[13] << 8);
        const uint16_t flag = (uint16_t)(p[14] | (p[15] << 8));
        const int32_t l_seq = le32(p + 16);
        const size_t need = 32 + l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
        if (l_seq < 0 || need > (size_t)bs) { o.err = "corrupt BAM record (field sizes)"; return; }
        if (flag & r->flag_off) continue;
        const Ref me{p, bs, my};
        if (!r->collapse) { emit(me); continue; }
        const size_t nl = l_rn ? l_rn - 1 : 0;
        if (!have_run || run_len != nl || memcmp(run_name, p + 32, nl) != 0) {
            if (have_run) flush();
            run_name = p + 32; run_len = nl; have_run = true;
        }
        const bool r1 = flag & 0x40, r2 = flag & 0x80;
        const int part = (r1 && !r2) ? 1 : (r2 && !r1) ? 2 : 0;
        const uint8_t *sq = p + 32 + l_rn + 4 * n_cig;
        const int sc = (l_seq > 0 && sq[((size_t)l_seq + 1) / 2] != 0xFF) ? 2 : 1;
        if (sc > score[part]) { best[part] = me; score[part] = sc; }
    }
    if (have_run) flush();
    o.off.push_back((int64_t)w.n);
    if (aux) { o.cigar_off.push_back((int64_t)o.cigar.size()); o.qual_off.push_back((int64_t)o.qual.size()); }
}

ParsedChunk *ParsePipe::front(std::string &e) {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [this] { return stop || (head < tail && state[head % raw.size()] == 2) || (eof && head == tail); });
    if (stop) { e = "reader closed"; return nullptr; }
    if (head == tail) { e = err; return nullptr; }           // end of file (err empty) or the chunker's error
    ParsedChunk &c = out[head % raw.size()];
    if (!c.err.empty()) { e = c.err; return nullptr; }
    return &c;
}

void ParsePipe::pop() {
    std::lock_guard<std::mutex> g(mu);
    state[head % raw.size()] = 0;
    ++head;
    cv_free.notify_one();
}

extern "C" {

void kdf_stream_words(uint64_t n_bases, uint64_t *packed_words, uint64_t *mask_words) {
    const uint64_t tiles = (n_bases + 63) / 64;        // a kernel tile = 64 window starts
    if (packed_words) *packed_words = tiles * 2 + 4;   // a tile reads packed words [2t, 2t+3]
    if (mask_words) *mask_words = tiles + 2;           // and mask words [t, t+1]
}

int kdf_canonical(const char *kmer, int k, uint64_t *lo, uint64_t *hi) {
    if (!kmer || k < 1 || k > 64) return KDF_ERR_INVALID;
    unsigned __int128 fwd = 0, rc = 0;
    for (int i = 0; i < k; ++i) {
        const uint8_t c = kCode.t[(uint8_t)kmer[i]];
        if (c > 3) return KDF_ERR_INVALID;
        fwd = (fwd << 2) | c;
        rc = (rc >> 2) | ((unsigned __int128)(3 - c) << (2 * (k - 1)));
    }
    const unsigned __int128 c = fwd < rc ? fwd : rc;
    if (lo) *lo = (uint64_t)c;
    if (hi) *hi = (uint64_t)(c >> 64);
    return KDF_OK;
}

int kdf_pack_reads(const char *ascii, const int64_t *offsets, int64_t n_reads, uint64_t *packed_out,
                   uint64_t *invalid_out, int64_t *stream_offsets_out, uint64_t *n_bases_out) {
    if (n_reads < 0 || (n_reads && (!ascii || !offsets)) || !packed_out || !invalid_out) return KDF_ERR_INVALID;
    StreamWriter w(packed_out, invalid_out);
    w.begin(n_reads ? (uint64_t)(offsets[n_reads] - offsets[0]) + (uint64_t)n_reads : 0);
    for (int64_t r = 0; r < n_reads; ++r) {
        if (stream_offsets_out) stream_offsets_out[r] = (int64_t)w.n;
        w.put_ascii(ascii + offsets[r], offsets[r + 1] - offsets[r]);
        w.put_sep();   // separator: no window spans two records
    }
    if (stream_offsets_out) stream_offsets_out[n_reads] = (int64_t)w.n;
    w.finish();
    if (n_bases_out) *n_bases_out = w.n;
    return KDF_OK;
}

const char *kdf_reader_error(const kdf_reader *r) { return r ? r->err.c_str() : g_host_err.c_str(); }

int kdf_bam_open(const char *path, uint32_t flag_off, int collapse, int threads, kdf_reader **out) {
    if (!path || !out) return rfail(nullptr, KDF_ERR_INVALID, "kdf_bam_open: NULL argument");
    *out = nullptr;
    kdf_reader *r = new kdf_reader();
    r->kind = kdf_reader::BAM;
    r->flag_off = flag_off; r->collapse = collapse != 0;
    r->fp = fopen(path, "rb");
    if (!r->fp) { rfail(nullptr, KDF_ERR_IO, "cannot open %s", path); delete r; return KDF_ERR_IO; }
    {   // CRAM by its magic bytes, whatever the file is called: it needs htslib + the reference, which this
        // library does not link (the reference passes --reference to samtools for it, jellyfish_wrappers.py:159-165)
        char magic[4] = {0, 0, 0, 0};
        const size_t got = fread(magic, 1, 4, r->fp);
        if (got == 4 && memcmp(magic, "CRAM", 4) == 0) {
            rfail(nullptr, KDF_ERR_IO, "%s is a CRAM file: CRAM input needs htslib, which the MI355X engine does not link; "
                                       "convert to BAM (samtools view -b) first", path);
            fclose(r->fp); r->fp = nullptr; delete r; return KDF_ERR_IO;
        }
        rewind(r->fp);
    }
    static const size_t kBuf = 1 << 20;
    setvbuf(r->fp, nullptr, _IOFBF, kBuf);
    r->threads = std::min(threads, 64);
    if (threads > 1) r->pool.reset(new BgzfPool(r->fp, r->threads));
    int rc = bam_read_header(r);
    if (rc) { g_host_err = std::string(path) + ": " + r->err; kdf_reader_close(r); return rc; }
    *out = r;
    return KDF_OK;
}

// ---- a reader of one BGZF RANGE of a BAM ------------------------------------------------------------------------
// The file's record blocks [data_off, file_size) are cut at parts - 1 byte offsets F_1 < F_2 < ...; the record stream is
// cut accordingly at S(F) = the first record that STARTS in a BGZF block at file offset >= F -- or, when QNAME runs are
// collapsed, the first KEPT record from there on whose name differs from the kept record before it, so that no run is
// split.  Part p reads [S(F_p), S(F_p+1)): the union of the parts is the whole file, record for record, whatever `parts`
// is.  The reader that ENDS at S(F) finds it by walking its records (kdf_reader::range_hi / stop_pos); the reader that
// STARTS there has no record boundary to walk from: it inflates a little before F, finds a record start by the shape
// of three consecutive records (what Hadoop-BAM's split guesser does), walks to S(F) and hands the real reader the
// BGZF block and the offset inside it.
namespace {

struct MiniBgzf {                     // sequential inflate from a file offset, remembering where every block's bytes begin
    FILE *fp; std::vector<uint8_t> buf; std::vector<std::pair<size_t, uint64_t>> blocks;   // (position in buf, file offset)
    BgzfBlock blk; bool eof = false;
    explicit MiniBgzf(FILE *f) : fp(f) {}
    bool more(std::string &err) {         // append one block; false at end of file or on error (err set)
        if (eof) return false;
        const int rc = bgzf_read_raw(fp, blk, err);
        if (rc != 0) { eof = true; return false; }
        if (!bgzf_inflate(blk)) { err = blk.err; eof = true; return false; }
        blocks.emplace_back(buf.size(), blk.file_off);
        buf.insert(buf.end(), blk.data.begin(), blk.data.end());
        return true;
    }
};

// first BGZF block at file offset >= from: gzip magic with the BC subfield, confirmed by the block that follows it
bool find_bgzf_block(FILE *fp, uint64_t from, uint64_t file_size, uint64_t &at) {
    std::vector<uint8_t> w(1 << 17);
    for (uint64_t base = from; base + 18 <= file_size;) {
        const size_t n = (size_t)std::min<uint64_t>(w.size(), file_size - base);
        if (fseeko(fp, (off_t)base, SEEK_SET) != 0 || fread(w.data(), 1, n, fp) != n) return false;
        for (size_t i = 0; i + 18 <= n; ++i) {
            if (w[i] != 0x1f || w[i + 1] != 0x8b || w[i + 2] != 8 || w[i + 3] != 4) continue;
            if ((w[i + 10] | (w[i + 11] << 8)) != 6 || w[i + 12] != 'B' || w[i + 13] != 'C' || w[i + 14] != 2 || w[i + 15] != 0) continue;
            const uint64_t bsize = (uint64_t)(w[i + 16] | (w[i + 17] << 8)) + 1, next = base + i + bsize;
            if (next > file_size) continue;
            if (next < file_size) {                             // the next block must look like one too
                uint8_t h[4];
                if (fseeko(fp, (off_t)next, SEEK_SET) != 0 || fread(h, 1, 4, fp) != 4) continue;
                if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) continue;
            }
            at = base + i; return true;
        }
        if (n < 18 + 1) break;
        base += n - 17;
    }
    return false;
}

// does a BAM alignment record plausibly start at q?  returns its total length (4 + block_size), 0 if not, and
// (size_t)-1 when the bytes at hand do not reach far enough to tell
size_t plausible_record(const uint8_t *q, size_t avail, int32_t n_ref) {
    if (avail < 36) return (size_t)-1;
    const int32_t bs = le32(q);
    if (bs < 32 || bs > (1 << 28)) return 0;
    const uint8_t *p = q + 4;
    const int32_t ref = le32(p), pos = le32(p + 4), l_seq = le32(p + 16), nref = le32(p + 20), npos = le32(p + 24);
    const unsigned l_rn = p[8], n_cig = p[12] | (p[13] << 8);
    if (ref < -1 || ref >= n_ref || nref < -1 || nref >= n_ref || pos < -1 || npos < -1 || l_seq < 0 || l_rn < 2) return 0;
    const size_t need = 32 + (size_t)l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (need > (size_t)bs) return 0;
    if (avail < 36 + (size_t)l_rn) return (size_t)-1;
    if (p[32 + l_rn - 1] != 0) return 0;                        // the name ends with its NUL ...
    for (unsigned i = 0; i + 1 < l_rn; ++i) if (p[32 + i] < 0x21 || p[32 + i] > 0x7e) return 0;   // ... and is printable before it
    return 4 + (size_t)bs;
}

// locate S(F) (see above).  found = false: the range starts past the last record.  data_block / data_uoff: the
// virtual offset of the file's first record (a true record boundary: no guessing when the look-back reaches it).
int locate_range_start(const char *path, uint64_t F, uint64_t file_size, int32_t n_ref, uint32_t flag_off, bool collapse,
                       uint64_t data_block, size_t data_uoff, uint64_t &blk_off, size_t &uoff, bool &found, std::string &err) {
    found = false;
    FILE *fp = fopen(path, "rb");
    if (!fp) { err = "cannot open the file again"; return KDF_ERR_IO; }
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{fp};
    uint64_t blockF = 0;
    if (!find_bgzf_block(fp, F, file_size, blockF)) return KDF_OK;       // nothing starts at or after F
    for (uint64_t k = 2;; k *= 4) {
        const uint64_t back = k * 65536;
        uint64_t lb = F > back ? F - back : 0;
        bool from_data_start = lb <= data_block;
        if (!from_data_start) { uint64_t b = 0; if (!find_bgzf_block(fp, lb, file_size, b) || b >= blockF) from_data_start = true; else lb = b; }
        if (from_data_start) lb = data_block;
        if (fseeko(fp, (off_t)lb, SEEK_SET) != 0) { err = "seek failed"; return KDF_ERR_IO; }
        MiniBgzf mz(fp);
        // inflate up to blockF and a little beyond
        size_t posF = (size_t)-1;
        std::string e2;
        while (posF == (size_t)-1 || mz.buf.size() < posF + (1u << 17)) {
            if (!mz.more(e2)) break;
            if (posF == (size_t)-1 && mz.blocks.back().second >= blockF) posF = mz.blocks.back().first;
        }
        if (!e2.empty()) { err = e2; return KDF_ERR_IO; }
        if (posF == (size_t)-1) return KDF_OK;                            // (blockF vanished: treat as past the end)
        // a record start at or before posF
        size_t start = (size_t)-1;
        if (from_data_start) start = data_uoff;
        else {
            // three records in a row that look like records (fewer only where the inflated bytes end)
            auto chain_ok = [&](size_t o) {
                size_t q = o;
                for (int i = 0; i < 3; ++i) {
                    const size_t len = plausible_record(mz.buf.data() + q, mz.buf.size() - q, n_ref);
                    if (len == 0) return false;
                    if (len == (size_t)-1) return i > 0;
                    q += len;
                    if (q >= mz.buf.size()) return true;
                }
                return true;
            };
            for (size_t o = 0; o < posF; ++o) if (chain_ok(o)) { start = o; break; }
            if (start == (size_t)-1) continue;                            // (inside one huge record?) look further back
        }
        // walk the records to S(F)
        size_t q = start; bool have_kept = false; std::string last_kept;
        for (;;) {
            while (mz.buf.size() - q < 4 || mz.buf.size() - q < 4 + (size_t)std::max(le32(mz.buf.data() + q), 0)) {
                std::string e3;
                if (!mz.more(e3)) { if (!e3.empty()) { err = e3; return KDF_ERR_IO; } return KDF_OK; }   // ran off the file: nothing starts in this range
            }
            const int32_t bs = le32(mz.buf.data() + q);
            if (bs < 32) { err = "corrupt BAM record while locating a range start"; return KDF_ERR_IO; }
            const uint8_t *p = mz.buf.data() + q + 4;
            const unsigned l_rn = p[8];
            const uint16_t flag = (uint16_t)(p[14] | (p[15] << 8));
            const bool kept = !(flag & flag_off);
            const size_t nl = l_rn ? l_rn - 1 : 0;
            if (q >= posF) {
                bool here = !collapse;
                if (collapse && kept) {
                    if (!have_kept && !from_data_start) break;           // no kept record seen before the cut: look further back
                    here = !have_kept || last_kept.size() != nl || memcmp(last_kept.data(), p + 32, nl) != 0;
                }
                if (here) {
                    // the block that holds byte q
                    size_t bi = mz.blocks.size() - 1;
                    while (mz.blocks[bi].first > q) --bi;
                    blk_off = mz.blocks[bi].second; uoff = q - mz.blocks[bi].first; found = true;
                    return KDF_OK;
                }
            }
            if (kept) { last_kept.assign((const char *)p + 32, nl); have_kept = true; }
            q += 4 + (size_t)bs;
        }
        // (fell out of the walk: retry with a longer look-back)
    }
}

}  // namespace

int kdf_bam_open_range(const char *path, uint32_t flag_off, int collapse, int threads, int part, int parts, kdf_reader **out) {
    if (!path || !out) return rfail(nullptr, KDF_ERR_INVALID, "kdf_bam_open_range: NULL argument");
    if (parts < 1 || part < 0 || part >= parts) return rfail(nullptr, KDF_ERR_INVALID, "kdf_bam_open_range: part %d of %d", part, parts);
    *out = nullptr;
    // the header (reference names, where the records begin) through a plain sequential reader
    kdf_reader *hr = nullptr;
    int rc = kdf_bam_open(path, flag_off, collapse, 1, &hr);
    if (rc) return rc;
    if (parts == 1) { kdf_reader_close(hr); return kdf_bam_open(path, flag_off, collapse, threads, out); }
    // the first record's virtual offset: the header ended at inbuf[inpos]; the sequential reader appended whole blocks
    uint64_t data_block = 0; size_t data_uoff = 0;
    {
        FILE *fp = fopen(path, "rb");
        if (!fp) { kdf_reader_close(hr); return rfail(nullptr, KDF_ERR_IO, "cannot open %s", path); }
        MiniBgzf mz(fp); std::string e;
        const size_t hdr_end = hr->inpos;
        while (mz.buf.size() <= hdr_end) if (!mz.more(e)) break;
        fclose(fp);
        if (!e.empty()) { kdf_reader_close(hr); return rfail(nullptr, KDF_ERR_IO, "%s: %s", path, e.c_str()); }
        if (mz.buf.size() <= hdr_end) {                          // a BAM without records: part 0 reads it (nothing), the others are empty
            kdf_reader_close(hr);
            rc = kdf_bam_open(path, flag_off, collapse, 1, out);
            if (rc == KDF_OK && part > 0) { (*out)->range_done = true; (*out)->eof = true; (*out)->inbuf.clear(); (*out)->inpos = 0; }
            return rc;
        }
        size_t bi = mz.blocks.size() - 1;
        while (mz.blocks[bi].first > hdr_end) --bi;
        data_block = mz.blocks[bi].second; data_uoff = hdr_end - mz.blocks[bi].first;
    }
    const int32_t n_ref = (int32_t)hr->ref_names.size();
    uint64_t file_size = 0;
    { FILE *fp = fopen(path, "rb"); if (fp) { fseeko(fp, 0, SEEK_END); file_size = (uint64_t)ftello(fp); fclose(fp); } }
    const uint64_t span = file_size > data_block ? file_size - data_block : 0;
    auto cut = [&](int p) { return data_block + (uint64_t)((unsigned __int128)span * (unsigned)p / (unsigned)parts); };
    uint64_t blk_off = data_block; size_t uoff = data_uoff; bool found = true;
    std::string e;
    if (part > 0) {
        rc = locate_range_start(path, cut(part), file_size, n_ref, flag_off, collapse != 0, data_block, data_uoff, blk_off, uoff, found, e);
        if (rc) { kdf_reader_close(hr); return rfail(nullptr, rc, "%s: %s", path, e.c_str()); }
    }
    // the real reader: the header reader's tables, a file positioned on the start block, uoff bytes to skip
    kdf_reader *r = new kdf_reader();
    r->kind = kdf_reader::BAM;
    r->flag_off = flag_off; r->collapse = collapse != 0;
    r->ref_names = hr->ref_names; r->ref_lens = hr->ref_lens; r->header_raw = hr->header_raw;
    kdf_reader_close(hr);
    r->fp = fopen(path, "rb");
    if (!r->fp) { delete r; return rfail(nullptr, KDF_ERR_IO, "cannot open %s", path); }
    setvbuf(r->fp, nullptr, _IOFBF, 1 << 20);
    r->threads = std::min(threads, 64);
    if (part + 1 < parts) r->range_hi = cut(part + 1);
    if (!found) { r->range_done = true; r->eof = true; *out = r; return KDF_OK; }      // an empty range
    if (fseeko(r->fp, (off_t)blk_off, SEEK_SET) != 0) { kdf_reader_close(r); return rfail(nullptr, KDF_ERR_IO, "%s: seek failed", path); }
    if (r->threads > 1) r->pool.reset(new BgzfPool(r->fp, r->threads));
    bool er = false;
    if (!bam_need(r, uoff, &er) && er) { g_host_err = std::string(path) + ": " + r->err; kdf_reader_close(r); return KDF_ERR_IO; }
    r->inpos = std::min(uoff, r->inbuf.size());
    // (a start that already lies past this range's end -- a range smaller than a run -- is an empty range: the walk's
    // first record is in the end zone and opens a run)
    *out = r;
    return KDF_OK;
}

int kdf_fasta_open(const char *path, int k, kdf_reader **out) {
    if (!path || !out || k < 1) return rfail(nullptr, KDF_ERR_INVALID, "kdf_fasta_open: NULL argument");
    *out = nullptr;
    kdf_reader *r = new kdf_reader();
    r->kind = kdf_reader::FASTA;
    r->fasta_k = k;
    r->gz = gzopen(path, "rb");
    if (!r->gz) { rfail(nullptr, KDF_ERR_IO, "cannot open %s", path); delete r; return KDF_ERR_IO; }
    gzbuffer(r->gz, 1 << 20);
    *out = r;
    return KDF_OK;
}

void kdf_reader_close(kdf_reader *r) {
    if (!r) return;
    if (r->pool) r->pool->shutdown();  // unblocks a chunker waiting for inflated bytes
    r->pipe.reset();                   // joins the chunker and parser threads
    r->pool.reset();                   // joins the I/O and inflate threads before the file closes
    if (r->fp) fclose(r->fp);
    if (r->gz) gzclose(r->gz);
    delete r;
}

int kdf_reader_next(kdf_reader *r, uint64_t max_bases, int64_t max_reads, uint64_t *packed_out,
                    uint64_t *invalid_out, int64_t *stream_offsets_out, int64_t *n_reads_out,
                    uint64_t *n_bases_out) {
    if (!r || !packed_out || !invalid_out || !n_reads_out || !n_bases_out || max_reads < 1)
        return rfail(r, KDF_ERR_INVALID, "kdf_reader_next: bad argument");
    r->m_flags.clear(); r->m_ref.clear(); r->m_pos.clear(); r->m_names.clear(); r->m_name_off.clear();
    r->m_cigar.clear(); r->m_cigar_off.clear(); r->m_sa.clear(); r->m_sa_off.clear();
    r->m_qual.clear(); r->m_qual_off.clear(); r->m_mapq.clear(); r->m_ordinal.clear();
    StreamWriter w(packed_out, invalid_out);
    w.begin(max_bases);
    int64_t n = 0;
    if (r->kind == kdf_reader::BAM && !r->started) {
        r->started = true;
        // inflate takes ~60 % of the CPU time of a pass, parsing ~40 %: half as many parser threads
        if (r->pool) r->pipe.reset(new ParsePipe(r, std::max(2, (r->threads + 1) / 2)));
    }
    if (r->kind == kdf_reader::BAM && r->pipe) {
        // parallel parsing: append whole records of the parsed chunks, in file order
        while (n < max_reads) {
            std::string e;
            ParsedChunk *c = r->pipe->front(e);
            if (!c) { if (!e.empty()) return rfail(r, KDF_ERR_IO, "%s", e.c_str()); break; }
            const size_t nr = c->n();
            if (c->next < nr) {
                const size_t i = c->next;
                const uint64_t first_len = (uint64_t)(c->off[i + 1] - c->off[i]);
                if (first_len > max_bases)
                    return rfail(r, KDF_ERR_INVALID, "read %s (%llu bases) exceeds max_bases", c->names.c_str() + c->name_off[i],
                                 (unsigned long long)(first_len - 1));
                const uint64_t room = max_bases - w.n;
                if (first_len > room) break;
                // largest j with off[j] - off[i] <= room and j - i <= max_reads - n
                size_t j = std::min(nr, i + (size_t)(max_reads - n));
                if ((uint64_t)(c->off[j] - c->off[i]) > room)
                    j = (size_t)(std::upper_bound(c->off.begin() + (long)i, c->off.begin() + (long)j + 1, c->off[i] + (int64_t)room) - c->off.begin()) - 1;
                const int64_t base = (int64_t)w.n - c->off[i];
                if (stream_offsets_out) for (size_t t = i; t < j; ++t) stream_offsets_out[n + (int64_t)(t - i)] = c->off[t] + base;
                w.append_stream(c->packed.data(), c->invalid.data(), (uint64_t)c->off[i], (uint64_t)(c->off[j] - c->off[i]));
                r->m_flags.insert(r->m_flags.end(), c->flags.begin() + (long)i, c->flags.begin() + (long)j);
                r->m_ref.insert(r->m_ref.end(), c->ref.begin() + (long)i, c->ref.begin() + (long)j);
                r->m_pos.insert(r->m_pos.end(), c->pos.begin() + (long)i, c->pos.begin() + (long)j);
                r->m_ordinal.insert(r->m_ordinal.end(), c->ordinal.begin() + (long)i, c->ordinal.begin() + (long)j);
                const int64_t nb0 = c->name_off[i], nb1 = j < nr ? c->name_off[j] : (int64_t)c->names.size();
                const int64_t shift = (int64_t)r->m_names.size() - nb0;
                for (size_t t = i; t < j; ++t) r->m_name_off.push_back(c->name_off[t] + shift);
                r->m_names.append(c->names, (size_t)nb0, (size_t)(nb1 - nb0));
                if (r->want_aux) {
                    const int64_t cs = (int64_t)r->m_cigar.size() - c->cigar_off[i], qs = (int64_t)r->m_qual.size() - c->qual_off[i];
                    for (size_t t = i; t < j; ++t) { r->m_cigar_off.push_back(c->cigar_off[t] + cs); r->m_qual_off.push_back(c->qual_off[t] + qs); }
                    r->m_cigar.insert(r->m_cigar.end(), c->cigar.begin() + c->cigar_off[i], c->cigar.begin() + c->cigar_off[j]);
                    r->m_qual.insert(r->m_qual.end(), c->qual.begin() + c->qual_off[i], c->qual.begin() + c->qual_off[j]);
                    r->m_mapq.insert(r->m_mapq.end(), c->mapq.begin() + (long)i, c->mapq.begin() + (long)j);
                    for (size_t t = i; t < j; ++t) {
                        if (c->sa_off[t] < 0) { r->m_sa_off.push_back(-1); continue; }
                        r->m_sa_off.push_back((int64_t)r->m_sa.size());
                        r->m_sa.append(c->sa.c_str() + c->sa_off[t]); r->m_sa.push_back('\0');
                    }
                }
                n += (int64_t)(j - i);
                c->next = j;
            }
            if (c->next == nr) r->pipe->pop(); else break;     // the batch is full
        }
    } else if (r->kind == kdf_reader::BAM) {
        while (n < max_reads) {
            if (bam_pump(r) < 0) return KDF_ERR_IO;
            if (r->ready.empty()) break;
            Record &rec = r->ready.front();
            if ((uint64_t)rec.l_seq + 1 > max_bases)
                return rfail(r, KDF_ERR_INVALID, "read %s (%d bases) exceeds max_bases", rec.name.c_str(), rec.l_seq);
            if (w.n + (uint64_t)rec.l_seq + 1 > max_bases) break;
            if (stream_offsets_out) stream_offsets_out[n] = (int64_t)w.n;
            w.put_seq4(rec.seq4.data(), rec.l_seq);
            w.put_sep();
            r->m_flags.push_back(rec.flag); r->m_ref.push_back(rec.ref_id); r->m_pos.push_back(rec.pos);
            r->m_name_off.push_back((int64_t)r->m_names.size());
            r->m_names.append(rec.name); r->m_names.push_back('\0');
            r->m_ordinal.push_back(rec.ordinal);
            if (r->want_aux) {
                r->m_cigar_off.push_back((int64_t)r->m_cigar.size());
                r->m_cigar.insert(r->m_cigar.end(), rec.cigar.begin(), rec.cigar.end());
                if (rec.has_sa) { r->m_sa_off.push_back((int64_t)r->m_sa.size()); r->m_sa.append(rec.sa); r->m_sa.push_back('\0'); }
                else r->m_sa_off.push_back(-1);
                r->m_qual_off.push_back((int64_t)r->m_qual.size());
                r->m_qual.insert(r->m_qual.end(), rec.qual.begin(), rec.qual.end());
                r->m_mapq.push_back(rec.mapq);
            }
            recycle(r, std::move(rec));
            r->ready.pop_front();
            ++n;
        }
    } else {
        // FASTA: one stream record per sequence; a sequence longer than the
        // batch continues in the next batch, restarting k-1 bases back so that
        // no window is lost or counted twice.
        const size_t ov = r->fasta_k > 1 ? (size_t)r->fasta_k - 1 : 0;
        while (n < max_reads) {
            if (!r->fa_in_record) {
                if (!r->fa_have_header) {
                    if (r->fa_eof) break;
                    // skip anything before the first header
                    r->fa_loaded_all = false; r->fa_codes.clear();
                    while (!r->fa_loaded_all) { fasta_fill(r, 1 << 20); r->fa_codes.clear(); }
                    if (!r->fa_have_header) break;          // no record in the file
                }
                r->fa_name = r->fa_pending_header.substr(0, r->fa_pending_header.find_first_of(" \t"));
                r->fa_have_header = false;
                r->fa_in_record = true; r->fa_continued = false; r->fa_loaded_all = r->fa_eof;
                r->fa_codes.clear();
            }
            const uint64_t room = max_bases > w.n + 1 ? max_bases - w.n - 1 : 0;
            if (room <= ov) {
                if (w.n > 0) break;                          // start this piece in the next batch
                return rfail(r, KDF_ERR_INVALID, "max_bases too small for k");
            }
            fasta_fill(r, (size_t)room + 1);
            const size_t take = std::min<size_t>(r->fa_codes.size(), (size_t)room);
            const bool whole = r->fa_loaded_all && take == r->fa_codes.size();
            if (stream_offsets_out) stream_offsets_out[n] = (int64_t)w.n;
            for (size_t i = 0; i < take; ++i) w.put(r->fa_codes[i]);
            w.put_sep();
            r->m_flags.push_back(r->fa_continued ? 1 : 0); r->m_ref.push_back(-1); r->m_pos.push_back(-1);
            r->m_name_off.push_back((int64_t)r->m_names.size());
            r->m_names.append(r->fa_name); r->m_names.push_back('\0');
            ++n;
            if (whole) { r->fa_in_record = false; r->fa_codes.clear(); continue; }
            // keep the overlap for the continuation piece; the batch is full
            r->fa_codes.erase(r->fa_codes.begin(), r->fa_codes.begin() + (take - ov));
            r->fa_continued = true;
            break;
        }
    }
    if (stream_offsets_out) stream_offsets_out[n] = (int64_t)w.n;
    if (r->want_aux) { r->m_cigar_off.push_back((int64_t)r->m_cigar.size()); r->m_qual_off.push_back((int64_t)r->m_qual.size()); }
    w.finish();
    *n_reads_out = n;
    *n_bases_out = w.n;
    return KDF_OK;
}

int kdf_reader_last_meta(kdf_reader *r, const uint16_t **flags, const int32_t **ref_ids, const int32_t **positions,
                         const char **name_buf, const int64_t **name_offsets) {
    if (!r) return KDF_ERR_INVALID;
    if (flags) *flags = r->m_flags.data();
    if (ref_ids) *ref_ids = r->m_ref.data();
    if (positions) *positions = r->m_pos.data();
    if (name_buf) *name_buf = r->m_names.data();
    if (name_offsets) *name_offsets = r->m_name_off.data();
    return KDF_OK;
}

int kdf_reader_want_aux(kdf_reader *r, int enable) {
    if (!r || r->kind != kdf_reader::BAM) return KDF_ERR_INVALID;
    if (r->started) return rfail(r, KDF_ERR_STATE, "kdf_reader_want_aux: call it before the first kdf_reader_next");
    r->want_aux = enable != 0;
    return KDF_OK;
}

int kdf_reader_last_aux(kdf_reader *r, const uint32_t **cigar, const int64_t **cigar_offsets,
                        const char **sa_buf, const int64_t **sa_offsets) {
    if (!r || !r->want_aux) return KDF_ERR_INVALID;
    if (cigar) *cigar = r->m_cigar.data();
    if (cigar_offsets) *cigar_offsets = r->m_cigar_off.data();
    if (sa_buf) *sa_buf = r->m_sa.data();
    if (sa_offsets) *sa_offsets = r->m_sa_off.data();
    return KDF_OK;
}

int kdf_reader_last_quals(kdf_reader *r, const uint8_t **qual, const int64_t **qual_offsets, const uint8_t **mapq) {
    if (!r || !r->want_aux) return KDF_ERR_INVALID;
    if (qual) *qual = r->m_qual.data();
    if (qual_offsets) *qual_offsets = r->m_qual_off.data();
    if (mapq) *mapq = r->m_mapq.data();
    return KDF_OK;
}

int kdf_reader_last_ordinals(kdf_reader *r, const uint64_t **ordinals) {
    if (!r || r->kind != kdf_reader::BAM || !ordinals) return KDF_ERR_INVALID;
    *ordinals = r->m_ordinal.data();
    return KDF_OK;
}

int kdf_reader_ref_count(kdf_reader *r) { return r ? (int)r->ref_names.size() : -1; }

const char *kdf_reader_ref_name(kdf_reader *r, int i) {
    if (!r || i < 0 || i >= (int)r->ref_names.size()) return nullptr;
    return r->ref_names[(size_t)i].c_str();
}

}  // extern "C"

// ---- N4: subset BAM writer (BGZF + coordinate sort + BAI) ----------------------

namespace {

struct BgzfWriter {
    FILE *fp = nullptr;
    std::vector<uint8_t> buf;            // pending uncompressed bytes (< BLOCK)
    std::vector<uint8_t> comp;
    uint64_t coff = 0;                   // compressed offset of the block `buf` will become
    bool failed = false;
    static constexpr size_t BLOCK = 0xff00;
    uint64_t tell() const { return (coff << 16) | (uint64_t)buf.size(); }
    void flush_block() {
        comp.resize(BLOCK + 1024);
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { failed = true; return; }
        zs.next_in = buf.data(); zs.avail_in = (uInt)buf.size();
        zs.next_out = comp.data() + 18; zs.avail_out = (uInt)(comp.size() - 18 - 8);
        const int zr = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (zr != Z_STREAM_END) { failed = true; return; }
        const size_t bsize = 18 + clen + 8;                       // always < 65536 for <= 0xff00 input bytes
        static const uint8_t h[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
        memcpy(comp.data(), h, 12);
        comp[12] = 'B'; comp[13] = 'C'; comp[14] = 2; comp[15] = 0;
        comp[16] = (uint8_t)((bsize - 1) & 0xff); comp[17] = (uint8_t)((bsize - 1) >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf.data(), (uInt)buf.size());
        const uint32_t isz = (uint32_t)buf.size();
        uint8_t *t = comp.data() + 18 + clen;
        for (int i = 0; i < 4; ++i) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)(isz >> (8 * i)); }
        if (fwrite(comp.data(), 1, bsize, fp) != bsize) failed = true;
        coff += bsize;
        buf.clear();
    }
    void write(const uint8_t *p, size_t n) {
        while (n) {
            const size_t take = std::min(n, BLOCK - buf.size());
            buf.insert(buf.end(), p, p + take);
            p += take; n -= take;
            if (buf.size() == BLOCK) flush_block();
        }
    }
    void finish() {
        if (!buf.empty()) flush_block();
        flush_block();                                            // empty block = the BGZF EOF marker
    }
};

inline int reg2bin(int64_t beg, int64_t end) {                    // SAM spec section 5.3
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

inline void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
inline void put64(std::vector<uint8_t> &v, uint64_t x) { for (int i = 0; i < 8; ++i) v.push_back((uint8_t)(x >> (8 * i))); }

struct RefIndex {
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> linear;
    uint64_t off_beg = 0, off_end = 0, n_mapped = 0, n_unmapped = 0;
    bool any = false;
};

// header text with @HD ... SO:coordinate (what `samtools sort` leaves, reference :2069)
std::string sorted_header_text(const std::string &text) {
    std::string out;
    size_t eol = text.find('\n');
    std::string first = text.substr(0, eol == std::string::npos ? text.size() : eol);
    if (first.compare(0, 3, "@HD") == 0) {
        std::string hd;
        size_t a = 0;
        bool had_so = false;
        while (a <= first.size()) {
            size_t b = first.find('\t', a);
            if (b == std::string::npos) b = first.size();
            std::string f = first.substr(a, b - a);
            if (f.compare(0, 3, "SO:") == 0) { f = "SO:coordinate"; had_so = true; }
            if (f.compare(0, 3, "GO:") != 0) { if (!hd.empty()) hd += '\t'; hd += f; }
            a = b + 1;
        }
        if (!had_so) hd += "\tSO:coordinate";
        out = hd + "\n" + (eol == std::string::npos ? std::string() : text.substr(eol + 1));
    } else {
        out = "@HD\tVN:1.6\tSO:coordinate\n" + text;
    }
    return out;
}

}  // namespace

extern "C" {

int kdf_bam_write_subset(const char *src_bam, const char *dst_bam, const uint64_t *ordinals, uint64_t n,
                         const uint8_t *aux, const uint64_t *aux_offsets, int sort_and_index, int threads,
                         uint64_t *n_written) {
    if (!src_bam || !dst_bam || (n && !ordinals) || (aux && !aux_offsets))
        return rfail(nullptr, KDF_ERR_INVALID, "kdf_bam_write_subset: bad argument");
    for (uint64_t i = 1; i < n; ++i)
        if (ordinals[i] <= ordinals[i - 1]) return rfail(nullptr, KDF_ERR_INVALID, "kdf_bam_write_subset: ordinals must be strictly ascending");
    kdf_reader *r = nullptr;
    int rc = kdf_bam_open(src_bam, 0, 0, threads, &r);
    if (rc) return rc;
    struct Rec { std::vector<uint8_t> raw; uint64_t key_hi, key_lo; };
    std::vector<Rec> recs;
    recs.reserve((size_t)n);
    {
        Record scratch; bool hq;
        uint64_t want = 0;
        while (want < n) {
            const int pr = bam_next_raw(r, scratch, hq);
            if (pr < 0) { g_host_err = r->err; kdf_reader_close(r); return KDF_ERR_IO; }
            if (pr == 1) break;
            if (scratch.ordinal != ordinals[want]) continue;
            Rec x;
            x.raw.assign(r->last_raw, r->last_raw + r->last_raw_len);
            if (aux) x.raw.insert(x.raw.end(), aux + aux_offsets[want], aux + aux_offsets[want + 1]);
            // samtools sort order: tid as unsigned (unplaced last), pos + 1, reverse strand; stable
            x.key_hi = (uint64_t)(uint32_t)scratch.ref_id;
            x.key_lo = ((uint64_t)(uint32_t)(scratch.pos + 1) << 1) | ((scratch.flag & 0x10) ? 1u : 0u);
            recs.push_back(std::move(x));
            ++want;
        }
        if (want < n) { kdf_reader_close(r); return rfail(nullptr, KDF_ERR_INVALID, "kdf_bam_write_subset: record %llu is past the end of %s", (unsigned long long)ordinals[want], src_bam); }
    }
    std::vector<uint8_t> header = r->header_raw;
    const std::vector<int32_t> ref_lens = r->ref_lens;
    kdf_reader_close(r);
    if (sort_and_index) {
        std::stable_sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) {
            return a.key_hi != b.key_hi ? a.key_hi < b.key_hi : a.key_lo < b.key_lo; });
        const int32_t l_text = le32(header.data() + 4);
        const std::string text = sorted_header_text(std::string((const char *)header.data() + 8, (size_t)l_text));
        std::vector<uint8_t> h2(header.begin(), header.begin() + 4);
        put32(h2, (uint32_t)text.size());
        h2.insert(h2.end(), text.begin(), text.end());
        h2.insert(h2.end(), header.begin() + 8 + l_text, header.end());
        header.swap(h2);
    }
    BgzfWriter w;
    w.fp = fopen(dst_bam, "wb");
    if (!w.fp) return rfail(nullptr, KDF_ERR_IO, "cannot create %s", dst_bam);
    w.write(header.data(), header.size());
    w.flush_block();                                             // records start on a block boundary, as htslib writes them
    std::vector<RefIndex> idx(ref_lens.size());
    uint64_t n_no_coor = 0;
    uint64_t prev_end = w.tell();
    for (auto &x : recs) {
        // keep a record inside one block when it fits (htslib does the same).  The index takes the offset
        // where the PREVIOUS record ended as this record's start, as hts_idx_push is fed: a run of one bin
        // then stays one chunk across block boundaries (a reader landing on a block's end moves on to the next).
        if (w.buf.size() + 4 + x.raw.size() > BgzfWriter::BLOCK && !w.buf.empty()) w.flush_block();
        const uint64_t vo0 = prev_end;
        uint8_t bs[4]; const uint32_t l = (uint32_t)x.raw.size();
        for (int i = 0; i < 4; ++i) bs[i] = (uint8_t)(l >> (8 * i));
        const uint8_t *p = x.raw.data();
        const int32_t tid = le32(p), pos = le32(p + 4);
        const unsigned l_rn = p[8], n_cig = p[12] | (p[13] << 8);
        const uint16_t flag = (uint16_t)(p[14] | (p[15] << 8));
        int64_t rlen = 0;
        for (unsigned i = 0; i < n_cig; ++i) {
            const uint32_t c = (uint32_t)le32(p + 32 + l_rn + 4 * i);
            const unsigned op = c & 15;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4;
        }
        const int64_t beg = pos, end = pos + ((flag & 4) || rlen == 0 ? 1 : rlen);
        const int bin = tid >= 0 && pos >= 0 ? reg2bin(beg, end) : 4680;
        x.raw[10] = (uint8_t)(bin & 0xff); x.raw[11] = (uint8_t)(bin >> 8);
        w.write(bs, 4);
        w.write(x.raw.data(), x.raw.size());
        const uint64_t vo1 = prev_end = w.tell();
        if (!sort_and_index) continue;
        if (tid < 0 || (size_t)tid >= idx.size() || pos < 0) { ++n_no_coor; continue; }
        RefIndex &ri = idx[(size_t)tid];
        auto &ch = ri.bins[(uint32_t)bin];
        if (!ch.empty() && ch.back().second == vo0) ch.back().second = vo1; else ch.emplace_back(vo0, vo1);
        const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
        if (ri.linear.size() <= w1) ri.linear.resize(w1 + 1, 0);
        for (size_t k = w0; k <= w1; ++k) if (ri.linear[k] == 0) ri.linear[k] = vo0;
        if (!ri.any) { ri.off_beg = vo0; ri.any = true; }
        ri.off_end = vo1;
        if (flag & 4) ++ri.n_unmapped; else ++ri.n_mapped;
    }
    w.finish();
    const bool wfail = w.failed || fclose(w.fp) != 0;
    if (wfail) return rfail(nullptr, KDF_ERR_IO, "write to %s failed", dst_bam);
    if (sort_and_index) {
        std::vector<uint8_t> b = {'B', 'A', 'I', 1};
        put32(b, (uint32_t)idx.size());
        for (auto &ri : idx) {
            // htslib's compress_binning (hts.c): a bin whose chunks span less than 64 KB of the file is merged
            // into its parent bin when that parent exists, deepest level first; then chunks that touch the same
            // BGZF block are joined.  Queries are unaffected (parents are always searched); samtools' own
            // indexes look like this, and so do ours.
            for (int l = 5; l > 0; --l) {
                const uint32_t first = ((1u << (3 * l)) - 1) / 7;
                for (auto it = ri.bins.begin(); it != ri.bins.end();) {
                    const uint32_t key = it->first;
                    auto &pl = it->second;
                    if (key < first || key >= 37449 || pl.empty()) { ++it; continue; }
                    if (l < 5 && pl.size() > 1) std::sort(pl.begin(), pl.end());
                    if ((pl.back().second >> 16) - (pl.front().first >> 16) < 0x10000) {
                        auto parent = ri.bins.find((key - 1) >> 3);
                        if (parent == ri.bins.end()) { ++it; continue; }
                        parent->second.insert(parent->second.end(), pl.begin(), pl.end());
                        it = ri.bins.erase(it);
                    } else ++it;
                }
            }
            for (auto &kv : ri.bins) {
                auto &pl = kv.second;
                std::sort(pl.begin(), pl.end());
                size_t m = 0;
                for (size_t q = 1; q < pl.size(); ++q) {
                    if ((pl[m].second >> 16) >= (pl[q].first >> 16)) { if (pl[m].second < pl[q].second) pl[m].second = pl[q].second; }
                    else pl[++m] = pl[q];
                }
                if (!pl.empty()) pl.resize(m + 1);
            }
            put32(b, (uint32_t)(ri.bins.size() + (ri.any ? 1 : 0)));
            for (auto &kv : ri.bins) {
                put32(b, kv.first); put32(b, (uint32_t)kv.second.size());
                for (auto &c : kv.second) { put64(b, c.first); put64(b, c.second); }
            }
            if (ri.any) {                                          // samtools' metadata pseudo-bin
                put32(b, 37450); put32(b, 2);
                put64(b, ri.off_beg); put64(b, ri.off_end); put64(b, ri.n_mapped); put64(b, ri.n_unmapped);
            }
            // an empty 16 kb window takes the offset of the next filled one (htslib fills backwards)
            for (size_t k = ri.linear.size(); k-- > 1;) if (ri.linear[k - 1] == 0) ri.linear[k - 1] = ri.linear[k];
            put32(b, (uint32_t)ri.linear.size());
            for (uint64_t v : ri.linear) put64(b, v);
        }
        put64(b, n_no_coor);
        const std::string bai = std::string(dst_bam) + ".bai";
        FILE *fi = fopen(bai.c_str(), "wb");
        if (!fi || fwrite(b.data(), 1, b.size(), fi) != b.size() || fclose(fi) != 0)
            return rfail(nullptr, KDF_ERR_IO, "cannot write %s", bai.c_str());
    }
    if (n_written) *n_written = (uint64_t)recs.size();
    return KDF_OK;
}

}  // extern "C"
