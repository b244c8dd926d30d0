// kdf_host.cpp -- host side of libkdf.so: ASCII -> 2-bit stream packer and the
// streaming read feeders (BGZF/BAM with `samtools fasta -F` semantics, FASTA).
// Replaces the `samtools fasta` half of the reference's pipes
// (core/jellyfish_wrappers.py:159-165, discovery/pipeline.py:106-112,369-375)
// and Jellyfish's FASTA parsing of the reference genome
// (core/jellyfish_wrappers.py:313-321).  zlib only (no htslib in the image);
// CRAM is not supported.
#include <zlib.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <string>
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

// Appends bases to a (packed, invalid) stream under construction.
struct StreamWriter {
    uint64_t *packed;
    uint64_t *invalid;
    uint64_t n = 0;
    StreamWriter(uint64_t *p, uint64_t *m) : packed(p), invalid(m) {}
    inline void put(uint8_t code) {           // code 0..3 valid, >= 4 invalid
        const uint64_t i = n++;
        if ((i & 31) == 0) packed[i >> 5] = 0;
        if ((i & 63) == 0) invalid[i >> 6] = 0;
        if (code < 4) packed[i >> 5] |= (uint64_t)code << ((i & 31) * 2);
        else invalid[i >> 6] |= 1ull << (i & 63);
    }
    // finish: mark the tail of the last mask word invalid
    void finish() {
        if (n & 63) invalid[n >> 6] |= ~0ull << (n & 63);
    }
};

// ----------------------------------------------------------------- records --

struct Record {
    std::vector<uint8_t> codes;   // one 2-bit code (or 4) per base
    std::string name;
    uint16_t flag = 0;
    int32_t ref_id = -1, pos = -1;
};

}  // namespace

struct kdf_reader {
    enum Kind { BAM, FASTA } kind = BAM;
    std::string err;
    // ---- BAM
    FILE *fp = nullptr;
    std::vector<uint8_t> inbuf;        // decompressed, not yet consumed
    size_t inpos = 0;
    bool eof = false;
    uint32_t flag_off = 0;
    bool collapse = false;
    // collapse state (samtools bam2fq: best record per read part of a QNAME run)
    bool have_run = false;
    std::string run_name;
    Record best[3];
    int score[3] = {-1, -1, -1};
    std::deque<Record> ready;          // records ready to be emitted
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
};

namespace {

int rfail(kdf_reader *r, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (r) r->err = buf; else g_host_err = buf;
    return code;
}

// ---- BGZF ------------------------------------------------------------------

// read + inflate one BGZF block, append to r->inbuf.  returns 0 ok, 1 eof, <0 error
int bgzf_read_block(kdf_reader *r) {
    uint8_t hdr[18];
    size_t got = fread(hdr, 1, 18, r->fp);
    if (got == 0) return 1;
    if (got != 18 || hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4))
        return rfail(r, -1, "not a BGZF block (bad gzip header)");
    const unsigned xlen = hdr[10] | (hdr[11] << 8);
    // the BC subfield is normally first (xlen == 6); handle the general case
    std::vector<uint8_t> extra(xlen);
    memcpy(extra.data(), hdr + 12, std::min<unsigned>(6, xlen));
    if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, r->fp) != xlen - 6) return rfail(r, -1, "truncated BGZF extra field");
    int bsize = -1;
    for (unsigned o = 0; o + 4 <= xlen;) {
        const unsigned slen = extra[o + 2] | (extra[o + 3] << 8);
        if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = extra[o + 4] | (extra[o + 5] << 8);
        o += 4 + slen;
    }
    if (bsize < 0) return rfail(r, -1, "BGZF block without BC subfield");
    const int cdata = bsize - (int)xlen - 19;
    if (cdata < 0) return rfail(r, -1, "corrupt BGZF block size");
    std::vector<uint8_t> comp((size_t)cdata + 8);
    if (fread(comp.data(), 1, comp.size(), r->fp) != comp.size()) return rfail(r, -1, "truncated BGZF block");
    const uint32_t isize = comp[cdata + 4] | (comp[cdata + 5] << 8) | (comp[cdata + 6] << 16) | ((uint32_t)comp[cdata + 7] << 24);
    if (isize == 0) return 0;
    // compact the consumed prefix now and then
    if (r->inpos > (1u << 20) && r->inpos * 2 > r->inbuf.size()) {
        r->inbuf.erase(r->inbuf.begin(), r->inbuf.begin() + r->inpos);
        r->inpos = 0;
    }
    const size_t old = r->inbuf.size();
    r->inbuf.resize(old + isize);
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return rfail(r, -1, "inflateInit2 failed");
    zs.next_in = comp.data(); zs.avail_in = (uInt)cdata;
    zs.next_out = r->inbuf.data() + old; zs.avail_out = isize;
    const int zr = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (zr != Z_STREAM_END || zs.avail_out != 0) return rfail(r, -1, "BGZF inflate failed (%d)", zr);
    return 0;
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
    if (!bam_need(r, 12, &e)) { std::string m = e ? r->err : std::string("empty or truncated BAM"); return rfail(r, KDF_ERR_IO, "%s", m.c_str()); }
    const uint8_t *p = r->inbuf.data() + r->inpos;
    if (memcmp(p, "BAM\1", 4) != 0) return rfail(r, KDF_ERR_IO, "not a BAM file (bad magic)");
    const int32_t l_text = le32(p + 4);
    if (!bam_need(r, 12 + (size_t)l_text, &e)) return rfail(r, KDF_ERR_IO, "truncated BAM header");
    r->inpos += 8 + (size_t)l_text;
    p = r->inbuf.data() + r->inpos;
    const int32_t n_ref = le32(p);
    r->inpos += 4;
    for (int32_t i = 0; i < n_ref; ++i) {
        if (!bam_need(r, 4, &e)) return rfail(r, KDF_ERR_IO, "truncated BAM reference list");
        const int32_t l_name = le32(r->inbuf.data() + r->inpos);
        if (!bam_need(r, 8 + (size_t)l_name, &e)) return rfail(r, KDF_ERR_IO, "truncated BAM reference list");
        r->inpos += 8 + (size_t)l_name;
    }
    return KDF_OK;
}

// parse the next raw alignment; returns 1 at EOF, 0 ok, <0 error.
int bam_next_raw(kdf_reader *r, Record &rec, bool &has_qual) {
    bool e;
    if (!bam_need(r, 4, &e)) return e ? -1 : 1;
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
    rec.codes.resize((size_t)l_seq);
    for (int32_t i = 0; i < l_seq; ++i) {
        const uint8_t b = sq[i >> 1];
        rec.codes[i] = kNt16[(i & 1) ? (b & 0xF) : (b >> 4)];
    }
    has_qual = l_seq > 0 && sq[((size_t)l_seq + 1) / 2] != 0xFF;
    r->inpos += 4 + (size_t)bs;
    return 0;
}

void flush_run(kdf_reader *r) {
    for (int i = 0; i < 3; ++i)
        if (r->score[i] >= 0) { r->ready.push_back(std::move(r->best[i])); r->best[i] = Record(); r->score[i] = -1; }
    r->have_run = false;
}

// fill r->ready with at least one record if any remain.  returns <0 on error.
int bam_pump(kdf_reader *r) {
    while (r->ready.empty()) {
        Record rec; bool hq = false;
        const int rc = bam_next_raw(r, rec, hq);
        if (rc < 0) return -1;
        if (rc == 1) { if (r->have_run) flush_run(r); return 0; }
        if (rec.flag & r->flag_off) continue;
        if (!r->collapse) { r->ready.push_back(std::move(rec)); return 0; }
        if (!r->have_run || rec.name != r->run_name) {
            if (r->have_run) flush_run(r);
            r->run_name = rec.name; r->have_run = true;
        }
        const bool r1 = rec.flag & 0x40, r2 = rec.flag & 0x80;
        const int part = (r1 && !r2) ? 1 : (r2 && !r1) ? 2 : 0;
        const int sc = hq ? 2 : 1;
        if (sc > r->score[part]) { r->best[part] = std::move(rec); r->score[part] = sc; }
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

extern "C" {

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
    for (int64_t r = 0; r < n_reads; ++r) {
        if (stream_offsets_out) stream_offsets_out[r] = (int64_t)w.n;
        for (int64_t i = offsets[r]; i < offsets[r + 1]; ++i) w.put(kCode.t[(uint8_t)ascii[i]]);
        w.put(4);   // separator: no window spans two records
    }
    if (stream_offsets_out) stream_offsets_out[n_reads] = (int64_t)w.n;
    w.finish();
    if (n_bases_out) *n_bases_out = w.n;
    return KDF_OK;
}

const char *kdf_reader_error(const kdf_reader *r) { return r ? r->err.c_str() : g_host_err.c_str(); }

int kdf_bam_open(const char *path, uint32_t flag_off, int collapse, int threads, kdf_reader **out) {
    (void)threads;
    if (!path || !out) return rfail(nullptr, KDF_ERR_INVALID, "kdf_bam_open: NULL argument");
    *out = nullptr;
    kdf_reader *r = new kdf_reader();
    r->kind = kdf_reader::BAM;
    r->flag_off = flag_off; r->collapse = collapse != 0;
    r->fp = fopen(path, "rb");
    if (!r->fp) { rfail(nullptr, KDF_ERR_IO, "cannot open %s", path); delete r; return KDF_ERR_IO; }
    static const size_t kBuf = 1 << 20;
    setvbuf(r->fp, nullptr, _IOFBF, kBuf);
    int rc = bam_read_header(r);
    if (rc) { g_host_err = std::string(path) + ": " + r->err; kdf_reader_close(r); return rc; }
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
    StreamWriter w(packed_out, invalid_out);
    int64_t n = 0;
    if (r->kind == kdf_reader::BAM) {
        while (n < max_reads) {
            if (bam_pump(r) < 0) return KDF_ERR_IO;
            if (r->ready.empty()) break;
            Record &rec = r->ready.front();
            if (rec.codes.size() + 1 > max_bases)
                return rfail(r, KDF_ERR_INVALID, "read %s (%zu bases) exceeds max_bases", rec.name.c_str(), rec.codes.size());
            if (w.n + rec.codes.size() + 1 > max_bases) break;
            if (stream_offsets_out) stream_offsets_out[n] = (int64_t)w.n;
            for (uint8_t c : rec.codes) w.put(c);
            w.put(4);
            r->m_flags.push_back(rec.flag); r->m_ref.push_back(rec.ref_id); r->m_pos.push_back(rec.pos);
            r->m_name_off.push_back((int64_t)r->m_names.size());
            r->m_names.append(rec.name); r->m_names.push_back('\0');
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
            w.put(4);
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

}  // extern "C"
