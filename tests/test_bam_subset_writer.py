"""N4 host side (no GPU): kdf_bam_write_subset against an independent reading of
its output -- Python's gzip module for the BGZF container, struct for the BAM
records and a BAI parser written from the SAM specification (sections 4.2, 5.2).
No samtools / pysam in this image: the byte layout of the index is "parity
unpinned" against htslib, the test pins that the index is *correct* (every
region query through it returns exactly the overlapping records)."""
import gzip
import os
import struct

import numpy as np

GIAB = os.path.join(os.path.dirname(__file__), "golden", "giab")
SRC = os.path.join(GIAB, "HG002_child.bam")


def _parse_bam(raw):
    assert raw[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].decode()
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]; o += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, o)[0]
        refs.append((raw[o + 4:o + 4 + ln - 1].decode(), struct.unpack_from("<i", raw, o + 4 + ln)[0]))
        o += 8 + ln
    recs = []
    while o < len(raw):
        bs = struct.unpack_from("<i", raw, o)[0]
        recs.append(raw[o + 4:o + 4 + bs]); o += 4 + bs
    return text, refs, recs


def _fields(rec):
    tid, pos, l_rn, mapq, bin_, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", rec, 0)
    name = rec[32:32 + l_rn - 1].decode()
    cig = struct.unpack_from(f"<{n_cig}I", rec, 32 + l_rn)
    rlen = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
    aux_at = 32 + l_rn + 4 * n_cig + (l_seq + 1) // 2 + l_seq
    return dict(tid=tid, pos=pos, bin=bin_, flag=flag, name=name, end=pos + (rlen if rlen and not flag & 4 else 1),
                aux=rec[aux_at:])


def _all_records(path):
    with gzip.open(path) as fh:
        return _parse_bam(fh.read())


def _pick(n_step=5):
    from kmer_denovo_filter_amd.reads import bam_reader
    ords = []
    with bam_reader(SRC, flag_off=0, collapse=False, want_meta=True) as rd:
        for b in rd:
            assert b.ordinals is not None
            ords.extend(int(x) for x in b.ordinals[::n_step])
    return ords


def test_subset_copy_is_byte_exact_and_sorted(tmp_path):
    from kmer_denovo_filter_amd.reads import write_bam_subset
    text0, refs0, recs0 = _all_records(SRC)
    ords = _pick()
    assert ords[:3] == [0, 5, 10] and max(ords) < len(recs0)
    out = str(tmp_path / "sub.bam")
    tags = [b"DVZchr1:%d:A:T\0" % o for o in ords]
    assert write_bam_subset(SRC, out, ords[::-1], tags[::-1]) == len(ords)          # any input order
    text, refs, recs = _all_records(out)
    assert refs == refs0 and text.split("\n")[0].endswith("SO:coordinate")
    assert text.split("\n")[1:] == text0.split("\n")[1:]
    want = {}
    for o, t in zip(ords, tags):
        r = bytearray(recs0[o] + t)
        want[bytes(r[:10]) + bytes(r[12:])] = want.get(bytes(r[:10]) + bytes(r[12:]), 0) + 1
    got = {}
    for r in recs:
        got[r[:10] + r[12:]] = got.get(r[:10] + r[12:], 0) + 1
    assert got == want                                       # every byte but the (recomputed) bin field
    keys = [((f["tid"] & 0xFFFFFFFF), f["pos"] + 1, bool(f["flag"] & 16)) for f in map(_fields, recs)]
    assert keys == sorted(keys)
    with open(out, "rb") as fh:                              # BGZF EOF marker block
        assert fh.read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _reg2bins(beg, end):
    end -= 1
    bins = [0]
    for shift, off in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins.extend(range(off + (beg >> shift), off + (end >> shift) + 1))
    return bins


def test_bai_region_queries_match_brute_force(tmp_path):
    from kmer_denovo_filter_amd.reads import write_bam_subset
    out = str(tmp_path / "sub.bam")
    ords = _pick(3)
    write_bam_subset(SRC, out, ords, None)
    _text, refs, recs = _all_records(out)
    fields = [_fields(r) for r in recs]
    bai = open(out + ".bai", "rb").read()
    assert bai[:4] == b"BAI\1" and struct.unpack_from("<i", bai, 4)[0] == len(refs)
    o = 8
    index = []
    for _ in refs:
        n_bin = struct.unpack_from("<i", bai, o)[0]; o += 4
        bins = {}
        for _b in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", bai, o); o += 8
            bins[b] = [struct.unpack_from("<QQ", bai, o + 16 * c) for c in range(n_chunk)]
            o += 16 * n_chunk
        n_intv = struct.unpack_from("<i", bai, o)[0]; o += 4
        lin = list(struct.unpack_from(f"<{n_intv}Q", bai, o)); o += 8 * n_intv
        index.append((bins, lin))
    n_no_coor = struct.unpack_from("<Q", bai, o)[0]
    assert o + 8 == len(bai)
    assert n_no_coor == sum(1 for f in fields if f["tid"] < 0)

    comp = open(out, "rb").read()

    def read_chunk(vbeg, vend):
        """Record payloads between two virtual offsets (inflating block by block)."""
        import zlib
        res = []
        coff, uoff = vbeg >> 16, vbeg & 0xFFFF
        buf = b""
        def block(at):
            bsize = struct.unpack_from("<H", comp, at + 16)[0] + 1
            return zlib.decompress(comp[at + 18:at + bsize - 8], -15), bsize
        data, bsize = block(coff)
        while (coff << 16 | uoff) < vend:
            while len(data) - uoff < 4:
                rest = data[uoff:]; coff += bsize; data, bsize = block(coff); data = rest + data; uoff = 0
            bs = struct.unpack_from("<i", data, uoff)[0]
            while len(data) - uoff < 4 + bs:
                rest = data[uoff:]; coff += bsize; nd, bsize = block(coff); data = rest + nd; uoff = 0
            res.append(data[uoff + 4:uoff + 4 + bs]); uoff += 4 + bs
            if uoff == len(data):
                coff += bsize; uoff = 0
                if coff >= len(comp): break
                data, bsize = block(coff)
        return res

    rng = np.random.default_rng(5)
    mapped = [f for f in fields if f["tid"] >= 0]
    checked = 0
    for f in [mapped[i] for i in rng.choice(len(mapped), 25, replace=False)]:
        tid = f["tid"]
        beg = max(0, f["pos"] - int(rng.integers(0, 400))); end = f["pos"] + int(rng.integers(1, 400))
        bins, lin = index[tid]
        assert 37450 in bins and bins[37450][1][0] == sum(1 for g in mapped if g["tid"] == tid and not g["flag"] & 4)
        min_off = lin[beg >> 14] if (beg >> 14) < len(lin) else (lin[-1] if lin else 0)
        got = set()
        for b in _reg2bins(beg, end):
            for cb, ce in bins.get(b, []):
                if ce <= min_off:
                    continue
                for rec in read_chunk(cb, ce):
                    g = _fields(rec)
                    if g["tid"] != tid:                       # (a chunk joined with its neighbour may run over other records: readers filter)
                        continue
                    if g["pos"] < end and g["end"] > beg:
                        got.add((g["name"], g["flag"], g["pos"]))
        exp = {(g["name"], g["flag"], g["pos"]) for g in mapped if g["tid"] == tid and g["pos"] < end and g["end"] > beg}
        assert got == exp and exp
        checked += len(exp)
    assert checked >= 25


def test_subset_writer_errors(tmp_path):
    import pytest
    from kmer_denovo_filter_amd._native import KdfError
    from kmer_denovo_filter_amd.reads import write_bam_subset
    out = str(tmp_path / "x.bam")
    assert write_bam_subset(SRC, out, [], None) == 0                       # empty selection: valid empty BAM + index
    assert _all_records(out)[2] == [] and os.path.exists(out + ".bai")
    with pytest.raises(KdfError):
        write_bam_subset(SRC, out, [10 ** 9], None)                        # past the end
    with pytest.raises(KdfError):
        write_bam_subset(SRC, out, [3, 3], None)                           # duplicates
    with pytest.raises(KdfError):
        write_bam_subset(str(tmp_path / "missing.bam"), out, [0], None)


def _parse_bai(raw):
    assert raw[:4] == b"BAI\1"
    n_ref = struct.unpack_from("<i", raw, 4)[0]
    o = 8; refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", raw, o)[0]; o += 4
        bins = {}
        for _b in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", raw, o); o += 8
            bins[b] = [struct.unpack_from("<QQ", raw, o + 16 * c) for c in range(n_chunk)]; o += 16 * n_chunk
        n_intv = struct.unpack_from("<i", raw, o)[0]; o += 4
        lin = struct.unpack_from(f"<{n_intv}Q", raw, o); o += 8 * n_intv
        refs.append((bins, lin))
    n_no_coor = struct.unpack_from("<Q", raw, o)[0] if o + 8 <= len(raw) else None
    return refs, n_no_coor


def test_bai_structure_equals_samtools_index_of_the_fixture(tmp_path):
    """The reference ships HG002_child.bam.bai, written by samtools.  Re-writing ALL records of that BAM through
    kdf_bam_write_subset (already coordinate sorted) must give an index with the same STRUCTURE: the same bins
    per reference, the same number of 16 kb linear-index windows, the same mapped / unmapped counts in the
    metadata pseudo-bin and the same count of unplaced reads.  (Virtual offsets differ: the BGZF blocks are ours.)"""
    from kmer_denovo_filter_amd.reads import write_bam_subset
    _t, _r, recs0 = _all_records(SRC)
    out = str(tmp_path / "all.bam")
    assert write_bam_subset(SRC, out, list(range(len(recs0))), None) == len(recs0)
    got, got_nc = _parse_bai(open(out + ".bai", "rb").read())
    exp, exp_nc = _parse_bai(open(SRC + ".bai", "rb").read())
    assert len(got) == len(exp) and got_nc == exp_nc
    for (gb, gl), (eb, el) in zip(got, exp):
        assert set(gb) == set(eb)                                            # bins incl. the pseudo-bin 37450 (after htslib's merging of small bins)
        assert {k: len(v) for k, v in gb.items()} == {k: len(v) for k, v in eb.items()}      # and the same number of chunks in each
        assert len(gl) == len(el)
        if 37450 in eb:
            assert gb[37450][1] == eb[37450][1]                              # (n_mapped, n_unmapped)
        # the same windows are empty / filled in the linear index
        assert [i for i, v in enumerate(gl) if v == 0] == [i for i, v in enumerate(el) if v == 0]
    # records come out in the order samtools sort would leave them: the file was sorted already
    keys = [((f["tid"] & 0xFFFFFFFF), f["pos"]) for f in map(_fields, _all_records(out)[2])]
    assert keys == [((f["tid"] & 0xFFFFFFFF), f["pos"]) for f in map(_fields, recs0)]
