"""Host feeders of libkdf.so (no GPU needed): the packer, the BAM reader with
`samtools fasta -F 0xD00` semantics and the FASTA reader, checked against the
oracle's independent pure-Python readers on the reference's fixtures."""
import os
import struct

import numpy as np
import pytest

from conftest import GIAB


def unpack(st):
    """ReadStream -> list of strings (N for invalid)."""
    bits = np.unpackbits(st.packed.view(np.uint8), bitorder="little")
    codes = bits[0::2] | (bits[1::2] << 1)
    inv = np.unpackbits(st.invalid.view(np.uint8), bitorder="little").astype(bool)
    chars = np.frombuffer(b"ACGT", np.uint8)[codes[:st.n_bases]].copy()
    chars[inv[:st.n_bases]] = ord("N")
    s = chars.tobytes().decode()
    out = []
    for r in range(st.n_reads):
        a, b = int(st.offsets[r]), int(st.offsets[r + 1])
        assert b > a and inv[b - 1]          # separator
        out.append(s[a:b - 1])
    return out


def test_pack_reads_roundtrip():
    from kmer_denovo_filter_amd import ReadStream
    reads = ["ACGTN", "", "acgtacgtac", "A" * 64, "C" * 63, "G" * 65, "RYKM", "T"]
    st = ReadStream.from_strings(reads)
    assert st.n_bases == sum(len(r) for r in reads) + len(reads)
    exp = [r.upper().replace("R", "N").replace("Y", "N").replace("K", "N").replace("M", "N") for r in reads]
    assert unpack(st) == exp
    # padding contract: packed tail zero, mask tail all ones
    assert (st.packed[(st.n_bases + 31) // 32:] == 0).all()
    assert (st.invalid[(st.n_bases + 63) // 64:] == ~np.uint64(0)).all()
    e = ReadStream.empty()
    assert e.n_bases == 0 and e.n_reads == 0


@pytest.mark.parametrize("sample", ["HG002_child", "HG004_mother", "HG003_father"])
def test_bam_reader_matches_samtools_fasta_semantics(oracle, sample):
    from kmer_denovo_filter_amd import bam_reader
    path = os.path.join(GIAB, sample + ".bam")
    exp = oracle.samtools_fasta_reads(path)
    got = []
    # small batches: exercises batch boundaries and the collapse run-over-batch logic
    for st in bam_reader(path, max_bases=20000, max_reads=64):
        got.extend(unpack(st))
    assert len(got) == len(exp)
    # samtools emits READ_OTHER, READ1, READ2 of a run in that order, as the oracle does
    assert got == [s if set(s) <= set("ACGTN") else "".join(c if c in "ACGT" else "N" for c in s) for s in exp]
    if sample == "HG002_child":
        assert len(got) == 10741


def test_bam_reader_module3_mode(oracle):
    """flag_off=0x500, no collapse == pysam iteration of core/bam_scanner.py:405-409."""
    from kmer_denovo_filter_amd import FLAG_OFF_MODULE3, bam_reader
    path = os.path.join(GIAB, "HG002_child.bam")
    _, recs = oracle.read_bam(path)
    exp = [(r.qname, r.flag, r.ref_id, r.pos) for r in recs if not (r.is_secondary or r.is_duplicate)]
    got = []
    for st in bam_reader(path, flag_off=FLAG_OFF_MODULE3, collapse=False, max_bases=1 << 20, want_meta=True):
        got.extend(zip(st.names, st.flags.tolist(), st.ref_ids.tolist(), st.positions.tolist()))
    assert got == exp


def test_fasta_reader_and_split_overlap(oracle, tmp_path):
    from kmer_denovo_filter_amd import fasta_reader
    path = os.path.join(GIAB, "mini_ref.fa")
    exp = [s.upper() for _, s in oracle.read_fasta(path)]
    got = []
    for st in fasta_reader(path, 31, max_bases=1 << 20):
        got.extend(unpack(st))
    assert got == ["".join(c if c in "ACGT" else "N" for c in s) for s in exp]
    # a sequence longer than the batch is split with k-1 overlap: the multiset
    # of windows is unchanged
    k = 31
    whole = oracle.OracleTable(k).count_reads(exp)
    pieces = []
    for st in fasta_reader(path, k, max_bases=997, max_reads=5):
        pieces.extend(unpack(st))
    assert len(pieces) > len(exp)
    split = oracle.OracleTable(k).count_reads(pieces)
    a, b = whole.export_ge(0), split.export_ge(0)
    assert (a[0] == b[0]).all() and (a[2] == b[2]).all()
    # lower case, multi-line, gz, blank lines, '>' only header
    p = tmp_path / "x.fa"
    p.write_text(">s1 desc\nacgt\nNNAC\n\n>s2\nGG\n>\nTT\n")
    got = []
    for st in fasta_reader(str(p), 3, want_meta=True):
        got.extend(zip(st.names, unpack(st)))
    assert got == [("s1", "ACGTNNAC"), ("s2", "GG"), ("", "TT")]


def test_reader_errors(tmp_path):
    from kmer_denovo_filter_amd import bam_reader
    from kmer_denovo_filter_amd._native import KdfError
    with pytest.raises(KdfError):
        bam_reader(str(tmp_path / "missing.bam"))
    bad = tmp_path / "bad.bam"
    bad.write_bytes(b"not a bam file at all")
    with pytest.raises(KdfError):
        bam_reader(str(bad))


def test_key_codec_matches_oracle(oracle):
    from kmer_denovo_filter_amd import keys_to_kmers, kmers_to_keys
    rng = np.random.default_rng(3)
    for k in (3, 31, 32, 33, 63):
        ks = ["".join(rng.choice(list("ACGT"), size=k)) for _ in range(50)]
        lo, hi = kmers_to_keys(ks, k)
        for s, l, h in zip(ks, lo, hi):
            assert ((int(h) << 64) | int(l)) == oracle.canonical_key(s)
        assert keys_to_kmers(lo, hi, k) == [oracle.canonicalize(s) for s in ks]
    with pytest.raises(ValueError):
        kmers_to_keys(["ACGN"], 4)


def _drain(path, **kw):
    """Everything a reader yields, flattened: (ASCII-equivalent bases per read, flags, ref, pos, names, ordinals)."""
    from kmer_denovo_filter_amd import bam_reader
    seqs, meta = [], []
    for st in bam_reader(path, **kw):
        bits = np.unpackbits(st.packed.view(np.uint8), bitorder="little")
        codes = (bits[0::2] | (bits[1::2] << 1))
        inv = np.unpackbits(st.invalid.view(np.uint8), bitorder="little").astype(bool)
        chars = np.frombuffer(b"ACGT", np.uint8)[codes[:len(inv)][:st.n_bases]].copy()
        chars[inv[:st.n_bases]] = ord("N")
        for i in range(st.n_reads):
            a, b = int(st.offsets[i]), int(st.offsets[i + 1])
            assert inv[b - 1]                                            # the separator after every read
            seqs.append(bytes(chars[a:b - 1]))
            if st.flags is not None:
                m = (int(st.flags[i]), int(st.ref_ids[i]), int(st.positions[i]), st.name(i), int(st.ordinals[i]))
                if st.cigar is not None:
                    q = st.qualities(i)
                    m += (tuple(st.cigartuples(i)), st.sa_tag(i), None if q is None else bytes(q), int(st.mapq[i]))
                meta.append(m)
    return seqs, meta


@pytest.mark.parametrize("collapse,flag_off", [(True, 0xD00), (False, 0x500), (False, 0)])
def test_parallel_parsing_equals_sequential(collapse, flag_off):
    """threads > 1 parses records in parallel chunks (cut on QNAME-run boundaries when runs are
    collapsed) and stitches chunk-local streams into the batch: the result must be identical to
    the single-threaded reader for any batch geometry, including batches that end inside a chunk."""
    path = os.path.join(GIAB, "HG002_child.bam")
    ref = _drain(path, flag_off=flag_off, collapse=collapse, threads=1, want_meta=True, max_bases=1 << 22)
    assert len(ref[0]) > 10_000
    for kw in (dict(max_bases=1 << 22, max_reads=1 << 20), dict(max_bases=70_001, max_reads=1 << 20),
               dict(max_bases=1 << 22, max_reads=333), dict(max_bases=251 * 3, max_reads=2)):
        got = _drain(path, flag_off=flag_off, collapse=collapse, threads=4, want_meta=True, **kw)
        assert got[0] == ref[0], kw
        assert got[1] == ref[1], kw
    # alignment details (CIGAR, SA, qualities, MAPQ) through the parallel parser
    ref_aux = _drain(path, flag_off=flag_off, collapse=collapse, threads=1, want_aux=True, max_bases=1 << 22)
    assert all(len(m) == 9 for m in ref_aux[1]) and any(m[5] for m in ref_aux[1])
    for kw in (dict(max_bases=1 << 22), dict(max_bases=9_001, max_reads=17)):
        assert _drain(path, flag_off=flag_off, collapse=collapse, threads=4, want_aux=True, **kw) == ref_aux, kw
    # no metadata wanted; early close in mid-stream must not hang
    from kmer_denovo_filter_amd import bam_reader
    assert _drain(path, flag_off=flag_off, collapse=collapse, threads=3, max_bases=1 << 20)[0] == ref[0]
    rd = bam_reader(path, flag_off=flag_off, collapse=collapse, threads=4, max_bases=50_000)
    next(iter(rd)); rd.close()


def test_parallel_parsing_synthetic_runs(tmp_path):
    """QNAME runs that straddle the chunker's 2048-record target, with dropped records inside runs."""
    from helpers import write_bam
    rng = np.random.default_rng(9)
    B = np.frombuffer(b"ACGTN", np.uint8)
    reads = []
    for i in range(9000):
        name = f"q{i // 3}"                                               # runs of 3 records
        flag = [0x41, 0x81, 0x941][i % 3] if i % 7 else 0x141               # supplementary third record; some secondary R1s
        seq = B[rng.integers(0, 5, int(rng.integers(1, 90)), dtype=np.int64).clip(0, 4)].tobytes().decode()
        aux = b""
        if i % 5 == 0:       # optional fields of several types in front of SA:Z (the walker must step over them)
            aux = (b"NMC\x03" + b"ASi" + struct.pack("<i", -7) + b"XBBs" + struct.pack("<i", 2) + struct.pack("<hh", 1, 2)
                   + b"MDZ10A5\0" + b"SAZ" + f"chr1,{i + 1},+,20M30S,60,1;".encode() + b"\0" + b"XSf" + struct.pack("<f", 1.5))
        reads.append({"name": name, "seq": seq, "pos": i, "flag": flag, "aux": aux, "qual": bool(i % 4)})
    path = str(tmp_path / "runs.bam")
    write_bam(path, [("chr1", 1_000_000)], reads)
    for collapse, flag_off in ((True, 0xD00), (True, 0x900), (False, 0x100)):
        ref = _drain(path, flag_off=flag_off, collapse=collapse, threads=1, want_meta=True)
        for kw in (dict(), dict(max_bases=4096, max_reads=50)):
            assert _drain(path, flag_off=flag_off, collapse=collapse, threads=5, want_meta=True, **kw) == ref
        ref_aux = _drain(path, flag_off=flag_off, collapse=collapse, threads=1, want_aux=True)
        sa = [m[6] for m in ref_aux[1] if m[6]]
        assert sa and all(x.startswith("chr1,") and x.endswith(",20M30S,60,1;") for x in sa)
        assert any(m[7] is None for m in ref_aux[1]) and any(m[7] is not None for m in ref_aux[1])
        assert _drain(path, flag_off=flag_off, collapse=collapse, threads=5, want_aux=True, max_bases=5000, max_reads=64) == ref_aux


def test_cram_is_recognised_by_its_magic_bytes(tmp_path):
    """A CRAM file under any name is refused with the CRAM message (the check used to look at the
    file name only, so a CRAM called x.bam failed later as 'not a BAM file')."""
    from kmer_denovo_filter_amd._native import KdfError
    from kmer_denovo_filter_amd.reads import bam_reader
    p = tmp_path / "sample.bam"
    p.write_bytes(b"CRAM\x03\x00" + b"\0" * 64)
    with pytest.raises(KdfError) as ei:
        with bam_reader(str(p)) as rd:
            next(iter(rd))
    assert "CRAM" in str(ei.value) and "samtools view -b" in str(ei.value)


@pytest.mark.parametrize("collapse,flag_off", [(True, 0xD00), (False, 0x500)])
def test_bgzf_ranges_together_are_the_whole_file(tmp_path, collapse, flag_off):
    """kdf_bam_open_range: the parts 0 .. n - 1 of a BAM, read one after the other, yield exactly the records the
    plain reader yields -- same sequences, same order -- for any number of parts (more parts than BGZF blocks: the
    surplus ones are empty), with 1 or several threads per part, and no QNAME run is split at a cut (the synthetic file
    has runs of three records, dropped records inside runs, and cuts fall inside runs)."""
    from helpers import write_bam
    rng = np.random.default_rng(31)
    B = np.frombuffer(b"ACGTN", np.uint8)
    reads = []
    for i in range(30_000):
        flag = [0x41, 0x81, 0x941][i % 3] if i % 7 else 0x141
        seq = B[rng.integers(0, 5, int(rng.integers(1, 160)), dtype=np.int64).clip(0, 4)].tobytes().decode()
        reads.append({"name": f"read{i // 3}", "seq": seq, "pos": i, "flag": flag, "aux": b"NMC\x03", "qual": bool(i % 4)})
    syn = str(tmp_path / "runs.bam")
    write_bam(syn, [("chr1", 1_000_000), ("chr2", 5000)], reads)
    for path in (os.path.join(GIAB, "HG002_child.bam"), syn):
        whole = _drain(path, flag_off=flag_off, collapse=collapse, threads=1, max_bases=1 << 22)[0]
        assert len(whole) > 5000
        for parts, threads in ((2, 1), (3, 4), (7, 1), (16, 3), (61, 1), (400, 1)):
            got, sizes = [], []
            for p in range(parts):
                seqs = _drain(path, flag_off=flag_off, collapse=collapse, threads=threads, max_bases=1 << 20, part=p, parts=parts)[0]
                got += seqs; sizes.append(len(seqs))
            assert got == whole, (path, parts, threads, sizes)
            if parts <= 16:
                assert min(sizes) > 0, (parts, sizes)                      # every range of a file this size holds records
