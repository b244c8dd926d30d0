"""Host feeders of libkdf.so (no GPU needed): the packer, the BAM reader with
`samtools fasta -F 0xD00` semantics and the FASTA reader, checked against the
oracle's independent pure-Python readers on the reference's fixtures."""
import os

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
