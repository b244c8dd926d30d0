"""Host-only pieces of N3 (no GPU): the pysam-like alignment view, the
variant-spanning k-mer producer and read_supports_alt, replaying the reference's
own unit cases (tests/test_kmer_utils.py:47-530 use a MockRead with the same
members)."""
import numpy as np


def _read(seq, start, cigar=None, quals=None):
    from kmer_denovo_filter_amd.alignment import AlignedRead
    return AlignedRead("r", 0, 0, "chr1", start, 60, cigar or [(0, len(seq))], seq,
                       None if quals is None else np.array(quals, np.uint8))


def test_alignment_view_matches_pysam_semantics():
    r = _read("ACGTACGTAC", 100, [(4, 2), (0, 3), (1, 2), (0, 1), (2, 2), (0, 2)])
    assert r.reference_end == 108
    assert r.get_reference_positions(full_length=True) == [None, None, 100, 101, 102, None, None, 103, 106, 107]
    assert r.get_reference_positions() == [100, 101, 102, 103, 106, 107]
    pairs = r.get_aligned_pairs()
    assert pairs[:3] == [(0, None), (1, None), (2, 100)] and (None, 104) in pairs and (None, 105) in pairs
    assert r.get_aligned_pairs(matches_only=True) == [(2, 100), (3, 101), (4, 102), (7, 103), (8, 106), (9, 107)]


def test_extract_variant_spanning_kmers_reference_cases():
    from kmer_denovo_filter_amd.kmer_utils import canonicalize, extract_variant_spanning_kmers
    r = _read("ACGTACGT", 100)
    assert len(extract_variant_spanning_kmers(r, 102, 4, min_baseq=0)) == 3          # starts 0, 1, 2
    assert extract_variant_spanning_kmers(_read("ACGT", 100), 200, 3, min_baseq=0) == set()
    q = [30, 30, 5, 30, 30, 30, 30, 30]
    assert extract_variant_spanning_kmers(_read("ACGTACGT", 100, quals=q), 102, 4, min_baseq=20) == set()
    assert extract_variant_spanning_kmers(_read("ACNTACGT", 100), 102, 4, min_baseq=0) == set()
    ks = extract_variant_spanning_kmers(_read("TTTTAAAA", 100), 103, 4, min_baseq=0)
    assert ks and all(k == canonicalize(k) for k in ks)
    # insertion: window extends over the inserted bases
    r = _read("ACGTTTTACGT", 100, [(0, 4), (1, 3), (0, 4)])
    ks = extract_variant_spanning_kmers(r, 103, 4, 0, ref="T", alt="TTTT")
    assert canonicalize("TTTA") in ks and canonicalize("GTTT") in ks


def test_read_supports_alt_reference_cases():
    from kmer_denovo_filter_amd.kmer_utils import read_supports_alt
    assert read_supports_alt(_read("ACGTACGT", 100), 102, "G", "G") is True
    assert read_supports_alt(_read("ACGTACGT", 100), 102, "G", "T") is False
    assert read_supports_alt(_read("ACGTACGT", 100), 102, "G", "<DEL>") is False
    assert read_supports_alt(_read("ACGTACGT", 100), 102, "G", None) is False
    assert read_supports_alt(_read("ACGTACGT", 100), 300, "G", "G") is False
    ins = _read("ACGTTTTACGT", 100, [(0, 4), (1, 3), (0, 4)])
    assert read_supports_alt(ins, 103, "T", "TTTT") is True
    dele = _read("ACGACGT", 100, [(0, 3), (2, 2), (0, 4)])                       # deletes ref 103-104
    assert read_supports_alt(dele, 102, "GTA", "G") is True
    low = _read("ACGTACGT", 100, quals=[30, 30, 5, 30, 30, 30, 30, 30])
    assert read_supports_alt(low, 102, "G", "G", min_baseq=20) is False


def test_parse_vcf_and_annotation(tmp_path):
    from kmer_denovo_filter_amd.vcf.pipeline import _parse_vcf_variants, annotate_variants
    p = tmp_path / "x.vcf"
    p.write_text("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tHG002\n"
                 "chr1\t101\t.\tA\tT\t50\tPASS\t.\tGT\t0/1\nchr1\t200\trs1\tC\tG,T\t50\tPASS\t.\tGT\t0/2\n")
    v = _parse_vcf_variants(str(p), proband_id="HG002")
    assert v[0] == {"chrom": "chr1", "pos": 100, "ref": "A", "alts": ("T",), "alt": "T", "id": None}
    assert v[1]["alt"] == "T" and v[1]["id"] == "rs1"
    assert _parse_vcf_variants(str(p))[1]["alt"] == "G"
    vk = {"chr1:100:A:T": [("r1", {"AAAC", "AACC"}, True), ("r1", {"AAAC"}, False), ("r2", {"CCCC"}, False)]}
    ann = annotate_variants(v[:1], vk, {"AAAC": 5, "CCCC": 2})
    assert ann["chr1:100:A:T"] == {"dku": 1, "dkt": 2, "dka": 1, "dku_dkt": 0.5, "dka_dkt": 0.5, "max_pkc": 5,
                                   "avg_pkc": 3.5, "min_pkc": 2, "max_pkc_alt": 5, "avg_pkc_alt": 5, "min_pkc_alt": 5}


def test_records_over_positions_matches_the_per_record_loop():
    """The vectorised matching of a batch's records against sorted variant positions (CIGAR reference spans +
    searchsorted per chromosome) against the obvious loop -- pysam's `fetch(chrom, pos, pos + 1)` semantics as the
    reference uses them (vcf/pipeline.py:619-726): a record is taken iff start <= pos < reference_end."""
    import bisect
    import os
    import numpy as np
    from kmer_denovo_filter_amd.core.bam_scanner import reference_end
    from kmer_denovo_filter_amd.reads import bam_reader
    from kmer_denovo_filter_amd.vcf.pipeline import _parse_vcf_variants, _records_over_positions
    giab = os.path.join(os.path.dirname(__file__), "golden", "giab")
    variants = _parse_vcf_variants(os.path.join(giab, "candidates.vcf.gz"), proband_id="HG002")
    by_chrom = {}
    for v in variants:
        by_chrom.setdefault(v["chrom"], set()).add(v["pos"])
    rng = np.random.default_rng(4)
    for c in list(by_chrom):                                    # plus random positions, so that most batches have hits and misses
        by_chrom[c] |= set(int(x) for x in rng.integers(min(by_chrom[c]) - 2000, max(by_chrom[c]) + 2000, 40))
    vpos = {c: np.asarray(sorted(p), dtype=np.int64) for c, p in by_chrom.items()}
    rd = bam_reader(os.path.join(giab, "HG002_child.bam"), flag_off=0, collapse=False, max_bases=1 << 16, threads=2, want_aux=True)
    refs = rd.references()
    n_hit = n_all = 0
    with rd:
        for batch in rd:
            n = batch.n_reads
            elig = (np.asarray(batch.flags[:n]) & 0x4) == 0
            for empty_one in (False, True):
                keep, first = _records_over_positions(batch, refs, vpos, elig, empty_span_is_one=empty_one)
                exp_keep, exp_first = [], []
                for i in range(n):
                    rid = int(batch.ref_ids[i])
                    if not elig[i] or rid < 0 or refs[rid] not in vpos:
                        continue
                    start = int(batch.positions[i]); end = reference_end(start, batch.cigartuples(i))
                    if empty_one and end <= start:
                        end = start + 1
                    pl = vpos[refs[rid]]
                    j = bisect.bisect_left(pl.tolist(), start)
                    if j < len(pl) and pl[j] < end:
                        exp_keep.append(i); exp_first.append(j)
                assert keep.tolist() == exp_keep and first.tolist() == exp_first
            n_hit += len(keep); n_all += n
    assert 0 < n_hit < n_all
