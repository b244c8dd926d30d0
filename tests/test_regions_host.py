"""Host-only pieces of N1 (no GPU): CIGAR walker, soft clips, clustering, run
merging in the writers."""
import numpy as np


def test_cigar_walker_and_softclips():
    from kmer_denovo_filter_amd.core import bam_scanner as B
    cig = [(4, 3), (0, 5), (1, 2), (0, 4), (2, 3), (0, 2), (4, 1)]          # 3S5M2I4M3D2M1S
    q2r = B._query_to_ref(100, cig, 17)
    assert q2r.tolist() == [-1] * 3 + [100, 101, 102, 103, 104] + [-1, -1] + [105, 106, 107, 108] + [112, 113] + [-1]
    assert B.reference_end(100, cig) == 114
    assert B._extract_softclips(cig) == (3, 1)
    assert B._extract_softclips([(5, 2), (4, 7), (0, 10), (5, 1)]) == (7, 0)
    assert B._extract_softclips([(4, 9)]) == (9, 0)
    assert B._extract_softclips(None) == (0, 0)
    cov = B._collect_kmer_ref_positions(100, cig, 17, np.array([3, 4]), 5)
    # k-mer at q3 covers q3..7 -> r100..104 ; k-mer at q4 covers q4..8 -> r101..104 (q8 inserted)
    assert dict(cov) == {100: 1, 101: 2, 102: 2, 103: 2, 104: 2}


def test_clustering_and_runs(tmp_path):
    from kmer_denovo_filter_amd.discovery import regions as R
    hits = [("chr2", 100, 150, "a", {"K1"}, False), ("chr10", 5, 60, "b", {"K2"}, False),
            ("chr2", 640, 700, "c", {"K1", "K3"}, False), ("chr2", 1300, 1350, "d", {"K4"}, False)]
    regions, rr, rk = R._cluster_read_hits(hits, 500)
    assert regions == [("chr10", 5, 60), ("chr2", 100, 700), ("chr2", 1300, 1350)]       # chrom NAME order
    assert rr[("chr2", 100, 700)] == {"a", "c"} and rk[("chr2", 100, 700)] == {"K1", "K3"}
    assert R._cluster_read_hits(hits, 0)[0] == [("chr10", 5, 60), ("chr2", 100, 150), ("chr2", 640, 700), ("chr2", 1300, 1350)]
    kc = {"chr1": {10: 3, 11: 3, 12: 6, 13: 6, 15: 6, 16: 1}}
    rc = {"chr1": {10: 3, 11: 3, 12: 3, 13: 2, 15: 3, 16: 3}}
    R._write_bedgraph(kc, str(tmp_path / "g"), read_coverage=rc, min_reads=3)
    assert open(tmp_path / "g").read().splitlines()[1:] == ["chr1\t10\t12\t3", "chr1\t12\t13\t6", "chr1\t15\t16\t6", "chr1\t16\t17\t1"]
    R._write_read_coverage_bed(kc, rc, str(tmp_path / "r"), min_reads=3)
    assert open(tmp_path / "r").read().splitlines()[2:] == ["chr1\t10\t12\t3\t1.0", "chr1\t12\t13\t3\t2.0", "chr1\t15\t16\t3\t2.0", "chr1\t16\t17\t3\t0.3"]
    ann = {("chr1", 0, 5): {"split_reads": 1, "discordant_pairs": 0, "unmapped_mates": 0}}
    R._classify_regions([("chr1", 0, 5)], ann, [])
    assert ann[("chr1", 0, 5)]["class"] == "AMBIGUOUS"
