"""GPU path (through the C ABI and the drop-in mirrors) against the reference's
committed goldens on the GIAB mini trio:
reference index == real Jellyfish file, 51125 -> 6679 -> 630, 195 / 11."""
import json
import os

import numpy as np
import pytest

from conftest import GIAB, GOLDEN

pytestmark = pytest.mark.gpu


def test_reference_index_equals_real_jellyfish(oracle, tmp_path):
    """_ensure_ref_jf builds mini_ref.fa's index; its (key, count) set must be
    bit-equal to the real Jellyfish binary/sorted fixture."""
    import shutil
    from kmer_denovo_filter_amd import jf_io
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _ensure_ref_jf
    fa = str(tmp_path / "mini_ref.fa")
    shutil.copy(os.path.join(GIAB, "mini_ref.fa"), fa)
    path = _ensure_ref_jf(fa, 31, 4)
    assert path == fa + ".k31.jf" and os.path.isfile(path)
    k, lo, hi, cnt = jf_io.read_index(path)
    jk, jlo, jhi, jcnt = jf_io.read_index(os.path.join(GIAB, "mini_ref.fa.k31.jf"))
    assert k == jk == 31 and len(lo) == 45275
    order = np.argsort(jlo)
    np.testing.assert_array_equal(lo, jlo[order])
    np.testing.assert_array_equal(cnt, jcnt[order])
    assert int(cnt.sum()) == 45804 and int(cnt.max()) == 12
    # an existing file is returned untouched
    before = os.path.getmtime(path)
    assert _ensure_ref_jf(fa, 31, 4) == path and os.path.getmtime(path) == before


def test_reference_index_written_as_jellyfish_is_the_real_file(tmp_path, monkeypatch):
    """KDF_JF_FORMAT=jellyfish: `_ensure_ref_jf` writes a Jellyfish binary/sorted file (records in the order of their hash
    position under the header's GF(2) matrix).  mini_ref.fa counted on the GPU and written under the REAL file's header
    (hash size and matrix of `jellyfish count -m 31 -s 100M -C`) must be that file, record for record, byte for byte."""
    import shutil
    from kmer_denovo_filter_amd import jf_io
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _ensure_ref_jf
    real = os.path.join(GIAB, "mini_ref.fa.k31.jf")
    fa = str(tmp_path / "mini_ref.fa")
    shutil.copy(os.path.join(GIAB, "mini_ref.fa"), fa)
    monkeypatch.setenv("KDF_JF_FORMAT", "jellyfish")
    path = _ensure_ref_jf(fa, 31, 4)
    header, off = jf_io.read_header(path)
    assert header["format"] == "binary/sorted" and header["size"] >= 2 * 45275
    k, lo, hi, cnt = jf_io.read_index(path)                          # (our own matrix: the same records, our order)
    pos = jf_io.jf_positions(header["matrix1"]["columns"], 62, lo) & np.uint64(header["size"] - 1)
    assert (np.diff(pos.astype(np.int64)) >= 0).all()
    rheader, roff = jf_io.read_header(real)
    again = str(tmp_path / "as_jellyfish.jf")
    jf_io.write_jellyfish_index(again, 31, lo, None, cnt, header=rheader)
    _, aoff = jf_io.read_header(again)
    assert open(again, "rb").read()[aoff:] == open(real, "rb").read()[roff:]
    # ... and the records themselves are the real file's
    monkeypatch.delenv("KDF_JF_FORMAT")
    _, jlo, _, jcnt = jf_io.read_index(real)
    o = np.argsort(jlo)
    np.testing.assert_array_equal(np.sort(lo), jlo[o]); np.testing.assert_array_equal(cnt[np.argsort(lo)], jcnt[o])


@pytest.fixture(scope="module")
def discovery(tmp_path_factory):
    from kmer_denovo_filter_amd.discovery.pipeline import (
        _extract_child_kmers_discovery, _filter_parents_discovery, _subtract_reference_kmers)
    from kmer_denovo_filter_amd.kmer_fasta import read_kmer_fasta_keys
    tmp = str(tmp_path_factory.mktemp("disc"))
    out = {}
    fa, n = _extract_child_kmers_discovery(os.path.join(GIAB, "HG002_child.bam"), None, 31, 3, 4, tmp)
    out["candidates"] = (n, read_kmer_fasta_keys(fa, 31))
    # the user-supplied real Jellyfish index (--ref-jf), as in tests/conftest.py:99-111
    fa2, n2 = _subtract_reference_kmers(os.path.join(GIAB, "mini_ref.fa.k31.jf"), fa, tmp)
    assert not os.path.exists(fa)                      # input FASTA is deleted
    out["non_ref"] = (n2, read_kmer_fasta_keys(fa2, 31))
    n3, fa3 = _filter_parents_discovery(os.path.join(GIAB, "HG004_mother.bam"),
                                        os.path.join(GIAB, "HG003_father.bam"), None, fa2, 31, 4, tmp, 0)
    out["proband_unique"] = (n3, read_kmer_fasta_keys(fa3, 31))
    out["proband_fa"] = fa3
    out["tmp"] = tmp
    return out


def test_discovery_chain_goldens(oracle, trio_reads, discovery):
    m = json.load(open(os.path.join(GOLDEN, "example_output_discovery", "giab_discovery.metrics.json")))
    assert discovery["candidates"][0] == m["child_candidate_kmers"] == 51125
    assert discovery["non_ref"][0] == m["non_ref_kmers"] == 6679
    assert discovery["proband_unique"][0] == m["proband_unique_kmers"] == 630
    # and the surviving SETS are bit-exact against the oracle chain
    ref = oracle.read_fasta(os.path.join(GIAB, "mini_ref.fa"))
    rt = oracle.OracleTable(31).count_reads([s for _, s in ref])
    st = oracle.discovery_chain(trio_reads["child"], trio_reads["mother"], trio_reads["father"], rt, 31, 3, 0)
    for stage in ("candidates", "non_ref", "proband_unique"):
        got = np.sort(discovery[stage][1][0])
        np.testing.assert_array_equal(got, st[stage][0])


def test_child_extraction_in_key_space_slices(discovery, tmp_path, monkeypatch):
    """KDF_KEY_PARTS: the child count done in three slices of the key space (three passes over the BAM)
    yields the same candidate set as the single pass."""
    from kmer_denovo_filter_amd.discovery.pipeline import _extract_child_kmers_discovery
    from kmer_denovo_filter_amd.kmer_fasta import read_kmer_fasta_keys
    monkeypatch.setenv("KDF_KEY_PARTS", "3")
    fa, n = _extract_child_kmers_discovery(os.path.join(GIAB, "HG002_child.bam"), None, 31, 3, 4, str(tmp_path))
    assert n == 51125
    lo, _ = read_kmer_fasta_keys(fa, 31)
    np.testing.assert_array_equal(np.sort(lo), np.sort(discovery["candidates"][1][0]))


def test_intermediate_fasta_contract(discovery):
    lines = open(discovery["proband_fa"]).read().split("\n")
    assert lines[0] == ">0" and lines[2] == ">1" and len(lines[1]) == 31
    assert lines[-1] == "" and len(lines) == 2 * 630 + 1
    assert lines[2 * 629] == ">629"


def test_module3_goldens(oracle, discovery):
    from kmer_denovo_filter_amd.core import bam_scanner
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _build_proband_jf_index
    m = json.load(open(os.path.join(GOLDEN, "example_output_discovery", "giab_discovery.metrics.json")))
    jf = _build_proband_jf_index(discovery["proband_fa"], 31, discovery["tmp"], 630)
    assert jf.endswith("proband_unique.jf")
    bam_scanner._init_scan_worker(jf, 31, m["filters"]["min_distinct_kmers_per_read"])
    total, unmapped, scanned, kept = bam_scanner.count_informative_reads(os.path.join(GIAB, "HG002_child.bam"))
    assert total == m["informative_reads"] == 195
    assert unmapped == m["unmapped_informative_reads"] == 11
    # per-read hit positions equal the oracle's
    lo, hi = discovery["proband_unique"][1]
    _, _, ohits = oracle.module3_scan(os.path.join(GIAB, "HG002_child.bam"), lo, hi, 31, 7)
    assert len(ohits) == len(kept)
    okey = {(r.qname, r.flag, r.pos): (pos, d) for r, pos, d in ohits}
    for inf in kept:
        pos, d = okey[(inf.query_name, inf.flag, inf.pos)]
        np.testing.assert_array_equal(inf.kmer_hit_indices, pos)
        assert inf.n_distinct == d


def test_scan_parent_jellyfish_counts(oracle, trio_reads, discovery, tmp_path):
    """VCF-mode Step 3 wrapper: dict of k-mer -> count (only count >= 1)."""
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _scan_parent_jellyfish
    from kmer_denovo_filter_amd.kmer_fasta import write_kmer_fasta
    lo, hi = discovery["non_ref"][1]
    fa = str(tmp_path / "child_kmers.fa")
    write_kmer_fasta(fa, lo, hi, 31)
    got = _scan_parent_jellyfish(os.path.join(GIAB, "HG004_mother.bam"), None, fa, 31, str(tmp_path / "mother"), 4,
                                 n_filter_kmers=len(lo))
    ot = oracle.OracleTable(31).load_filter(lo, hi).count_reads_filtered(trio_reads["mother"])
    elo, ehi, ecnt = ot.export_ge(1)
    exp = {oracle.int_to_kmer(int(l), 31): int(c) for l, c in zip(elo, ecnt)}
    assert got == exp and len(got) == 6679 - 1513


def test_jellyfish_kmer_query_mirror(oracle, discovery):
    """reference tests/test_kmer_utils.py:594-709 on the engine-backed class."""
    from kmer_denovo_filter_amd import jf_io
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _build_proband_jf_index
    from kmer_denovo_filter_amd.kmer_utils import JellyfishKmerQuery, _extract_read_kmers, canonicalize
    import tempfile
    d = tempfile.mkdtemp()
    fa = os.path.join(d, "test.fa")
    open(fa, "w").write(">seq\nACGTACGTACGTACGTACGT\n")
    jf = _build_proband_jf_index(fa, 5, d)
    q = JellyfishKmerQuery(jf)
    assert len(q.query_batch([canonicalize("ACGTA")])) > 0
    assert len(q.query_batch([canonicalize("TTTTT")])) == 0
    assert canonicalize("ACGTA") in q._cache
    hits = q.query_batch([canonicalize("ACGTA"), canonicalize("TTTTT")])
    assert canonicalize("ACGTA") in hits and canonicalize("TTTTT") not in hits
    unique, idx = q.scan_read("ACGTACGTACGTACGTACGT", 5)
    cap, cands = _extract_read_kmers("ACGTACGTACGTACGTACGT", 5)
    assert unique == set(cands) and idx == set(cap)
    assert q.scan_read("TTTTTTTTTTTTTTT", 5) == (set(), set())
    assert q.query_batch([]) == set()
    q.close()
    assert len(q._cache) == 0


def test_discovery_region_outputs_match_goldens(discovery, tmp_path):
    """N1: scan hits -> reference coordinates -> clusters -> annotated BED,
    bedGraph, read-coverage BED and BEDPE, byte for byte equal to the reference's
    committed outputs (tests/example_output_discovery/giab_discovery.*;
    flags of tests/conftest.py:99-111: defaults, k=31, min_dk_per_read = 7)."""
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _build_proband_jf_index
    from kmer_denovo_filter_amd.discovery import regions as R
    gold = os.path.join(GOLDEN, "example_output_discovery")
    m = json.load(open(os.path.join(gold, "giab_discovery.metrics.json")))
    jf = _build_proband_jf_index(discovery["proband_fa"], 31, discovery["tmp"], 630)
    (regions, region_reads, total_inf, region_kmers, unmapped_inf, sv_meta, kcov, rcov) = R._anchor_and_cluster(
        os.path.join(GIAB, "HG002_child.bam"), None, None, 31, merge_distance=500, threads=4,
        min_distinct_kmers_per_read=7, proband_jf=jf, n_proband_unique=630)
    assert total_inf == m["informative_reads"] and unmapped_inf == m["unmapped_informative_reads"]
    assert len(regions) == m["candidate_regions"] == 21
    regions = R._filter_regions(regions, region_reads, region_kmers, 1, 1)
    ann, links = R._annotate_and_link_from_metadata(regions, region_reads, sv_meta)
    R._classify_regions(regions, ann, links)
    out = {n: str(tmp_path / n) for n in ("bed", "bedgraph", "readcov", "bedpe")}
    R._write_bed(regions, region_reads, region_kmers, out["bed"], region_annotations=ann,
                 filters={"min_distinct_kmers_per_read": 7, "min_supporting_reads": 1, "min_distinct_kmers": 1})
    R._write_bedgraph(kcov, out["bedgraph"], read_coverage=rcov, min_reads=3)
    R._write_read_coverage_bed(kcov, rcov, out["readcov"], min_reads=3)
    R._write_bedpe(links, out["bedpe"])
    for ours, theirs in (("bed", "giab_discovery.bed"), ("bedgraph", "giab_discovery.kmer_coverage.bedgraph"),
                         ("readcov", "giab_discovery.read_coverage.bed"), ("bedpe", "giab_discovery.sv.bedpe")):
        assert open(out[ours]).read() == open(os.path.join(gold, theirs)).read(), theirs
    for reg, g in zip(regions, m["regions"]):
        assert (reg[0], reg[1], reg[2]) == (g["chrom"], g["start"], g["end"])
        assert len(region_reads[reg]) == g["reads"] and len(region_kmers[reg]) == g["unique_kmers"]
        assert ann[reg]["class"] == g["class"] and ann[reg]["max_clip_len"] == g["max_clip_len"]


def test_vcf_mode_goldens(tmp_path):
    """N3: VCF mode end to end on the engine (tests/conftest.py:53-64: defaults
    k=31, --min-baseq 20, --min-mapq 20).  Pins the count --if COUNT VALUES
    (summed over the parents) through MIN/AVG/MAX_PKC[_ALT] of all 22 variants
    against the reference's committed summary (tests/example_output/summary.txt:31-52)
    and metrics.json (1484 / 1294 / 190 / 12)."""
    from kmer_denovo_filter_amd.vcf.pipeline import (_collect_child_kmers, _parse_vcf_variants, annotate_variants,
                                                     scan_parents)
    gold = os.path.join(GOLDEN, "example_output")
    m = json.load(open(os.path.join(gold, "metrics.json")))
    variants = _parse_vcf_variants(os.path.join(GIAB, "candidates.vcf.gz"), proband_id="HG002")
    assert len(variants) == m["total_variants"] == 22
    fa = str(tmp_path / "child_kmers.fa")
    total, per_variant = _collect_child_kmers(os.path.join(GIAB, "HG002_child.bam"), None, variants, 31, 20, 20,
                                              False, fa)
    assert total == m["total_child_kmers"] == 1484
    found = scan_parents(os.path.join(GIAB, "HG004_mother.bam"), os.path.join(GIAB, "HG003_father.bam"), None, fa,
                         31, str(tmp_path), 4, total)
    assert len(found) == m["parent_found_kmers"] == 1294
    assert max(0, total - len(found)) == m["child_unique_kmers"] == 190
    ann = annotate_variants(variants, per_variant, found)
    assert sum(1 for a in ann.values() if a["dku"] > 0) == m["variants_with_unique_reads"] == 12
    rows = {}
    for line in open(os.path.join(gold, "summary.txt")):
        f = line.split()
        if len(f) == 14 and f[0].startswith("chr") and ">" in f[1]:
            chrom, pos = f[0].split(":")
            ref, alt = f[1].split(">")
            rows[f"{chrom}:{int(pos) - 1}:{ref}:{alt}"] = f[2:13]
    assert len(rows) == 22
    for key, g in rows.items():
        a = ann[key]
        got = [a["dku"], a["dkt"], a["dka"], a["dku_dkt"], a["dka_dkt"], a["max_pkc"], a["avg_pkc"], a["min_pkc"],
               a["max_pkc_alt"], a["avg_pkc_alt"], a["min_pkc_alt"]]
        exp = [int(g[0]), int(g[1]), int(g[2]), float(g[3]), float(g[4]), int(g[5]), float(g[6]), int(g[7]),
               int(g[8]), float(g[9]), int(g[10])]
        assert got == exp, (key, got, exp)
    assert ann["chr19:15018719:G:A"]["max_pkc"] == 2177 and ann["chr18:62805215:A:ATAATATACACTGCATAGGTTATACATATACAGTG"]["avg_pkc"] == 683.58


def test_informative_reads_bam_discovery(oracle, discovery, tmp_path):
    """N4, discovery mode (reference tests/test_example_output_discovery.py:112-121 only
    ask for a non-empty BAM; here the selection is pinned against the oracle: every
    primary/supplementary, non-duplicate record with >= 1 proband-unique k-mer, first
    per (name, is_supplementary), each tagged dk, coordinate sorted, indexed)."""
    from kmer_denovo_filter_amd.discovery.pipeline import _write_informative_reads_discovery
    child = os.path.join(GIAB, "HG002_child.bam")
    out = str(tmp_path / "giab_discovery.informative.bam")
    n = _write_informative_reads_discovery(child, None, discovery["proband_fa"], 31, out)
    lo, hi = discovery["proband_unique"][1]
    kmers = [oracle.int_to_kmer((int(h) << 64) | int(l), 31) for l, h in zip(lo, hi)]
    table = oracle.OracleTable(31, 4096).count_reads(kmers)
    _refs, recs = oracle.read_bam(child)
    recs = [r for r in recs if not (r.is_secondary or r.is_duplicate)]
    _hit, distinct = table.scan_reads(oracle.concat_reads([r.seq for r in recs]))
    exp, seen = [], set()
    for r, d in zip(recs, distinct):
        key = (r.qname, r.is_supplementary)
        if d >= 1 and key not in seen:
            seen.add(key); exp.append((r.qname, r.flag, r.ref_id, r.pos))
    assert n == len(exp) > 195                       # min_distinct 1 here, 7 in Module 3's metric
    _refs2, got = oracle.read_bam(out)
    got = list(got)
    assert sorted((r.qname, r.flag, r.ref_id, r.pos) for r in got) == sorted(exp)
    keys = [((r.ref_id & 0xFFFFFFFF), r.pos) for r in got]
    assert keys == sorted(keys) and os.path.getsize(out + ".bai") > 8 and not os.path.exists(out + ".unsorted.bam")
    import gzip
    raw = gzip.open(out).read()
    assert raw.count(b"dkC\x01") >= n


def test_informative_reads_bam_vcf_mode(tmp_path):
    """N4, VCF mode: reads named in informative_reads_by_variant, one record per name
    (first met walking the variant positions in sorted order), DV:Z tag."""
    import gzip
    from kmer_denovo_filter_amd.vcf.pipeline import (_collect_child_kmers, _parse_vcf_variants, _write_informative_reads,
                                                     informative_reads_by_variant, scan_parents)
    child = os.path.join(GIAB, "HG002_child.bam")
    variants = _parse_vcf_variants(os.path.join(GIAB, "candidates.vcf.gz"), proband_id="HG002")
    fa = str(tmp_path / "child_kmers.fa")
    total, per_variant = _collect_child_kmers(child, None, variants, 31, 20, 20, False, fa)
    found = scan_parents(os.path.join(GIAB, "HG004_mother.bam"), os.path.join(GIAB, "HG003_father.bam"), None, fa,
                         31, str(tmp_path), 4, total)
    by_var = informative_reads_by_variant(variants, per_variant, found)
    assert len(by_var) == 12                                        # metrics.json variants_with_unique_reads
    out = str(tmp_path / "informative.bam")
    n = _write_informative_reads(child, None, by_var, out)
    names = set().union(*by_var.values())
    assert n == len(names)
    raw = gzip.open(out).read()
    for key, rn in by_var.items():
        assert raw.count(key.encode()) >= len(rn)
    assert os.path.exists(out + ".bai")


def test_child_key_parts_rule(tmp_path, monkeypatch):
    """Automatic choice of KDF_KEY_PARTS: one pass for the fixture, several for a BAM too large for the free HBM."""
    from kmer_denovo_filter_amd.discovery import pipeline as P
    monkeypatch.delenv("KDF_KEY_PARTS", raising=False)
    assert P._child_key_parts(os.path.join(GIAB, "HG002_child.bam")) == 1
    monkeypatch.setattr(os.path, "getsize", lambda p: 400 << 30)                 # a 400 GB BAM
    assert P._child_key_parts("whatever.bam") >= 7                               # 1.8 TB of table on <= 288 GB
    monkeypatch.setenv("KDF_KEY_PARTS", "5")
    assert P._child_key_parts("whatever.bam") == 5
