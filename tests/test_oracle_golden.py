"""Pins the ORACLE to the reference's own fixtures:
a real Jellyfish binary/sorted file, the discovery-chain goldens and the
canonicalize KATs.  If these fail the oracle is wrong, not the GPU path."""
import json
import os

import numpy as np
import pytest

from conftest import GIAB, GOLDEN


def test_canonicalize_kats(oracle):
    # reference tests/test_kmer_utils.py:14-44
    assert oracle.reverse_complement("ACGT") == "ACGT"
    assert oracle.reverse_complement("AAAA") == "TTTT"
    assert oracle.reverse_complement("AACG") == "CGTT"
    assert oracle.reverse_complement("acgt") == "acgt"
    assert oracle.canonicalize("ACGT") == "ACGT"
    assert oracle.canonicalize("TTTT") == "AAAA"
    assert oracle.canonicalize("AAAA") == "AAAA"
    assert oracle.canonicalize("ACCC") == "ACCC"
    assert oracle.canonicalize("GGGT") == "ACCC"
    # numeric min in the Jellyfish code == lexicographic min of the strings
    for s in ("ACGT", "TTTT", "GGGT", "ACCC", "CATG", "TGCA"):
        assert oracle.canonical_key(s) == oracle.kmer_to_int(oracle.canonicalize(s))


def test_extract_read_kmers_kats(oracle):
    # reference tests/test_kmer_utils.py:537-584
    cap, uniq = oracle.extract_read_kmers("ACGTACGT", 5)
    assert len(cap) == 4 and len(uniq) > 0
    assert oracle.extract_read_kmers("ACG", 5) == ({}, [])
    cap, _ = oracle.extract_read_kmers("ACNGTACGT", 5)
    assert all("N" not in v for v in cap.values())
    _, uniq = oracle.extract_read_kmers("AAAAAAAAAA", 5)
    assert len(uniq) == len(set(uniq))
    cap, _ = oracle.extract_read_kmers("ACGTAC", 4)
    assert cap == {i: oracle.canonicalize("ACGTAC"[i:i + 4]) for i in range(3)}


def test_c_oracle_matches_python_restatement(oracle):
    rng = np.random.default_rng(1)
    reads = ["".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04], size=int(rng.integers(0, 80))))
             for _ in range(60)] + ["acgtnACGTacgt", "ACGTRYACGTACGT"]
    for k in (3, 5, 11, 31, 33, 41):
        d = oracle.py_count(reads, k)
        t = oracle.OracleTable(k).count_reads(reads)
        lo, hi, c = t.export_ge(0)
        got = {oracle.int_to_kmer((int(h) << 64) | int(l), k): int(v) for l, h, v in zip(lo, hi, c)}
        assert got == d
        filt = set(list(d)[::3])
        df = oracle.py_count(reads[::2], k, filt=filt)
        flo, fhi = oracle.keys_to_arrays([oracle.kmer_to_int(x) for x in filt])
        tf = oracle.OracleTable(k).load_filter(flo, fhi).count_reads_filtered(reads[::2])
        lo, hi, c = tf.export_ge(0)
        got = {oracle.int_to_kmer((int(h) << 64) | int(l), k): int(v) for l, h, v in zip(lo, hi, c)}
        assert got == df


def test_jellyfish_fixture_bit_equal(oracle):
    """mini_ref.fa.k31.jf is real `jellyfish count -m 31 -s 100M -t 2 -C` output."""
    hdr, keys, counts = oracle.read_jf_binary_sorted(os.path.join(GIAB, "mini_ref.fa.k31.jf"))
    assert hdr["canonical"] is True and hdr["key_len"] == 62 and hdr["counter_len"] == 4
    assert len(keys) == 45275 and int(counts.sum()) == 45804 and int(counts.max()) == 12
    ref = oracle.read_fasta(os.path.join(GIAB, "mini_ref.fa"))
    assert len(ref) == 24 and sum(len(s) for _, s in ref) == 52034
    t = oracle.OracleTable(31).count_reads([s for _, s in ref])
    lo, hi, c = t.export_ge(0)
    assert sorted(zip(keys, counts.tolist())) == list(zip(lo.tolist(), c.tolist()))
    # threaded variant agrees
    t2 = oracle.OracleTable(31).count_reads([s for _, s in ref], threads=3)
    lo2, _, c2 = t2.export_ge(0)
    assert (lo2 == lo).all() and (c2 == c).all()
    # and so does the partitioned count + tally that bench.py times as its CPU baseline
    for threads in (1, 3, 8):
        assert oracle.count_tally_mt([s for _, s in ref], 31, threads, 2) == (45275, 45804, int((counts >= 2).sum()))


def test_partitioned_tally_equals_the_plain_count_on_the_child_reads(oracle, trio_reads):
    """kdfo_count_tally_mt (the CPU baseline of bench.py) against kdfo_count_reads + kdfo_export_ge on the
    golden child reads: 282 880 distinct 31-mers, 51 125 of them seen >= 3 times (metrics.json), and a
    wide-key (k = 63) and a tiny-k case against the plain count."""
    reads = trio_reads["child"]
    d, total, ge3 = oracle.count_tally_mt(reads, 31, 4, 3)
    assert (d, ge3) == (282880, 51125) and total == oracle.count_windows(reads, 31)
    for k in (63, 5):
        lo, hi, c = oracle.OracleTable(k).count_reads(reads).export_ge(0)
        assert oracle.count_tally_mt(reads, k, 5, 3) == (len(lo), int(c.astype("int64").sum()), int((c >= 3).sum()))


@pytest.fixture(scope="module")
def chain(oracle, trio_reads):
    ref = oracle.read_fasta(os.path.join(GIAB, "mini_ref.fa"))
    rt = oracle.OracleTable(31).count_reads([s for _, s in ref])
    return oracle.discovery_chain(trio_reads["child"], trio_reads["mother"], trio_reads["father"], rt, 31, 3, 0)


def test_discovery_chain_goldens(oracle, trio_reads, chain):
    m = json.load(open(os.path.join(GOLDEN, "example_output_discovery", "giab_discovery.metrics.json")))
    assert len(trio_reads["child"]) == 10741
    assert oracle.count_windows(trio_reads["child"], 31) == 2348843
    assert len(chain["candidates"][0]) == m["child_candidate_kmers"] == 51125
    assert len(chain["non_ref"][0]) == m["non_ref_kmers"] == 6679
    assert len(chain["after_mother"][0]) == 1513
    assert len(chain["proband_unique"][0]) == m["proband_unique_kmers"] == 630


def test_module3_goldens(oracle, chain):
    m = json.load(open(os.path.join(GOLDEN, "example_output_discovery", "giab_discovery.metrics.json")))
    lo, hi = chain["proband_unique"]
    total, unmapped, _ = oracle.module3_scan(os.path.join(GIAB, "HG002_child.bam"), lo, hi, 31,
                                             m["filters"]["min_distinct_kmers_per_read"])
    assert total == m["informative_reads"] == 195
    assert unmapped == m["unmapped_informative_reads"] == 11
