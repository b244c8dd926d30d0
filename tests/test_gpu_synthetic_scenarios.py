"""The reference's synthetic k=5 discovery scenarios
(tests/discovery/test_pipeline.py:39-163,228-300,717-856) replayed on the
drop-in mirrors: Modules 1-2 (count, ref subtraction, parent filter) and the
Module-3 probe.  Everything the reference asserts about the k-mer stages is
asserted here; clustering/BED output is outside the hot path."""
import os

import numpy as np
import pytest

from helpers import make_ref_fasta, write_bam

pytestmark = pytest.mark.gpu
K = 5


def _bam(tmp, name, seqs, pos=30):
    path = os.path.join(tmp, name)
    write_bam(path, [("chr1", 300)], [{"name": f"{name}_{i}", "seq": s, "pos": pos} for i, s in enumerate(seqs)])
    return path


def _run(tmp, child, mother, father, ref_fa, min_child_count=3, parent_max_count=0, ref_jf=None):
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _build_proband_jf_index, _ensure_ref_jf
    from kmer_denovo_filter_amd.discovery.pipeline import (
        _extract_child_kmers_discovery, _filter_parents_discovery, _subtract_reference_kmers)
    ref_jf = _ensure_ref_jf(ref_fa, K, 4, ref_jf)
    fa, n_cand = _extract_child_kmers_discovery(child, ref_fa, K, min_child_count, 4, tmp)
    fa2, n_nonref = _subtract_reference_kmers(ref_jf, fa, tmp)
    n_pu, fa3 = _filter_parents_discovery(mother, father, ref_fa, fa2, K, 4, tmp, parent_max_count)
    return n_cand, n_nonref, n_pu, fa3


@pytest.fixture()
def scene(tmp_path):
    tmp = str(tmp_path)
    ref_fa = os.path.join(tmp, "ref.fa")
    ref = make_ref_fasta(ref_fa, "chr1", 200)
    mut = list(ref[30:90])
    mut[20] = "G" if ref[50] != "G" else "T"
    return tmp, ref_fa, ref, "".join(mut)


def test_denovo_detected(scene, oracle):
    from kmer_denovo_filter_amd.core import bam_scanner
    from kmer_denovo_filter_amd.kmer_fasta import read_kmer_fasta_keys
    tmp, ref_fa, ref, mut = scene
    child = _bam(tmp, "child.bam", [mut] * 4)
    mother = _bam(tmp, "mother.bam", [ref[30:90]] * 3)
    father = _bam(tmp, "father.bam", [ref[30:90]] * 3)
    n_cand, n_nonref, n_pu, fa = _run(tmp, child, mother, father, ref_fa)
    assert n_cand > 0 and 0 < n_pu <= n_nonref <= n_cand
    # the survivors are exactly the windows covering the mutated base that are absent from ref and parents
    exp = oracle.discovery_chain([mut] * 4, [ref[30:90]] * 3, [ref[30:90]] * 3,
                                 oracle.OracleTable(K).count_reads([ref]), K, 3, 0)
    got = np.sort(read_kmer_fasta_keys(fa, K)[0])
    np.testing.assert_array_equal(got, exp["proband_unique"][0])
    # Module 3: every child read carries the k-mers -> informative (reference asserts informative_reads > 0)
    bam_scanner._init_scan_worker(fa, K, 1)
    total, unmapped, scanned, kept = bam_scanner.count_informative_reads(child)
    assert scanned == 4 and total == 4 and unmapped == 0
    for inf in kept:
        assert inf.n_distinct == n_pu or inf.n_distinct >= 1
        assert all(15 < p < 25 for p in inf.kmer_hit_indices)      # windows around query position 20


def test_inherited_variant_yields_nothing(scene):
    tmp, ref_fa, ref, mut = scene
    child = _bam(tmp, "child.bam", [mut] * 4)
    mother = _bam(tmp, "mother.bam", [mut] * 3)                      # mother carries the variant
    father = _bam(tmp, "father.bam", [ref[30:90]] * 3)
    n_cand, n_nonref, n_pu, fa = _run(tmp, child, mother, father, ref_fa)
    assert n_nonref > 0 and n_pu == 0 and fa is None                # reference: proband_unique_kmers == 0


def test_parent_max_count_zero_vs_one(scene):
    tmp, ref_fa, ref, mut = scene
    child = _bam(tmp, "child.bam", [mut] * 4)
    mother = _bam(tmp, "mother.bam", [ref[30:90]] * 3 + [mut])       # ONE parental read with the variant
    father = _bam(tmp, "father.bam", [ref[30:90]] * 3)
    assert _run(tmp, child, mother, father, ref_fa, parent_max_count=0)[2] == 0
    os.makedirs(os.path.join(tmp, "b"), exist_ok=True)
    n = _run(os.path.join(tmp, "b"), child, mother, father, ref_fa, parent_max_count=1)[2]
    assert n > 0                                                      # count 1 <= parent_max_count 1 survives


def test_min_child_count_is_inclusive(scene, oracle):
    tmp, ref_fa, ref, mut = scene
    child = _bam(tmp, "child.bam", [mut] * 3)
    mother = _bam(tmp, "mother.bam", [ref[30:90]])
    father = _bam(tmp, "father.bam", [ref[30:90]])
    rt = oracle.OracleTable(K).count_reads([ref])
    for sub, mcc in (("a", 3), ("b", 4)):
        os.makedirs(os.path.join(tmp, sub), exist_ok=True)
        got = _run(os.path.join(tmp, sub), child, mother, father, ref_fa, min_child_count=mcc)
        exp = oracle.discovery_chain([mut] * 3, [ref[30:90]], [ref[30:90]], rt, K, mcc, 0)
        assert got[:3] == (len(exp["candidates"][0]), len(exp["non_ref"][0]), len(exp["proband_unique"][0]))
        if mcc == 3:
            assert got[2] > 0            # count 3 >= 3 survives: -L is inclusive
        else:
            assert got[2] == 0           # the variant k-mers occur exactly 3 times


def test_empty_child(scene):
    tmp, ref_fa, ref, mut = scene
    child = _bam(tmp, "child.bam", [])
    mother = _bam(tmp, "mother.bam", [ref[30:90]])
    father = _bam(tmp, "father.bam", [ref[30:90]])
    assert _run(tmp, child, mother, father, ref_fa) == (0, 0, 0, None)


def test_prebuilt_ref_jf_is_reused(scene):
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _ensure_ref_jf
    tmp, ref_fa, ref, mut = scene
    custom = os.path.join(tmp, "custom_ref.jf")
    assert _ensure_ref_jf(ref_fa, K, 2, custom) == custom and os.path.isfile(custom)
    mtime = os.path.getmtime(custom)
    assert _ensure_ref_jf(ref_fa, K, 2, custom) == custom and os.path.getmtime(custom) == mtime
    assert not os.path.exists(ref_fa + f".k{K}.jf")                  # default path untouched


def test_automaton_and_utils_mirrors(scene, oracle):
    from kmer_denovo_filter_amd.kmer_utils import build_kmer_automaton, canonicalize, reverse_complement
    from kmer_denovo_filter_amd.utils import (_estimate_fasta_sequence_count, _load_kmers_from_fasta,
                                              _write_kmer_fasta)
    assert build_kmer_automaton([]) is None
    A = build_kmer_automaton(["AAAAC", "ACGTA"])
    assert len(A) == 4
    hits = list(A.iter("TTGTTTTACGTAGG"))
    # windows: GTTTT (= rc AAAAC) at 2 -> end 6; ACGTA at 7 -> end 11; TACGT is rc(ACGTA) at 6 -> end 10
    assert sorted(hits) == sorted([(6, "AAAAC"), (10, "ACGTA"), (11, "ACGTA")])
    assert canonicalize(reverse_complement("AAAAC")) == "AAAAC"
    tmp = scene[0]
    p = os.path.join(tmp, "k.fa")
    _write_kmer_fasta(["AAAAC", "ACGTA"], p)
    assert open(p).read() == ">0\nAAAAC\n>1\nACGTA\n"
    assert _load_kmers_from_fasta(p) == {"AAAAC", "ACGTA"}
    assert _estimate_fasta_sequence_count(p) == (2, False)
