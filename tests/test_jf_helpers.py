"""A13: the sizing / chunk-discovery / merge helpers of
core/jellyfish_wrappers.py (reference jellyfish_wrappers.py:59-107,335-366) and the k-mer FASTA helpers kept
by name (reference utils.py:150-222).  Host logic runs everywhere; the merge goes through the engine."""
import os

import numpy as np
import pytest


def test_find_jf_files_lists_the_base_file_and_its_overflow_chunks(tmp_path):
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _find_jf_files
    base = str(tmp_path / "child.jf")
    assert _find_jf_files(base) == []
    for name in ("child.jf_1", "child.jf_0", "child.jf_10", "child.jfx", "other.jf_0"):
        (tmp_path / name).write_bytes(b"x")
    assert _find_jf_files(base) == [base + "_0", base + "_1", base + "_10"]          # chunks only, name order
    (tmp_path / "child.jf").write_bytes(b"x")
    assert _find_jf_files(base) == [base, base + "_0", base + "_1", base + "_10"]
    weird = str(tmp_path / "a[1].jf")                                                   # glob characters in the path
    open(weird + "_0", "wb").close()
    assert _find_jf_files(weird) == [weird + "_0"]


def test_hash_size_heuristic_and_log_formats(tmp_path, monkeypatch):
    from kmer_denovo_filter_amd.core import jellyfish_wrappers as W
    assert W._estimate_jf_hash_size(str(tmp_path / "missing.bam"), 31) == "1G"
    assert W._estimate_jf_hash_size(str(tmp_path / "missing.bam"), 31, default="7G") == "7G"
    for size, want in ((0, "100M"), (333_333_333, "100M"), (1_000_000_000, "300M"), (3_333_333_334, "1G"),
                       (10_000_000_000, "3G"), (50_000_000_000, "4G")):
        monkeypatch.setattr(os.path, "getsize", lambda p, _s=size: _s)
        assert W._estimate_jf_hash_size("x.bam", 31) == want, size
    for size, want in ((0, "0.0 B"), (1023, "1023.0 B"), (1024, "1.0 KB"), (1536, "1.5 KB"), (5 << 30, "5.0 GB"),
                       (3 << 50, "3.0 PB"), (1 << 62, "4096.0 PB")):
        monkeypatch.setattr(os.path, "getsize", lambda p, _s=size: _s)
        assert W._format_file_size("x") == want
    monkeypatch.undo()
    assert W._format_file_size(str(tmp_path / "missing")) == "?"
    for sec, want in ((0, "0.0s"), (59.94, "59.9s"), (60, "1m 0.0s"), (61.25, "1m 1.2s"), (3599.9, "59m 59.9s"),
                      (3600, "1h 0m 0s"), (3725.4, "1h 2m 5s")):
        assert W._format_elapsed(sec) == want
    assert W._parse_hash_size("3G") == 3_000_000_000 and W._parse_hash_size("250M") == 250_000_000 and W._parse_hash_size(77) == 77


def test_kmer_fasta_helpers_by_name(tmp_path):
    from kmer_denovo_filter_amd.utils import _estimate_fasta_sequence_count, _load_kmers_from_fasta, _write_kmer_fasta
    p = str(tmp_path / "k.fa")
    kmers = ["ACGTA", "TTTTT", "ACGTA", "GGGCC"]
    _write_kmer_fasta(iter(kmers), p)
    assert open(p).read() == ">0\nACGTA\n>1\nTTTTT\n>2\nACGTA\n>3\nGGGCC\n"
    assert _load_kmers_from_fasta(p) == {"ACGTA", "TTTTT", "GGGCC"}
    assert _estimate_fasta_sequence_count(p) == (4, False)                    # the file ends inside the sample: exact
    assert _estimate_fasta_sequence_count(p, sample_lines=8) == (4, True)     # exactly the sample: scaled by 1
    assert _estimate_fasta_sequence_count(p, sample_lines=4) == (4, True)     # half the file sampled, two records seen
    _write_kmer_fasta(("ACGTACGTAC" for _ in range(100_000)), p)             # more than one write block
    n, extrapolated = _estimate_fasta_sequence_count(p)
    assert extrapolated and 100_000 <= n < 115_000                            # longer record numbers further on: the estimate runs high, as the reference's does
    assert len(open(p).read().splitlines()) == 200_000
    open(p, "w").close()
    assert _estimate_fasta_sequence_count(p) == (0, False)
    assert _estimate_fasta_sequence_count(str(tmp_path / "missing.fa")) == (0, False)
    with pytest.raises(ValueError):
        _estimate_fasta_sequence_count(p, sample_lines=0)


@pytest.mark.gpu
def test_merge_jf_files_sums_the_chunks(tmp_path, oracle):
    """`jellyfish merge` (reference :335-366): counts of keys present in several chunks add up, keys of one chunk
    stay, the chunks are removed, a single file is returned as it is."""
    from kmer_denovo_filter_amd import jf_io
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _find_jf_files, _merge_jf_files
    k = 31
    rng = np.random.default_rng(1)
    keys = np.unique(rng.integers(0, 1 << 60, 5000, dtype=np.uint64))
    base = str(tmp_path / "child.jf")
    parts = []
    for i, (sel, add) in enumerate(((slice(0, 3000), 1), (slice(2000, 5000), 10), (slice(0, None, 7), 100))):
        lo = keys[sel]
        path = base if i == 0 else f"{base}_{i - 1}"
        jf_io.write_index(path, k, lo, np.zeros(len(lo), np.uint64), np.full(len(lo), add, np.uint32))
        parts.append((lo, add))
    files = _find_jf_files(base)
    assert len(files) == 3
    assert _merge_jf_files(files[:1], str(tmp_path / "unused.jf")) == files[0]
    assert _merge_jf_files([], str(tmp_path / "unused.jf")) is None
    merged = _merge_jf_files(files, str(tmp_path / "merged.jf"))
    kk, lo, hi, cnt = jf_io.read_index(merged, expect_k=k)
    exp = {}
    for plo, add in parts:
        for x in plo.tolist():
            exp[x] = exp.get(x, 0) + add
    assert kk == k and dict(zip(lo.tolist(), cnt.tolist())) == exp and not hi.any()
    assert np.all(lo[1:] > lo[:-1])                                           # the on-disk index is sorted
    assert [f for f in files if os.path.exists(f)] == []                      # the chunks are gone


def test_index_is_streamed_block_by_block(tmp_path, oracle):
    """`iter_index` (memory-mapped blocks) gives what the oracle's reader of the real Jellyfish fixture gives, for any
    block size; unusual key / counter widths go through the general decoder; `index_records` never reads the body."""
    import os
    from kmer_denovo_filter_amd import jf_io
    path = os.path.join(os.path.dirname(__file__), "golden", "giab", "mini_ref.fa.k31.jf")
    hdr, keys, counts = oracle.read_jf_binary_sorted(path)
    assert jf_io.index_records(path) == len(keys) == 45275
    for chunk in (1000, 45275, 1 << 20):
        los, cnts = [], []
        for k, lo, hi, c in jf_io.iter_index(path, expect_k=31, chunk_records=chunk):
            assert k == 31 and hi is None and len(lo) <= chunk
            los.append(lo); cnts.append(c)
        np.testing.assert_array_equal(np.concatenate(los), np.asarray(keys, dtype=np.uint64))
        np.testing.assert_array_equal(np.concatenate(cnts), np.asarray(counts, dtype=np.uint32))
    # wide keys + the package's own writer, read back in blocks of 7
    rng = np.random.default_rng(1)
    lo = np.sort(rng.integers(0, 1 << 62, 50, dtype=np.uint64)); hi = rng.integers(0, 1 << 20, 50, dtype=np.uint64)
    cnt = rng.integers(1, 1000, 50).astype(np.uint32)
    p2 = str(tmp_path / "w.jf")
    jf_io.write_index(p2, 47, lo, hi, cnt)
    got = list(jf_io.iter_index(p2, chunk_records=7))
    assert len(got) == 8 and all(g[0] == 47 for g in got)
    np.testing.assert_array_equal(np.concatenate([g[1] for g in got]), lo)
    np.testing.assert_array_equal(np.concatenate([g[2] for g in got]), hi)
    np.testing.assert_array_equal(np.concatenate([g[3] for g in got]), cnt)
    k, rlo, rhi, rcnt = jf_io.read_index(p2)
    assert k == 47 and np.array_equal(rlo, lo) and np.array_equal(rhi, hi) and np.array_equal(rcnt, cnt)


# ---- Jellyfish's own binary/sorted writer ----------------------------------------------------
_FIXTURE_JF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "giab", "mini_ref.fa.k31.jf")


def test_real_jellyfish_file_is_in_matrix_position_order():
    """The record order of the reference's real Jellyfish file (its `count -m 31 -s 100M -C` of mini_ref.fa): ascending
    hash position under the header's matrix1, equal positions in ascending key order -- the rule the writer follows."""
    from kmer_denovo_filter_amd import jf_io
    header, _ = jf_io.read_header(_FIXTURE_JF)
    k, lo, hi, cnt = jf_io.read_index(_FIXTURE_JF)
    assert (k, len(lo)) == (31, 45275)
    m = header["matrix1"]
    assert (m["c"], m["r"], 1 << m["r"]) == (62, 27, header["size"])
    assert header["reprobes"] == jf_io.jf_reprobes(header["max_reprobe"])
    pos = jf_io.jf_positions(m["columns"], 62, lo) & np.uint64(header["size"] - 1)
    d = np.diff(pos.astype(np.int64))
    assert (d >= 0).all()
    ties = np.flatnonzero(d == 0)
    assert len(ties) > 0 and (lo[ties] < lo[ties + 1]).all()
    assert jf_io._gf2_rank(m["columns"], m["r"]) == m["r"]                               # (Jellyfish's matrix has full row rank too)


def test_jellyfish_writer_gives_the_real_file_back_byte_for_byte(tmp_path):
    """The fixture's records, shuffled, written under the fixture's own header: the same file."""
    from kmer_denovo_filter_amd import jf_io
    header, off = jf_io.read_header(_FIXTURE_JF)
    k, lo, hi, cnt = jf_io.read_index(_FIXTURE_JF)
    perm = np.random.default_rng(1).permutation(len(lo))
    out = str(tmp_path / "again.jf")
    jf_io.write_jellyfish_index(out, 31, lo[perm], None, cnt[perm], header=header)
    want, got = open(_FIXTURE_JF, "rb").read(), open(out, "rb").read()
    _, off2 = jf_io.read_header(out)
    assert got[off2:] == want[off:]                                                     # every record, in Jellyfish's order
    h2, _ = jf_io.read_header(out)
    assert h2 == header                                                                 # (the JSON text may be spaced differently)


@pytest.mark.parametrize("k", [5, 31, 32, 47, 63])
def test_jellyfish_writer_with_its_own_matrix_round_trips(tmp_path, k):
    from kmer_denovo_filter_amd import jf_io
    rng = np.random.default_rng(k)
    n = 700 if k == 5 else 5000
    if k <= 32:
        lo = np.unique(rng.integers(0, 1 << min(2 * k, 63), n, dtype=np.uint64)); hi = None
    else:
        lo = rng.integers(0, 1 << 63, n, dtype=np.uint64); hi = rng.integers(0, 1 << (2 * k - 64), n, dtype=np.uint64)
    cnt = rng.integers(1, 1 << 32, len(lo), dtype=np.uint64).astype(np.uint32)
    out = str(tmp_path / "own.jf")
    jf_io.write_jellyfish_index(out, k, lo, hi, cnt, cmdline=["count", "-m", str(k), "-C"])
    header, off = jf_io.read_header(out)
    assert header["format"] == "binary/sorted" and header["canonical"] and off % 8 == 0
    m = header["matrix1"]
    assert m["c"] == 2 * k and (1 << m["r"]) == header["size"] >= 2 * len(lo)
    assert jf_io._gf2_rank(m["columns"], m["r"]) == min(m["r"], 2 * k)
    k2, rlo, rhi, rcnt = jf_io.read_index(out, expect_k=k)
    pos = jf_io.jf_positions(m["columns"], 2 * k, rlo, rhi) & np.uint64(header["size"] - 1)
    assert (np.diff(pos.astype(np.int64)) >= 0).all()
    key = lambda a, b: sorted(zip((b if b is not None else np.zeros(len(a), np.uint64)).tolist(), a.tolist()))
    o_w = np.lexsort((lo, hi)) if hi is not None else np.argsort(lo)
    o_r = np.lexsort((rlo, rhi))
    np.testing.assert_array_equal(rlo[o_r], lo[o_w]); np.testing.assert_array_equal(rcnt[o_r], cnt[o_w])
    if hi is not None:
        np.testing.assert_array_equal(rhi[o_r], hi[o_w])
    again = str(tmp_path / "own2.jf")                                                   # deterministic: same input, same bytes
    jf_io.write_jellyfish_index(again, k, lo, hi, cnt, cmdline=["count", "-m", str(k), "-C"])
    assert open(again, "rb").read() == open(out, "rb").read()
