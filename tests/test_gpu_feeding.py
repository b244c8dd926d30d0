"""The streamed-sample feeder (`samtools fasta | jellyfish count`, reference core/jellyfish_wrappers.py:166-199) as three
overlapping stages -- reader thread into pinned batches, copy stream, count -- must count exactly what the plain
batch-by-batch loop counts, for count -C and count --if, over many small batches (every staging slot is reused)."""
import numpy as np
import pytest

from helpers import write_bam

pytestmark = pytest.mark.gpu


def _bam(tmp_path, n=6000, seed=3):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, 200_000)
    B = np.frombuffer(b"ACGTN", np.uint8)
    reads = []
    for i, s in enumerate(np.sort(rng.integers(0, len(genome) - 150, n))):
        L = int(rng.integers(20, 151))
        codes = genome[s:s + L].copy()
        if i % 17 == 0:
            codes[rng.integers(0, L)] = 4                       # an N
        reads.append({"name": f"r{i}", "seq": B[codes].tobytes().decode(), "pos": int(s),
                      "flag": 0x900 if i % 29 == 0 else (0x41 if i & 1 else 0x81)})
    path = str(tmp_path / "feed.bam")
    write_bam(path, [("chr1", len(genome))], reads)
    return path


@pytest.mark.parametrize("k", [21, 47])
def test_overlapped_feeding_equals_the_plain_loop(tmp_path, k):
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.reads import bam_reader, stream_batches_overlapped
    path = _bam(tmp_path)
    with KmerEngine(k, capacity_hint=1 << 20) as a, KmerEngine(k, capacity_hint=1 << 20) as b:
        n_a = 0
        with bam_reader(path, max_bases=1 << 16, max_reads=1 << 10, threads=2) as rd:
            for batch in rd:
                a.count(batch)
                n_a += batch.n_reads
        with bam_reader(path, max_bases=1 << 16, max_reads=1 << 10, threads=2) as rd:
            n_b = stream_batches_overlapped(b, rd, filtered=False)
        assert n_a == n_b > 0
        assert a.stats()[1:] == b.stats()[1:]
        ra, rb = a.export_ge(0), b.export_ge(0)
        for x, y in zip(ra, rb):
            np.testing.assert_array_equal(x, y)
        # count --if through the same pipeline: the filter is every 3rd key
        lo, hi, cnt = ra
        flo, fhi = lo[::3].copy(), hi[::3].copy()
        with KmerEngine(k, capacity_hint=len(flo)) as f:
            f.load_filter(flo, fhi)
            with bam_reader(path, max_bases=1 << 16, max_reads=1 << 10, threads=2) as rd:
                stream_batches_overlapped(f, rd, filtered=True)
            np.testing.assert_array_equal(f.query(flo, fhi), cnt[::3])


def test_stream_bam_mirror_uses_the_pipeline(tmp_path):
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.core.jellyfish_wrappers import _stream_bam
    from kmer_denovo_filter_amd.reads import bam_reader
    path = _bam(tmp_path, n=2000)
    with KmerEngine(31, capacity_hint=1 << 20) as a, KmerEngine(31, capacity_hint=1 << 20) as b:
        with bam_reader(path, threads=1) as rd:
            for batch in rd:
                a.count(batch)
        n = _stream_bam(b, path, None, 2, filtered=False)
        assert n > 0 and a.stats()[1:] == b.stats()[1:]
        for x, y in zip(a.export_ge(0), b.export_ge(0)):
            np.testing.assert_array_equal(x, y)
