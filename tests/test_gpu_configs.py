"""The BASELINE.json configurations at FULL size on the GPU.  The oracle cannot follow at these
sizes inside a test, so the checks are: the totals the 16-core oracle produced for the exact bench workload
(BENCH_r01.json cpu_baseline: it counted the whole batch), bit-for-bit agreement of the independent GPU paths
(direct global-table kernels / binned LDS-bucket pipeline / super-k-mer pipeline), conservation (sum of counts ==
valid windows), and a slice of the same reads against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# configs[1]: 10 M x 150 bp, k = 31, 100 Mbp uniform genome, seed 20260417 -- the workload bench.py times
BENCH_WINDOWS, BENCH_DISTINCT, BENCH_GE3 = 1_163_397_354, 266_204_130, 99_718_792


def _sorted_dump(e, min_count, n):
    """ascending (keys..., counts) of the table as device tensors"""
    import torch
    lo = torch.empty(n, dtype=torch.int64, device="cuda:0")
    hi = torch.empty(n, dtype=torch.int64, device="cuda:0") if e.wide else None
    cnt = torch.empty(n, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    got = e.export_ge_dev(min_count, lo.data_ptr(), hi.data_ptr() if hi is not None else None, cnt.data_ptr(), n, sorted_=True)
    assert got == n
    return lo, hi, cnt


def test_config1_exact_bench_workload_all_paths():
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.synth import synth_stream
    ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda:0", genome_seed=20260417)
    torch.cuda.synchronize()
    ref = None
    for path in (2, 3, 1):                                   # binned, super-k-mer, direct
        with KmerEngine(31, capacity_hint=1 << 28) as e:
            e.set_option("force_path", path)
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
            cap, distinct, windows = e.stats()
            assert (windows, distinct, e.count_ge(3)) == (BENCH_WINDOWS, BENCH_DISTINCT, BENCH_GE3), path
            lo, _, cnt = _sorted_dump(e, 0, distinct)
            assert int((cnt.to(torch.int64) & 0xFFFFFFFF).sum().item()) == windows     # conservation
            assert bool((lo[1:] > lo[:-1]).all())                                       # ascending (keys < 2^62), no duplicate
            if ref is None:
                ref = (lo, cnt)
            else:
                assert torch.equal(lo, ref[0]) and torch.equal(cnt, ref[1]), f"path {path} differs from the binned path"
            # the materialised dump -L 3 the bench step takes: same set whatever the path
            n3 = e.count_ge(3)
            l3, _, c3 = _sorted_dump(e, 3, n3)
            keep = (ref[1].to(torch.int64) & 0xFFFFFFFF) >= 3
            assert torch.equal(l3, ref[0][keep]) and torch.equal(c3, ref[1][keep])
            del lo, cnt, l3, c3


def test_config4_k63_full_size_and_oracle_slice(oracle):
    """k = 63 (128-bit keys) at 10 M x 150 bp: binned == direct bit for bit, conservation; the first 300 k reads
    against the oracle."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.synth import stream_to_ascii, synth_stream
    ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda:0", genome_seed=20260417)
    torch.cuda.synchronize()
    ref = None
    for path in (2, 1):
        with KmerEngine(63, capacity_hint=1 << 28) as e:
            e.set_option("force_path", path)
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
            cap, distinct, windows = e.stats()
            lo, hi, cnt = _sorted_dump(e, 0, distinct)
            assert int((cnt.to(torch.int64) & 0xFFFFFFFF).sum().item()) == windows
            if ref is None:
                ref = (lo, hi, cnt, windows)
            else:
                assert windows == ref[3] and torch.equal(lo, ref[0]) and torch.equal(hi, ref[1]) and torch.equal(cnt, ref[2])
    del ref
    n_slice = 300_032                                        # x 151 positions = a whole number of 64-position tiles: a prefix
    buf, offs = stream_to_ascii(ds, n_slice)                 # of the stream is then a valid stream of its own
    ot = oracle.OracleTable(63, 1 << 25).count_reads((buf, offs), threads=8)
    olo, ohi, ocnt = ot.export_ge(0)
    n_pos = n_slice * 151                                    # the stream holds one separator per read
    with KmerEngine(63, capacity_hint=1 << 24) as e:
        e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), n_pos)
        assert e.stats()[2] == oracle.count_windows((buf, offs), 63)
        glo, ghi, gcnt = e.export_ge(0)
    np.testing.assert_array_equal(glo, olo); np.testing.assert_array_equal(ghi, ohi); np.testing.assert_array_equal(gcnt, ocnt)


def test_config2_parent_filter_chain_full_64mbp():
    """configs[2] substitute at FULL size (64 Mbp, 30x trio): the chain's sets shrink monotonically, the planted
    SNVs survive, and the sieve path's per-key parent counts equal the binned path's and the direct path's."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
    import parent_filter
    from kmer_denovo_filter_amd import KmerEngine, devkeys
    res, (lo, hi), streams = parent_filter.run(64_000_000, 30, 31, 20260418, "cuda:0")
    assert res["candidates"] > res["non_ref"] > res["after_mother"] > res["after_father"] == res["proband_unique"] > 0
    # 64 000 planted SNVs x up to 31 k-mers each, seen >= 3 times at 30x: the bulk of the non-reference set
    assert 1_500_000 < res["proband_unique"] < 64_000 * 31
    assert res["stages"]["mother_count_if"]["path"] == "sieve"
    assert np.all(lo[1:] != lo[:-1])
    dlo, _ = devkeys.from_host(lo, None, False)
    mother = streams["mother"]
    counts = []
    for path in (4, 2, 1):                                   # sieve, binned, direct
        with KmerEngine(31, capacity_hint=len(lo)) as e:
            e.load_filter_dev(dlo.data_ptr(), None, len(lo))
            e.set_option("force_path", path)
            e.count_filtered_dev(mother.packed.data_ptr(), mother.invalid.data_ptr(), mother.n_bases)
            counts.append(devkeys.query(e, dlo, None))
            assert e.stats()[2] == res["stages"]["mother_count_if"]["windows"]
    assert all(torch.equal(counts[0], c) for c in counts[1:])
    assert int(counts[0].sum().item()) == 0                   # the survivors are absent from the mother by construction
