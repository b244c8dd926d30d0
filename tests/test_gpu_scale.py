"""Full-size properties (BASELINE.json configs[1] scale where the oracle cannot
follow): the two independent GPU paths (direct global-table kernels vs binned
LDS-bucket pipeline) must agree bit for bit, totals are conserved, and a
mid-size sample is checked against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev_stream(n_reads, seed=20260417):
    import torch
    from kmer_denovo_filter_amd.synth import synth_stream
    ds = synth_stream(n_reads, 150, 20_000_000, seed=seed, device="cuda:0")
    torch.cuda.synchronize()      # the engine launches on its own stream: the data must be complete
    return ds


def test_binned_vs_direct_vs_oracle_midsize(oracle):
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.synth import stream_to_ascii
    ds = _dev_stream(300_000)
    buf, offs = stream_to_ascii(ds, ds.n_reads)
    ot = oracle.OracleTable(31, 1 << 24).count_reads((buf, offs), threads=8)
    olo, ohi, ocnt = ot.export_ge(0)
    res = []
    for path in (1, 2):
        with KmerEngine(31, capacity_hint=1 << 23) as e:
            e.set_option("force_path", path)
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
            res.append(e.export_ge(0))
            assert e.stats()[2] == oracle.count_windows((buf, offs), 31)
    for lo, hi, cnt in res:
        np.testing.assert_array_equal(lo, olo)
        np.testing.assert_array_equal(cnt, ocnt)


def test_full_size_paths_agree_and_conserve():
    """10 M x 150 bp: sum of counts == valid windows; binned == direct."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    ds = _dev_stream(10_000_000)
    out = []
    for path in (2, 1):
        with KmerEngine(31, capacity_hint=1 << 27) as e:
            e.set_option("force_path", path)
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
            cap, distinct, windows = e.stats()
            lo, hi, cnt = e.export_ge(0)
            assert len(lo) == distinct
            assert int(cnt.astype(np.uint64).sum()) == windows
            assert (np.diff(lo.astype(np.uint64)) > 0).all()          # sorted, no duplicates
            out.append((lo, cnt, windows))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2]
