"""The two ends of the multi-GPU merge on ONE GPU (csrc/kdf_merge.h): the owner-ordered dump of a counted table and
the owners' LDS bucket merge of what `world` source ranks would send.  Reference: `jellyfish merge` of partial indexes,
kmer_denovo_filter/core/jellyfish_wrappers.py:335-366 -- the merged index holds every key once with the SUM of its
counts, which is what is checked here (against the engine's own unordered dump, itself oracle-checked elsewhere)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _counted_engine(k, n_reads=400_000, seed=5):
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.synth import synth_stream
    ds = synth_stream(n_reads, 150, 5_000_000, seed=seed, device="cuda:0")
    torch.cuda.synchronize()
    e = KmerEngine(k, capacity_hint=1 << 27)              # 2^28 slots: the smallest table the owner-ordered dump takes
    e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
    e.synchronize()
    return e


def _hash(lo, hi):
    """csrc/kdf_device.h kdf_hash on uint64 arrays."""
    x = lo.copy()
    if hi is not None:
        x ^= (hi << np.uint64(37)) | (hi >> np.uint64(27))
    with np.errstate(over="ignore"):
        return (x ^ (x >> np.uint64(32))) * np.uint64(0x9FB21C651E98DF25)


def _dump_parts(e, world):
    import torch
    from kmer_denovo_filter_amd.distributed import EngineOps
    ops = EngineOps(e, torch.device("cuda:0"))
    got = ops.export_pairs_by_owner(world)
    assert got is not None
    return got


@pytest.mark.parametrize("k", [31, 63])
def test_owner_ordered_dump_is_the_table_in_hash_order(k):
    import torch
    e = _counted_engine(k)
    try:
        world = 8
        lo, hi, cnt, counts = _dump_parts(e, world)
        assert sum(counts) == lo.numel() == e.stats()[1]
        rlo, rhi, rcnt = e.export_ge(0)                     # the sorted host dump: same multiset
        hl = lo.cpu().numpy().view(np.uint64)
        hh = hi.cpu().numpy().view(np.uint64) if hi is not None else None
        hc = cnt.cpu().numpy().view(np.uint32)
        order = np.lexsort((hl, hh)) if hh is not None else np.argsort(hl, kind="stable")
        np.testing.assert_array_equal(hl[order], rlo)
        if hh is not None:
            np.testing.assert_array_equal(hh[order], rhi)
        np.testing.assert_array_equal(hc[order], rcnt)
        h = _hash(hl, hh)
        own = ((h >> np.uint64(48)) * np.uint64(world)) >> np.uint64(16)
        np.testing.assert_array_equal(own, np.repeat(np.arange(world, dtype=np.uint64), counts))
        pre = h >> np.uint64(64 - (e.get_stat("log2cap") - 6))     # grouped by the top log2cap - 6 hash bits, ascending
        assert np.all(pre[1:] >= pre[:-1])
    finally:
        e.close()


@pytest.mark.parametrize("k,world", [(31, 8), (63, 4), (31, 3)])
def test_owner_bucket_merge_sums_the_segments(k, world):
    """Every owner receives its part from `world` sources (here: the same part, so counts must come out x world),
    first into a table that was only `clear`ed (the merge is also the deferred clear), then into the live table."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.distributed import EngineOps
    dev = torch.device("cuda:0")
    e = _counted_engine(k)
    try:
        lo, hi, cnt, counts = _dump_parts(e, world)
        a = 0
        total_distinct = 0
        for r, n in enumerate(counts):
            plo, pcnt = lo[a:a + n], cnt[a:a + n]
            phi = hi[a:a + n] if hi is not None else None
            a += n
            with KmerEngine(k, capacity_hint=1 << 16) as own:          # grows to fit the first merge
                ops = EngineOps(own, dev)
                ops.prepare_owner(world)
                assert own.get_stat("hash_shift") == world.bit_length() - 1
                own.clear()
                ops.add_pairs_segments([(plo, phi, pcnt)] * world)
                if world & (world - 1) == 0:
                    assert own.get_stat("last_merge_path") == 1, "a hash-ordered dump must take the LDS bucket merge"
                assert own.stats()[1] == n
                q = ops.query(plo, phi)
                assert torch.equal(q, pcnt * world)
                # live table, one more source, in an order that is NOT grouped: the device-side test must notice
                perm = torch.randperm(n, device=dev)
                ops.add_pairs_segments([(plo[perm], phi[perm] if phi is not None else None, pcnt[perm])])
                assert own.get_stat("last_merge_path") == 2
                ops.add_pairs_segments([(plo, phi, pcnt)])              # grouped again, into a table that holds keys
                assert own.stats()[1] == n
                assert torch.equal(ops.query(plo, phi), pcnt * (world + 2))
                # nothing else got in: the owner's dump is exactly its part
                olo, ohi_, ocnt = own.export_ge(0)
                assert len(olo) == n
                total_distinct += n
        assert total_distinct == e.stats()[1]
    finally:
        e.close()


def test_owner_table_uses_its_whole_length():
    """Without hash_shift an owner of 8 would crowd its keys into an eighth of the table and overflow its buckets at
    a nominal load of 0.5; with it the same keys fit."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd._native import KdfError
    from kmer_denovo_filter_amd.distributed import EngineOps
    dev = torch.device("cuda:0")
    e = _counted_engine(31)
    try:
        lo, hi, cnt, counts = _dump_parts(e, 8)
        plo, pcnt = lo[:counts[0]:8].contiguous(), cnt[:counts[0]:8].contiguous()     # every 8th pair of owner 0: still in hash order
        n = plo.numel()
        assert n >= 1 << 16
        with KmerEngine(31, capacity_hint=n) as own:                    # load <= 0.5 for n keys
            ops = EngineOps(own, dev)
            ops.prepare_owner(8)
            ops.add_pairs_segments([(plo, None, pcnt)])
            assert own.stats()[1] == n and own.stats()[0] < 4 * n
        with KmerEngine(31, capacity_hint=n) as crowded:
            with pytest.raises(KdfError):
                EngineOps(crowded, dev).add_pairs_segments([(plo, None, pcnt)])
    finally:
        e.close()


def test_hash_shift_only_on_an_empty_table():
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd._native import KdfError
    e = _counted_engine(31, n_reads=20_000)
    try:
        with pytest.raises(KdfError):
            e.set_option("hash_shift", 2)
        e.clear()
        e.set_option("hash_shift", 2)
        assert e.get_stat("hash_shift") == 2
    finally:
        e.close()


@pytest.mark.parametrize("k", [31, 63])
def test_packed_dump_is_the_send_buffer(k):
    """kdf_export_parts_packed_dev writes what the exchange would otherwise assemble with copies: per owner one byte
    segment [lo | hi | counts], starts on multiples of 8."""
    import torch
    from kmer_denovo_filter_amd.distributed import EngineOps
    e = _counted_engine(k)
    try:
        ops = EngineOps(e, torch.device("cuda:0"))
        world = 8
        lo, hi, cnt, counts = ops.export_pairs_by_owner(world)
        buf, pcounts, offs = ops.export_packed_by_owner(world)
        assert pcounts == counts and offs[0] == 0 and buf.numel() == offs[world]
        esz = 20 if k > 32 else 12
        a = 0
        for p, n in enumerate(counts):
            assert offs[p] % 8 == 0 and offs[p + 1] - offs[p] == (n * esz + 7) // 8 * 8
            seg = buf[offs[p]:offs[p + 1]]
            slo = seg[:8 * n].view(torch.int64)
            o = 8 * n
            # (inside a 4096-slot block the order of equal sub-bins is not fixed between two dumps: compare as sorted triples)
            if k > 32:
                shi = seg[o:o + 8 * n].view(torch.int64); o += 8 * n
            scnt = seg[o:o + 4 * n].view(torch.int32)
            def ordered(l, h, c):
                o = torch.argsort(l, stable=True)
                if h is not None:
                    o = o[torch.argsort(h[o], stable=True)]
                return l[o], (h[o] if h is not None else None), c[o]
            rl, rh, rc = ordered(lo[a:a + n], hi[a:a + n] if k > 32 else None, cnt[a:a + n])
            gl, gh, gc = ordered(slo, shi if k > 32 else None, scnt)
            assert torch.equal(rl, gl) and torch.equal(rc, gc) and (k <= 32 or torch.equal(rh, gh))
            a += n
    finally:
        e.close()
