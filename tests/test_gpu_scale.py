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


def test_parent_filter_chain_synthetic_trio(oracle):
    """BASELINE configs[2] at reduced size (1 Mbp genome, 20x): the surviving
    proband-unique SET equals the oracle's discovery chain on the same reads."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
    import parent_filter
    from kmer_denovo_filter_amd.synth import stream_to_ascii
    res, (lo, hi), streams = parent_filter.run(1_000_000, 20, 31, 20260418, "cuda:0")
    asc = {}
    for name in ("child", "mother", "father"):
        asc[name] = stream_to_ascii(streams[name], streams[name].n_reads)
    gbuf, goffs = stream_to_ascii(streams["ref"], 1)
    rt = oracle.OracleTable(31, 1 << 21).count_reads((gbuf, goffs))
    exp = oracle.discovery_chain(asc["child"], asc["mother"], asc["father"], rt, 31, 3, 0)
    assert res["candidates"] == len(exp["candidates"][0])
    assert res["non_ref"] == len(exp["non_ref"][0])
    assert res["after_mother"] == len(exp["after_mother"][0])
    np.testing.assert_array_equal(np.sort(lo), exp["proband_unique"][0])
    assert res["proband_unique"] > 0                      # the planted SNVs are found


def test_owner_exchange_gpu_pieces_two_virtual_ranks(oracle):
    """Everything OwnerPartitionedCount does on the GPU except the RCCL call:
    two engines count two read shards, their pairs are dumped on the device,
    split by owner_of, routed in-process, and summed by two owner engines.  The
    union must equal the oracle's global count; ownership must be exclusive."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.distributed import EngineOps, ShardedFilterCount, owner_of
    from kmer_denovo_filter_amd.synth import stream_to_ascii, synth_genome, synth_stream
    dev = torch.device("cuda:0")
    world = 2
    g = synth_genome(2_000_000, 5, dev)
    shards = [synth_stream(40_000, 150, seed=100 + r, device=dev, genome=g) for r in range(world)]
    torch.cuda.synchronize()
    for k in (31, 45):
        local = [EngineOps(KmerEngine(k, capacity_hint=1 << 22), dev) for _ in range(world)]
        owner = [EngineOps(KmerEngine(k, capacity_hint=1 << 22), dev) for _ in range(world)]
        outbox = [[None] * world for _ in range(world)]
        for r in range(world):
            local[r].count_stream(shards[r].packed, shards[r].invalid, shards[r].n_bases)
            lo, hi, cnt = local[r].export_pairs(0)
            own = owner_of(lo, hi, world)
            for d in range(world):
                m = own == d
                outbox[r][d] = (lo[m], hi[m] if hi is not None else None, cnt[m])
        for d in range(world):
            for r in range(world):
                owner[d].add_pairs(*outbox[r][d])
        allreads = []
        for r in range(world):
            buf, offs = stream_to_ascii(shards[r], shards[r].n_reads)
            allreads.append((buf, offs))
        ot = oracle.OracleTable(k, 1 << 22)
        for br in allreads:
            ot.count_reads(br)
        elo, ehi, ecnt = ot.export_ge(0)
        got_lo, got_hi, got_cnt = [], [], []
        for d in range(world):
            lo, hi, cnt = owner[d].e.export_ge(0)
            t_lo = torch.from_numpy(lo.view(np.int64))
            t_hi = torch.from_numpy(hi.view(np.int64)) if k > 32 else None
            assert bool((owner_of(t_lo, t_hi, world) == d).all())
            got_lo.append(lo); got_hi.append(hi); got_cnt.append(cnt)
        lo = np.concatenate(got_lo); hi = np.concatenate(got_hi); cnt = np.concatenate(got_cnt)
        order = np.lexsort((lo, hi))
        np.testing.assert_array_equal(lo[order], elo)
        np.testing.assert_array_equal(hi[order], ehi)
        np.testing.assert_array_equal(cnt[order], ecnt)
        # count --if merge with a 1-rank group object: query path + clamp
        sfc = ShardedFilterCount(local[0])
        tl = torch.from_numpy(elo[:1000].view(np.int64).copy()).to(dev)
        th = torch.from_numpy(ehi[:1000].view(np.int64).copy()).to(dev) if k > 32 else None
        merged = sfc.merged_counts(tl, th).cpu().numpy()
        exp_local = oracle.OracleTable(k, 1 << 22).count_reads(allreads[0]).query(elo[:1000], ehi[:1000])
        np.testing.assert_array_equal(merged, exp_local.astype(np.int64))
        for o in local + owner:
            o.e.close()


def test_owner_ordered_dump_matches_owner_of():
    """kdf_export_parts_dev (the exchange's send side): every pair appears once, in the
    segment of the rank owner_of names, for power-of-two and odd world sizes, narrow and
    wide keys; a table too small for contiguous owner ranges is refused, not mis-grouped."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd._native import KdfError
    from kmer_denovo_filter_amd.distributed import EngineOps, owner_of
    from kmer_denovo_filter_amd.synth import synth_stream
    dev = torch.device("cuda:0")
    ds = synth_stream(300_000, 150, 3_000_000, seed=77, device=dev)
    torch.cuda.synchronize()
    for k in (31, 47):
        with KmerEngine(k, capacity_hint=1 << 27) as e:                 # 2^28 slots
            ops = EngineOps(e, dev)
            ops.count_stream(ds.packed, ds.invalid, ds.n_bases)
            ref_lo, ref_hi, ref_cnt = ops.export_pairs(0)
            key = lambda lo, hi, c: torch.stack([lo, hi if hi is not None else torch.zeros_like(lo), c.to(torch.int64)], 1)
            ref = key(ref_lo, ref_hi, ref_cnt)
            ref = ref[torch.argsort(ref[:, 0] * 31 + ref[:, 1])].cpu().numpy()
            ref = ref[np.lexsort((ref[:, 2], ref[:, 1], ref[:, 0]))]
            for world in (2, 3, 8):
                lo, hi, cnt, counts = ops.export_pairs_by_owner(world)
                assert sum(counts) == lo.numel() == ref_lo.numel() and len(counts) == world
                own = owner_of(lo, hi, world).cpu().numpy()
                exp = np.repeat(np.arange(world), counts)
                np.testing.assert_array_equal(own, exp)
                got = key(lo, hi, cnt).cpu().numpy()
                got = got[np.lexsort((got[:, 2], got[:, 1], got[:, 0]))]
                np.testing.assert_array_equal(got, ref)
    with KmerEngine(31, capacity_hint=1 << 20) as e:
        ops = EngineOps(e, dev)
        ops.count_stream(ds.packed, ds.invalid, 151 * 1000)
        assert ops.export_pairs_by_owner(2) is None                     # adapter falls back to the sort-based split
        lo = torch.empty(1 << 20, dtype=torch.int64, device=dev); cnt = torch.empty(1 << 20, dtype=torch.int32, device=dev)
        with pytest.raises(KdfError):
            e.export_parts_dev(0, 2, lo.data_ptr(), None, cnt.data_ptr(), 1 << 20)


def _two_rank_worker(rank, world, port, k, q):
    """One rank of test_owner_partitioned_count_two_processes (both share cuda:0; gloo, collectives on host copies)."""
    import torch
    import torch.distributed as dist
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.distributed import EngineOps, OwnerPartitionedCount
    from kmer_denovo_filter_amd.synth import synth_genome, synth_stream
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        g = synth_genome(3_000_000, 11, dev)
        shards = [synth_stream(150_000, 150, seed=500 + r, device=dev, genome=g) for r in range(world)]
        torch.cuda.synchronize()
        with KmerEngine(k, capacity_hint=1 << 27) as le, KmerEngine(k, capacity_hint=1 << 27) as oe:
            m = OwnerPartitionedCount(EngineOps(le, dev), dist.group.WORLD, dev, owner_ops=EngineOps(oe, dev),
                                      stage_through_host=True)
            m.clear()
            for _ in range(2):                                   # a streamed shard: two batches, one merge
                m.count_local(shards[rank].packed, shards[rank].invalid, shards[rank].n_bases)
            assert le.get_stat("log2cap") >= 28                   # the owner-ordered dump is the path taken
            g5 = m.merge(5)                                       # ONE merge closes the job
            assert oe.get_stat("hash_shift") == 1 and oe.get_stat("last_merge_path") == 1   # hash-ordered segments, LDS bucket merge
            got = [oe.count_ge(c) for c in (1, 2, 5)] + [g5]      # this rank's share + the global figure
            owned = oe.stats()[1]
        if rank == 0:                                            # the same job on one engine
            with KmerEngine(k, capacity_hint=1 << 27) as e:
                ops = EngineOps(e, dev)
                for _ in range(2):
                    for r in range(world):
                        ops.count_stream(shards[r].packed, shards[r].invalid, shards[r].n_bases)
                exp = [e.count_ge(c) for c in (1, 2, 5)]
            q.put(("ok", got, exp, owned))
        else:
            q.put(("ok", got, None, owned))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as ex:  # noqa: BLE001
        import traceback
        q.put(("err", f"rank {rank}: {ex}\n{traceback.format_exc()}", None, 0))


@pytest.mark.parametrize("k", [31, 47])
def test_owner_partitioned_count_two_processes(k):
    """bench.py's N > 1 job with real engines: two PROCESSES (sharing the one GPU of the test box, gloo with
    host-staged collectives standing in for RCCL) count their shards in two batches, merge once through the
    owner-ordered device dump, and agree with one engine that counted everything."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[0] == "ok", r[1]
    exp = next(r[2] for r in res if r[2] is not None)
    for i in range(3):
        assert sum(r[1][i] for r in res) == exp[i], (i, [r[1] for r in res], exp)
    assert all(r[1][3] == exp[2] for r in res)                                  # merge() returns the global count on every rank
    assert sum(r[3] for r in res) == exp[0] and all(r[3] > 0 for r in res)     # ownership is exclusive and shared out


def test_path_choice_follows_batch_and_table_size():
    """Kernel C rewrites every bucket of the table whatever the pending passes hold, so a FLUSH of little into a big table
    takes the direct kernels and a flush of much the binned pipeline (csrc/kdf_engine.hip use_binned: ~14 M positions per
    GB); without flushes in between the small batches ride along with the big one; the tables are the same."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    small, big = _dev_stream(100_000, seed=1), _dev_stream(2_000_000, seed=2)        # 15 M / 302 M positions
    dumps = {}
    for name, force in (("auto", 0), ("auto-deferred", 0), ("direct", 1), ("binned", 2)):
        with KmerEngine(31, capacity_hint=1 << 27) as e:                          # 2^28 slots = 3.2 GB: crossover ~46 M positions
            e.set_option("force_path", force)
            taken = []
            for ds in (small, big, small):
                e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
                if name != "auto-deferred":
                    e.flush()
                taken.append(e.last_count_path())
            if name == "auto":
                assert taken == ["direct", "binned", "direct"], taken
            if name == "auto-deferred":
                e.flush()
                assert e.get_stat("flushes") == 1 and e.get_stat("binned_passes") == 2 and e.last_count_path() == "binned"
            _, distinct, windows = e.stats()
            lo = torch.empty(distinct, dtype=torch.int64, device="cuda:0"); cnt = torch.empty(distinct, dtype=torch.int32, device="cuda:0")
            n = e.export_ge_dev(0, lo.data_ptr(), None, cnt.data_ptr(), distinct, sorted_=True); e.synchronize()
            dumps[name] = (lo[:n], cnt[:n], windows)
    for name in ("direct", "binned", "auto-deferred"):
        assert dumps[name][2] == dumps["auto"][2]
        assert torch.equal(dumps[name][0], dumps["auto"][0]) and torch.equal(dumps[name][1], dumps["auto"][1])


@pytest.mark.parametrize("bigb", [31, 64])
def test_widest_partition_geometry_binned_equals_direct(bigb):
    """A table of 2^32 slots takes the widest partition the pipeline has (1024 coarse bins, 512 fine bins, groups of
    1024 slabs -- one run per thread of the piece kernel, the case in which round 3's first strong-scaling job read past
    its run table); 300 k reads in two passes, deferred, against the direct path on a small table.  With 8192-slot
    buckets (bigb = 31; the default takes them from 2^32 slots on) that resolves the table; with 4096-slot buckets (bigb = 64) kernel C
    takes two sub-buckets per partition bucket."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    ds = _dev_stream(300_000, seed=9)
    dumps = []
    for hint, force in ((1 << 31, 2), (1 << 24, 1)):
        with KmerEngine(31, capacity_hint=hint) as e:
            e.set_option("big_bucket_log2cap", bigb)
            if force == 2:
                assert e.get_stat("bucket_bits") == (13 if bigb == 31 else 12)
            e.set_option("force_path", force)
            third = ds.n_bases // 3 // 64 * 64
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), third)
            e.flush()                                   # (the engine now knows the windows per position: groups of 1024 slabs from here on)
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), third)
            _, distinct, windows = e.stats()
            if force == 2:
                assert e.get_stat("log2cap") == 32 and e.get_stat("binned_passes") == 3 and e.get_stat("flushes") == 2
            lo = torch.empty(distinct, dtype=torch.int64, device="cuda:0"); cnt = torch.empty(distinct, dtype=torch.int32, device="cuda:0")
            n = e.export_ge_dev(0, lo.data_ptr(), None, cnt.data_ptr(), distinct, sorted_=True); e.synchronize()
            dumps.append((lo[:n], cnt[:n], windows))
    assert dumps[0][2] == dumps[1][2]
    assert torch.equal(dumps[0][0], dumps[1][0]) and torch.equal(dumps[0][1], dumps[1][1])


@pytest.mark.parametrize("k", [31, 63])
def test_big_buckets_binned_equals_direct_and_merge(k):
    """Tables from 2^big_bucket_log2cap slots on have buckets of twice the slots (8192 narrow / 4096 wide: one
    1024-thread workgroup of kernel C per CU).  Forced here on a mid-size table: the binned path into big buckets in two
    deferred passes, a growth in between (rehash into big buckets), the LDS bucket merge of its dump into a second
    big-bucket table, and the same reads through the direct path into ordinary buckets must give the same sorted dump."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine
    ds = _dev_stream(300_000, seed=21)
    wide = k > 32

    def dump(e):
        _, distinct, windows = e.stats()
        lo = torch.empty(distinct, dtype=torch.int64, device="cuda:0"); cnt = torch.empty(distinct, dtype=torch.int32, device="cuda:0")
        hi = torch.empty(distinct, dtype=torch.int64, device="cuda:0") if wide else None
        n = e.export_ge_dev(0, lo.data_ptr(), hi.data_ptr() if wide else None, cnt.data_ptr(), distinct, sorted_=True); e.synchronize()
        assert n == distinct
        return lo, hi, cnt, windows

    half = ds.n_bases // 2 // 64 * 64
    with KmerEngine(k, capacity_hint=1 << 22) as e:                 # too small for the sample: it grows under load
        e.set_option("big_bucket_log2cap", 12)
        assert e.get_stat("bucket_bits") == (13 if not wide else 12)
        e.set_option("force_path", 2)
        e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), half)
        e.flush()
        e.count_dev(ds.packed.data_ptr() + half // 4, ds.invalid.data_ptr() + half // 8, ds.n_bases - half)
        big = dump(e)
        assert e.get_stat("bucket_bits") == (13 if not wide else 12) and e.get_stat("binned_passes") >= 2
        # the dump in ascending order of the stored form (what the owner-ordered dump of a big table is): the LDS bucket merge
        # takes segments grouped by table bucket
        nd = big[0].numel()
        x = big[0].clone()
        if wide:
            x ^= (big[1] << 37) | ((big[1] >> 27) & ((1 << 37) - 1))
        hsh = (x ^ ((x >> 32) & 0xFFFFFFFF)) * (0x9FB21C651E98DF25 - (1 << 64))
        order = torch.argsort(hsh ^ (-(1 << 63)))               # unsigned order
        plo, pcnt = big[0][order].contiguous(), big[2][order].contiguous()
        phi = big[1][order].contiguous() if wide else None
        torch.cuda.synchronize()
        with KmerEngine(k, capacity_hint=1 << 12) as own:
            own.set_option("big_bucket_log2cap", 12); own.set_option("merge_min_pairs", 1)
            own.clear()
            own.add_pairs_multi_dev([(plo.data_ptr(), phi.data_ptr() if wide else None, pcnt.data_ptr(), nd)])
            own.synchronize()
            assert own.get_stat("last_merge_path") == 1 and own.get_stat("bucket_bits") == (13 if not wide else 12)
            merged = dump(own)
    with KmerEngine(k, capacity_hint=1 << 24) as d:
        d.set_option("force_path", 1)
        d.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), half)
        d.count_dev(ds.packed.data_ptr() + half // 4, ds.invalid.data_ptr() + half // 8, ds.n_bases - half)
        ref = dump(d)
    for got in (big, merged):
        assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])
        if wide:
            assert torch.equal(got[1], ref[1])
    assert big[3] == ref[3]
