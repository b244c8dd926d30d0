"""GPU parity (through the C ABI) against the oracle: count, count --if, query,
dump, scan -- narrow (k<=32) and wide (33<=k<=63) keys, ragged/empty inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rand_reads(rng, n, lo, hi, n_frac=0.01, genome=None):
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        if genome is not None and L <= len(genome):
            s = int(rng.integers(0, len(genome) - L + 1))
            a = genome[s:s + L].copy()
        else:
            a = rng.integers(0, 4, L).astype(np.uint8)
        chars = np.frombuffer(b"ACGT", np.uint8)[a].copy()
        nmask = rng.random(L) < n_frac
        chars[nmask] = ord("N")
        s = chars.tobytes().decode()
        if rng.random() < 0.3:
            s = s.lower()
        out.append(s)
    return out


def oracle_sorted(O, k, reads):
    t = O.OracleTable(k, 1 << 12).count_reads(reads)
    return t, t.export_ge(0)


@pytest.mark.parametrize("k", [5, 15, 31, 32, 33, 47, 63])
def test_count_export_matches_oracle(oracle, k):
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(100 + k)
    genome = rng.integers(0, 4, 20000).astype(np.uint8)
    reads = rand_reads(rng, 400, 0, 300, genome=genome) + ["", "A", "N" * 70, "ACGT" * 40]
    t, (lo, hi, cnt) = oracle_sorted(oracle, k, reads)
    with KmerEngine(k, capacity_hint=1024) as e:      # small hint: exercises grow/rehash
        e.count(ReadStream.from_strings(reads))
        glo, ghi, gcnt = e.export_ge(0)
        cap, distinct, windows = e.stats()
        assert windows == oracle.count_windows(reads, k)
        assert distinct == len(lo)
        np.testing.assert_array_equal(glo, lo)
        np.testing.assert_array_equal(ghi, hi)
        np.testing.assert_array_equal(gcnt, cnt)
        # dump -L 2
        lo2, hi2, c2 = t.export_ge(2)
        g2 = e.export_ge(2)
        np.testing.assert_array_equal(g2[0], lo2)
        np.testing.assert_array_equal(g2[1], hi2)
        np.testing.assert_array_equal(g2[2], c2)
        # query: present + absent keys, input order
        q_lo = np.concatenate([lo[::-1][:100], rng.integers(0, 1 << 62, 50, dtype=np.uint64)])
        q_hi = np.concatenate([hi[::-1][:100], np.zeros(50, np.uint64)])
        np.testing.assert_array_equal(e.query(q_lo, q_hi), t.query(q_lo, q_hi))


@pytest.mark.parametrize("k", [7, 31, 41, 63])
def test_count_filtered_and_scan(oracle, k):
    from kmer_denovo_filter_amd import KmerEngine, ReadStream, hit_positions
    rng = np.random.default_rng(7 + k)
    genome = rng.integers(0, 4, 8000).astype(np.uint8)
    child = rand_reads(rng, 200, 20, 260, genome=genome)
    parent = rand_reads(rng, 300, 0, 260, genome=genome)
    _, (lo, hi, cnt) = oracle_sorted(oracle, k, child)
    sel = rng.random(len(lo)) < 0.5
    flo, fhi = lo[sel], hi[sel]
    ot = oracle.OracleTable(k, 1 << 12).load_filter(flo, fhi).count_reads_filtered(parent)
    with KmerEngine(k) as e:
        e.load_filter(flo, fhi)
        st = ReadStream.from_strings(parent)
        e.count_filtered(st)
        np.testing.assert_array_equal(e.query(lo, hi), ot.query(lo, hi))
        glo, ghi, gcnt = e.export_ge(0)      # filter keys never seen report 0
        olo, ohi, ocnt = ot.export_ge(0)
        np.testing.assert_array_equal(glo, olo)
        np.testing.assert_array_equal(gcnt, ocnt)
        # scan: hit iff stored with count > 0
        hits, distinct = e.scan(st)
        ohit, odist = ot.scan_reads(parent)
        np.testing.assert_array_equal(distinct, odist)
        buf_off = 0
        for r, read in enumerate(parent):
            s, eoff = int(st.offsets[r]), int(st.offsets[r]) + len(read)
            got = hit_positions(hits, s, eoff)
            exp = np.nonzero(ohit[buf_off:buf_off + len(read)])[0]
            np.testing.assert_array_equal(got, exp)
            buf_off += len(read)


def test_empty_inputs(oracle):
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    with KmerEngine(31) as e:
        e.count(ReadStream.from_strings([]))
        assert e.stats()[1:] == (0, 0)
        assert e.count_ge(0) == 0
        lo, hi, c = e.export_ge(1)
        assert len(lo) == 0
        assert len(e.query(np.zeros(0, np.uint64))) == 0
        e.load_filter(np.zeros(0, np.uint64))
        e.count_filtered(ReadStream.from_strings(["ACGT" * 20]))
        assert e.count_ge(0) == 0


def test_errors_are_loud():
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    from kmer_denovo_filter_amd._native import KdfError
    with pytest.raises(ValueError):
        KmerEngine(64)
    with KmerEngine(31) as e:
        with pytest.raises(KdfError):
            e.count_filtered(ReadStream.from_strings(["ACGT" * 20]))   # no filter loaded
        e.load_filter(np.array([5], np.uint64))
        with pytest.raises(KdfError):
            e.count(ReadStream.from_strings(["ACGT" * 20]))            # filter mode: insert refused


# --------------------------------------------------------------------------
# binned (LDS-bucket) pipeline, forced on small inputs
# --------------------------------------------------------------------------

@pytest.mark.parametrize("k,hint", [(31, 1 << 12), (31, 1 << 17), (21, 1 << 20), (47, 1 << 13), (63, 1 << 17)])
def test_binned_count_matches_oracle(oracle, k, hint):
    """force_path=2: every count call goes through A0/A1/B/C.  Small hints make
    buckets overflow, which exercises the transactional failure + replay."""
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(900 + k + hint)
    genome = rng.integers(0, 4, 60000).astype(np.uint8)
    reads = rand_reads(rng, 3000, 0, 300, genome=genome) + ["", "A" * 500, "N" * 70, "ACGT" * 80]
    t, (lo, hi, cnt) = oracle_sorted(oracle, k, reads)
    with KmerEngine(k, capacity_hint=hint) as e:
        e.set_option("force_path", 2)
        st = ReadStream.from_strings(reads)
        half = len(reads) // 2
        # two batches: the second one finds a non-empty table
        e.count(ReadStream.from_strings(reads[:half]))
        e.count(ReadStream.from_strings(reads[half:]))
        assert e.get_stat("binned_passes") == 2
        glo, ghi, gcnt = e.export_ge(0)
        cap, distinct, windows = e.stats()
        assert windows == oracle.count_windows(reads, k)
        assert distinct == len(lo)
        np.testing.assert_array_equal(glo, lo)
        np.testing.assert_array_equal(ghi, hi)
        np.testing.assert_array_equal(gcnt, cnt)
        if hint <= 1 << 13:
            assert e.get_stat("replayed_buckets") > 0     # the small table really overflowed
        np.testing.assert_array_equal(e.query(lo[::7], hi[::7]), cnt[::7])


@pytest.mark.parametrize("k", [31, 55])
def test_binned_count_filtered(oracle, k):
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(77 + k)
    genome = rng.integers(0, 4, 50000).astype(np.uint8)
    child = rand_reads(rng, 1500, 50, 260, genome=genome)
    parent = rand_reads(rng, 3000, 0, 260, genome=genome)
    _, (lo, hi, _) = oracle_sorted(oracle, k, child)
    sel = rng.random(len(lo)) < 0.4
    flo, fhi = lo[sel], hi[sel]
    ot = oracle.OracleTable(k, 1 << 12).load_filter(flo, fhi).count_reads_filtered(parent)
    with KmerEngine(k) as e:
        e.load_filter(flo, fhi)
        e.set_option("force_path", 2)
        e.count_filtered(ReadStream.from_strings(parent[:1000]))
        e.count_filtered(ReadStream.from_strings(parent[1000:]))
        assert e.get_stat("binned_passes") == 2
        np.testing.assert_array_equal(e.query(lo, hi), ot.query(lo, hi))
        glo, _, gcnt = e.export_ge(0)
        olo, _, ocnt = ot.export_ge(0)
        np.testing.assert_array_equal(glo, olo)
        np.testing.assert_array_equal(gcnt, ocnt)


def test_binned_skewed_multiplicity(oracle):
    """Heavy hitters (poly-A, short tandem repeats) land in one bucket: chunks
    and buckets must cope with runs far above the mean."""
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(5)
    reads = ["A" * 300] * 400 + ["ACACACACAC" * 30] * 300 + rand_reads(rng, 500, 100, 200)
    t, (lo, hi, cnt) = oracle_sorted(oracle, 31, reads)
    with KmerEngine(31, capacity_hint=1 << 16) as e:
        e.set_option("force_path", 2)
        e.count(ReadStream.from_strings(reads))
        glo, _, gcnt = e.export_ge(0)
        np.testing.assert_array_equal(glo, lo)
        np.testing.assert_array_equal(gcnt, cnt)
        assert int(gcnt.max()) == 400 * 270


@pytest.mark.parametrize("k", [31, 45])
def test_binned_stream_walked_in_several_passes(oracle, k):
    """A stream longer than a binned pass may cover (2^31 positions by default; 8192 here) is
    walked pass by pass, each starting on a tile boundary: windows that start in one pass and
    end in the next are counted exactly once, insert mode and count --if."""
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(12)
    genome = rng.integers(0, 4, 30_000).astype(np.uint8)
    reads = rand_reads(rng, 600, 60, 260, genome=genome)
    st = ReadStream.from_strings(reads)
    t, (lo, hi, cnt) = oracle_sorted(oracle, k, reads)
    with KmerEngine(k, capacity_hint=1 << 15) as e:
        e.set_option("force_path", 2); e.set_option("binned_max_positions", 8192)
        e.count(st)
        assert e.get_stat("binned_passes") == -(-st.n_bases // 8192) > 5
        glo, ghi, gcnt = e.export_ge(0)
        np.testing.assert_array_equal(glo, lo); np.testing.assert_array_equal(ghi, hi); np.testing.assert_array_equal(gcnt, cnt)
    sel = slice(0, len(lo), 3)
    ot = oracle.OracleTable(k, 1 << 12).load_filter(lo[sel], hi[sel]).count_reads_filtered(reads)
    with KmerEngine(k, capacity_hint=1 << 15) as e:
        e.load_filter(lo[sel], hi[sel] if k > 32 else None)
        e.set_option("force_path", 2); e.set_option("binned_max_positions", 8192)
        e.count_filtered(st)
        np.testing.assert_array_equal(e.query(lo[sel], hi[sel] if k > 32 else None), ot.query(lo[sel], hi[sel]))
        with pytest.raises(Exception):
            e.set_option("binned_max_positions", 1 << 40)


@pytest.mark.parametrize("k,path", [(31, 1), (31, 2), (45, 1), (45, 2)])
def test_counts_saturate_at_uint32_max(oracle, k, path):
    """Jellyfish's 4-byte counter: a count that would pass 2^32 - 1 stays there, in the direct
    kernels (atomic add + max) and in kernel C (wrapping LDS adds, saturated at write-back by
    comparing with the HBM count) -- both probe variants of C, narrow and wide keys."""
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(3)
    reads = ["A" * 200] * 50 + rand_reads(rng, 300, 80, 160)
    st = ReadStream.from_strings(reads)
    t, (lo, hi, cnt) = oracle_sorted(oracle, k, reads)
    a_idx = int(np.flatnonzero((lo == 0) & (hi == 0))[0])                      # poly-A is key 0
    n_a = int(cnt[a_idx]); assert n_a == 50 * (200 - k + 1)
    other = (a_idx + 1) % len(lo)
    for flags in ((0,) if path == 1 else (0, 8)):                                # 8: kernel C's plain probe loop
        with KmerEngine(k, capacity_hint=1 << 14) as e:
            e.set_option("force_path", path); e.set_option("debug_flags", flags)
            pre = np.array([0xFFFFFFFF - n_a + 7, 0xFFFFFFF0], np.uint32)          # 7 short of saturating; already near the top
            e.add_pairs(np.array([lo[a_idx], lo[other]]), np.array([hi[a_idx], hi[other]]), pre)
            e.count(st)
            got = e.query(np.array([lo[a_idx], lo[other]]), np.array([hi[a_idx], hi[other]]) if k > 32 else None)
            assert int(got[0]) == 0xFFFFFFFF, hex(int(got[0]))
            assert int(got[1]) == min(0xFFFFFFFF, 0xFFFFFFF0 + int(cnt[other]))
            glo, ghi, gcnt = e.export_ge(0)
            exp = cnt.astype(np.uint64).copy(); exp[a_idx] += int(pre[0]); exp[other] += int(pre[1])
            np.testing.assert_array_equal(gcnt, np.minimum(exp, 0xFFFFFFFF).astype(np.uint32))


@pytest.mark.parametrize("k,path", [(31, 1), (31, 2), (47, 2)])
def test_count_in_key_space_slices(oracle, k, path):
    """"key_parts" / "key_part": a sample whose distinct k-mers do not fit one table is counted slice by slice
    over the same stream.  The slices partition the key space (ranges of the low hash bits), so their dumps are
    disjoint, their union is the full count, and their window totals add up -- also when the table has to grow
    in the middle of a slice."""
    from kmer_denovo_filter_amd import KmerEngine, ReadStream

    def slice_of(lo_, hi_, parts):                         # kdf_slice(kdf_hash(lo, hi), parts) of csrc/kdf_device.h
        M = (1 << 64) - 1
        x = lo_ ^ (((hi_ << 37) | (hi_ >> 27)) & M)
        h = ((x ^ (x >> 32)) * 0x9FB21C651E98DF25) & M
        return ((h & 0xFFFF) * parts) >> 16
    rng = np.random.default_rng(21)
    genome = rng.integers(0, 4, 60_000).astype(np.uint8)
    reads = rand_reads(rng, 1500, 40, 300, genome=genome) + ["A" * 200] * 5
    st = ReadStream.from_strings(reads)
    t, (lo, hi, cnt) = oracle_sorted(oracle, k, reads)
    total_windows = oracle.count_windows(reads, k)
    for parts in (2, 5):
        got, windows = {}, 0
        with KmerEngine(k, capacity_hint=1 << 9) as e:                       # small: every slice grows the table
            e.set_option("force_path", path); e.set_option("key_parts", parts)
            for p in range(parts):
                e.clear(); e.set_option("key_part", p)
                e.count(st)
                glo, ghi, gcnt = e.export_ge(0)
                windows += e.stats()[2]
                for a, b, c in zip(glo.tolist(), ghi.tolist(), gcnt.tolist()):
                    assert (b, a) not in got and slice_of(a, b, parts) == p
                    got[(b, a)] = c
            with pytest.raises(Exception):
                e.set_option("key_part", parts)
        assert windows == total_windows
        assert got == {(int(b), int(a)): int(c) for a, b, c in zip(lo, hi, cnt)}


def test_table_too_large_is_a_clear_error():
    """Out of device memory names the way out (key-space slices) instead of a bare HIP error."""
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd._native import KdfError
    with pytest.raises(KdfError) as ei:
        KmerEngine(31, capacity_hint=1 << 40)            # 2^41 slots = 26 TB
    assert "key_parts" in str(ei.value) and "does not fit" in str(ei.value)
    with KmerEngine(31, capacity_hint=1 << 10) as e:      # the process is still usable afterwards
        assert e.stats()[1] == 0


@pytest.mark.parametrize("k", [21, 31, 47])
def test_count_filtered_through_the_sieve(oracle, k):
    """count --if with the membership sieve (auto path; force_path 4 insists on it): every filter size from empty
    to "every k-mer of the stream" (all windows survive the sieve and drain through the wave queues), a filter
    loaded from device memory, counts kept across batches and zeroed by reset_counts."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(500 + k)
    genome = rng.integers(0, 4, 50000).astype(np.uint8)
    child = rand_reads(rng, 1500, 50, 260, genome=genome)
    parent = rand_reads(rng, 3000, 0, 260, genome=genome) + ["A" * 300] * 20
    _, (lo, hi, _) = oracle_sorted(oracle, k, child + ["A" * 100])
    st1, st2 = ReadStream.from_strings(parent[:1000]), ReadStream.from_strings(parent[1000:])
    for frac in (0.0, 0.01, 0.4, 1.0):
        sel = rng.random(len(lo)) < frac if frac < 1.0 else np.ones(len(lo), bool)
        flo, fhi = lo[sel], hi[sel]
        ot = oracle.OracleTable(k, 1 << 12).load_filter(flo, fhi).count_reads_filtered(parent)
        with KmerEngine(k) as e:
            if frac == 0.4:                                   # device-resident filter keys
                dl = torch.from_numpy(flo.view(np.int64).copy()).cuda()
                dh = torch.from_numpy(fhi.view(np.int64).copy()).cuda() if k > 32 else None
                e.load_filter_dev(dl.data_ptr(), dh.data_ptr() if dh is not None else None, len(flo))
            else:
                e.load_filter(flo, fhi)
            e.set_option("force_path", 4)
            e.count_filtered(st1); e.count_filtered(st2)
            assert e.last_count_path() == "sieve"
            assert e.stats()[2] == oracle.count_windows(parent, k)
            np.testing.assert_array_equal(e.query(lo, hi), ot.query(lo, hi))
            glo, _, gcnt = e.export_ge(0)
            olo, _, ocnt = ot.export_ge(0)
            np.testing.assert_array_equal(glo, olo); np.testing.assert_array_equal(gcnt, ocnt)
            e.reset_counts()
            assert int(e.query(lo, hi).sum()) == 0 and e.stats()[1] == len(flo)
            e.set_option("force_path", 0)
            e.count_filtered(st1)
            o1 = oracle.OracleTable(k, 1 << 12).load_filter(flo, fhi).count_reads_filtered(parent[:1000])
            np.testing.assert_array_equal(e.query(lo, hi), o1.query(lo, hi))


@pytest.mark.parametrize("k", [31, 47])
def test_deferred_flush_equals_eager_equals_direct(oracle, k):
    """A sample streamed in many batches (discovery/pipeline.py:106-172: ONE `jellyfish count` over the whole
    `samtools fasta` pipe).  Kernel C deferred over the pending partition passes == a flush after every count call ==
    the direct global-atomic path == the oracle, with `query` / `count_ge` issued in the middle of the stream (they see
    exactly the batches counted so far), a table that is far too small (it grows, buckets are replayed) and a heavy
    homopolymer batch (the skew instantiation of kernel C and its heavy-bucket split)."""
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(1300 + k)
    genome = rng.integers(0, 4, 60000).astype(np.uint8)
    reads = rand_reads(rng, 4000, 0, 300, genome=genome) + ["", "A" * 500, "N" * 70, "ACGT" * 80] + ["A" * 300] * 3000
    batches = [reads[i::8] for i in range(8)]
    t, (lo, hi, cnt) = oracle_sorted(oracle, k, reads)
    part = [b for bt in batches[:3] for b in bt]
    tp, (plo, phi, pcnt) = oracle_sorted(oracle, k, part)
    dumps = {}
    for name, path, defer, hint in (("deferred", 2, 1, 1 << 13), ("eager", 2, 0, 1 << 13), ("direct", 1, 1, 1 << 13),
                                    ("deferred-big", 2, 1, 1 << 20)):
        with KmerEngine(k, capacity_hint=hint) as e:
            e.set_option("force_path", path); e.set_option("defer", defer)
            for i, bt in enumerate(batches):
                e.count(ReadStream.from_strings(bt))
                if i == 2:                                        # mid-stream reads of the table: everything counted so far, nothing else
                    if path == 2 and defer:
                        assert e.get_stat("pending_passes") == 3 and e.get_stat("flushes") == 0
                    assert e.count_ge(2) == int((pcnt >= 2).sum())
                    np.testing.assert_array_equal(e.query(plo[::7], phi[::7] if k > 32 else None), pcnt[::7])
                    assert e.get_stat("pending_passes") == 0
            glo, ghi, gcnt = e.export_ge(0)
            if path == 2:
                assert e.get_stat("binned_passes") == 8
                # (deferred: the mid-stream read and the final one; a ring sized for the small table's geometry may add one)
                assert (2 <= e.get_stat("flushes") <= 4) if defer else e.get_stat("flushes") == 8
            assert e.stats()[2] == oracle.count_windows(reads, k)
            if name == "deferred":
                assert e.get_stat("replayed_buckets") > 0         # the table started far too small
            np.testing.assert_array_equal(glo, lo); np.testing.assert_array_equal(ghi, hi); np.testing.assert_array_equal(gcnt, cnt)
            dumps[name] = (glo, gcnt)


@pytest.mark.parametrize("k", [27, 63])
def test_pending_stream_of_small_batches(oracle, k):
    """Auto path: small batches are concatenated in the pending stream (every batch starts on a tile boundary; the bits
    past a batch's end read invalid whatever the source holds), partitioned when the stream reaches `l1_positions`,
    bigger batches are partitioned where they lie; `kdf_clear` drops what is pending; a flush of little goes direct."""
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(1700 + k)
    genome = rng.integers(0, 4, 80000).astype(np.uint8)
    reads = rand_reads(rng, 6000, 0, 300, genome=genome) + ["", "N" * 70, "ACGT" * 80]
    t, (lo, hi, cnt) = oracle_sorted(oracle, k, reads)
    sizes = [1, 7, 300, 64, 1500, 2, 900, 33]                     # reads per batch, cycled: ragged batches
    with KmerEngine(k, capacity_hint=1 << 18) as e:
        e.set_option("l1_positions", 150_000); e.set_option("l1_direct_positions", 400_000); e.set_option("binned_min_positions", 10_000)
        e.set_option("binned_bytes_per_position", 1 << 20)
        e.count(ReadStream.from_strings(reads[:50]))              # ... and dropped:
        e.clear()
        assert e.stats()[1:] == (0, 0)
        a, i = 0, 0
        while a < len(reads):
            n = sizes[i % len(sizes)]; i += 1
            e.count(ReadStream.from_strings(reads[a:a + n])); a += n
        assert e.get_stat("binned_passes") >= 2                   # the pending stream filled up more than once
        glo, ghi, gcnt = e.export_ge(0)
        assert e.stats()[2] == oracle.count_windows(reads, k)
        np.testing.assert_array_equal(glo, lo); np.testing.assert_array_equal(ghi, hi); np.testing.assert_array_equal(gcnt, cnt)
        # little pending at flush time: the direct kernels
        e.set_option("binned_min_positions", 1 << 22)
        e.count(ReadStream.from_strings(reads[:40]))
        t2 = oracle.OracleTable(k, 1 << 12).count_reads(reads).count_reads(reads[:40])
        lo2, hi2, cnt2 = t2.export_ge(0)
        glo, ghi, gcnt = e.export_ge(0)
        assert e.last_count_path() == "direct"
        np.testing.assert_array_equal(glo, lo2); np.testing.assert_array_equal(gcnt, cnt2)


@pytest.mark.parametrize("k", [31, 63])
def test_heavy_buckets_are_split_and_stay_exact(oracle, k):
    """The passes that follow a skewed one (one coarse bin far above its share) run the second instantiation of kernel C;
    buckets whose runs hold more than 65 536 entries are left to kb_heavy_slice_kernel / kb_heavy_combine_kernel (32
    workgroups per bucket, private LDS tables, transactional fold).  Homopolymer and microsatellite reads by the hundred thousand, in two
    batches (the second into the live table), against the oracle and against the direct path."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(12)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    fl = lambda n: acgt[rng.integers(0, 4, n)].tobytes().decode()
    heavy = []
    longer = 0 if k <= 32 else 40                                   # (the repeats have to outlast a 63-mer)
    for _ in range(60_000):
        heavy.append(fl(int(rng.integers(5, 40))) + "A" * int(rng.integers(40 + longer, 100 + longer)) + fl(int(rng.integers(5, 30))))
    for _ in range(30_000):
        heavy.append(fl(10) + "CA" * int(rng.integers(25 + longer // 2, 50 + longer // 2)) + fl(12))
    for _ in range(20_000):
        heavy.append("GAA" * int(rng.integers(15 + longer // 3, 35 + longer // 3)) + fl(20))
    reads = heavy + rand_reads(rng, 60_000, 100, 151)
    rng.shuffle(reads)
    lo, hi, cnt = oracle.OracleTable(k, 1 << 12).count_reads(reads, threads=8).export_ge(0)
    assert int(cnt.max()) > 500_000
    st = ReadStream.from_strings(reads)
    half = len(reads) // 2
    for hint, batches in ((1 << 22, 1), (1 << 22, 2), (1 << 16, 1)):
        with KmerEngine(k, capacity_hint=hint) as e:
            e.set_option("force_path", 2)
            e.count(ReadStream.from_strings(reads[:half])); e.clear()           # the engine has now seen a skewed pass
            if batches == 1:
                e.count(st)
            else:
                e.count(ReadStream.from_strings(reads[:half])); e.count(ReadStream.from_strings(reads[half:]))
            glo, ghi, gcnt = e.export_ge(0)
            np.testing.assert_array_equal(glo, lo); np.testing.assert_array_equal(ghi, hi); np.testing.assert_array_equal(gcnt, cnt)
            assert e.stats()[2] == oracle.count_windows(reads, k)
            np.testing.assert_array_equal(e.query(lo[::7], hi[::7] if k > 32 else None), cnt[::7])
            if hint == 1 << 22 and batches == 1:
                assert e.get_stat("heavy_buckets") > 0, "the skewed instantiation did not split any bucket: the test does not reach the code"


@pytest.mark.parametrize("k", [31, 63])
def test_dump_fused_into_the_flush_equals_the_table_dump(oracle, k):
    """`jellyfish count` followed by `dump -L n` (discovery/pipeline.py:114-196): with partition passes pending, the flush that
    applies them can write the dump out of the buckets it holds (kb_bucket_kernel<.., DUMP>; option fused_dump).  Fused == dump from
    the table (fused_dump 0, the default) == oracle, for several thresholds, into a fresh table and into a live one (counts from an earlier
    flush), with the heavy-bucket split taking the dump back to the table pass, and with a buffer that is too small."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(4100 + k)
    genome = rng.integers(0, 4, 40000).astype(np.uint8)
    first = rand_reads(rng, 6000, k, 260, genome=genome)
    second = rand_reads(rng, 5000, k, 260, genome=genome) + ["ACGT" * 70, "N" * 80, ""]
    wide = k > 32

    def dump(e, min_count, cap):
        lo = torch.zeros(max(cap, 1), dtype=torch.int64, device="cuda:0"); cnt = torch.zeros(max(cap, 1), dtype=torch.int32, device="cuda:0")
        hi = torch.zeros(max(cap, 1), dtype=torch.int64, device="cuda:0") if wide else None
        torch.cuda.synchronize()
        n = e.export_ge_dev(min_count, lo.data_ptr(), hi.data_ptr() if wide else None, cnt.data_ptr(), cap, sorted_=True)
        glo = lo[:n].cpu().numpy().view(np.uint64); gcnt = cnt[:n].cpu().numpy().view(np.uint32)
        ghi = hi[:n].cpu().numpy().view(np.uint64) if wide else np.zeros(n, np.uint64)
        return glo, ghi, gcnt

    def want(reads, min_count):
        _, (lo, hi, cnt) = oracle_sorted(oracle, k, reads)
        keep = cnt >= min_count
        return lo[keep], hi[keep], cnt[keep]

    for min_count in (1, 2, 3):
        for fused in (1, 0):
            with KmerEngine(k, capacity_hint=1 << 20) as e:
                e.set_option("force_path", 2); e.set_option("fused_dump", fused)
                e.count(ReadStream.from_strings(first[0::4]))
                assert e.count_ge(1) > 0                          # (a tally left behind by an earlier read must not leak into the dump's size: the fuzz found that)
                for i in range(1, 4):
                    e.count(ReadStream.from_strings(first[i::4]))
                assert e.get_stat("pending_passes") == 3
                got = dump(e, min_count, 1 << 21)
                assert e.get_stat("pending_passes") == 0 and e.get_stat("flushes") == 2
                assert e.get_stat("fused_dumps") == fused
                for g, w in zip(got, want(first, min_count)):
                    np.testing.assert_array_equal(g, w)
                # more reads into the LIVE table (kernel C reads the buckets back and adds), dumped again
                e.count(ReadStream.from_strings(second))
                got = dump(e, min_count, 1 << 21)
                assert e.get_stat("fused_dumps") == 2 * fused
                for g, w in zip(got, want(first + second, min_count)):
                    np.testing.assert_array_equal(g, w)
                # nothing pending: the table pass
                got = dump(e, min_count + 1, 1 << 21)
                assert e.get_stat("fused_dumps") == 2 * fused
                for g, w in zip(got, want(first + second, min_count + 1)):
                    np.testing.assert_array_equal(g, w)
    # a buffer that is too small: the error the table dump gives, nothing written past the end
    with KmerEngine(k, capacity_hint=1 << 20) as e:
        e.set_option("force_path", 2); e.set_option("fused_dump", 1)
        e.count(ReadStream.from_strings(first))
        guard = torch.full((1000 + 64,), -7, dtype=torch.int64, device="cuda:0"); cnt = torch.zeros(1000, dtype=torch.int32, device="cuda:0")
        hi = torch.zeros(1000, dtype=torch.int64, device="cuda:0") if wide else None
        torch.cuda.synchronize()
        with pytest.raises(Exception, match="room for 1000"):
            e.export_ge_dev(1, guard.data_ptr(), hi.data_ptr() if wide else None, cnt.data_ptr(), 1000)
        assert bool((guard[1000:] == -7).all())
        got = dump(e, 1, 1 << 21)                                 # (the table is intact)
        for g, w in zip(got, want(first, 1)):
            np.testing.assert_array_equal(g, w)
    # heavy buckets (a homopolymer flood: the skew instantiation leaves them to the heavy-bucket kernels): no fused dump, same result
    flood = first[:2000] + ["A" * 400] * 6000
    with KmerEngine(k, capacity_hint=1 << 20) as e:
        e.set_option("force_path", 2); e.set_option("fused_dump", 1)
        e.count(ReadStream.from_strings(flood)); e.flush()        # (the first flush of an engine learns the skew)
        e.count(ReadStream.from_strings(flood))
        got = dump(e, 2, 1 << 21)
        assert e.get_stat("heavy_buckets") > 0 and e.get_stat("fused_dumps") == 0
        for g, w in zip(got, want(flood + flood, 2)):
            np.testing.assert_array_equal(g, w)
