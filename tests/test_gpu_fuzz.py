"""Seeded randomised parity of the engine against the oracle (a short slice of scratch/fuzz_parity.py, which ran
6 000+ cases at the end of round 1): k, read sets, table sizes, kernel path (direct / binned), kernel C variant
(lookahead + queue / plain loop) and pass splitting are drawn at random; count, two-batch count, the threshold
dump, query and count --if must agree bit for bit."""
import numpy as np
import pytest

from test_gpu_parity_basic import rand_reads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_random_cases_match_oracle(oracle, seed):
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    rng = np.random.default_rng(seed)
    for it in range(60):
        k = int(rng.choice([5, 11, 21, 31, 32, 33, 45, 47, 63, int(rng.integers(1, 64))]))
        n = int(rng.integers(50, 2500))
        genome = rng.integers(0, 4, int(rng.integers(2000, 150000))).astype(np.uint8) if rng.random() < 0.8 else None
        reads = rand_reads(rng, n, max(1, k - 3), int(rng.integers(k + 1, 400)), n_frac=float(rng.choice([0, 0.002, 0.05])), genome=genome)
        if rng.random() < 0.3:
            reads += ["A" * int(rng.integers(k, 500))] * int(rng.integers(1, 60)) + ["ACGT" * 100] * int(rng.integers(0, 20))
        path, flags = int(rng.choice([1, 2])), int(rng.choice([0, 8]))
        hint, maxpos = int(rng.choice([1 << 8, 1 << 12, 1 << 16, 1 << 20])), int(rng.choice([1 << 31, 4096, 65536]))
        tag = f"seed {seed} case {it}: k={k} reads={len(reads)} path={path} flags={flags} hint={hint} maxpos={maxpos}"
        lo, hi, cnt = oracle.OracleTable(k, 1 << 12).count_reads(reads).export_ge(0)
        with KmerEngine(k, capacity_hint=hint) as e:
            e.set_option("force_path", path); e.set_option("debug_flags", flags); e.set_option("binned_max_positions", maxpos)
            if rng.random() < 0.5:
                half = len(reads) // 2
                e.count(ReadStream.from_strings(reads[:half])); e.count(ReadStream.from_strings(reads[half:]))
            else:
                e.count(ReadStream.from_strings(reads))
            glo, ghi, gcnt = e.export_ge(0)
            assert np.array_equal(glo, lo) and np.array_equal(ghi, hi) and np.array_equal(gcnt, cnt), tag
            thr = int(rng.integers(1, 6))
            assert e.count_ge(thr) == int((cnt >= thr).sum()), tag
            if len(lo):
                sel = rng.choice(len(lo), size=min(len(lo), 300), replace=False)
                assert np.array_equal(e.query(lo[sel], hi[sel] if k > 32 else None), cnt[sel]), tag
        if len(lo) > 4:
            sel = np.sort(rng.choice(len(lo), size=max(1, len(lo) // int(rng.integers(2, 6))), replace=False))
            other = rand_reads(rng, int(rng.integers(50, 1500)), max(1, k - 3), 300, genome=genome) + reads[: len(reads) // 3]
            ot = oracle.OracleTable(k, 1 << 12).load_filter(lo[sel], hi[sel]).count_reads_filtered(other)
            with KmerEngine(k, capacity_hint=hint) as e:
                e.load_filter(lo[sel], hi[sel] if k > 32 else None)
                e.set_option("force_path", path); e.set_option("debug_flags", flags); e.set_option("binned_max_positions", maxpos)
                e.count_filtered(ReadStream.from_strings(other))
                assert np.array_equal(e.query(lo[sel], hi[sel] if k > 32 else None), ot.query(lo[sel], hi[sel])), tag + " (count --if)"


def khash(lo, hi):
    x = lo.copy()
    if hi is not None:
        x ^= (hi << np.uint64(37)) | (hi >> np.uint64(27))
    with np.errstate(over="ignore"):
        return (x ^ (x >> np.uint64(32))) * np.uint64(0x9FB21C651E98DF25)


def round2_case(O, rng, scale, tag0):
    """One random case over the round-2 code paths (see test_random_cases_round2_paths); returns the merge path taken."""
    import torch
    from kmer_denovo_filter_amd import KmerEngine, ReadStream
    k = int(rng.choice([16, 21, 27, 31, 32, 33, 47, 63, int(rng.integers(1, 64))]))
    n = int(rng.integers(50, 4000)) * scale
    genome = rng.integers(0, 4, int(rng.integers(2000, 200000)) * scale).astype(np.uint8) if rng.random() < 0.8 else None
    reads = rand_reads(rng, n, max(1, k - 3), int(rng.integers(k + 1, 400)), n_frac=float(rng.choice([0, 0.002, 0.05])), genome=genome)
    if rng.random() < 0.3:
        reads += ["A" * int(rng.integers(k, 500))] * int(rng.integers(1, 60)) + ["ACGT" * 100] * int(rng.integers(0, 20))
    wide = k > 32
    paths = [0, 1, 2, 2]
    path = int(rng.choice(paths))
    defer = int(rng.choice([0, 1, 1]))                       # kernel C deferred over the batches, or a flush per count call
    l1 = int(rng.choice([1 << 12, 1 << 16, 1 << 30]))        # (auto path) pending-stream size that triggers a partition
    hshift = int(rng.choice([0, 0, 1, 3]))
    hint = int(rng.choice([1 << 8, 1 << 12, 1 << 16, 1 << 20]))
    maxpos = int(rng.choice([1 << 31, 4096, 65536]))
    tag = f"{tag0} k={k} reads={len(reads)} path={path} defer={defer} l1={l1} hshift={hshift} hint={hint} maxpos={maxpos}"
    lo, hi, cnt = O.OracleTable(k, 1 << 12).count_reads(reads).export_ge(0)

    skewvar = 4096 if rng.random() < 0.5 else 0              # kernel C's skew instantiation (wave-aggregated count adds), forced

    bigb = (31, 10, 14)[(len(reads) + k) % 3]                # tables from 2^bigb slots on take the big-bucket instantiation of kernel C
                                                             # (not drawn from rng: the cases of earlier rounds stay what they were)

    def opts(e, p=path):
        e.set_option("big_bucket_log2cap", bigb)
        e.set_option("force_path", p); e.set_option("binned_max_positions", maxpos); e.set_option("debug_flags", skewvar)
        e.set_option("defer", defer); e.set_option("l1_positions", l1); e.set_option("l1_direct_positions", 4 * l1)
        e.set_option("binned_min_positions", int(rng.choice([1 << 10, 1 << 22])))
        e.set_option("fused_dump", (len(reads) + k) % 2)     # `dump -L` written by the flush that applies the pending passes (not drawn from rng either)

    with KmerEngine(k, capacity_hint=hint) as e:
        opts(e)
        if hshift:
            e.set_option("hash_shift", hshift)
        half = len(reads) // 2
        r = rng.random()
        if r < 0.35:
            e.count(ReadStream.from_strings(reads[:half])); e.count(ReadStream.from_strings(reads[half:]))
        elif r < 0.7:                                        # a streamed sample: many small batches, something read in between
            step = max(1, len(reads) // int(rng.integers(3, 12)))
            for a in range(0, len(reads), step):
                e.count(ReadStream.from_strings(reads[a:a + step]))
                if rng.random() < 0.2:
                    e.count_ge(1)
        else:
            e.count(ReadStream.from_strings(reads))
        if len(lo):                                          # `dump -L n` into caller-sized device buffers, whatever is still pending
            dthr = 1 + len(reads) % 3
            dl = torch.zeros(len(lo), dtype=torch.int64, device="cuda:0"); dc = torch.zeros(len(lo), dtype=torch.int32, device="cuda:0")
            dh = torch.zeros(len(lo), dtype=torch.int64, device="cuda:0") if wide else None
            torch.cuda.synchronize()
            nd = e.export_ge_dev(dthr, dl.data_ptr(), dh.data_ptr() if wide else None, dc.data_ptr(), len(lo), sorted_=True)
            keep = cnt >= dthr
            assert nd == int(keep.sum()), tag
            assert np.array_equal(dl[:nd].cpu().numpy().view(np.uint64), lo[keep]) and np.array_equal(dc[:nd].cpu().numpy().view(np.uint32), cnt[keep]), tag
            if wide:
                assert np.array_equal(dh[:nd].cpu().numpy().view(np.uint64), hi[keep]), tag
        glo, ghi, gcnt = e.export_ge(0)
        assert np.array_equal(glo, lo) and np.array_equal(ghi, hi) and np.array_equal(gcnt, cnt), tag
        thr = int(rng.integers(1, 6))
        assert e.count_ge(thr) == int((cnt >= thr).sum()), tag
        if len(lo):
            sel = rng.choice(len(lo), size=min(len(lo), 500), replace=False)
            assert np.array_equal(e.query(lo[sel], hi[sel] if wide else None), cnt[sel]), tag

    # ---- the merge: the reads in 2-4 parts, each counted by its own engine, all dumps summed into one owner table
    m = int(rng.integers(2, 5))
    cut = sorted(rng.integers(0, len(reads) + 1, m - 1).tolist())
    parts = [reads[a:b] for a, b in zip([0] + cut, cut + [len(reads)])]
    own_shift = int(rng.choice([0, 1, 2, 3]))
    ordered = rng.random() < 0.7
    dev = torch.device("cuda:0")
    segs, keep = [], []
    for pr in parts:
        if not pr:
            continue
        with KmerEngine(k, capacity_hint=hint) as e:
            opts(e, int(rng.choice(paths)))
            e.count(ReadStream.from_strings(pr))
            plo, phi, pcnt = e.export_ge(0)
        if ordered and len(plo):                             # what kdf_export_parts_dev guarantees: ascending hash order
            with np.errstate(over="ignore"):
                h = khash(plo, phi if wide else None) << np.uint64(own_shift)
            o = np.argsort(h, kind="stable")
            plo, phi, pcnt = plo[o], phi[o], pcnt[o]
        tl = torch.from_numpy(plo.view(np.int64).copy()).to(dev)
        th = torch.from_numpy(phi.view(np.int64).copy()).to(dev) if wide else None
        tc = torch.from_numpy(pcnt.view(np.int32).copy()).to(dev)
        keep.append((tl, th, tc))
        segs.append((tl.data_ptr(), th.data_ptr() if wide else None, tc.data_ptr(), len(plo)))
    torch.cuda.synchronize()
    with KmerEngine(k, capacity_hint=int(rng.choice([1 << 8, 1 << 16]))) as own:
        own.set_option("big_bucket_log2cap", bigb)
        own.set_option("hash_shift", own_shift)
        own.set_option("merge_min_pairs", int(rng.choice([1, 1, 1 << 16])))
        pre = rng.random() < 0.3 and len(parts[0]) > 0      # a live table: the first part is counted in directly, not merged
        if pre:
            own.set_option("force_path", 1)
            own.count(ReadStream.from_strings(parts[0]))
            segs_m = segs[1:]
        else:
            if rng.random() < 0.5:
                own.clear()                                  # deferred clear: the merge writes every bucket
            segs_m = segs
        own.add_pairs_multi_dev(segs_m)
        own.synchronize()
        mp = own.get_stat("last_merge_path")
        glo, ghi, gcnt = own.export_ge(0)
        assert np.array_equal(glo, lo) and np.array_equal(ghi, hi) and np.array_equal(gcnt, cnt), tag + f" merge m={m} own_shift={own_shift} ordered={ordered} pre={pre} path={mp}"

    # ---- count --if through the sieve / binned / direct
    if len(lo) > 4:
        sel = np.sort(rng.choice(len(lo), size=max(1, len(lo) // int(rng.integers(2, 6))), replace=False))
        other = rand_reads(rng, int(rng.integers(50, 2000)) * scale, max(1, k - 3), 300, genome=genome) + reads[: len(reads) // 3]
        ot = O.OracleTable(k, 1 << 12).load_filter(lo[sel], hi[sel]).count_reads_filtered(other)
        fp = int(rng.choice([0, 1, 2, 4, 4])); sb = int(rng.choice([0, 8, 16, 32]))
        with KmerEngine(k, capacity_hint=hint) as e:
            e.set_option("sieve_bits", sb); e.set_option("big_bucket_log2cap", bigb)
            e.load_filter(lo[sel], hi[sel] if wide else None)
            e.set_option("force_path", fp); e.set_option("binned_max_positions", maxpos)
            e.count_filtered(ReadStream.from_strings(other))
            assert np.array_equal(e.query(lo[sel], hi[sel] if wide else None), ot.query(lo[sel], hi[sel])), tag + f" filtered fp={fp} sb={sb}"
            assert e.stats()[2] == O.count_windows(other, k), tag + " filtered windows"
    return mp


@pytest.mark.parametrize("seed", [404, 505])
def test_random_cases_round2_paths(oracle, seed):
    """The round-2 / round-3 paths (scratch/fuzz_round2.py): deferred kernel C over streamed batches, the pending stream
    of small batches, owner tables (hash_shift), the multi-segment merge in hash order (LDS buckets) and out
    of order (atomic fallback), into fresh, cleared and live tables, and count --if through the sieve at every width."""
    rng = np.random.default_rng(seed)
    taken = set()
    for it in range(40):
        taken.add(round2_case(oracle, rng, 1, f"seed {seed} case {it}"))
    assert {1, 2} <= taken                                  # both merge kernels were exercised

