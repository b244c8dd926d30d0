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
