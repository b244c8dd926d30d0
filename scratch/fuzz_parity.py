"""Randomised parity of the engine against the oracle (GPU box): k, read sets, table sizes, kernel paths and
variants are drawn at random; count, count --if, query and the >= threshold dump are compared bit for bit."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from kmer_denovo_filter_amd import KmerEngine, ReadStream
from oracle import oracle as O
O.build()
from test_gpu_parity_basic import rand_reads

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # > 1: more reads, longer genomes (many buckets per table)
rng = np.random.default_rng(seed)
t0 = time.time(); it = 0
while time.time() - t0 < budget:
    it += 1
    k = int(rng.choice([5, 11, 21, 31, 32, 33, 45, 47, 63, int(rng.integers(1, 64))]))
    n = int(rng.integers(50, 4000)) * scale
    genome = rng.integers(0, 4, int(rng.integers(2000, 200000)) * scale).astype(np.uint8) if rng.random() < 0.8 else None
    reads = rand_reads(rng, n, max(1, k - 3), int(rng.integers(k + 1, 400)), n_frac=float(rng.choice([0, 0.002, 0.05])), genome=genome)
    if rng.random() < 0.3:
        reads += ["A" * int(rng.integers(k, 500))] * int(rng.integers(1, 60)) + ["ACGT" * 100] * int(rng.integers(0, 20))
    path = int(rng.choice([0, 1, 2, 2])); flags = int(rng.choice([0, 4096])); hint = int(rng.choice([1 << 8, 1 << 12, 1 << 16, 1 << 20]))
    maxpos = int(rng.choice([1 << 31, 4096, 65536]))
    st = ReadStream.from_strings(reads)
    t = O.OracleTable(k, 1 << 12).count_reads(reads)
    lo, hi, cnt = t.export_ge(0)
    tag = f"it {it} k={k} reads={len(reads)} path={path} flags={flags} hint={hint} maxpos={maxpos}"
    with KmerEngine(k, capacity_hint=hint) as e:
        e.set_option("force_path", path); e.set_option("debug_flags", flags); e.set_option("binned_max_positions", maxpos)
        half = len(reads) // 2
        if rng.random() < 0.5:                      # two batches into the same table
            e.count(ReadStream.from_strings(reads[:half])); e.count(ReadStream.from_strings(reads[half:]))
        else:
            e.count(st)
        glo, ghi, gcnt = e.export_ge(0)
        assert np.array_equal(glo, lo) and np.array_equal(ghi, hi) and np.array_equal(gcnt, cnt), tag
        thr = int(rng.integers(1, 6))
        assert e.count_ge(thr) == int((cnt >= thr).sum()), tag
        if len(lo):
            sel = rng.choice(len(lo), size=min(len(lo), 500), replace=False)
            q = e.query(lo[sel], hi[sel] if k > 32 else None)
            assert np.array_equal(q, cnt[sel]), tag
    if len(lo) > 4:
        sel = np.sort(rng.choice(len(lo), size=max(1, len(lo) // int(rng.integers(2, 6))), replace=False))
        other = rand_reads(rng, int(rng.integers(50, 2000)), max(1, k - 3), 300, genome=genome) + reads[: len(reads) // 3]
        ot = O.OracleTable(k, 1 << 12).load_filter(lo[sel], hi[sel]).count_reads_filtered(other)
        with KmerEngine(k, capacity_hint=hint) as e:
            e.load_filter(lo[sel], hi[sel] if k > 32 else None)
            e.set_option("force_path", path); e.set_option("debug_flags", flags); e.set_option("binned_max_positions", maxpos)
            e.count_filtered(ReadStream.from_strings(other))
            assert np.array_equal(e.query(lo[sel], hi[sel] if k > 32 else None), ot.query(lo[sel], hi[sel])), tag + " filtered"
    if it % (20 if scale == 1 else 2) == 0:
        print(f"{it} cases ok ({time.time() - t0:.0f}s)", flush=True)
print(f"done: {it} cases, seed {seed}", flush=True)
