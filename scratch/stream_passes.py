"""Cost of consecutive count passes into a NON-empty table (what steps 2..K of the multi-GPU job do)."""
import sys, time, torch
sys.path.insert(0, '.')
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda:0"); torch.cuda.synchronize()
with KmerEngine(31, capacity_hint=1 << 28) as e:
    for rep in range(2):
        e.clear(); e.synchronize()
        ts = []
        for p in range(4):
            t0 = time.perf_counter()
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("passes into the same table (ms):", [round(t, 2) for t in ts], "count_ge(3*4)", e.count_ge(12), e.stats(), flush=True)
