import sys, numpy as np, torch
sys.path.insert(0, '.')
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
ds = synth_stream(10_000_000, 150, 20_000_000, seed=20260417, device="cuda:0")
for path in (2, 1):
    with KmerEngine(31, capacity_hint=1 << 27) as e:
        e.set_option("force_path", path)
        e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
        cap, distinct, windows = e.stats()
        n0 = e.count_ge(0); n1 = e.count_ge(1)
        lo, hi, cnt = e.export_ge(0)
        u = np.unique(lo)
        print("path", path, "cap", cap, "distinct", distinct, "windows", windows, "count_ge0", n0, "ge1", n1,
              "len", len(lo), "unique", len(u), "sum", int(cnt.astype(np.uint64).sum()),
              "sorted", bool((lo[1:] >= lo[:-1]).all()), "replayed", e.get_stat("replayed_buckets"), "log2cap", e.get_stat("log2cap"), flush=True)
