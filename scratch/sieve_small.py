"""count --if with small filters (VCF mode / Module 3 sizes): sieve in LDS vs in L2 (debug flag 2048), direct, binned"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from kmer_denovo_filter_amd import KmerEngine, devkeys
from kmer_denovo_filter_amd.synth import synth_stream
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda", genome_seed=20260417); torch.cuda.synchronize()
e0 = KmerEngine(31, capacity_hint=1 << 22)
e0.count_dev(ds.packed.data_ptr(), 0 + ds.invalid.data_ptr(), 64 * 200_000)     # a slice: its k-mers are the filter pool
lo, hi, cnt = e0.export_ge(0); e0.close()
for n in (630, 1484, 10_000, 60_000, 1_000_000):
    sel = lo[:: max(1, len(lo) // n)][:n]
    dlo, _ = devkeys.from_host(sel, None, False)
    row = {"keys": len(sel)}
    ref = None
    for name, path, flags in (("sieve_lds", 4, 0), ("sieve_l2", 4, 2048), ("direct", 1, 0)):
        e = KmerEngine(31, capacity_hint=len(sel)); e.set_option("debug_flags", flags)
        e.load_filter_dev(dlo.data_ptr(), None, len(sel)); e.set_option("force_path", path)
        for it in range(3):
            e.reset_counts(); e.synchronize(); e.profile(True)
            e.count_filtered_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
            ms, _, _ = e.profile_read(); e.profile(False)
        c = devkeys.query(e, dlo, None)
        if ref is None: ref = c
        assert torch.equal(c, ref), name
        row[name] = round(e.stats()[2] / ms / 1e6, 1)
        e.close()
    print(json.dumps(row))
# Module-3 scan: hit bit per window, sieve path vs direct kernel (force_path 1)
for n in (630, 100_000):
    sel = lo[:: max(1, len(lo) // n)][:n]
    row = {"scan_keys": len(sel)}
    ref = None
    hits = torch.empty(ds.invalid.numel(), dtype=torch.int64, device="cuda")
    for name, path in (("sieve", 0), ("direct", 1)):
        e = KmerEngine(31, capacity_hint=len(sel))
        e.add_pairs(sel, np.zeros(len(sel), np.uint64), np.ones(len(sel), np.uint32)); e.set_option("force_path", path)
        import time
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            e.scan_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases, hits.data_ptr()); e.synchronize()
            dt = time.perf_counter() - t0
        n_tiles = (ds.n_bases + 63) // 64
        h = hits[:n_tiles].clone()
        if ref is None: ref = h
        assert torch.equal(h, ref), name
        row[name] = round(1163397354 / dt / 1e9, 1)
        e.close()
    print(json.dumps(row))
