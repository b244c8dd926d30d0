"""count --if through the sieve on the parent-filter workload: kernel time for several sieve sizes"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
import parent_filter
from kmer_denovo_filter_amd import KmerEngine, devkeys
res, (lo, hi), streams = parent_filter.run(64_000_000, 30, 31, 20260418, "cuda:0")
# the mother stage's filter = non_ref set: rebuild it (child - ref)
ds = streams["mother"]
dlo, _ = devkeys.from_host(lo, None, False)      # the survivors (1.86 M keys): same size class as the stage's filter
for bits in (0, 8, 16, 32, 64):
    e = KmerEngine(31, capacity_hint=len(lo))
    e.set_option("sieve_bits", bits)
    e.load_filter_dev(dlo.data_ptr(), None, len(lo))
    for it in range(3):
        e.reset_counts(); e.synchronize(); e.profile(True)
        e.count_filtered_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
        ms, n, pos = e.profile_read(); e.profile(False)
    w = e.stats()[2]
    print(json.dumps({"sieve_bits": bits, "kernel_ms": round(ms, 3), "Gkmer_s": round(w / ms / 1e6, 1), "path": e.last_count_path()}))
    e.close()
