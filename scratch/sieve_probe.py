"""count --if through the sieves on the parent-filter workload: kernel time per form / size"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
import parent_filter
from kmer_denovo_filter_amd import KmerEngine, devkeys
res, (lo, hi), streams = parent_filter.run(64_000_000, 30, 31, 20260418, "cuda:0")
print(json.dumps(res["stages"]))
ds = streams["mother"]
dlo, _ = devkeys.from_host(lo, None, False)
for form, lg in ((1, 0), (2, 0), (2, 18), (2, 19), (2, 21), (2, 22)):
    e = KmerEngine(31, capacity_hint=len(lo))
    e.set_option("sieve_form", form); e.set_option("sieve2_log2words", lg)
    e.load_filter_dev(dlo.data_ptr(), None, len(lo))
    for it in range(3):
        e.reset_counts(); e.synchronize(); e.profile(True)
        e.count_filtered_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
        ms, n, pos = e.profile_read(); e.profile(False)
    w = e.stats()[2]
    print(json.dumps({"form": form, "log2words": lg, "kernel_ms": round(ms, 3), "Gkmer_s": round(w / ms / 1e6, 1), "path": e.last_count_path()}))
    e.close()
