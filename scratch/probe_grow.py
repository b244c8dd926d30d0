import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
streams = [synth_stream(440_000, 150, 3_000_000_000, seed=5 + i, device="cuda", genome_seed=1) for i in range(8)]
torch.cuda.synchronize()
e = KmerEngine(31, capacity_hint=1 << 28)
for it in range(2):
    e.clear(); e.flush(); e.synchronize()
    t0 = time.perf_counter(); marks = []
    for b in range(64):
        ds = streams[b % 8]
        e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
        if b % 8 == 7:
            e.synchronize(); marks.append(round((time.perf_counter() - t0) * 1e3, 1))
    e.flush(); e.synchronize()
    print("iter", it, "total ms", round((time.perf_counter() - t0) * 1e3, 1), "marks", marks, {k: e.get_stat(k) for k in ("flushes", "binned_passes", "replayed_buckets", "log2cap", "bucket_bits", "ring_bytes")}, flush=True)
