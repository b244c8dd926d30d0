"""a streamed sample into a BIG table (BASELINE configs[3] proxy on one GPU): many _stream_bam-sized batches (2^26 positions)
into one table, amortised rate of the deferred binned pipeline (pending stream -> partition passes -> ONE kernel C per flush)
against the direct global-atomic kernels, by table size"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
g_len = 3_000_000_000 if len(sys.argv) < 2 else int(sys.argv[1])
n_batches = 64 if len(sys.argv) < 3 else int(sys.argv[2])
streams = [synth_stream(440_000, 150, g_len, seed=5 + i, device="cuda", genome_seed=1) for i in range(8)]
torch.cuda.synchronize()
for log2hint in (28, 30, 32):
    for path, pname in ((1, "direct"), (0, "auto (deferred)")):
        e = KmerEngine(31, capacity_hint=1 << log2hint); e.set_option("force_path", path)
        best, w = 1e9, 0
        for it in range(2):
            e.clear(); e.flush(); e.synchronize()
            t0 = time.perf_counter()
            for b in range(n_batches):
                ds = streams[b % len(streams)]
                e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
            e.flush(); e.synchronize()
            best = min(best, time.perf_counter() - t0)
        cap, distinct, w = e.stats()
        print(json.dumps({"table_GB": round(12 * cap / 1e9, 1), "path": pname, "batches": n_batches, "positions_per_batch": streams[0].n_bases,
                          "total_ms": round(best * 1e3, 1), "ms_per_batch": round(best * 1e3 / n_batches, 3), "Gkmer_per_s": round(w / best / 1e9, 2),
                          "flushes": e.get_stat("flushes"), "partition_passes": e.get_stat("binned_passes"), "last_path": e.last_count_path()}), flush=True)
        e.close()
