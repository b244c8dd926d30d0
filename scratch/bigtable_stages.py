"""bigtable_probe with the stage breakdown (profile_stages) and the partition-only run (debug flag 2048), per table size"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
n_batches = int(os.environ.get('NB', '64'))
streams = [synth_stream(440_000, 150, 3_000_000_000, seed=5 + i, device="cuda", genome_seed=1) for i in range(8)]
torch.cuda.synchronize()
for log2hint in [int(a) for a in sys.argv[1:]] or [30, 31, 32]:
    for dbg in (0, 2048):
        e = KmerEngine(31, capacity_hint=1 << log2hint)
        if dbg: e.set_option("debug_flags", dbg)
        best = 1e9
        for it in range(3):
            e.clear(); e.flush(); e.synchronize()
            if it == 2: e.profile(True)
            t0 = time.perf_counter()
            for b in range(n_batches):
                ds = streams[b % len(streams)]
                e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
            e.flush(); e.synchronize()
            dt = time.perf_counter() - t0
            if it < 2: best = min(best, dt)
        ms, n = e.profile_stages()
        cap, distinct, w = e.stats()
        print(json.dumps({"slots_log2": cap.bit_length() - 1, "table_GB": round(12 * cap / 1e9, 1), "dbg": dbg, "total_ms": round(best * 1e3, 1),
                          "profiled_ms": round(dt * 1e3, 1), "stages_ms": [round(x, 2) for x in ms], "passes": n, "batches": n_batches, "Gkmer_per_s": round(w / best / 1e9, 2), "ring_GB": round(e.get_stat("ring_bytes") / 1e9, 1),
                          "flushes": e.get_stat("flushes"), "distinct": distinct}), flush=True)
        e.close()
