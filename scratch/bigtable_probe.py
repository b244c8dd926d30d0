"""a streamed sample into a BIG table: cost of one batch through the binned pipeline (which rewrites every bucket of the table)
against the direct global-atomic kernels, by batch size and table size"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
g_len = 3_000_000_000 if len(sys.argv) < 2 else int(sys.argv[1])
streams = {n: synth_stream(n, 150, g_len, seed=5 + n, device="cuda", genome_seed=1) for n in (440_000, 1_760_000, 7_000_000)}
torch.cuda.synchronize()
for log2hint in (28, 30, 32):
    for path, pname in ((1, "direct"), (2, "binned"), (0, "auto")):
        e = KmerEngine(31, capacity_hint=1 << log2hint); e.set_option("force_path", path)
        row = {"table_GB": round(12 * 2 ** (log2hint + 1) / 1e9, 1), "path": pname}
        for n, ds in streams.items():
            best = 1e9
            for it in range(3):
                t0 = time.perf_counter()
                e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
                best = min(best, (time.perf_counter() - t0) * 1e3)
            row[f"{ds.n_bases >> 20}M_pos_ms"] = round(best, 2)
            row[f"{ds.n_bases >> 20}M_path"] = e.last_count_path()
        print(json.dumps(row), flush=True)
        e.close()
