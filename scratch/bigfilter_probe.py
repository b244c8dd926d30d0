"""count --if of one streamed batch against a BIG filter (no sieve: above 16 MB), binned vs direct vs the engine's choice"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
big = synth_stream(20_000_000, 150, 3_000_000_000, seed=3, device="cuda", genome_seed=1)
batches = {n: synth_stream(n, 150, 3_000_000_000, seed=5 + n, device="cuda", genome_seed=1) for n in (440_000, 7_000_000)}
torch.cuda.synchronize()
src = KmerEngine(31, capacity_hint=1 << 31)
src.count_dev(big.packed.data_ptr(), big.invalid.data_ptr(), big.n_bases); src.synchronize()
n = src.stats()[1]
lo = torch.empty(n, dtype=torch.int64, device="cuda"); cnt = torch.empty(n, dtype=torch.int32, device="cuda")
src.export_ge_dev(0, lo.data_ptr(), None, cnt.data_ptr(), n); src.synchronize(); src.close(); del cnt
ref = {}
for path, pname in ((1, "direct"), (2, "binned"), (0, "auto")):
    e = KmerEngine(31, capacity_hint=n); e.load_filter_dev(lo.data_ptr(), None, n); e.set_option("force_path", path)
    row = {"filter_keys": n, "table_GB": round(e.stats()[0] * 12 / 1e9, 1), "path": pname}
    for nr, ds in batches.items():
        best = 1e9
        for it in range(3):
            t0 = time.perf_counter()
            e.count_filtered_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        row[f"{ds.n_bases >> 20}M_pos_ms"] = round(best, 2); row[f"{ds.n_bases >> 20}M_path"] = e.last_count_path()
    q = torch.empty(1 << 20, dtype=torch.int32, device="cuda"); e.query_dev(lo.data_ptr(), None, 1 << 20, q.data_ptr()); e.synchronize()
    s_ = int(q.to(torch.int64).sum()); ref.setdefault("s", s_); row["same_counts"] = s_ == ref["s"]
    print(json.dumps(row), flush=True); e.close()
