"""Cost of the multi-GPU merge pieces on ONE GPU (no RCCL here): owner-ordered dump of the
local table, and the owner-side add of as many pairs as an 8-rank exchange delivers."""
import sys, time, torch
sys.path.insert(0, '.')
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.distributed import EngineOps
from kmer_denovo_filter_amd.synth import synth_stream
dev = torch.device("cuda:0")
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device=dev); torch.cuda.synchronize()
def T(f, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
with KmerEngine(31, capacity_hint=1 << 28) as e, KmerEngine(31, capacity_hint=1 << 28) as o:
    ops, oops = EngineOps(e, dev), EngineOps(o, dev)
    ops.count_stream(ds.packed, ds.invalid, ds.n_bases); e.synchronize()
    t, r = T(lambda: ops.export_pairs_by_owner(8)); lo, hi, cnt, counts = r
    print("owner-ordered dump: %.2f ms for %d pairs" % (t, lo.numel()), counts[:3], flush=True)
    t2, _ = T(lambda: ops.export_pairs(0), 2)
    print("plain dump: %.2f ms" % t2, flush=True)
    def add():
        o.clear(); oops.add_pairs(lo, hi, cnt); o.synchronize()
    t3, _ = T(add, 2)
    print("owner add_pairs of %d pairs: %.2f ms (%.1f Gpairs/s)" % (lo.numel(), t3, lo.numel() / t3 / 1e6), flush=True)
    t4, n3 = T(lambda: o.count_ge(3))
    print("count_ge: %.2f ms -> %d (local %d)" % (t4, n3, e.count_ge(3)), flush=True)
