"""owner-side merge cost on ONE GPU: dump the bench table grouped by owner (as a rank would before the all-to-all), then
add every owner part into an owner table the way the receiving rank does (part by part, source by source)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.distributed import EngineOps
from kmer_denovo_filter_amd.synth import synth_stream
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda", genome_seed=20260417)
torch.cuda.synchronize()
e = KmerEngine(31, capacity_hint=1 << 28)
own = KmerEngine(31, capacity_hint=1 << 28)
ops, oops = EngineOps(e, dev), EngineOps(own, dev)
e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
for it in range(3):
    own.clear()
    (lo, hi, cnt, counts), t_exp = T(lambda: ops.export_pairs_by_owner(world))
    _, t_pack = T(lambda: ops.export_packed_by_owner(world))
    # `world` sources, each a hash-ordered eighth of the dump (pair i -> source i % world): what an owner sees, at full size
    segs = [(lo[s::world].contiguous(), None, cnt[s::world].contiguous()) for s in range(world)]
    _, t_add = T(lambda: oops.add_pairs_segments(segs))
    path = own.get_stat("last_merge_path")
    _, t_add2 = T(lambda: oops.add_pairs_segments(segs))          # into the live table (load + store of every bucket)
    t_rand = None
    if it == 2:
        own.clear()
        p = torch.randperm(lo.numel(), device=dev)
        l2, c2 = lo[p].contiguous(), cnt[p].contiguous()
        t_rand = T(lambda: oops.add_pairs(l2, None, c2))[1]
    print(json.dumps({"world": world, "pairs": int(lo.numel()), "export_parts_ms": round(t_exp, 2), "export_packed_ms": round(t_pack, 2), "merge_fresh_ms": round(t_add, 2),
                      "merge_live_ms": round(t_add2, 2), "path": path, "add_pairs_random_order_ms": t_rand and round(t_rand, 2),
                      "owner_stats": own.stats()}))
