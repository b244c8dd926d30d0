"""Repeat-rich genome at bench size, the paths skew_check.py does not cover: k = 63 (binned vs direct), count --if
(sieve vs binned vs direct), and the two ends of the merge (hash-ordered dump -> 8 owners -> union == the table)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from kmer_denovo_filter_amd import KmerEngine, devkeys
from kmer_denovo_filter_amd.distributed import EngineOps
from kmer_denovo_filter_amd.synth import synth_stream
from skew_probe_lib import repeat_rich
dev = torch.device("cuda:0")
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = torch.from_numpy(repeat_rich(100_000_000, 7)).cuda()
ds = synth_stream(reads, 150, seed=20260417, device="cuda", genome=g)
ds2 = synth_stream(reads, 150, seed=99, device="cuda", genome=g)
torch.cuda.synchronize()

def dump(e, wide):
    _, distinct, windows = e.stats()
    lo = torch.empty(distinct, dtype=torch.int64, device=dev); hi = torch.empty(distinct, dtype=torch.int64, device=dev) if wide else None
    cnt = torch.empty(distinct, dtype=torch.int32, device=dev)
    n = e.export_ge_dev(0, lo.data_ptr(), hi.data_ptr() if wide else None, cnt.data_ptr(), distinct, sorted_=True); e.synchronize()
    assert n == distinct
    return lo, hi, cnt, windows

# 1. k = 63
res = []
for path in (1, 2):
    e = KmerEngine(63, capacity_hint=1 << 28); e.set_option("force_path", path)
    e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
    res.append(dump(e, True)); e.close()
same = all(torch.equal(a, b) for a, b in zip(res[0][:3], res[1][:3])) and res[0][3] == res[1][3]
print(json.dumps({"k63 binned == direct": same, "distinct": int(res[0][0].numel()), "windows": res[0][3]}), flush=True)
del res

# 2. count --if: filter = every 50th key of this sample that occurs >= 3 times (heavy keys included), probe = another sample
e = KmerEngine(31, capacity_hint=1 << 28)
e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
lo, _, cnt, _ = dump(e, False)
f = lo[cnt >= 3][::50].contiguous()
heavy = lo[cnt >= 100000]
f = torch.unique(torch.cat([f, heavy]))
out = []
for name, path in (("sieve", 4), ("binned", 2), ("direct", 1)):
    fe = KmerEngine(31, capacity_hint=int(f.numel()))
    fe.load_filter_dev(f.data_ptr(), None, int(f.numel())); fe.set_option("force_path", path)
    fe.count_filtered_dev(ds2.packed.data_ptr(), ds2.invalid.data_ptr(), ds2.n_bases); fe.synchronize()
    out.append((name, devkeys.query(fe, f, None), fe.last_count_path())); fe.close()
print(json.dumps({"filter_keys": int(f.numel()), "heavy_in_filter": int(heavy.numel()), "paths": [o[2] for o in out],
                  "sieve == direct": bool(torch.equal(out[0][1], out[2][1])), "binned == direct": bool(torch.equal(out[1][1], out[2][1])),
                  "sum": int(out[2][1].to(torch.int64).sum())}), flush=True)

# 3. merge: the table's hash-ordered dump by 8 owners, each owner merges its part from 3 "sources" (the part split 3 ways, in order)
ops = EngineOps(e, dev)
plo, _, pcnt, counts = ops.export_pairs_by_owner(8)
a = 0; tot = 0; ok = True
for r, n in enumerate(counts):
    l, c = plo[a:a + n], pcnt[a:a + n]; a += n
    own = KmerEngine(31, capacity_hint=max(n, 1)); oo = EngineOps(own, dev); oo.prepare_owner(8); own.clear()
    oo.add_pairs_segments([(l[i::3].contiguous(), None, c[i::3].contiguous()) for i in range(3)])
    ok = ok and own.get_stat("last_merge_path") == 1 and own.stats()[1] == n and bool(torch.equal(oo.query(l, None), c))
    tot += own.stats()[1]; own.close()
print(json.dumps({"owners hold exactly the table": ok and tot == e.stats()[1], "pairs": int(plo.numel())}), flush=True)
