"""Which path is right on the repeat-rich genome?  direct vs binned vs super-k-mer: distinct, sum of counts, sorted dumps."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from skew_probe_lib import repeat_rich
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
g = torch.from_numpy(repeat_rich(G, 7)).cuda()
ds = synth_stream(reads, 150, seed=20260417, device="cuda", genome=g)
torch.cuda.synchronize()
dumps = {}
for pname, path in (("direct", 1), ("binned", 2), ("superkmer", 3)):
    e = KmerEngine(31, capacity_hint=1 << 28 if reads >= 5_000_000 else reads * 40)
    e.set_option("force_path", path)
    e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
    cap, distinct, windows = e.stats()
    lo = torch.empty(distinct, dtype=torch.int64, device="cuda"); cnt = torch.empty(distinct, dtype=torch.int32, device="cuda")
    n = e.export_ge_dev(0, lo.data_ptr(), None, cnt.data_ptr(), distinct, sorted_=True); e.synchronize()
    lo, cnt = lo[:n], cnt[:n]
    dup = int((lo[1:] == lo[:-1]).sum()) if n > 1 else 0
    print(json.dumps({"path": pname, "distinct": distinct, "dumped": n, "windows": windows, "sum_counts": int(cnt.to(torch.int64).sum()),
                      "duplicate_keys_in_dump": dup, "spills": e.get_stat("sk_spills"), "failed": e.get_stat("sk_failed_buckets"),
                      "ovf_log2cap": e.get_stat("ovf_log2cap"), "sk_passes": e.get_stat("sk_passes")}), flush=True)
    dumps[pname] = (lo, cnt)
    e.close()
for a, b in (("direct", "binned"), ("direct", "superkmer")):
    la, ca = dumps[a]; lb, cb = dumps[b]
    same = la.numel() == lb.numel() and bool(torch.equal(la, lb)) and bool(torch.equal(ca, cb))
    print(a, "==", b, ":", same, flush=True)
    if not same and b == "superkmer":
        # keys that appear twice in the super-k-mer dump
        d = lb[1:][lb[1:] == lb[:-1]]
        print("  duplicated keys:", d.numel(), [hex(int(x)) for x in d[:5].tolist()])
