import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
from skew_probe_lib import repeat_rich
for reads, G, hint in ((200_000, 2_000_000, 1 << 23), (1_000_000, 10_000_000, 1 << 25), (3_000_000, 30_000_000, 1 << 27)):
    g = torch.from_numpy(repeat_rich(G, 7)).cuda()
    ds = synth_stream(reads, 150, seed=20260417, device="cuda", genome=g); torch.cuda.synchronize()
    for bal in (1, 0):
        e = KmerEngine(31, capacity_hint=hint); e.set_option("force_path", 3); e.set_option("sk_balance", bal)
        e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
        cap, distinct, windows = e.stats()
        lo = torch.empty(distinct, dtype=torch.int64, device="cuda"); cnt = torch.empty(distinct, dtype=torch.int32, device="cuda")
        n = e.export_ge_dev(0, lo.data_ptr(), None, cnt.data_ptr(), distinct, sorted_=True); e.synchronize()
        lo = lo[:n]
        dmask = lo[1:] == lo[:-1]
        d = lo[1:][dmask]
        uniq, c = torch.unique(d, return_counts=True)
        top = sorted(zip(c.tolist(), [hex(int(x) & (2**64 - 1)) for x in uniq.tolist()]), reverse=True)[:4]
        print(json.dumps({"reads": reads, "balance": bal, "distinct": distinct, "dups": int(dmask.sum()), "dup_keys": int(uniq.numel()), "top": top,
                          "spills": e.get_stat("sk_spills"), "failed": e.get_stat("sk_failed_buckets"), "ovf": e.get_stat("ovf_log2cap"),
                          "passes": e.get_stat("sk_passes"), "log2cap": e.get_stat("log2cap")}), flush=True)
        e.close()
