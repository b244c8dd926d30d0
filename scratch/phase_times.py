"""phase cycle sums of the one-piece-per-workgroup piece sort (variant build -DKB_TIMING; debug flag 64 selects that kernel)"""
import sys, torch
sys.path.insert(0, '.')
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda:0"); torch.cuda.synchronize()
with KmerEngine(k, capacity_hint=1 << 28) as e:
    e.set_option("debug_flags", 64)
    for it in range(2):
        e.clear(); e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.flush(); e.synchronize()
    t0 = [e.get_stat(f"trash{8 + i}") for i in range(48)]
    e.clear(); e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.flush(); e.synchronize()
    t1 = [e.get_stat(f"trash{8 + i}") for i in range(48)]
    d = [b - a for a, b in zip(t0, t1)]
    n = max(d[9], 1)
    names = ["offsets loaded", "run scan", "run table", "gather (issue + arrival)", "rank", "fine scan", "scatter", "write-out issue", "write-out drain"]
    tot = sum(d[:9])
    print("pieces", n, "cycles per piece", round(tot / n))
    for nm, v in zip(names, d[:9]):
        print(f"  {nm:28s} {v / n:9.0f} cycles  {100 * v / tot:5.1f} %")
    na = max(d[19], 1)
    names_a = ["keys + bins + rank atomics issued", "ranks back + B1", "scan (one wave)", "B2", "scatter + offset row", "next words retired", "B3", "write-out issued", "B4 + turn-around"]
    tot = sum(d[10:19])
    print("slabs", na, "cycles per slab", round(tot / na))
    for nm, v in zip(names_a, d[10:19]):
        print(f"  {nm:36s} {v / na:9.0f} cycles  {100 * v / tot:5.1f} %")
    nc = max(d[40], 1)
    names_c = ["slice init", "pass descriptors (setup)", "run bounds requested", "bounds arrived + scan", "run table + entry loads issued", "entries arrived", "lookahead resolve", "queue drain", "barrier (other waves)", "write-back issued"]
    tot = sum(d[30:40])
    print("buckets", nc, "cycles per bucket", round(tot / nc))
    for nm, v in zip(names_c, d[30:40]):
        print(f"  {nm:36s} {v / nc:9.0f} cycles  {100 * v / tot:5.1f} %")
