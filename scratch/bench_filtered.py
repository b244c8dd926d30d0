import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
dev = "cuda:0"
parent = synth_stream(10_000_000, 150, 100_000_000, seed=777, device=dev, genome_seed=20260417); torch.cuda.synchronize()
child = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device=dev, genome_seed=20260417); torch.cuda.synchronize()
with KmerEngine(31, capacity_hint=1 << 28) as e:
    e.count_dev(child.packed.data_ptr(), child.invalid.data_ptr(), child.n_bases)
    lo, hi, cnt = e.export_ge(2)
print("child keys with count>=2:", len(lo), flush=True)
rng = np.random.default_rng(0)
for nkeys in (1000, 100_000, 4_000_000, 30_000_000, len(lo)):
    sel = np.sort(rng.choice(len(lo), size=min(nkeys, len(lo)), replace=False))
    flo = lo[sel]
    res = {}
    for path in (1, 2):
        with KmerEngine(31) as e:
            e.load_filter(flo)
            e.set_option("force_path", path)
            ts = []
            for it in range(3):
                e.synchronize(); t0 = time.perf_counter()
                e.count_filtered_dev(parent.packed.data_ptr(), parent.invalid.data_ptr(), parent.n_bases)
                e.synchronize(); ts.append(time.perf_counter() - t0)
            c = e.query(flo)
            res[path] = (min(ts), int(c.astype(np.uint64).sum()) // 3, e.get_stat("log2cap"))
    assert res[1][1] == res[2][1], res
    w = 1163e6
    print(f"filter {len(flo):>10d} keys log2cap {res[1][2]}: direct {res[1][0]*1e3:7.2f} ms ({w/res[1][0]/1e9:6.1f} Gk/s)  binned {res[2][0]*1e3:7.2f} ms ({w/res[2][0]/1e9:6.1f} Gk/s)  hits {res[1][1]}", flush=True)
