"""binned `count --if` of a repeat-rich parent at bench size (10 M x 150 bp from the repeat-rich 100 Mbp genome of skew_probe.py)
against a filter of 100 M k-mers of a uniform genome (a whole-genome filter is too big for the sieve: binned path): every window is partitioned, the homopolymer / microsatellite buckets
are heavy although their k-mers are not in the filter (kb_heavy_filtered_kernel)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream, synth_genome
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
def repeat_rich(n_bases, seed):
    rng = np.random.default_rng(seed)
    alu = rng.integers(0, 4, 300)
    out, n = [], 0
    while n < n_bases:
        piece = rng.integers(0, 4, int(rng.integers(600, 2400))); out.append(piece); n += len(piece)
        copy = alu.copy(); mut = rng.random(300) < 0.10; copy[mut] = rng.integers(0, 4, int(mut.sum()))
        out.append(copy if rng.random() < 0.5 else (3 - copy)[::-1]); n += 300
        unit = [np.array([1, 0]), np.array([2, 0, 0]), np.array([0])][int(rng.integers(0, 3))]
        sat = np.tile(unit, int(rng.integers(40, 200)) // len(unit) + 1); out.append(sat); n += len(sat)
    return np.concatenate(out).astype(np.uint8)[:n_bases]
k = int(os.environ.get("K", "31"))
G = 100_000_000
g = torch.from_numpy(repeat_rich(G, 7)).cuda()
parent = synth_stream(10_000_000, 150, seed=20260417, device="cuda", genome=g)
child = synth_stream(10_000_000, 150, G, seed=3, device="cuda")
torch.cuda.synchronize()
with KmerEngine(k, capacity_hint=1 << 28) as c:
    c.count_dev(child.packed.data_ptr(), child.invalid.data_ptr(), child.n_bases)
    lo, hi, cnt = c.export_ge(0)
lo, hi = lo[:100_000_000].copy(), hi[:100_000_000].copy()
del cnt
with KmerEngine(k, capacity_hint=len(lo)) as e:
    e.load_filter(lo, hi if k > 32 else None)
    best = None
    for it in range(3):
        e.reset_counts(); e.synchronize()
        t0 = time.perf_counter()
        e.count_filtered_dev(parent.packed.data_ptr(), parent.invalid.data_ptr(), parent.n_bases); e.flush(); e.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    print(json.dumps({"k": k, "filter_keys": len(lo), "wall_ms": round(best, 2), "heavy_buckets": e.get_stat("heavy_buckets"), "path": e.last_count_path(), "hits": int(e.query(lo, hi if k > 32 else None).sum())}), flush=True)
