"""Cost of bucket skew at bench size (VERDICT r1 weak point 9): the bench pass on 10 M reads from a REPEAT-RICH 100 Mbp
genome (an Alu-like 300 bp family every ~1.5 kb at 10 % divergence, microsatellites of 40-200 bp, poly-A runs) against the
uniform genome: time of the FIRST pass of a fresh engine (the skew is known before kernel C runs, since C is deferred) and of
a steady pass, heavy buckets, replays, heaviest k-mer."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream, synth_genome

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

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
G = 100_000_000
genomes = {"uniform": synth_genome(G, 20260417, "cuda"), "repeat_rich": torch.from_numpy(repeat_rich(G, 7)).cuda()}
for gname, g in genomes.items():
    ds = synth_stream(reads, 150, seed=20260417, device="cuda", genome=g)
    torch.cuda.synchronize()
    for pname, path in (("binned", 2),):
        e = KmerEngine(int(os.environ.get("K", "31")), capacity_hint=1 << 28)
        e.set_option("force_path", path)
        best = first = None
        for it in range(4):
            e.clear(); e.flush(); e.synchronize()
            t0 = time.perf_counter()
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.flush(); e.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            if it == 0: first = dt
            else: best = dt if best is None else min(best, dt)
        e.clear(); e.profile(True)
        e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.flush(); e.synchronize()
        stage_ms, _ = e.profile_stages(); names = e.profile_stage_names(); e.profile(False)
        cap, distinct, windows = e.stats()
        lo, hi, cnt = e.export_ge(1000)
        row = {"genome": gname, "path": pname, "first_pass_wall_ms": round(first, 2), "wall_ms": round(best, 2), "Gkmer_per_s": round(windows / best / 1e6, 1), "windows": windows,
               "distinct": distinct, "slots": cap, "kmers_ge1000": int(len(lo)), "max_count": int(cnt.max()) if len(cnt) else 0,
               "stage_ms": {n: round(x, 2) for n, x in zip(names, stage_ms)}}
        for s in ("heavy_buckets", "replayed_buckets", "flushes"):
            try: row[s] = e.get_stat(s)
            except Exception: pass
        print(json.dumps(row), flush=True)
        e.close()
