"""the direct kernels on repeat-rich input: a streamed batch (63 M positions) into a big table, uniform vs repeat-rich genome"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream, synth_genome
from skew_probe_lib import repeat_rich
G = 100_000_000
genomes = {"uniform": synth_genome(G, 20260417, "cuda"), "repeat_rich": torch.from_numpy(repeat_rich(G, 7)).cuda()}
for gname, g in genomes.items():
    batches = [synth_stream(440_000, 150, seed=100 + i, device="cuda", genome=g) for i in range(6)]
    torch.cuda.synchronize()
    for k in (31, 63):
        for path, pname in ((0, "auto"), (2, "binned")):
            e = KmerEngine(k, capacity_hint=1 << 30); e.set_option("force_path", path)
            ts = []
            for ds in batches:
                t0 = time.perf_counter()
                e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
                ts.append(round((time.perf_counter() - t0) * 1e3, 2))
            print(json.dumps({"genome": gname, "k": k, "path": pname, "taken": e.last_count_path(), "batch_ms": ts, "stats": e.stats()}), flush=True)
            e.close()
