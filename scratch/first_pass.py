import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
k = int(sys.argv[1]) if len(sys.argv) > 1 else 63
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda"); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); e = KmerEngine(k, capacity_hint=1 << 28); e.synchronize(); t1 = time.perf_counter()
    e.set_option("force_path", 2)
    for it in range(3):
        e.clear(); e.synchronize()
        ta = time.perf_counter(); e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); tb = time.perf_counter(); e.synchronize(); tc = time.perf_counter()
        e.flush(); e.synchronize(); td = time.perf_counter()
        print(f"k={k} engine {rep} create {1e3*(t1-t0):.1f} ms; pass {it}: count call {1e3*(tb-ta):.1f} + sync {1e3*(tc-tb):.1f} + flush {1e3*(td-tc):.1f} ms", flush=True)
    e.close()
