"""Fixed cost of kernel C: a tiny stream into a 2^29-slot table (every bucket workgroup only opens and writes back its slice)."""
import sys, torch
sys.path.insert(0, '.')
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
ds = synth_stream(4096, 150, 1_000_000, seed=3, device="cuda:0"); torch.cuda.synchronize()
for flags in (0, 8):
    with KmerEngine(31, capacity_hint=1 << 28) as e:
        e.set_option("force_path", 2); e.set_option("debug_flags", flags)
        for it in range(3):
            e.clear(); e.profile(True)
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
        print("flags", flags, "stages ms", [round(x, 3) for x in e.profile_stages()[0]], "passes", e.profile_stages()[1], e.stats(), flush=True)
