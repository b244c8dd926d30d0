import sys, time, torch
sys.path.insert(0, '.')
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
flags_list = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 8, 0, 8]
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda:0"); torch.cuda.synchronize()
for flags in flags_list:
    with KmerEngine(int(sys.argv[2]) if len(sys.argv) > 2 else 31, capacity_hint=1 << 28) as e:
        e.set_option("debug_flags", flags)
        best = 1e9
        for it in range(4):
            e.clear(); e.synchronize()
            t0 = time.perf_counter()
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
            e.synchronize()
            best = min(best, time.perf_counter() - t0)
        sig = [e.count_ge(c) for c in (1, 2, 3, 10, 40)]
        e.profile(True); e.clear()
        e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
        print("flags", flags, "pass ms %.2f" % (best * 1e3), [round(x, 2) for x in e.profile_stages()[0]], sig, flush=True)
