import sys, time, torch
sys.path.insert(0, '.')
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda:0"); torch.cuda.synchronize()
for flags in (0, 0, 0):
    with KmerEngine(31, capacity_hint=1 << 28) as e:
        e.set_option("debug_flags", flags)
        for it in range(3):
            e.clear(); e.synchronize()
            t0 = time.perf_counter()
            try:
                e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
            except Exception as ex:
                pass
            e.synchronize()
            dt = time.perf_counter() - t0
        e.profile(True); e.clear()
        try:
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
        except Exception as ex: pass
        print("flags", flags, "pass ms %.2f" % (dt * 1e3), [round(x,2) for x in e.profile_stages()[0]], flush=True)
