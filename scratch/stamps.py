import sys, torch
sys.path.insert(0, '.')
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda:0"); torch.cuda.synchronize()
with KmerEngine(31, capacity_hint=1 << 28) as e:
    e.set_option("debug_flags", 2048)
    for it in range(2):
        e.clear(); e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
    t = [e.get_stat(f"dbg_t{i}") for i in range(6)]
    names = ["extract+rank (wave0)", "wait B1", "scan + B2", "LDS scatter + B3", "copy-out (wave0)", "wait B4"]
    nwg = 4012; tot = sum(t)
    for n, v in zip(names, t):
        print(f"{n:24s} {v/nwg/100e6*1e6:8.1f} us per WG   {100*v/tot:5.1f}%")
    print("sum per WG %.1f us (memtime ticks assumed 100 MHz)" % (tot / nwg / 100e6 * 1e6))
