"""stage times of the bench pass under ablation flags (WRONG results: timing only; needs a variant build with -DKB_ABLATE:
scratch/build_variant.sh abl -DKB_ABLATE, then copy scratch/variants/libkdf_abl.so over kmer_denovo_filter_amd/libkdf.so).  64 = one-piece-per-workgroup piece sort;
256 = A without write-out / B gather from 8 KB / C entries from 8 KB; 512 = B without write-out / C without write-back;
1024 = A and B without rank return values"""
import sys, time, torch
sys.path.insert(0, '.')
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
flags_list = [int(x) for x in sys.argv[1].split(",")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda:0"); torch.cuda.synchronize()
for flags in flags_list:
    with KmerEngine(k, capacity_hint=1 << 28) as e:
        e.set_option("debug_flags", flags)
        for it in range(2):
            e.clear(); e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.flush(); e.synchronize()
        e.profile(True)
        for it in range(4):
            e.clear(); e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.flush()
        e.synchronize()
        st, n = e.profile_stages()
        print("flags", flags, "stages ms", [round(x / max(n, 1), 3) for x in st], flush=True)
