"""super-k-mer path on the bench workload: stage times, spills, failed buckets, S3 phase stamps (debug flag 16)"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
hint = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 28
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ds = synth_stream(reads, 150, 100_000_000, seed=20260417, device="cuda", genome_seed=20260417)
torch.cuda.synchronize()
e = KmerEngine(31, capacity_hint=hint)
e.set_option("debug_flags", flags); e.set_option("force_path", 3)
for it in range(3):
    e.clear()
    e.profile(True)
    t0 = time.perf_counter()
    e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
    e.synchronize()
    dt = time.perf_counter() - t0
    ms, n = e.profile_stages()
    e.profile(False)
    st = {k: e.get_stat(k) for k in ("sk_passes", "sk_spills", "sk_failed_buckets", "sk_fallbacks", "ovf_log2cap", "log2cap", "bucket_bits")}
    ph = [e.get_stat("dbg_t%d" % i) for i in range(6)]
    print(json.dumps({"wall_ms": round(dt * 1e3, 2), "stage_ms": [round(x, 3) for x in ms], "stats": st, "phase_kcyc64": ph, "stats2": e.stats()}))
