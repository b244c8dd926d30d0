"""End to end from a BAM file on disk: BGZF inflate + parse + H2D + count (the rate a caller of the mirrors sees)."""
import sys, time, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import write_bam
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.core.jellyfish_wrappers import _stream_bam
path = '/tmp/synth_1m.bam'
if not os.path.exists(path):
    rng = np.random.default_rng(1); genome = rng.integers(0, 4, 5_000_000); B = np.frombuffer(b"ACGT", np.uint8)
    starts = np.sort(rng.integers(0, len(genome) - 150, 1_000_000))
    write_bam(path, [("chr1", 5_000_000)], [{"name": f"r{i}", "seq": B[genome[s:s + 150]].tobytes().decode(), "pos": int(s), "flag": 0x41 if i & 1 else 0x81} for i, s in enumerate(starts)])
for threads in (1, 4, 8, 16):
    with KmerEngine(31, capacity_hint=1 << 24) as e:
        _stream_bam(e, path, None, threads, filtered=False)          # warm
        e.clear(); e.synchronize()
        t = time.time(); n = _stream_bam(e, path, None, threads, filtered=False); e.synchronize(); dt = time.time() - t
        w = e.stats()[2]
    print(f"threads={threads}: {n} reads, {w / dt / 1e9:.3f} Gk-mer/s end to end ({150 * n / dt / 1e9:.2f} Gbase/s)", flush=True)
