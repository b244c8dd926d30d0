"""End to end from a BAM file on disk: BGZF inflate + parse + H2D + count (the rate a caller of the mirrors sees), by host
threads and by the number of reader pipelines (BGZF ranges read side by side, kdf_bam_open_range)."""
import sys, time, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import write_bam
from kmer_denovo_filter_amd import KmerEngine, bam_reader
from kmer_denovo_filter_amd.core.jellyfish_wrappers import _stream_bam
path = '/tmp/synth_4m.bam'
N = 4_000_000
if not os.path.exists(path):
    rng = np.random.default_rng(1); genome = rng.integers(0, 4, 20_000_000); B = np.frombuffer(b"ACGT", np.uint8)
    starts = np.sort(rng.integers(0, len(genome) - 150, N))
    t = time.time()
    write_bam(path, [("chr1", 20_000_000)], [{"name": f"r{i}", "seq": B[genome[s:s + 150]].tobytes().decode(), "pos": int(s), "flag": 0x41 if i & 1 else 0x81} for i, s in enumerate(starts)])
    print("wrote", os.path.getsize(path) / 1e6, "MB in", round(time.time() - t, 1), "s", flush=True)
# the reader alone (no GPU): one pipeline, by threads
for threads in (1, 4, 16):
    t = time.time(); nb = 0
    for st in bam_reader(path, max_bases=1 << 26, threads=threads):
        nb += st.n_bases
    dt = time.time() - t
    print(f"reader only, 1 pipeline, threads={threads}: {nb / dt / 1e9:.2f} Gbase/s", flush=True)
for threads, pipes in ((1, 1), (4, 1), (8, 1), (16, 1), (8, 2), (16, 2), (16, 4), (16, 8), (32, 8)):
    os.environ["KDF_READER_PIPELINES"] = str(pipes)
    with KmerEngine(31, capacity_hint=1 << 26) as e:
        _stream_bam(e, path, None, threads, filtered=False); e.flush()          # warm
        e.clear(); e.synchronize()
        t = time.time(); n = _stream_bam(e, path, None, threads, filtered=False); e.flush(); e.synchronize(); dt = time.time() - t
        w = e.stats()[2]
    assert n == N
    print(f"threads={threads} pipelines={pipes}: {n} reads, {w / dt / 1e9:.3f} Gk-mer/s end to end ({150 * n / dt / 1e9:.2f} Gbase/s)", flush=True)
